"""Autograd wrappers over the HIP kernels for the dense layers (MLP, fcn, fusion head).

Backward runs on PyTorch's autograd worker thread: every call is stateless, device and stream are
taken from the tensors / current stream at call time, and `backward(retain_graph=True)`
(reference src/train.py:553) is tolerated because nothing is freed or mutated in backward.
"""
import torch
from . import ops, gradsink, lib


class LinearActFn(torch.autograd.Function):
    """y = act(x @ w.T + b) with act in {none, ReLU, LeakyReLU(slope)}  (src/model.py:10-24)."""

    @staticmethod
    def forward(ctx, x, w, b, slope):
        act, sl = ops.act_code(slope)
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        wc = w if w.is_contiguous() else w.contiguous()
        y = ops.linear_fwd(x2, wc, b, act=act, slope=sl)
        ctx.act, ctx.slope, ctx.xshape = act, sl, x.shape
        ctx.has_bias = b is not None
        ctx.sinks = (gradsink.of(w) if wc is w else None, gradsink.of(b))
        ctx.save_for_backward(x2, wc, y)
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, w, y = ctx.saved_tensors
        g = gy.reshape(-1, gy.shape[-1])
        if not g.is_contiguous():
            g = g.contiguous()
        if ctx.act != ops.ACT_NONE:
            g = ops.act_bwd(g, y, ctx.act, ctx.slope)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_dgrad(g, w).reshape(ctx.xshape)
        if ctx.needs_input_grad[1] and ctx.has_bias and ctx.needs_input_grad[2]:
            # weight and bias gradient from one pass over g
            dw, db = gradsink.deliver_pair(ctx.sinks[0], ctx.sinks[1],
                                           lambda ow, ob: ops.linear_wgrad(g, x2, dw=ow, db=ob, with_bias=True))
        elif ctx.needs_input_grad[1]:
            dw = gradsink.deliver(ctx.sinks[0], lambda out: ops.linear_wgrad(g, x2, dw=out))
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            db = gradsink.deliver(ctx.sinks[1], lambda out: ops.colsum(g, out=out))
        return dx, dw, db, None


def linear_act(x, w, b=None, slope=None):
    """act(x @ w.T + b); slope None = no activation, 0 = ReLU, >0 = LeakyReLU(slope)."""
    return LinearActFn.apply(x, w, b, slope)


class GatherRowsFn(torch.autograd.Function):
    """out[i] = table[idx[i]] (idx int32 on the device); backward = segmented row sums in a fixed order (bitwise
    reproducible): by binary search when the caller states that idx is ascending (endpoints ordered by level), through a
    stable sort otherwise.  Tables above 4096 rows are not on the reference path and are refused rather than summed with
    float atomics."""

    @staticmethod
    def forward(ctx, table, idx, ascending):
        t = table if table.is_contiguous() else table.contiguous()
        if t.shape[0] > 4096:
            raise NotImplementedError('gather_rows: tables above 4096 rows have no deterministic gradient kernel')
        ctx.shape, ctx.ascending = t.shape, bool(ascending)
        ctx.save_for_backward(idx)
        return ops.gather_rows(t, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        gg = ops.strided_rows(g)                         # a column slice of the concatenated head input: read in place
        D = ctx.shape[1]
        if ctx.ascending and D % 4 == 0 and D <= 256 and 256 % (D // 4) == 0:
            out = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
            lib.call('mmft_seg_sum_sorted', gg, gg.stride(0), idx, gg.shape[0], ctx.shape[0], D, out, out.stride(0), 0,
                     *lib.stream_args(out))
            return out, None, None
        out = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        ops.scatter_add_rows_det(out, idx, gg)           # small table, many duplicates: fixed summation order
        return out, None, None


class ConcatColsFn(torch.autograd.Function):
    """torch.cat(parts, 1) of the fusion head (src/model.py:285-290) as one library launch; the backward hands out column
    slices of the incoming gradient (views: the consumers read them with a row stride)."""

    @staticmethod
    def forward(ctx, *parts):
        ps = [ops._rows2d(p if p.stride(-1) == 1 else p.contiguous(), 'part') for p in parts]
        T = ps[0].shape[0]
        widths = [p.shape[1] for p in ps]
        if any(p.shape[0] != T for p in ps) or any(w % 4 for w in widths) or not 2 <= len(ps) <= 3:
            raise ValueError('concat_cols: two or three blocks of equal row count, widths multiples of 4')
        ps = [p if (p.stride(0) % 4 == 0 and p.storage_offset() % 4 == 0) else p.contiguous() for p in ps]
        out = torch.empty((T, sum(widths)), dtype=torch.float32, device=ps[0].device)
        c = ps[2] if len(ps) == 3 else None
        lib.call('mmft_concat_cols', ps[0], ps[0].stride(0), widths[0], ps[1], ps[1].stride(0), widths[1], c,
                 c.stride(0) if c is not None else 0, widths[2] if c is not None else 0, out, out.stride(0), T, *lib.stream_args(out))
        ctx.widths = widths
        return out

    @staticmethod
    def backward(ctx, g):
        outs, o = [], 0
        for w in ctx.widths:
            outs.append(g[:, o:o + w])
            o += w
        return tuple(outs)


def concat_cols(*parts):
    return ConcatColsFn.apply(*parts)


def gather_rows(table, idx, ascending=False):
    """ascending=True: the caller guarantees idx[i] <= idx[i + 1] (its gradient then needs no sort)."""
    return GatherRowsFn.apply(table, idx, ascending)
