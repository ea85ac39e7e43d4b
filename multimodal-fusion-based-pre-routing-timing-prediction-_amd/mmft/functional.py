"""Autograd wrappers over the HIP kernels for the dense layers (MLP, fcn, fusion head).

Backward runs on PyTorch's autograd worker thread: every call is stateless, device and stream are
taken from the tensors / current stream at call time, and `backward(retain_graph=True)`
(reference src/train.py:553) is tolerated because nothing is freed or mutated in backward.
"""
import torch
from . import ops, gradsink


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
    """out[i] = table[idx[i]] (idx int32 on the device); backward = segmented row sums in a fixed order for small
    tables (bitwise reproducible), float atomics otherwise."""

    @staticmethod
    def forward(ctx, table, idx):
        t = table if table.is_contiguous() else table.contiguous()
        ctx.shape = t.shape
        ctx.save_for_backward(idx)
        return ops.gather_rows(t, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        out = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        gg = g if g.is_contiguous() else g.contiguous()
        if ctx.shape[0] <= 4096:
            ops.scatter_add_rows_det(out, idx, gg)       # small table, many duplicates: fixed summation order
        else:
            ops.scatter_add_rows(out, idx, gg)
        return out, None


def gather_rows(table, idx):
    return GatherRowsFn.apply(table, idx)
