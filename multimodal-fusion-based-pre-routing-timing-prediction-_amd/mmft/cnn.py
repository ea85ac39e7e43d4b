"""Autograd wrappers for the layout-image CNN ops (NHWC / torch channels_last, fp32).

Replaces the torch modules the reference composes in src/Unet.py:8-119 and src/model.py:216-247.
Every op takes and returns (N,C,H,W)-shaped tensors whose memory is NHWC; inputs in another layout
are converted once by the HIP layout kernel.
"""
import torch
from . import ops, gradsink
from .ops import POOL_MAX, POOL_AVG  # noqa: F401


class Conv2dFn(torch.autograd.Function):
    """y = act(conv2d(x, w, stride 1, 'same' padding) + b)  (src/Unet.py:16,19,75; src/model.py:227-243)."""

    @staticmethod
    def forward(ctx, x, w, b, pad, slope):
        act, sl = ops.act_code(slope)
        xn = ops.to_nhwc(x)
        y = ops.conv2d_fwd(xn, w, b, pad, act, sl)
        ctx.pad, ctx.act, ctx.slope = pad, act, sl
        ctx.has_bias = b is not None
        ctx.sinks = (gradsink.of(w), gradsink.of(b))
        ctx.save_for_backward(xn, w, y if act != ops.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        xn, w, y = ctx.saved_tensors
        g = ops.to_nhwc(gy)
        if ctx.act != ops.ACT_NONE:
            g = ops.act_bwd(g.permute(0, 2, 3, 1), y.permute(0, 2, 3, 1), ctx.act, ctx.slope).permute(0, 3, 1, 2)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_dgrad(g, w, ctx.pad)
        if ctx.needs_input_grad[1]:
            dw = gradsink.deliver(ctx.sinks[0] if ops.is_nhwc(w) else None,
                                  lambda out: ops.conv2d_wgrad(xn, g, w.shape[2], w.shape[3], ctx.pad, out=out),
                                  shape=(w.shape[0], w.shape[2], w.shape[3], w.shape[1]))
            dw = dw.permute(0, 3, 1, 2) if dw is not None else None          # [Co][KH][KW][Ci] memory -> OIHW shape
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = gradsink.deliver(ctx.sinks[1], lambda out: ops.colsum(ops.rows_view(g), out=out))
        return dx, dw, db, None, None


def conv2d(x, w, b=None, pad=0, act_slope=None):
    return Conv2dFn.apply(x, w, b, pad, act_slope)


class BnReluFn(torch.autograd.Function):
    """BatchNorm2d in train mode (+ReLU)  (src/Unet.py:17-18,20-21); running stats updated in place."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, relu, per_sample):
        xn = ops.to_nhwc(x)
        y, mean, invstd = ops.bn_train_fwd(xn, gamma.detach(), beta.detach(), running_mean, running_var, momentum, eps,
                                           relu, per_sample)
        ctx.relu = relu
        ctx.sinks = (gradsink.of(gamma), gradsink.of(beta))
        # the backward recomputes the ReLU mask from x with the forward's own affine: y is not kept for it
        ctx.save_for_backward(xn, gamma, beta, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        xn, gamma, beta, mean, invstd = ctx.saved_tensors
        sg, sb = ctx.sinks
        both = sg is not None and sb is not None and gradsink.fresh(sg) and gradsink.fresh(sb)
        outs = (gradsink.take(sg), gradsink.take(sb)) if both else (None, None)
        dx, dgamma, dbeta = ops.bn_train_bwd(ops.to_nhwc(gy), xn, None, gamma.detach(), mean, invstd, ctx.relu,
                                             dgamma=outs[0], dbeta=outs[1], beta=beta.detach())
        if both:
            gradsink.taken(sg)
            gradsink.taken(sb)
            dgamma = dbeta = None
        return dx, dgamma, dbeta, None, None, None, None, None, None


def bn_relu(x, bn, relu=True, per_sample=False, count=True):
    """`bn` is an nn.BatchNorm2d holding the parameters/buffers; always batch statistics."""
    if bn.momentum is None or not bn.affine:
        raise NotImplementedError('BatchNorm2d without momentum/affine is not on the reference path')
    if count and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += x.shape[0] if per_sample else 1
    return BnReluFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, relu, per_sample)


class Pool2x2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        xn = ops.to_nhwc(x)
        ctx.mode = mode
        ctx.save_for_backward(xn)
        return ops.pool2x2_fwd(xn, mode)

    @staticmethod
    def backward(ctx, gy):
        (xn,) = ctx.saved_tensors
        return ops.pool2x2_bwd(xn, ops.to_nhwc(gy), ctx.mode), None


def pool2x2(x, mode):
    return Pool2x2Fn.apply(x, mode)


class OutConvFn(torch.autograd.Function):
    """relu(pool(conv1x1(x) + b)) of OutConv (src/Unet.py:71-82) as one kernel per direction; the backward recomputes
    the pixel values from x, so only x is kept."""

    @staticmethod
    def forward(ctx, x, w, b, mode):
        xn = ops.to_nhwc(x)
        ctx.mode, ctx.has_bias = mode, b is not None
        ctx.sinks = (gradsink.of(w), gradsink.of(b))
        ctx.save_for_backward(xn, w, b)
        return ops.outconv_fwd(xn, w, b, mode)

    @staticmethod
    def backward(ctx, gout):
        xn, w, b = ctx.saved_tensors
        g = gout if gout.is_contiguous() else gout.contiguous()
        res = {}

        def compute(ow, ob):
            dx, dw, db = ops.outconv_bwd(xn, w, b, g, ctx.mode, dw=ow.reshape(-1) if ow is not None else None,
                                         db=ob.reshape(-1) if ob is not None else None)
            res['dx'] = dx
            return dw.reshape(w.shape), (db.reshape(b.shape) if b is not None else None)
        dw, db = gradsink.deliver_pair(ctx.sinks[0], ctx.sinks[1] if ctx.has_bias else None, compute)
        return res['dx'], dw, db, None


def outconv(x, w, b, mode):
    """OutConv body: fused when the shape allows it, else conv2d + pool2x2 + relu."""
    if w.shape[0] == 1 and ops.is_nhwc(x) and ops.outconv_supported(x, w):
        return OutConvFn.apply(x, w, b, mode)
    return relu(pool2x2(conv2d(x, w, b, pad=0), mode))


class UpsampleBilinear2xFn(torch.autograd.Function):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)  (src/Unet.py:50)."""

    @staticmethod
    def forward(ctx, x):
        return ops.upsample_bilinear2x_fwd(ops.to_nhwc(x))

    @staticmethod
    def backward(ctx, gy):
        return ops.upsample_bilinear2x_bwd(ops.to_nhwc(gy))


def upsample_bilinear2x(x):
    return UpsampleBilinear2xFn.apply(x)


def _flat(t):
    """contiguous-memory view of a contiguous or channels_last tensor (None otherwise)."""
    if t.is_contiguous():
        return t
    if t.dim() == 4 and ops.is_nhwc(t):
        return t.permute(0, 2, 3, 1)
    return None


class ActFn(torch.autograd.Function):
    """Standalone ReLU / LeakyReLU (OutConv's ReLU after the pool, src/Unet.py:74-78)."""

    @staticmethod
    def forward(ctx, x, slope):
        act, sl = ops.act_code(slope)
        xc = x if _flat(x) is not None else x.contiguous()
        y = torch.empty_like(xc)
        ops.act_fwd(_flat(xc), act, sl, out=_flat(y))
        ctx.act, ctx.slope = act, sl
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        if y.is_contiguous():
            g = gy if gy.is_contiguous() else gy.contiguous()
            return ops.act_bwd(g, y, ctx.act, ctx.slope), None
        g = ops.to_nhwc(gy)
        return ops.act_bwd(_flat(g), _flat(y), ctx.act, ctx.slope).permute(0, 3, 1, 2), None


def relu(x):
    return ActFn.apply(x, 0.0)


def _wr(w):
    """ConvTranspose2d weight (Ci,Co,2,2) -> [(a,b,co)][ci] matrix (free for parameters stored in that order)."""
    v = w.detach().permute(2, 3, 1, 0)
    v = v if v.is_contiguous() else v.contiguous()
    return v.reshape(-1, w.shape[0])


def _convt_param_grads(ctx, gu, xn, w, Co, Ci, g_rows):
    """(dw, db) of ConvTranspose2d(k=2, s=2) from the unshuffled gradient gu [pixels, (a,b,co)]: dw is the GEMM
    gu^T x; the bias gradient is the GEMM's column-sum by-product folded over the four (a,b) positions, so the
    gradient image is not read a second time.  Entries taken by a gradient sink come back as None."""
    want_w = ctx.needs_input_grad[1]
    want_b = ctx.has_bias and ctx.needs_input_grad[2]
    # the sink is usable when the parameter is stored in (a,b,co,ci) order (Unet.Up keeps it that way)
    sw = ctx.sinks[0] if w.permute(2, 3, 1, 0).is_contiguous() else None
    dw = db = None
    if want_w and want_b:
        col = {}

        def both(out):
            dwr, col['s'] = ops.linear_wgrad(gu, ops.rows_view(xn), dw=out, with_bias=True)     # [(a,b,co)][ci], [4Co]
            return dwr
        dwr = gradsink.deliver(sw, both, shape=(4 * Co, Ci))
        dw = dwr.reshape(2, 2, Co, Ci).permute(3, 2, 0, 1) if dwr is not None else None
        db = gradsink.deliver(ctx.sinks[1], lambda out: torch.sum(col['s'].view(4, Co), 0, out=out)
                              if out is not None else col['s'].view(4, Co).sum(0))
    elif want_w:
        dwr = gradsink.deliver(sw, lambda out: ops.linear_wgrad(gu, ops.rows_view(xn), dw=out), shape=(4 * Co, Ci))
        dw = dwr.reshape(2, 2, Co, Ci).permute(3, 2, 0, 1) if dwr is not None else None
    elif want_b:
        db = gradsink.deliver(ctx.sinks[1], lambda out: ops.colsum(g_rows(), out=out))
    return dw, db


class ConvT2x2Fn(torch.autograd.Function):
    """ConvTranspose2d(k=2, s=2) with bias  (src/Unet.py:53) = GEMM [pixels x Ci][Ci x 4Co] + pixel shuffle."""

    @staticmethod
    def forward(ctx, x, w, b):
        xn = ops.to_nhwc(x)
        N, Ci, H, W = xn.shape
        Co = w.shape[1]
        t = ops.linear_fwd(ops.rows_view(xn), _wr(w), None)
        y = ops.pixel_shuffle2(t, b, N, H, W, Co)
        ctx.has_bias = b is not None
        ctx.sinks = (gradsink.of(w), gradsink.of(b))
        ctx.save_for_backward(xn, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        xn, w = ctx.saved_tensors
        g = ops.to_nhwc(gy)
        N, Ci, H, W = xn.shape
        Co = w.shape[1]
        gu = ops.pixel_unshuffle2(g)                       # [pixels, 4Co]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_dgrad(gu, _wr(w)).reshape(N, H, W, Ci).permute(0, 3, 1, 2)
        dw, db = _convt_param_grads(ctx, gu, xn, w, Co, Ci, lambda: ops.rows_view(g))
        return dx, dw, db


def conv_transpose2x2(x, w, b):
    if tuple(w.shape[2:]) != (2, 2):
        raise NotImplementedError('only ConvTranspose2d(kernel_size=2, stride=2) is on the reference path')
    return ConvT2x2Fn.apply(x, w, b)


class UpCatFn(torch.autograd.Function):
    """torch.cat([x2, ConvTranspose2d(k=2, s=2)(x1)], dim=1) when no padding is needed (src/Unet.py:53-67, even
    sizes): the transposed convolution's pixel shuffle writes straight into its channel slice of the concatenated
    tensor and the backward reads the slice in place - the up-sampled half is never copied."""

    @staticmethod
    def forward(ctx, x1, w, b, x2):
        xn, a = ops.to_nhwc(x1), ops.to_nhwc(x2)
        N, Ci, H, W = xn.shape
        Co, C2 = w.shape[1], a.shape[1]
        t = ops.linear_fwd(ops.rows_view(xn), _wr(w), None)
        out = ops.empty_nhwc(N, C2 + Co, 2 * H, 2 * W, a.device)
        ops.copy_region(a, out, 0, 0, 0)
        ops.pixel_shuffle2(t, b, N, H, W, Co, out=out, c_off=C2)
        ctx.has_bias, ctx.C2 = b is not None, C2
        ctx.sinks = (gradsink.of(w), gradsink.of(b))
        ctx.save_for_backward(xn, w)
        return out

    @staticmethod
    def backward(ctx, g):
        xn, w = ctx.saved_tensors
        g = ops.to_nhwc(g)
        N, Ci, H, W = xn.shape
        Co, C2 = w.shape[1], ctx.C2
        g2 = dx = dw = db = None
        if ctx.needs_input_grad[3]:
            g2 = ops.empty_nhwc(N, C2, 2 * H, 2 * W, g.device)
            ops.copy_region(g2, g, 0, 0, 0, reverse=True)
        gu = ops.pixel_unshuffle2(g, Co=Co, c_off=C2)      # [pixels, 4Co] from the slice, in place
        if ctx.needs_input_grad[0]:
            dx = ops.linear_dgrad(gu, _wr(w)).reshape(N, H, W, Ci).permute(0, 3, 1, 2)
        dw, db = _convt_param_grads(ctx, gu, xn, w, Co, Ci, lambda: ops.rows_view(g)[:, C2:])
        return dx, dw, db, g2


def up_cat(x1, w, b, x2):
    """cat([x2, conv_transpose2x2(x1)], 1); falls back to the two-step form when the sizes call for padding."""
    if tuple(w.shape[2:]) != (2, 2):
        raise NotImplementedError('only ConvTranspose2d(kernel_size=2, stride=2) is on the reference path')
    if x2.shape[2] == 2 * x1.shape[2] and x2.shape[3] == 2 * x1.shape[3] and x2.shape[1] % 4 == 0:
        return UpCatFn.apply(x1, w, b, x2)
    return cat_pad(x2, conv_transpose2x2(x1, w, b))


class CatPadFn(torch.autograd.Function):
    """torch.cat([x2, F.pad(x1, centre)], dim=1)  (src/Unet.py:59-67)."""

    @staticmethod
    def forward(ctx, x2, x1):
        a, b = ops.to_nhwc(x2), ops.to_nhwc(x1)
        N, C2, H, W = a.shape
        C1, h1, w1 = b.shape[1], b.shape[2], b.shape[3]
        dy, dx = H - h1, W - w1
        if dy < 0 or dx < 0:
            raise NotImplementedError('Up: the upsampled map is larger than the skip connection')
        out = ops.empty_nhwc(N, C2 + C1, H, W, a.device)
        if dy or dx:
            out.zero_()
        ops.copy_region(a, out, 0, 0, 0)
        ops.copy_region(b, out, C2, dy // 2, dx // 2)
        ctx.geom = (C2, C1, h1, w1, dy // 2, dx // 2)
        return out

    @staticmethod
    def backward(ctx, g):
        C2, C1, h1, w1, yo, xo = ctx.geom
        g = ops.to_nhwc(g)
        N, _, H, W = g.shape
        g2 = ops.empty_nhwc(N, C2, H, W, g.device)
        g1 = ops.empty_nhwc(N, C1, h1, w1, g.device)
        ops.copy_region(g2, g, 0, 0, 0, reverse=True)
        ops.copy_region(g1, g, C2, yo, xo, reverse=True)
        return g2, g1


def cat_pad(x2, x1):
    return CatPadFn.apply(x2, x1)
