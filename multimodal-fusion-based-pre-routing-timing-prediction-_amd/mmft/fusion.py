"""Fusion-head glue on the HIP path: sparse masked projection (K7), MSE (K10), flat fused Adam (K18).

The reference materialises `path_mask.to_dense() * feat_map` (T x P fp32, 88 MB at defaults) and feeds it
to `fcn` (src/train.py:500-501, src/model.py:271-272).  MaskedPathMap carries the same information as
CSR rows + the 1 x P feature map; PathModel recognises it and never builds the dense map.  A dense
path_map tensor (what an unmodified reference loop passes) is still accepted and goes through the dense
GEMM kernel.
"""
import torch
from . import lib, ops


class PathMasks:
    """Device CSR of a design's path masks (num_paths x P, 0/1)  (src/verilog_parser_asap7.py:1302-1369)."""

    def __init__(self, indptr, cols, P, device):
        self.indptr = torch.as_tensor(indptr).to(torch.int32).to(device).contiguous()
        self.cols = torch.as_tensor(cols).to(torch.int32).to(device).contiguous()
        self.P = int(P)
        self.num_paths = self.indptr.numel() - 1
        if self.cols.numel() and (int(self.cols.max()) >= self.P or int(self.cols.min()) < 0):
            raise ValueError('mask column outside the map')

    @staticmethod
    def batch(masks):
        """Stack the designs' path rows (path ids offset by the running path count). Columns stay in
        [0, P): fcn's weight is shared, the per-design feature map is selected through f_off."""
        ip, cols, off_nnz = [masks[0].indptr[:1]], [], 0
        for m in masks:
            if m.P != masks[0].P:
                raise ValueError('PathMasks.batch: all designs must share the map size')
            ip.append(m.indptr[1:] + off_nnz)
            cols.append(m.cols)
            off_nnz += int(m.cols.numel())
        return PathMasks(torch.cat(ip), torch.cat(cols), masks[0].P, masks[0].indptr.device)


class MaskedPathMap:
    """Lazy `index_select(path_masks, 0, paths).to_dense() * feat_map`: rows = `paths`, values from feat_map."""

    def __init__(self, masks, paths, feat_map, f_off=None):
        self.masks = masks
        self.f_off = f_off            # int32 [T]: offset b*P of each row's design into feat_map [B, P] (None = 0)
        if not torch.is_tensor(paths):
            paths = torch.tensor(paths, dtype=torch.int32)
        self.paths = paths.to(torch.int32).to(feat_map.device).contiguous()
        self.feat_map = feat_map
        if feat_map.numel() % masks.P or (f_off is None and feat_map.numel() != masks.P):
            raise ValueError(f'feat_map has {feat_map.numel()} elements, masks expect (a multiple of) {masks.P}')

    def __len__(self):
        return self.paths.numel()

    @property
    def shape(self):
        return (self.paths.numel(), self.masks.P)


class MaskedFcFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat_map, w, b, masks, paths, f_off):
        f = feat_map.reshape(-1)
        f = f if f.is_contiguous() else f.contiguous()
        wc = w if w.is_contiguous() else w.contiguous()
        Dout, P = wc.shape
        T = paths.numel()
        dev, st = lib.stream_args(f)
        wT = torch.empty((P, Dout), dtype=torch.float32, device=f.device)
        lib.call('mmft_transpose', wc, wT, Dout, P, dev, st)
        out = torch.empty((T, Dout), dtype=torch.float32, device=f.device)
        lib.call('mmft_masked_fc_fwd', masks.indptr, masks.cols, paths, f_off, T, f, wT, b, out, P, Dout, dev, st)
        ctx.masks, ctx.paths, ctx.fshape, ctx.f_off = masks, paths, feat_map.shape, f_off
        ctx.has_bias = b is not None
        ctx.save_for_backward(f, wc)
        return out

    @staticmethod
    def backward(ctx, gout):
        f, wc = ctx.saved_tensors
        Dout, P = wc.shape
        g = gout if gout.is_contiguous() else gout.contiguous()
        T = ctx.paths.numel()
        dev, st = lib.stream_args(f)
        B = f.numel() // P
        S = torch.zeros((B * P, Dout), dtype=torch.float32, device=f.device)
        lib.call('mmft_masked_fc_bwd_scatter', ctx.masks.indptr, ctx.masks.cols, ctx.paths, ctx.f_off, T, g, S, P, Dout,
                 dev, st)
        dw = torch.empty_like(wc)
        df = torch.empty(B * P, dtype=torch.float32, device=f.device)
        lib.call('mmft_masked_fc_bwd_finish', S, f, wc, dw, df, B, P, Dout, dev, st)
        db = ops.colsum(g) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return (df.reshape(ctx.fshape) if ctx.needs_input_grad[0] else None,
                dw if ctx.needs_input_grad[1] else None, db, None, None, None)


def masked_fc(pm, w, b):
    """fcn(mask_rows.to_dense() * feat_map) without the dense map."""
    ops._chk(pm.feat_map, 'feat_map')
    if w.shape[1] != pm.masks.P:
        raise ValueError(f'fcn expects {w.shape[1]} map cells, masks have {pm.masks.P}')
    if w.shape[0] % 4:
        raise ValueError('masked_fc: cnn_outdim must be a multiple of 4')
    return MaskedFcFn.apply(pm.feat_map, w, b, pm.masks, pm.paths, pm.f_off)


class MseFn(torch.autograd.Function):
    """nn.MSELoss() (mean)  (src/train.py:32,522) with the gradient produced by the same kernel."""

    @staticmethod
    def forward(ctx, pred, target):
        p = pred if pred.is_contiguous() else pred.contiguous()
        t = target if target.is_contiguous() else target.contiguous()
        ops._chk(p, 'pred'); ops._chk(t, 'target')
        if p.shape != t.shape or p.dim() != 1:
            raise ValueError('mse: 1-D tensors of equal length required')
        loss = torch.empty(1, dtype=torch.float32, device=p.device)
        grad = torch.empty_like(p)
        dev, st = lib.stream_args(p)
        lib.call('mmft_mse_fwd_bwd', p, t, p.numel(), loss, grad, dev, st)
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        (grad,) = ctx.saved_tensors
        return grad * gl, None


def mse_loss(pred, target):
    return MseFn.apply(pred, target)


class FlatAdam:
    """torch.optim.Adam(params, lr, weight_decay) semantics (src/train.py:431-435) on ONE flat fp32 buffer.

    Parameters are re-pointed at views of `flat_param`, their .grad at views of `flat_grad`, so a data-parallel
    step is one all-reduce of flat_grad followed by one fused kernel.  Parameters that never receive a
    gradient (fc_net_drive, fc_attn2: unused in forward, src/model.py:52-54) must be left out, as
    torch's Adam skips them.
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError('FlatAdam: no parameters')
        dev = self.params[0].device
        if dev.type != 'cuda':
            raise RuntimeError('FlatAdam runs on the GPU only')
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        sizes = [p.numel() for p in self.params]
        # 16-byte aligned slots so that every view stays vector-load friendly
        self.offsets, off = [], 0
        for n in sizes:
            self.offsets.append(off)
            off += (n + 3) // 4 * 4
        self.n = off
        self.flat_param = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.step_count = 0
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            # keep the parameter's logical shape AND its memory order (channels_last conv weights etc.)
            src = p.data
            view = torch.as_strided(self.flat_param, src.shape, src.stride(), o) if _dense(src) else None
            if view is None:
                src = src.contiguous()
                view = self.flat_param[o:o + n].view(src.shape)
            view.copy_(src)
            p.data = view
            p.grad = torch.as_strided(self.flat_grad, view.shape, view.stride(), o)

    def zero_grad(self):
        self.flat_grad.zero_()

    def step(self, gscale=1.0):
        self.step_count += 1
        b1, b2 = self.betas
        bc1 = 1.0 - b1 ** self.step_count
        bc2 = 1.0 - b2 ** self.step_count
        dev, st = lib.stream_args(self.flat_param)
        lib.call('mmft_adam_step', self.flat_param, self.flat_grad, self.m, self.v, self.n, float(self.lr), float(b1),
                 float(b2), float(self.eps), float(self.wd), float(bc1), float(bc2), float(gscale), dev, st)


def _dense(t):
    """non-overlapping and dense (any permutation of a contiguous block)."""
    if t.numel() == 0:
        return True
    sizes_strides = sorted(((st, sz) for sz, st in zip(t.shape, t.stride()) if sz > 1))
    expect = 1
    for st, sz in sizes_strides:
        if st != expect:
            return False
        expect *= sz
    return True
