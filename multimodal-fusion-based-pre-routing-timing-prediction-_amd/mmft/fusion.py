"""Fusion-head glue on the HIP path: sparse masked projection (K7), MSE (K10), flat fused Adam (K18).

The reference materialises `path_mask.to_dense() * feat_map` (T x P fp32, 88 MB at defaults) and feeds it
to `fcn` (src/train.py:500-501, src/model.py:271-272).  MaskedPathMap carries the same information as
CSR rows + the 1 x P feature map; PathModel recognises it and never builds the dense map.  A dense
path_map tensor (what an unmodified reference loop passes) is still accepted and goes through the dense
GEMM kernel.
"""
import numpy as np
import torch
from . import lib, ops, gradsink


def _csc_from_csr(indptr, cols, P):
    """Transposed masks: for every map cell the ascending list of path ids covering it."""
    indptr = np.asarray(indptr, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    rows = np.repeat(np.arange(indptr.shape[0] - 1, dtype=np.int64), np.diff(indptr))
    order = np.argsort(cols, kind='stable')
    cptr = np.zeros(P + 1, dtype=np.int64)
    np.cumsum(np.bincount(cols, minlength=P), out=cptr[1:])
    return cptr, rows[order]


RUN_BLOCK = 64          # cells per prefix block (mmft_masked_fc_prefix); 0 disables the run form
USE_RUNS = True         # MaskedFcFn.forward: prefix sums + runs when the masks offer them (False: one gather per cell)


def _runs_from_csr(indptr, cols, S):
    """(run_ptr[rows+1], run_start[R], run_len[R]) of a CSR with ascending columns per row; a run ends where the next
    column is not the successor, at a row end, and at every multiple of S."""
    n = cols.shape[0]
    rows = indptr.shape[0] - 1
    if n == 0:
        return np.zeros(rows + 1, dtype=np.int64), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    new = np.ones(n, dtype=bool)
    new[1:] = (cols[1:] != cols[:-1] + 1) | (cols[1:] % S == 0)
    new[indptr[:-1][np.diff(indptr) > 0]] = True
    starts = np.nonzero(new)[0]
    run_len = np.diff(np.append(starts, n))
    run_ptr = np.searchsorted(starts, indptr, side='left')
    return run_ptr.astype(np.int64), cols[starts], run_len


class PathMasks:
    """Device CSR (+ transposed CSC) of path masks, num_paths x P, 0/1  (src/verilog_parser_asap7.py:1302-1369).

    For B stacked designs the rows are the designs' paths one after another (global path ids) and the
    CSC has B*P cells (cell = b*P + p); columns stay in [0, P) because fcn's weight is shared."""

    def __init__(self, indptr, cols, P, device, B=1, row_design=None):
        ip = np.asarray(torch.as_tensor(indptr).cpu().numpy(), dtype=np.int64)
        cc = np.asarray(torch.as_tensor(cols).cpu().numpy(), dtype=np.int64)
        self.P, self.B = int(P), int(B)
        self.num_paths = ip.shape[0] - 1
        if cc.size and (cc.max() >= self.P or cc.min() < 0):
            raise ValueError('mask column outside the map')
        if ip[-1] >= 2 ** 31:
            raise ValueError('mask nnz exceeds int32')
        self.host_indptr, self.host_cols = ip, cc
        self.row_design = np.zeros(self.num_paths, dtype=np.int64) if row_design is None else np.asarray(row_design)
        self.indptr = torch.from_numpy(ip.astype(np.int32)).to(device)
        self.cols = torch.from_numpy(cc.astype(np.int32)).to(device)
        cell = cc + np.repeat(self.row_design, np.diff(ip)) * self.P
        cptr, cpaths = _csc_from_csr(ip, cell, self.B * self.P)
        self.csc_indptr = torch.from_numpy(cptr.astype(np.int32)).to(device)
        self.csc_paths = torch.from_numpy(cpaths.astype(np.int32)).to(device)
        # run-length form of the rows for the prefix-sum projection: runs of consecutive cells that do not cross a
        # block of RUN_BLOCK cells (box masks: ~70 runs for ~580 cells per path)
        self.run_block = RUN_BLOCK if self.P % RUN_BLOCK == 0 else 0
        if self.run_block:
            rp, rs, rl = _runs_from_csr(ip, cc, self.run_block)
            self.run_ptr = torch.from_numpy(rp.astype(np.int32)).to(device)
            self.run_start = torch.from_numpy(rs.astype(np.int32)).to(device)
            self.run_len = torch.from_numpy(rl.astype(np.int32)).to(device)
            self.num_runs = int(rs.shape[0])
            # run boundaries per cell (design-major) for the backward: +path at a run's last cell, -(path) - 1 at the
            # cell before a run that starts inside a block
            rows_of_run = np.repeat(np.arange(self.num_paths), np.diff(rp))
            base = self.row_design[rows_of_run] * self.P
            end_cell = base + rs + rl - 1
            inner = (rs % self.run_block) != 0
            cells = np.concatenate([end_cell, (base + rs - 1)[inner]])
            codes = np.concatenate([rows_of_run, -rows_of_run[inner] - 1])
            order = np.argsort(cells, kind='stable')
            bptr = np.zeros(self.B * self.P + 1, dtype=np.int64)
            np.cumsum(np.bincount(cells, minlength=self.B * self.P), out=bptr[1:])
            self.bnd_ptr = torch.from_numpy(bptr.astype(np.int32)).to(device)
            self.bnd_code = torch.from_numpy(codes[order].astype(np.int32)).to(device)

    @staticmethod
    def from_sparse_coo(path_masks, device):
        """From the reference's own container: the sparse COO `(num_paths, P)` tensor of ones that
        src/verilog_parser_asap7.py:1353-1368 builds and the training loop indexes (src/train.py:500).  Zero entries are
        dropped, duplicates merged; rows keep their order, columns ascend inside a row."""
        sp = path_masks.coalesce().cpu()
        idx, val = sp.indices().numpy(), sp.values().numpy()
        keep = val != 0
        rows, cols = idx[0][keep].astype(np.int64), idx[1][keep].astype(np.int64)      # coalesce() sorts by (row, column)
        n, P = int(sp.shape[0]), int(sp.shape[1])
        indptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.bincount(rows, minlength=n), out=indptr[1:])
        return PathMasks(indptr, cols, P, device)

    @staticmethod
    def batch(masks):
        P, dev = masks[0].P, masks[0].indptr.device
        ip, cols, rd, off = [np.zeros(1, dtype=np.int64)], [], [], 0
        for b, m in enumerate(masks):
            if m.P != P or m.B != 1:
                raise ValueError('PathMasks.batch: single-design masks with one map size expected')
            ip.append(m.host_indptr[1:] + off)
            cols.append(m.host_cols)
            rd.append(np.full(m.num_paths, b, dtype=np.int64))
            off += int(m.host_cols.shape[0])
        return PathMasks(np.concatenate(ip), np.concatenate(cols), P, dev, B=len(masks), row_design=np.concatenate(rd))


def batch_links(paths, num_paths):
    """first[q] = first batch row holding path q (-1: none); next[t] = next row with the same path (-1: none)."""
    paths = np.asarray(paths, dtype=np.int64)
    T = paths.shape[0]
    first = np.full(num_paths, -1, dtype=np.int32)
    nxt = np.full(T, -1, dtype=np.int32)
    if T:
        order = np.argsort(paths, kind='stable')
        sp = paths[order]
        same = sp[1:] == sp[:-1]
        nxt[order[:-1][same]] = order[1:][same]
        starts = np.concatenate([[True], ~same])
        first[sp[starts]] = order[starts]
    return first, nxt


class MaskedPathMap:
    """Lazy `index_select(path_masks, 0, paths).to_dense() * feat_map` (src/train.py:500-501): rows = `paths`
    (global path ids), values from feat_map [B, P]; never materialised."""

    def __init__(self, masks, paths, feat_map, f_off=None, first=None, next_=None):
        self.masks = masks
        dev = feat_map.device
        if torch.is_tensor(paths) and paths.dtype == torch.int32 and paths.device == dev and paths.is_contiguous():
            self.paths = paths                                  # the usual case: a slice of the step's packed selection
        else:
            if not torch.is_tensor(paths):
                paths = torch.as_tensor(np.asarray(paths), dtype=torch.int32)
            self.paths = paths.to(torch.int32).to(dev).contiguous()
        if f_off is None and masks.B > 1:
            # feature-map offset of each row's design, looked up on the device (no host round trip per level call)
            rd = masks.__dict__.get('_row_off_dev')
            if rd is None or rd.device != dev:
                rd = masks.__dict__['_row_off_dev'] = torch.from_numpy((masks.row_design * masks.P).astype(np.int32)).to(dev)
            f_off = rd[self.paths.long()]
        self.f_off, self._first, self._next = f_off, first, next_
        self.feat_map = feat_map
        if feat_map.numel() != masks.B * masks.P:
            raise ValueError(f'feat_map has {feat_map.numel()} elements, masks expect {masks.B} x {masks.P}')

    def _links(self):
        """first / next chains over the rows that share a path (only the backward kernels read them).  Small ad-hoc
        batches derive them on the host - one device-to-host copy of the row list, paid at the first backward use."""
        if self._first is None:
            f_h, n_h = batch_links(self.paths.detach().cpu().numpy(), self.masks.num_paths)
            dev = self.paths.device
            self._first, self._next = torch.from_numpy(f_h).to(dev), torch.from_numpy(n_h).to(dev)
        return self._first, self._next

    @property
    def first(self):
        return self._links()[0]

    @property
    def next(self):
        return self._links()[1]

    def __len__(self):
        return self.paths.numel()

    @property
    def shape(self):
        return (self.paths.numel(), self.masks.P)


# Per-step cache of what the masked projection derives from (fcn.weight, feat_map) alone: the transposed weight and the
# block-prefix table.  The per-level drop-in loop calls fcn once per level with the SAME weight and feature map
# (src/train.py:500-503); without the cache every level re-transposes 8 MB and rebuilds the 67 MB prefix table.
# Valid while the same tensor objects are alive and unmodified (weak references + versions + the parameter epoch that the
# fused optimizer bumps, since it rewrites parameters through raw pointers) and the work was issued on the same stream.
_FC_CACHE = {}


def _fc_cached(kind, deps, build):
    import weakref
    if torch.cuda.is_current_stream_capturing():
        return build()          # a captured step must contain its own transpose / prefix launches: never reuse, never keep
    key = tuple((id(t), t._version) for t in deps) + (gradsink.param_epoch(), torch.cuda.current_stream().cuda_stream)
    e = _FC_CACHE.get(kind)
    if e is not None and e[0] == key and all(r() is t for r, t in zip(e[1], deps)):
        return e[2]
    val = build()
    _FC_CACHE[kind] = (key, [weakref.ref(t) for t in deps], val)
    return val


def masked_fc_prefix(feat_map, w, masks):
    """Block-prefix table GP[b * P + c] = sum of f * w^T over the cells <= c of c's RUN_BLOCK block (one launch pair per step and
    feature map, cached), or None when the masks offer no run form.  Shared by MaskedFcFn and the one-call level head."""
    f = feat_map.reshape(-1)
    if not (USE_RUNS and masks.run_block and f.numel() == masks.B * masks.P and f.is_contiguous() and w.is_contiguous()):
        return None
    Dout, P = w.shape
    dev, st = lib.stream_args(f)

    def _wT():
        t = torch.empty((P, Dout), dtype=torch.float32, device=f.device)
        lib.call('mmft_transpose', w, t, Dout, P, dev, st)
        return t
    wT = _fc_cached('wT', (w,), _wT)

    def _gp():
        t = torch.empty((masks.B * P, Dout), dtype=torch.float32, device=f.device)
        lib.call('mmft_masked_fc_prefix', f, wT, t, masks.B, P, Dout, masks.run_block, dev, st)
        return t
    return _fc_cached('GP', (w, feat_map), _gp)


class HeadLevelCtx:
    """mmft_head_level_fwd with everything that is the same for all level calls of a step resolved ONCE (the prefix table,
    raw pointers, sizes): a level call then costs one output allocation and one C call.  `ok` is False when the
    configuration is outside that entry point (no run form of the masks, a head that is not Linear-ReLU-Linear)."""

    def __init__(self, h, feat_map, masks, fcn, mlp_fuse, Da):
        self.ok = False
        mods = list(mlp_fuse.layers)
        if len(mods) != 3 or not isinstance(mods[0], torch.nn.Linear) or not isinstance(mods[2], torch.nn.Linear) or \
                getattr(mlp_fuse, 'negative_slope', 0) != 0 or not isinstance(fcn, torch.nn.Linear):
            return
        w, b = fcn.weight.detach(), fcn.bias
        GP = masked_fc_prefix(feat_map.detach(), w, masks)
        if GP is None:
            return
        l1, l2 = mods[0], mods[2]
        Dh, Dc = h.shape[1], w.shape[0]
        if l1.in_features != Dh + Dc + Da or Dh % 4 or Dc % 4 or Da % 4:
            return
        self.keep = (h, GP, masks, w, b, l1, l2)               # the tensors behind the raw pointers
        self.h, self.Dh, self.Dc, self.Da, self.H1, self.nout = h, Dh, Dc, Da, l1.out_features, l2.out_features
        ptr = lambda t: t.data_ptr() if t is not None else None
        self.head = (h.data_ptr(), h.stride(0))
        self.mid = (masks.run_ptr.data_ptr(), masks.run_start.data_ptr(), masks.run_len.data_ptr())
        self.tail_a = (GP.data_ptr(), ptr(b), Dc, masks.run_block)
        self.tail_b = (Da, l1.weight.data_ptr(), ptr(l1.bias), self.H1, l2.weight.data_ptr(), ptr(l2.bias), self.nout)
        self.dev, self.st = lib.stream_args(h)
        self.fn = lib.load().mmft_head_level_fwd
        self.ws, self.ws_rows = None, 0
        self.ok = True

    def run(self, tix, paths, f_off, alpha_row):
        T = tix.numel()
        if T > self.ws_rows:
            self.ws_rows = max(T, 2 * self.ws_rows, 1024)
            self.ws = lib.workspace(self.h.device, lib.query('mmft_head_level_workspace_bytes', self.ws_rows, self.Dh, self.Dc,
                                                             self.Da, self.H1))
        out = torch.empty((T, self.nout), dtype=torch.float32, device=self.h.device)
        rc = self.fn(*self.head, tix.data_ptr(), T, self.Dh, *self.mid, paths.data_ptr(),
                     f_off.data_ptr() if f_off is not None else None, *self.tail_a, alpha_row.data_ptr(), *self.tail_b,
                     self.ws.data_ptr(), self.ws.numel() * 4, out.data_ptr(), self.dev, self.st)
        if rc != 0:
            raise RuntimeError(f"mmft_head_level_fwd failed ({rc}): {lib.load().mmft_last_error().decode()}")
        return out.squeeze(-1)


class MaskedFcFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat_map, w, b, pm):
        f = feat_map.reshape(-1)
        f = f if f.is_contiguous() else f.contiguous()
        wc = w if w.is_contiguous() else w.contiguous()
        Dout, P = wc.shape
        T = pm.paths.numel()
        dev, st = lib.stream_args(f)

        def _wT():
            t = torch.empty((P, Dout), dtype=torch.float32, device=f.device)
            lib.call('mmft_transpose', wc, t, Dout, P, dev, st)
            return t
        wT = _fc_cached('wT', (w,), _wT)
        out = torch.empty((T, Dout), dtype=torch.float32, device=f.device)
        m = pm.masks
        if USE_RUNS and m.run_block and f.numel() == m.B * P:
            # block-prefix sums of f * wT once per step, then two reads per run of a path instead of one per cell
            def _gp():
                t = torch.empty((m.B * P, Dout), dtype=torch.float32, device=f.device)
                lib.call('mmft_masked_fc_prefix', f, wT, t, m.B, P, Dout, m.run_block, dev, st)
                return t
            GP = _fc_cached('GP', (w, feat_map), _gp)
            if lib.PROF_ON:         # runs of the sampled paths: the static average per path (the ids live on the device)
                # HBM side: every prefix row at most once (the two reads per run mostly hit L2 / MALL), the run lists, the output
                R = T * m.num_runs / max(m.num_paths, 1)
                lib.prof_hint(2.0 * R * Dout, min(2.0 * R, float(m.B * P)) * Dout * 4 + R * 8 + T * (Dout * 4 + 12))
            lib.call('mmft_masked_fc_fwd_runs', m.run_ptr, m.run_start, m.run_len, pm.paths, pm.f_off, T, GP, b, out, Dout,
                     m.run_block, dev, st)
        else:
            lib.call('mmft_masked_fc_fwd', m.indptr, m.cols, pm.paths, pm.f_off, T, f, wT, b, out, P, Dout, dev, st)
        ctx.pm, ctx.fshape = pm, feat_map.shape
        ctx.has_bias = b is not None
        ctx.sinks = (gradsink.of(w) if wc is w else None, gradsink.of(b))
        ctx.save_for_backward(f, wT)
        return out

    @staticmethod
    def backward(ctx, gout):
        f, wT = ctx.saved_tensors
        P, Dout = wT.shape
        pm = ctx.pm
        g = ops.strided_rows(gout)                       # a column slice of the concatenated head input: read in place
        dev, st = lib.stream_args(f)
        B = pm.masks.B
        dwT = torch.empty_like(wT)
        df = torch.empty(B * P, dtype=torch.float32, device=f.device)
        groups = Dout // 4
        if USE_RUNS and pm.masks.run_block and groups <= 64 and groups & (groups - 1) == 0 and \
                pm.masks.run_block * Dout * 4 <= 65536 and pm.masks.run_block % (512 // groups) == 0:
            ws = lib.workspace(f.device, lib.query('mmft_masked_fc_bwd_runs_workspace_bytes', B, P, Dout))
            if lib.PROF_ON:
                # boundary entries of the sampled paths (static average per path): one gout row each (HBM side: each of the T
                # rows at most once); per design the wT block is read and the dwT slab written once, f / df / the boundary
                # pointers once per cell
                E = g.shape[0] * pm.masks.bnd_code.numel() / max(pm.masks.num_paths, 1)
                lib.prof_hint(E * Dout + 4.0 * B * P * Dout,
                              min(E, float(g.shape[0])) * Dout * 4 + E * 4 + 2.0 * B * P * Dout * 4 + 12.0 * B * P)
            lib.call('mmft_masked_fc_bwd_runs', pm.masks.bnd_ptr, pm.masks.bnd_code, pm.first, pm.next, g, g.stride(0), f, wT, dwT, df,
                     B, P, Dout, pm.masks.run_block, ws, ws.numel() * 4, dev, st)
        else:
            ws = lib.workspace(f.device, B * P * Dout * 4 if B > 1 else 0)
            lib.call('mmft_masked_fc_bwd', pm.masks.csc_indptr, pm.masks.csc_paths, pm.first, pm.next, g, g.stride(0), f, wT, dwT, df,
                     B, P, Dout, ws, ws.numel() * 4, dev, st)
        dw = db = None
        if ctx.needs_input_grad[1]:
            def _dw(out):
                out = torch.empty((Dout, P), dtype=torch.float32, device=f.device) if out is None else out
                lib.call('mmft_transpose', dwT, out, P, Dout, dev, st)
                return out
            dw = gradsink.deliver(ctx.sinks[0], _dw)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = gradsink.deliver(ctx.sinks[1], lambda out: ops.colsum(g, out=out))
        return (df.reshape(ctx.fshape) if ctx.needs_input_grad[0] else None, dw, db, None)


def masked_fc(pm, w, b):
    """fcn(mask_rows.to_dense() * feat_map) without the dense map."""
    ops._chk(pm.feat_map, 'feat_map')
    if w.shape[1] != pm.masks.P:
        raise ValueError(f'fcn expects {w.shape[1]} map cells, masks have {pm.masks.P}')
    if w.shape[0] % 4:
        raise ValueError('masked_fc: cnn_outdim must be a multiple of 4')
    return MaskedFcFn.apply(pm.feat_map, w, b, pm)


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
        return (grad if is_unit_grad(gl) else grad * gl), None


_UNIT = {}


def unit_grad(device):
    """A cached scalar 1.0 on `device` to seed backward() with (loss.backward(unit_grad(dev))): autograd then launches no
    ones_like fill, and the loss Functions recognise it and skip the multiplication by the upstream gradient."""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    t = _UNIT.get(key)
    if t is None:
        t = _UNIT[key] = torch.ones((), dtype=torch.float32, device=dev)
    return t


def is_unit_grad(g):
    return any(g.data_ptr() == t.data_ptr() for t in _UNIT.values())


class MseGatherFn(torch.autograd.Function):
    """MSE(pred, table[idx]) with the gather inside the loss kernel (arrival_time[target_list], src/train.py:519-522)."""

    @staticmethod
    def forward(ctx, pred, table, idx):
        p = pred if pred.is_contiguous() else pred.contiguous()
        ops._chk(p, 'pred'); ops._chk(table, 'table')
        if p.dim() != 1 or idx.dtype != torch.int32 or idx.numel() != p.numel() or not idx.is_contiguous() or table.dim() > 2 or \
                (table.dim() == 2 and table.shape[1] != 1):
            raise ValueError('mse_gather: pred (T,), table (N,) or (N, 1), idx int32 (T,)')
        loss = torch.empty(1, dtype=torch.float32, device=p.device)
        grad = torch.empty_like(p)
        dev, st = lib.stream_args(p)
        lib.call('mmft_mse_gather_fwd_bwd', p, table, table.stride(0), idx, p.numel(), loss, grad, dev, st)
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        (grad,) = ctx.saved_tensors
        return (grad if is_unit_grad(gl) else grad * gl), None, None


def mse_loss(pred, target):
    return MseFn.apply(pred, target)


def mse_loss_gather(pred, table, idx):
    return MseGatherFn.apply(pred, table, idx)


class CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss() (mean) of the classification task (src/train.py:32,516-518), gradient from the same kernel."""

    @staticmethod
    def forward(ctx, logits, labels):
        z = logits if logits.is_contiguous() else logits.contiguous()
        ops._chk(z, 'logits')
        ops._chk(labels, 'labels', torch.int64)
        if z.dim() != 2 or labels.dim() != 1 or labels.numel() != z.shape[0] or not labels.is_contiguous():
            raise ValueError('cross_entropy: logits [T, C] and int64 labels [T] required')
        loss = torch.empty(1, dtype=torch.float32, device=z.device)
        grad = torch.empty_like(z)
        dev, st = lib.stream_args(z)
        lib.call('mmft_cross_entropy_fwd_bwd', z, labels, z.shape[0], z.shape[1], loss, grad, None, dev, st)
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        (grad,) = ctx.saved_tensors
        return grad * gl, None


def cross_entropy_loss(logits, labels):
    return CrossEntropyFn.apply(logits, labels)


def cls_eval_sums(logits, labels):
    """fp64[6]: n, sum of per-row losses, tp, fp, tn, fn (argmax prediction; positive = class != 0)."""
    z = logits if logits.is_contiguous() else logits.contiguous()
    ops._chk(z, 'logits'); ops._chk(labels, 'labels', torch.int64)
    out = torch.empty(6, dtype=torch.float64, device=z.device)
    dev, st = lib.stream_args(z)
    lib.call('mmft_cross_entropy_fwd_bwd', z, labels, z.shape[0], z.shape[1], None, None, out, dev, st)
    return out


class FlatAdam:
    """torch.optim.Adam(params, lr, weight_decay) semantics (src/train.py:431-435) on ONE flat fp32 buffer.

    Parameters are re-pointed at views of `flat_param`, their .grad at views of `flat_grad`, so a data-parallel
    step is one all-reduce of flat_grad followed by one fused kernel.  Parameters that never receive a
    gradient (fc_net_drive, fc_attn2: unused in forward, src/model.py:52-54) must be left out, as
    torch's Adam skips them.
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, buckets=None, zero_after_step=False):
        """buckets: optional [(name, [params...]), ...] partition of `params`; the flat buffers are laid out bucket
        after bucket, so each bucket is one contiguous range (one all-reduce + one Adam launch of its own, issued as
        soon as its gradients exist: mmft.dist.GradReducer).  Default: one bucket holding everything."""
        params = [p for p in params]
        if buckets is None:
            buckets = [('all', params)]
        self.params = [p for _, ps in buckets for p in ps]
        if not self.params:
            raise ValueError('FlatAdam: no parameters')
        if len(self.params) != len(params) or {id(p) for p in self.params} != {id(p) for p in params}:
            raise ValueError('FlatAdam: the buckets must partition the parameter list')
        dev = self.params[0].device
        if dev.type != 'cuda':
            raise RuntimeError('FlatAdam runs on the GPU only')
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        # zero_after_step: the Adam kernel clears the gradients it has consumed, so zero_grad() launches no fill (the .grad
        # views then read zero after step(); keep False where the caller inspects gradients after the step)
        self.zero_after_step = bool(zero_after_step)
        self._cleared = [False] * len(buckets)
        # 16-byte aligned slots so that every view stays vector-load friendly
        self.offsets, off = [], 0
        self.bucket_ranges, self.bucket_params = [], []
        for name, ps in buckets:
            lo = off
            for p in ps:
                self.offsets.append(off)
                off += (p.numel() + 3) // 4 * 4
            self.bucket_ranges.append((name, lo, off))
            self.bucket_params.append(list(ps))
        self.n = off
        self.flat_param = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.step_count = 0                # host mirror of state[:, 0]
        # per bucket: [optimizer steps taken, arrival ticket] - the step counter lives on the device
        self.state = torch.zeros((len(buckets), 2), dtype=torch.int32, device=dev)
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
            gradsink.attach(p, p.grad)          # backward kernels store into the flat buffer directly

    def zero_grad(self):
        if not all(self._cleared):
            self.flat_grad.zero_()
        self._cleared = [False] * len(self._cleared)      # the backward pass is about to write them
        gradsink.new_step()

    def step_bucket(self, i, gscale=1.0):
        """Adam over bucket i only (its own device-side step counter); the caller keeps `step_count` in step."""
        _, lo, hi = self.bucket_ranges[i]
        if hi == lo:
            return
        b1, b2 = self.betas
        dev, st = lib.stream_args(self.flat_param)
        lib.call('mmft_adam_step_counted', self.flat_param[lo:hi], self.flat_grad[lo:hi], self.m[lo:hi], self.v[lo:hi],
                 hi - lo, self.state[i], float(self.lr), float(b1), float(b2), float(self.eps), float(self.wd),
                 float(gscale), int(self.zero_after_step), dev, st)
        self._cleared[i] = self.zero_after_step
        gradsink.params_changed()        # raw-pointer update: Tensor._version does not see it

    def step(self, gscale=1.0):
        """One Adam step.  The step counter lives in device memory (`self.state[:, 0]`) and the kernel derives both
        bias corrections from it, so the same launch is valid eagerly, inside a HIP-graph capture and on every replay -
        the host uploads nothing per step and may run any number of steps ahead of the device."""
        self.step_count += 1
        for i in range(len(self.bucket_ranges)):
            self.step_bucket(i, gscale)

    def step_captured(self, gscale=1.0):
        """The same launch while a HIP graph is being captured: recording it takes no step."""
        self.step(gscale)
        self.step_count -= 1

    def note_replay(self):
        """Host mirror of the device counter after a graph replay that contains the Adam launch."""
        self.step_count += 1

    def device_step_count(self):
        return int(self.state[0, 0].item())


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
