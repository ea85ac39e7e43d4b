"""Tensor-level wrappers over the C ABI (no autograd here; see mmft.functional).

Every function checks on the host that operand shapes match what the kernels and their grids assume
before launching, and raises instead of falling back when the inputs are not device fp32 tensors.
"""
import torch
from . import lib

EPI_STORE, EPI_ACCUM, EPI_ADD_ACT = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
POOL_MAX, POOL_AVG = 0, 1
# rows with more edges than this get a whole workgroup in the level kernels (graph.hip): a cell row's in-edges (forward
# gather) / a node's out-edges (reverse pull) are a serial chain of dependent loads in one thread group otherwise
import os as _os
# rows with more in- / out-edges than this get a whole workgroup (MMFT_HEAVY_IN / MMFT_HEAVY_OUT override, read at import)
PAIR_HEAVY_IN = int(_os.environ.get('MMFT_HEAVY_IN', '16'))
PAIR_HEAVY_OUT = int(_os.environ.get('MMFT_HEAVY_OUT', '16'))


def _chk(t, name, dtype=torch.float32):
    if not torch.is_tensor(t):
        raise TypeError(f'{name}: expected a tensor')
    if not t.is_cuda:
        raise RuntimeError(f'{name}: the mmft hot path runs on the GPU only (got a {t.device} tensor); '
                           f'there is no CPU fallback')
    if t.dtype != dtype:
        raise TypeError(f'{name}: expected {dtype}, got {t.dtype}')
    return t


def _rows2d(t, name):
    _chk(t, name)
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f'{name}: expected a 2-D tensor with unit inner stride, got shape {tuple(t.shape)} '
                         f'strides {t.stride()}')
    return t


def _hid2d(t, name):
    """A [rows, 256] hidden tensor of the level MLP: fp32, or bf16 where the bf16-mode kernels store it so (sweep.HIDDEN_BF16)."""
    _chk(t, name, t.dtype if torch.is_tensor(t) and t.dtype == torch.bfloat16 else torch.float32)
    if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) % 4:
        raise ValueError(f'{name}: expected a 2-D tensor with unit inner stride and a row pitch that is a multiple of 4')
    return t


def _same_hid_dtype(*ts):
    d = {t.dtype for t in ts if t is not None}
    if len(d) > 1:
        raise TypeError('the hidden tensors of one call (mask, hid_out) must share their dtype')
    return int(bool(d) and d.pop() == torch.bfloat16)


def _launch(name, args):
    """lib.call, returning the argument list with tensors replaced by their addresses: a caller whose tensors are persistent
    (the sweep's buffers and static tables) keeps the list and repeats the launch with `relaunch`, skipping this wrapper's
    checks - the per-level kernels of the drop-in loop are issued eagerly, 63 times per step, and the checks cost more
    host time than the launch."""
    lib.call(name, *args)
    return [a.data_ptr() if torch.is_tensor(a) else a for a in args]


def relaunch(name, raw, dev, stream):
    """Repeat a launch recorded by `_launch` on the given stream (the last two arguments of every entry point)."""
    lib.call(name, *raw[:-2], dev, stream)


def strided_rows(t):
    """t itself when the kernels can read it as rows of a wider tensor (2-D, unit inner stride, rows 16-byte aligned - a column
    slice of a concatenation's gradient), else a contiguous copy."""
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.storage_offset() % 4 == 0:
        return t
    return t.contiguous()


def _idx(t, name, n=None):
    if t is None:
        return None
    _chk(t, name, torch.int32)
    if t.dim() != 1 or not t.is_contiguous():
        raise ValueError(f'{name}: expected a contiguous 1-D int32 tensor')
    if n is not None and t.numel() != n:
        raise ValueError(f'{name}: expected {n} entries, got {t.numel()}')
    return t


def act_code(slope_or_none):
    if slope_or_none is None:
        return ACT_NONE, 0.0
    if slope_or_none == 0:
        return ACT_RELU, 0.0
    return ACT_LEAKY, float(slope_or_none)


# ------------------------------------------------------------------------------------------ dense
def linear_fwd(x, w, bias, y=None, xidx=None, yidx=None, M=None, epi=EPI_STORE, act=ACT_NONE, slope=0.0):
    """y[yidx] = epi(x[xidx] @ w.T + bias). x:(R,K) w:(N,K) ; returns y."""
    _rows2d(x, 'x'); _rows2d(w, 'w')
    N, K = w.shape
    if x.shape[1] != K:
        raise ValueError(f'linear_fwd: x has {x.shape[1]} features, weight expects {K}')
    if M is None:
        M = xidx.numel() if xidx is not None else x.shape[0]
    _idx(xidx, 'xidx', M); _idx(yidx, 'yidx', M)
    if xidx is None and M > x.shape[0]:
        raise ValueError('linear_fwd: M exceeds rows of x')
    if bias is not None:
        _chk(bias, 'bias')
        if bias.numel() != N or not bias.is_contiguous():
            raise ValueError('linear_fwd: bias shape')
    if y is None:
        if yidx is not None or epi != EPI_STORE:
            raise ValueError('linear_fwd: output buffer required with yidx / accumulate modes')
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    _rows2d(y, 'y')
    if y.shape[1] < N or (yidx is None and y.shape[0] < M):
        raise ValueError('linear_fwd: output too small')
    dev, st = lib.stream_args(x)
    lib.call('mmft_linear_fwd', x, xidx, x.stride(0), w, w.stride(0), bias, y, yidx, y.stride(0), M, N, K,
             epi, act, float(slope), dev, st)
    return y


def linear_dgrad(g, w, dx=None, gidx=None, dxidx=None, M=None, mask=None, maskidx=None, epi=EPI_STORE):
    """dx[dxidx] (+)= g[gidx] @ w ; w:(K=out, N=in). Optional mask: keep where mask[maskidx] > 0."""
    _rows2d(g, 'g'); _rows2d(w, 'w')
    K, N = w.shape
    if g.shape[1] != K:
        raise ValueError(f'linear_dgrad: g has {g.shape[1]} columns, weight has {K} rows')
    if M is None:
        M = gidx.numel() if gidx is not None else g.shape[0]
    _idx(gidx, 'gidx', M); _idx(dxidx, 'dxidx', M); _idx(maskidx, 'maskidx', M)
    if dx is None:
        if dxidx is not None or epi != EPI_STORE:
            raise ValueError('linear_dgrad: output buffer required')
        dx = torch.empty((M, N), dtype=torch.float32, device=g.device)
    _rows2d(dx, 'dx')
    if dx.shape[1] < N or (dxidx is None and dx.shape[0] < M):
        raise ValueError('linear_dgrad: output too small')
    ldmask = 0
    if mask is not None:
        _rows2d(mask, 'mask')
        if mask.shape[1] < N or (maskidx is None and mask.shape[0] < M):
            raise ValueError('linear_dgrad: mask too small')
        ldmask = mask.stride(0)
    dev, st = lib.stream_args(g)
    lib.call('mmft_linear_dgrad', g, gidx, g.stride(0), w, w.stride(0), dx, dxidx, dx.stride(0), M, N, K,
             mask, maskidx, ldmask, epi, dev, st)
    return dx


ROWS_OUTER = True       # bf16 mode: mmft_rows_outer_bf16 for the fc_cell_neigh weight gradients (False: the generic engine)


def linear_wgrad(g, x, dw=None, gidx=None, xidx=None, rows=None, accumulate=False, db=None, with_bias=False):
    """dw[o][i] (+)= sum_r g[gidx[r]][o] * x[xidx[r]][i]; with_bias (or db given): also db[o] (+)= sum_r g[gidx[r]][o]
    from the same pass, and the pair (dw, db) is returned."""
    g16, x16 = g.dtype == torch.bfloat16, x.dtype == torch.bfloat16
    (_hid2d if g16 else _rows2d)(g, 'g'); (_hid2d if x16 else _rows2d)(x, 'x')
    out, inn = g.shape[1], x.shape[1]
    if (g16 or x16) and not (ROWS_OUTER and gidx is None and xidx is None and lib.get_math_mode() == 'bf16'
                             and lib.query('mmft_rows_outer_supported', out, inn)):
        # a bf16-stored hidden operand outside the row-contraction kernel's shapes: widen it (small / unusual cases only)
        g, x = (g.float() if g16 else g), (x.float() if x16 else x)
        g16 = x16 = False
    if rows is None:
        rows = gidx.numel() if gidx is not None else g.shape[0]
    _idx(gidx, 'gidx', rows); _idx(xidx, 'xidx', rows)
    if gidx is None and g.shape[0] < rows or xidx is None and x.shape[0] < rows:
        raise ValueError('linear_wgrad: fewer rows than requested')
    if dw is None:
        dw = torch.empty((out, inn), dtype=torch.float32, device=g.device)
        accumulate = False
    _rows2d(dw, 'dw')
    if tuple(dw.shape) != (out, inn):
        raise ValueError(f'linear_wgrad: dw shape {tuple(dw.shape)} != {(out, inn)}')
    dev, st = lib.stream_args(g)
    want_db = with_bias or db is not None
    if want_db:
        if db is None:
            db = torch.empty(out, dtype=torch.float32, device=g.device)
        _chk(db, 'db')
        if db.numel() != out or not db.is_contiguous():
            raise ValueError('linear_wgrad: db shape')
    if ROWS_OUTER and gidx is None and xidx is None and (rows >= 4096 or g16 or x16) and dw.is_contiguous() and g.stride(0) % 4 == 0 and \
            x.stride(0) % 4 == 0 and lib.get_math_mode() == 'bf16' and lib.query('mmft_rows_outer_supported', out, inn):
        # bf16 mode, plain row ranges, the 128 x 256 / 256 x 128 products of fc_cell_neigh: transposed LDS reads instead of
        # the engine's in-register transposes
        ws = lib.workspace(g.device, lib.query('mmft_rows_outer_workspace_bytes', rows, out, inn))
        lib.call('mmft_rows_outer_bf16', g, g.stride(0), x, x.stride(0), dw, db if want_db else None, rows, out, inn, int(accumulate),
                 ws, ws.numel() * 4, int(g16), int(x16), dev, st)
        return (dw, db) if want_db else dw
    if want_db:
        ws = lib.workspace(g.device, lib.query('mmft_linear_wgrad_bias_workspace_bytes', rows, out, inn))
        lib.call('mmft_linear_wgrad_bias', g, gidx, g.stride(0), x, xidx, x.stride(0), dw, dw.stride(0), db, rows, out,
                 inn, int(accumulate), ws, ws.numel() * 4, dev, st)
        return dw, db
    ws = lib.workspace(g.device, lib.query('mmft_linear_wgrad_workspace_bytes', rows, out, inn))
    lib.call('mmft_linear_wgrad', g, gidx, g.stride(0), x, xidx, x.stride(0), dw, dw.stride(0), rows, out, inn,
             int(accumulate), ws, ws.numel() * 4, dev, st)
    return dw


def first_layer_grads_fusable(fin, HD, D2):
    return fin <= 48 and HD == 256 and D2 == 128


def mlp2_first_layer_grads(g, hid, x, rows, w2, dw1=None, db1=None):
    """(dW1, db1) of a Linear-ReLU-Linear MLP from the gradient g of its output, over the node rows `rows`
    (int32 tensor | (start, n) | None = all): dH = (g w2) * (hid > 0) stays in registers (mmft_mlp2_first_layer_grads)."""
    _rows2d(g, 'g'); _rows2d(hid, 'hid'); _rows2d(x, 'x'); _rows2d(w2, 'w2')
    D2, HD = w2.shape
    fin = x.shape[1]
    if g.shape[1] != D2 or hid.shape[1] != HD or g.shape[0] != hid.shape[0] or g.shape[0] != x.shape[0]:
        raise ValueError('mlp2_first_layer_grads: shapes')
    idx, row0, n = _rowspec(rows, g.shape[0], 'rows')
    dw1 = torch.empty((HD, fin), dtype=torch.float32, device=g.device) if dw1 is None else dw1
    db1 = torch.empty(HD, dtype=torch.float32, device=g.device) if db1 is None else db1
    _chk(dw1, 'dw1'); _chk(db1, 'db1')
    if tuple(dw1.shape) != (HD, fin) or not dw1.is_contiguous() or db1.numel() != HD or not db1.is_contiguous():
        raise ValueError('mlp2_first_layer_grads: output shapes')
    ws = lib.workspace(g.device, lib.query('mmft_mlp2_first_layer_grads_workspace_bytes', fin, HD))
    dev, st = lib.stream_args(g)
    lib.call('mmft_mlp2_first_layer_grads', g, g.stride(0), hid, hid.stride(0), x, x.stride(0), idx, row0, n, w2,
             w2.stride(0), dw1, db1, fin, HD, D2, 0, ws, ws.numel() * 4, dev, st)
    return dw1, db1


def mlp2_fusable(K1, HD, D2):
    return (K1, HD, D2) == (128, 256, 128)


def mlp2_rows(x1, rows, w1, b1, w2, b2, out, kmajor=False, mask=None, hid_out=None, add_act=False, relu_out=False,
              active=None):
    """Fused Linear-ReLU-Linear over the gathered rows `rows` (see mmft_mlp2_rows in include/mmft.h)."""
    _rows2d(x1, 'x1'); _rows2d(w1, 'w1'); _rows2d(w2, 'w2'); _rows2d(out, 'out'); _idx(rows, 'rows')
    if kmajor:
        K1, HD = w1.shape
        HD2, D2 = w2.shape
    else:
        HD, K1 = w1.shape
        D2, HD2 = w2.shape
    if HD2 != HD or x1.shape[1] != K1 or out.shape[1] != D2:
        raise ValueError('mlp2_rows: inconsistent widths')
    for t, nm, w in ((mask, 'mask', HD), (hid_out, 'hid_out', HD)):
        if t is not None:
            _rows2d(t, nm)
            if t.shape[1] != w or t.shape[0] != x1.shape[0]:
                raise ValueError(f'mlp2_rows: {nm} must be [rows of x1, {w}]')
    if out.shape[0] != x1.shape[0]:
        raise ValueError('mlp2_rows: out must have one row per row of x1 (scatter by node id)')
    dev, st = lib.stream_args(x1)
    lib.call('mmft_mlp2_rows', x1, x1.stride(0), rows, rows.numel(), w1, w1.stride(0), b1, w2, w2.stride(0), b2,
             int(kmajor), mask, mask.stride(0) if mask is not None else 0, hid_out,
             hid_out.stride(0) if hid_out is not None else 0, out, out.stride(0), int(add_act), int(relu_out),
             K1, HD, D2, _active(active, x1.shape[0]), dev, st)
    return out


def pack_bf16(w, transpose=False, out=None):
    """bf16 copy of a small fp32 matrix ([R, C] -> [R, C], or [C, R] with transpose) for the pre-packed bf16 kernels."""
    _rows2d(w, 'w')
    R, C = w.shape
    shape = (C, R) if transpose else (R, C)
    if out is None:
        out = torch.empty(shape, dtype=torch.bfloat16, device=w.device)
    elif not (out.is_cuda and out.dtype == torch.bfloat16 and tuple(out.shape) == shape and out.is_contiguous()):
        raise ValueError(f'pack_bf16: out must be a contiguous bf16 CUDA tensor of shape {shape}')
    dev, st = lib.stream_args(w)
    lib.call('mmft_pack_bf16', w, w.stride(0), R, C, out, int(transpose), dev, st)
    return out


def mlp2_rows_bf16(x1, rows, w1p, b1, w2p, b2, out, mask=None, hid_out=None, add_act=False, relu_out=False, active=None):
    """Fused Linear-ReLU-Linear over gathered rows with pre-packed bf16 weights (mmft_mlp2_rows_bf16)."""
    _rows2d(x1, 'x1'); _rows2d(out, 'out'); _idx(rows, 'rows')
    for t, nm, shape in ((w1p, 'w1p', (256, 128)), (w2p, 'w2p', (128, 256))):
        if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.bfloat16 and tuple(t.shape) == shape and t.is_contiguous()):
            raise ValueError(f'mlp2_rows_bf16: {nm} must be a contiguous bf16 CUDA tensor of shape {shape}')
    if x1.shape[1] != 128 or out.shape[1] != 128 or out.shape[0] != x1.shape[0]:
        raise ValueError('mlp2_rows_bf16: 128 -> 256 -> 128 over node-indexed buffers')
    for t, nm in ((mask, 'mask'), (hid_out, 'hid_out')):
        if t is not None:
            _hid2d(t, nm)
            if t.shape[1] != 256 or t.shape[0] != x1.shape[0]:
                raise ValueError(f'mlp2_rows_bf16: {nm} must be [rows of x1, 256]')
    dev, st = lib.stream_args(x1)
    lib.call('mmft_mlp2_rows_bf16', x1, x1.stride(0), rows, rows.numel(), w1p, b1, w2p, b2, mask,
             mask.stride(0) if mask is not None else 0, hid_out, hid_out.stride(0) if hid_out is not None else 0, out,
             out.stride(0), int(add_act), int(relu_out), 128, 256, 128, _active(active, x1.shape[0]), _same_hid_dtype(mask, hid_out),
             dev, st)
    return out


def level_fwd_bf16(h, pre, in_net, in_cell, net_range, cell_rows, A, LSE, w1p, b1, w2p, b2, hid_out, relu=True, active=None,
                   alg_bytes=0, in_cell_driver=None):
    """Fused forward level kernel of the bf16 mode (mmft_level_fwd_bf16): folded gather + fc_cell_neigh in one launch."""
    for t, nm in ((h, 'h'), (pre, 'pre'), (A, 'A'), (LSE, 'LSE')):
        _rows2d(t, nm)
        if t.shape != h.shape or t.stride(0) != h.stride(0):
            raise ValueError(f'level_fwd_bf16: {nm} must have the layout of h')
    N = h.shape[0]
    if h.shape[1] != 128:
        raise ValueError('level_fwd_bf16: D = 128 only')
    _csr(in_net[0], in_net[1], N, 'in_net'); _csr(in_cell[0], in_cell[1], N, 'in_cell')
    _, nrow0, nn = _rowspec(net_range if net_range is not None else (0, 0), N, 'net_range')
    ct, crow0, nc = _rowspec(cell_rows, N, 'cell_rows')
    for t, nm, shape in ((w1p, 'w1p', (256, 128)), (w2p, 'w2p', (128, 256))):
        if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.bfloat16 and tuple(t.shape) == shape and t.is_contiguous()):
            raise ValueError(f'level_fwd_bf16: {nm} must be a contiguous bf16 CUDA tensor of shape {shape}')
    _hid2d(hid_out, 'hid_out')
    if hid_out.shape != (N, 256):
        raise ValueError('level_fwd_bf16: hid_out must be [N, 256]')
    dev, st = lib.stream_args(h)
    lib.call('mmft_level_fwd_bf16', h, pre, h.stride(0), 128, in_net[0], in_net[1], in_cell[0], in_cell[1], nrow0, nn, ct, crow0,
             nc, A, LSE, w1p, b1, w2p, b2, hid_out, hid_out.stride(0), int(relu), _active(active, N),
             _edge_drivers(in_cell_driver, in_cell[1]), int(alg_bytes), _same_hid_dtype(hid_out), dev, st)


def level_fwd_slots(h, pre, slots, net_driver, net_range, cell_range, A, LSE, w1p, b1, w2p, b2, hid_out, relu=True, active=None,
                    alg_bytes=0):
    """Slot-table form of level_fwd_bf16 (mmft_level_fwd_slots): contiguous net / cell row ranges, fan-in <= 4."""
    for t, nm in ((h, 'h'), (pre, 'pre'), (A, 'A'), (LSE, 'LSE')):
        _rows2d(t, nm)
        if t.shape != h.shape or t.stride(0) != h.stride(0):
            raise ValueError(f'level_fwd_slots: {nm} must have the layout of h')
    N = h.shape[0]
    if h.shape[1] != 128:
        raise ValueError('level_fwd_slots: D = 128 only')
    for t, nm, shape in ((slots, 'slots', (N, 8)), (net_driver, 'net_driver', (N,))):
        if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.int32 and tuple(t.shape) == shape and t.is_contiguous()):
            raise ValueError(f'level_fwd_slots: {nm} must be a contiguous int32 CUDA tensor of shape {shape}')
    nrow0, nn = net_range if net_range is not None else (0, 0)
    crow0, nc = cell_range
    if min(nrow0, nn, crow0, nc) < 0 or nrow0 + nn > N or crow0 + nc > N:
        raise ValueError('level_fwd_slots: row range outside the graph')
    for t, nm, shape in ((w1p, 'w1p', (256, 128)), (w2p, 'w2p', (128, 256))):
        if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.bfloat16 and tuple(t.shape) == shape and t.is_contiguous()):
            raise ValueError(f'level_fwd_slots: {nm} must be a contiguous bf16 CUDA tensor of shape {shape}')
    _hid2d(hid_out, 'hid_out')
    if hid_out.shape != (N, 256):
        raise ValueError('level_fwd_slots: hid_out must be [N, 256]')
    dev, st = lib.stream_args(h)
    return _launch('mmft_level_fwd_slots', [h, pre, h.stride(0), 128, slots, net_driver, nrow0, nn, crow0, nc, A, LSE, w1p, b1, w2p, b2,
                                            hid_out, hid_out.stride(0), int(relu), _active(active, N), int(alg_bytes),
                                            _same_hid_dtype(hid_out), dev, st])


def level_bwd_pair(G, h, A, LSE, DA, own, tiles, ntiles, out_net_indptr, sink_shift, cslots, out_cell, scratch, counters, w1p, w2p,
                   HN, DHN, relu=True,
                   has_mlp=True, alg_bytes=0):
    """Reverse sweep of one (cell level, net level above it) pair in one launch (mmft_level_bwd_pair); the tables come from
    PinGraph.level_bwd_pairs, which checks the layout the kernel assumes."""
    for t, nm in ((G, 'G'), (h, 'h'), (A, 'A'), (LSE, 'LSE'), (DA, 'DA')):
        _rows2d(t, nm)
        if t.shape != h.shape or t.stride(0) != h.stride(0):
            raise ValueError(f'level_bwd_pair: {nm} must have the layout of h')
    N = h.shape[0]
    if h.shape[1] != 128:
        raise ValueError('level_bwd_pair: D = 128 only')
    if not (torch.is_tensor(tiles) and tiles.is_cuda and tiles.dtype == torch.int32 and tiles.dim() == 2 and tiles.shape[1] == 8
            and tiles.is_contiguous() and tiles.shape[0] == ntiles):
        raise ValueError('level_bwd_pair: tiles must be a contiguous int32 CUDA tensor [ntiles, 8]')
    _chk(scratch, 'scratch')
    _chk(counters, 'counters', torch.int32)
    if scratch.dim() != 2 or scratch.shape[1] != 128 or not scratch.is_contiguous():
        raise ValueError('level_bwd_pair: scratch must be a contiguous fp32 [rows, 128] tensor')
    if not (torch.is_tensor(cslots) and cslots.is_cuda and cslots.dtype == torch.int32 and tuple(cslots.shape) == (N, 4)
            and cslots.is_contiguous()):
        raise ValueError('level_bwd_pair: cslots must be a contiguous int32 CUDA tensor [N, 4]')
    _chk(out_net_indptr, 'out_net_indptr', torch.int32)
    if out_net_indptr.numel() != N + 1:
        raise ValueError('level_bwd_pair: out_net_indptr must have N + 1 entries')
    _csr(out_cell[0], out_cell[1], N, 'out_cell')
    if own is not None:
        _chk(own, 'own', torch.uint8)
        if own.numel() != N:
            raise ValueError('level_bwd_pair: one own-gradient flag per node expected')
    if has_mlp:
        for t, nm, shape in ((w1p, 'w1p', (256, 128)), (w2p, 'w2p', (128, 256))):
            if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.bfloat16 and tuple(t.shape) == shape and t.is_contiguous()):
                raise ValueError(f'level_bwd_pair: {nm} must be a contiguous bf16 CUDA tensor of shape {shape}')
        for t, nm in ((HN, 'HN'), (DHN, 'DHN')):
            if t is None and nm == 'DHN':
                continue
            _hid2d(t, nm)
            if tuple(t.shape) != (N, 256):
                raise ValueError(f'level_bwd_pair: {nm} must be [N, 256]')
    dev, st = lib.stream_args(h)
    return _launch('mmft_level_bwd_pair', [G, h, A, LSE, DA, h.stride(0), 128, N, own, tiles, int(ntiles), out_net_indptr, int(sink_shift),
                                           cslots, out_cell[0], out_cell[1], scratch, counters, int(relu), int(bool(has_mlp)),
                                           w1p if has_mlp else None, w2p if has_mlp else None, HN if has_mlp else None,
                                           HN.stride(0) if has_mlp else 0, DHN if has_mlp else None,
                                           DHN.stride(0) if (has_mlp and DHN is not None) else 0, int(alg_bytes),
                                           _same_hid_dtype(HN, DHN) if has_mlp else 0, dev, st])


def mlp2_feat_fusable(fin, HD, D2):
    return 1 <= fin <= 64 and HD == 256 and D2 == 128


def mlp2_feat_fwd_bf16(x, rows, w1, b1, w2, b2, out, relu_out=False):
    """out[rows] = (relu)(w2 relu(w1 x[rows] + b1) + b2) over a contiguous node range rows = (row0, n), no hidden tensor
    stored (mmft_mlp2_feat_fwd_bf16; bf16 math mode)."""
    _rows2d(x, 'x'); _rows2d(out, 'out')
    for t, nm in ((w1, 'w1'), (b1, 'b1'), (w2, 'w2'), (b2, 'b2')):
        _chk(t, nm)
        if not t.is_contiguous():
            raise ValueError(f'mlp2_feat_fwd_bf16: {nm} must be contiguous')
    fin = x.shape[1]
    if tuple(w1.shape) != (256, fin) or tuple(w2.shape) != (128, 256) or out.shape[1] != 128 or out.shape[0] != x.shape[0]:
        raise ValueError('mlp2_feat_fwd_bf16: fin -> 256 -> 128 over node-indexed buffers')
    rt, row0, n = _rowspec(rows, x.shape[0], 'rows')
    if rt is not None:
        raise ValueError('mlp2_feat_fwd_bf16: rows must be a (row0, n) range')
    dev, st = lib.stream_args(x)
    lib.call('mmft_mlp2_feat_fwd_bf16', x, x.stride(0), row0, n, fin, w1, b1, w2, b2, out, out.stride(0), int(relu_out), dev, st)
    return out


def mlp2_feat_bwd_bf16(g, x, rows, w1, b1, w2, dw1=None, db1=None, dw2=None, db2=None):
    """(dw1, db1, dw2, db2) of that MLP from the gradient g of its output rows; the hidden activations are recomputed
    (mmft_mlp2_feat_bwd_bf16)."""
    _rows2d(g, 'g'); _rows2d(x, 'x')
    fin = x.shape[1]
    rt, row0, n = _rowspec(rows, x.shape[0], 'rows')
    if rt is not None or g.shape[0] != x.shape[0] or g.shape[1] != 128:
        raise ValueError('mlp2_feat_bwd_bf16: rows must be a (row0, n) range over node-indexed g [N, 128] / x [N, fin]')
    mk = lambda t, shape: torch.empty(shape, dtype=torch.float32, device=x.device) if t is None else t
    dw1, db1, dw2, db2 = mk(dw1, (256, fin)), mk(db1, (256,)), mk(dw2, (128, 256)), mk(db2, (128,))
    for t, nm in ((dw1, 'dw1'), (db1, 'db1'), (dw2, 'dw2'), (db2, 'db2'), (w1, 'w1'), (b1, 'b1'), (w2, 'w2')):
        _chk(t, nm)
        if not t.is_contiguous():
            raise ValueError(f'mlp2_feat_bwd_bf16: {nm} must be contiguous')
    if n == 0:
        for t in (dw1, db1, dw2, db2):
            t.zero_()
        return dw1, db1, dw2, db2
    ws = lib.workspace(x.device, lib.query('mmft_mlp2_feat_bwd_workspace_bytes', n, fin))
    dev, st = lib.stream_args(x)
    lib.call('mmft_mlp2_feat_bwd_bf16', g, g.stride(0), x, x.stride(0), row0, n, fin, w1, b1, w2, dw1, db1, dw2, db2, 0, ws,
             ws.numel() * 4, dev, st)
    return dw1, db1, dw2, db2


def _edge_drivers(drv, indices):
    if drv is None:
        return None
    _idx(drv, 'in_cell_driver')
    if drv.numel() != indices.numel():
        raise ValueError('in_cell_driver: one entry per cell in-edge expected')
    return drv


def _active(active, N):
    if active is None:
        return None
    _chk(active, 'active', torch.uint8)
    if active.numel() != N or not active.is_contiguous():
        raise ValueError('active: one uint8 flag per node expected')
    return active


def colsum(g, out=None, idx=None, rows=None, accumulate=False):
    _rows2d(g, 'g')
    cols = g.shape[1]
    if rows is None:
        rows = idx.numel() if idx is not None else g.shape[0]
    _idx(idx, 'idx', rows)
    if out is None:
        out = torch.empty(cols, dtype=torch.float32, device=g.device)
        accumulate = False
    _chk(out, 'out')
    if out.numel() != cols or not out.is_contiguous():
        raise ValueError('colsum: out shape')
    need = lib.query('mmft_colsum_workspace_bytes', rows, cols)
    ws = lib.workspace(g.device, need)
    dev, st = lib.stream_args(g)
    lib.call('mmft_colsum', g, idx, g.stride(0), rows, cols, out, int(accumulate), ws, ws.numel() * 4, dev, st)
    return out


def act_bwd(dy, y, act, slope=0.0):
    _chk(dy, 'dy'); _chk(y, 'y')
    if not (dy.is_contiguous() and y.is_contiguous()) or dy.numel() != y.numel():
        raise ValueError('act_bwd: contiguous tensors of equal size required')
    out = torch.empty_like(dy)
    dev, st = lib.stream_args(dy)
    lib.call('mmft_act_bwd', dy, y, out, dy.numel(), act, float(slope), dev, st)
    return out


def act_fwd(x, act, slope=0.0, out=None):
    _chk(x, 'x')
    if not x.is_contiguous():
        raise ValueError('act_fwd: contiguous tensor required')
    if out is None:
        out = torch.empty_like(x)
    dev, st = lib.stream_args(x)
    lib.call('mmft_act_fwd', x, out, x.numel(), act, float(slope), dev, st)
    return out


# ------------------------------------------------------------------------------------------ graph
def _csr(indptr, indices, n_nodes, name):
    _idx(indptr, name + '.indptr', n_nodes + 1)
    _idx(indices, name + '.indices')


def _rowspec(rows, N, name):
    """rows: int32 device tensor | (start, n) contiguous range | None (all rows) -> (tensor or None, row0, n)."""
    if rows is None:
        return None, 0, N
    if isinstance(rows, tuple):
        start, n = int(rows[0]), int(rows[1])
        if start < 0 or n < 0 or start + n > N:
            raise ValueError(f'{name}: row range outside the node set')
        return None, start, n
    _idx(rows, name)
    return rows, 0, rows.numel()


def seg_softmax_sum_fwd(h, in_csr, rows, A, LSE=None, alg_bytes=0):
    _rows2d(h, 'h'); _rows2d(A, 'A')
    _csr(in_csr[0], in_csr[1], h.shape[0], 'in_csr')
    if A.shape != h.shape or (LSE is not None and (LSE.shape != h.shape or LSE.stride(0) != A.stride(0))):
        raise ValueError('seg_softmax_sum_fwd: A/LSE must match h')
    rt, row0, n = _rowspec(rows, h.shape[0], 'rows')
    dev, st = lib.stream_args(h)
    lib.call('mmft_seg_softmax_sum_fwd', h, h.stride(0), in_csr[0], in_csr[1], rt, row0, n, h.shape[1], A, LSE,
             A.stride(0), int(alg_bytes), dev, st)
    return A


def seg_mean_add_act_fwd(h, in_csr, rows, relu=True, alg_bytes=0):
    _rows2d(h, 'h')
    _csr(in_csr[0], in_csr[1], h.shape[0], 'in_csr')
    rt, row0, n = _rowspec(rows, h.shape[0], 'rows')
    dev, st = lib.stream_args(h)
    lib.call('mmft_seg_mean_add_act_fwd', h, h.stride(0), in_csr[0], in_csr[1], rt, row0, n, h.shape[1], int(relu),
             int(alg_bytes), dev, st)
    return h


def seg_mean_fwd(src, in_csr, rows):
    _rows2d(src, 'src')
    _csr(in_csr[0], in_csr[1], src.shape[0], 'in_csr')
    _idx(rows, 'rows')
    n = rows.numel() if rows is not None else src.shape[0]
    out = torch.empty((n, src.shape[1]), dtype=torch.float32, device=src.device)
    dev, st = lib.stream_args(src)
    lib.call('mmft_seg_mean_fwd', src, src.stride(0), in_csr[0], in_csr[1], rows, n, src.shape[1], out, out.stride(0),
             dev, st)
    return out


def scatter_add_rows_det(dst, idx, src):
    """dst[idx[i]] += src[i], bitwise reproducible: rows are grouped by a stable sort of idx and every destination
    row sums its contributions in batch order (no float atomics).  Meant for small destination tables."""
    _rows2d(dst, 'dst'); _rows2d(src, 'src'); _idx(idx, 'idx', src.shape[0])
    R = dst.shape[0]
    sidx, perm = torch.sort(idx.long(), stable=True)
    indptr = torch.searchsorted(sidx, torch.arange(R + 1, device=idx.device)).to(torch.int32)
    dev, st = lib.stream_args(dst)
    # few long segments (a 64-row level table fed by ~10^4 batch rows): a workgroup per destination row
    name = 'mmft_seg_sum_rows_wg' if (idx.numel() >= 16 * R and dst.shape[1] <= 256 and 256 % (dst.shape[1] // 4) == 0) \
        else 'mmft_seg_sum_fwd'
    if name == 'mmft_seg_sum_rows_wg':
        lib.prof_hint(float(idx.numel()) * dst.shape[1], idx.numel() * (dst.shape[1] * 4 + 4) + 2.0 * R * dst.shape[1] * 4)
    lib.call(name, src, src.stride(0), indptr, perm.to(torch.int32), None, R, dst.shape[1], dst, dst.stride(0), 1, dev, st)
    return dst


def seg_sum_sorted(dst, sorted_idx, src):
    """dst[v] += sum of src rows e with sorted_idx[e] == v; sorted_idx ascending int32 (the caller guarantees the order)."""
    _rows2d(dst, 'dst'); _rows2d(src, 'src'); _idx(sorted_idx, 'sorted_idx', src.shape[0])
    dev, st = lib.stream_args(dst)
    lib.call('mmft_seg_sum_sorted', src, src.stride(0), sorted_idx, src.shape[0], dst.shape[0], dst.shape[1], dst, dst.stride(0), 1,
             dev, st)
    return dst


def target_rows_begin(G, idx, flags):
    """Zero the rows idx of G and flag them (uint8 per node) ahead of the endpoint-gradient scatter."""
    _rows2d(G, 'G'); _idx(idx, 'idx'); _chk(flags, 'flags', torch.uint8)
    if flags.numel() != G.shape[0]:
        raise ValueError('target_rows_begin: one flag per node expected')
    dev, st = lib.stream_args(G)
    lib.call('mmft_target_rows_begin', G, G.stride(0), idx, idx.numel(), G.shape[1], flags, dev, st)


def target_rows_end(idx, flags):
    _idx(idx, 'idx'); _chk(flags, 'flags', torch.uint8)
    dev, st = lib.stream_args(flags)
    lib.call('mmft_target_rows_end', idx, idx.numel(), flags, dev, st)


def level_bwd_pull(G, h, rows, out_net, out_net_w, out_cell, A, LSE, DA, relu=True, alg_bytes=0, own=None, heavy=None,
                   heavy_thresh=PAIR_HEAVY_OUT, active=None):
    for t, nm in ((G, 'G'), (h, 'h'), (A, 'A'), (LSE, 'LSE'), (DA, 'DA')):
        _rows2d(t, nm)
        if t.shape != h.shape or t.stride(0) != h.stride(0):
            raise ValueError(f'level_bwd_pull: {nm} must have the layout of h')
    N = h.shape[0]
    _csr(out_net[0], out_net[1], N, 'out_net'); _csr(out_cell[0], out_cell[1], N, 'out_cell')
    _chk(out_net_w, 'out_net_w')
    if out_net_w.numel() != out_net[1].numel() or not out_net_w.is_contiguous():
        raise ValueError('level_bwd_pull: one weight per out-net edge expected')
    rt, row0, n = _rowspec(rows, N, 'rows')
    if own is not None:
        _chk(own, 'own', torch.uint8)
        if own.numel() != N:
            raise ValueError('level_bwd_pull: one own-gradient flag per node expected')
    dev, st = lib.stream_args(h)
    if heavy is not None:
        _idx(heavy, 'heavy')
    lib.call('mmft_level_bwd_pull', G, h, h.stride(0), rt, row0, n, h.shape[1], out_net[0], out_net[1], out_net_w,
             out_cell[0], out_cell[1], A, LSE, DA, int(relu), own, heavy, heavy.numel() if heavy is not None else 0,
             int(heavy_thresh), _active(active, N), int(alg_bytes), dev, st)
    return G




def pair_fwd_gather(h, pre, in_net, in_cell, net_range, cell_rows, A, LSE, relu=True, heavy=None, heavy_thresh=PAIR_HEAVY_IN,
                    alg_bytes=0, active=None, in_cell_driver=None):
    """Folded forward gather of one (net level, cell level) pair: see mmft_pair_fwd_gather in include/mmft.h.
    net_range = (row0, n); cell_rows = int32 tensor | (row0, n) | None (no cell level)."""
    _rows2d(h, 'h'); _rows2d(pre, 'pre')
    N = h.shape[0]
    if pre.shape != h.shape or pre.stride(0) != h.stride(0):
        raise ValueError('pair_fwd_gather: pre must have the layout of h')
    _csr(in_net[0], in_net[1], N, 'in_net'); _csr(in_cell[0], in_cell[1], N, 'in_cell')
    _, nrow0, nn = _rowspec(net_range if net_range is not None else (0, 0), N, 'net_range')
    if cell_rows is None:
        ct, crow0, nc = None, 0, 0
    else:
        ct, crow0, nc = _rowspec(cell_rows, N, 'cell_rows')
        _rows2d(A, 'A'); _rows2d(LSE, 'LSE')
        if A.shape != h.shape or LSE.shape != h.shape or LSE.stride(0) != A.stride(0):
            raise ValueError('pair_fwd_gather: A / LSE must match h')
    if heavy is not None:
        _idx(heavy, 'heavy')
    dev, st = lib.stream_args(h)
    lib.call('mmft_pair_fwd_gather', h, pre, h.stride(0), h.shape[1], in_net[0], in_net[1], in_cell[0], in_cell[1], nrow0, nn,
             ct, crow0, nc, A if nc else None, LSE if nc else None, A.stride(0) if nc else h.stride(0), int(relu), heavy,
             heavy.numel() if heavy is not None else 0, int(heavy_thresh), _active(active, N),
             _edge_drivers(in_cell_driver, in_cell[1]), int(alg_bytes), dev, st)


ATTN_SLOPE = 0.01          # F.leaky_relu default (src/model.py:136)


def seg_attn_fwd(h, key, c12, in_csr, rows, A, alpha):
    """Attention aggregation of one cell level (mmft_seg_attn_fwd): A[v] = sum_i softmax_i(e_i) h[u_i]."""
    _rows2d(h, 'h'); _rows2d(A, 'A'); _chk(key, 'key'); _chk(c12, 'c12'); _chk(alpha, 'alpha')
    N = h.shape[0]
    _csr(in_csr[0], in_csr[1], N, 'in_csr')
    if key.numel() != N or not key.is_contiguous() or c12.numel() != 2 or not c12.is_contiguous():
        raise ValueError("seg_attn_fwd: key must hold one value per node, c12 two scalars")
    if alpha.numel() != in_csr[1].numel() or not alpha.is_contiguous() or A.shape != h.shape:
        raise ValueError('seg_attn_fwd: alpha needs one slot per in-edge, A the layout of h')
    rt, row0, n = _rowspec(rows, N, 'rows')
    dev, st = lib.stream_args(h)
    lib.call('mmft_seg_attn_fwd', h, h.stride(0), key, c12, ATTN_SLOPE, in_csr[0], in_csr[1], rt, row0, n, h.shape[1], A,
             A.stride(0), alpha, dev, st)
    return A


def level_bwd_pull_attn(G, h, rows, out_net, out_net_w, out_cell, o2i, alpha, DA, relu=True, own=None):
    for t, nm in ((G, 'G'), (h, 'h'), (DA, 'DA')):
        _rows2d(t, nm)
        if t.shape != h.shape or t.stride(0) != h.stride(0):
            raise ValueError(f'level_bwd_pull_attn: {nm} must have the layout of h')
    N = h.shape[0]
    _csr(out_net[0], out_net[1], N, 'out_net'); _csr(out_cell[0], out_cell[1], N, 'out_cell')
    _idx(o2i, 'o2i', out_cell[1].numel()); _chk(alpha, 'alpha'); _chk(out_net_w, 'out_net_w')
    if alpha.numel() != out_cell[1].numel() or out_net_w.numel() != out_net[1].numel():
        raise ValueError('level_bwd_pull_attn: one alpha per cell edge, one weight per net edge expected')
    rt, row0, n = _rowspec(rows, N, 'rows')
    if own is not None:
        _chk(own, 'own', torch.uint8)
    dev, st = lib.stream_args(h)
    lib.call('mmft_level_bwd_pull_attn', G, h, h.stride(0), rt, row0, n, h.shape[1], out_net[0], out_net[1], out_net_w,
             out_cell[0], out_cell[1], o2i, alpha, DA, int(relu), own, dev, st)
    return G


def seg_attn_bwd_scores(DA, h, A, alpha, key, c12, in_csr, rows, dcp):
    for t, nm in ((DA, 'DA'), (h, 'h'), (A, 'A')):
        _rows2d(t, nm)
        if t.shape != h.shape or t.stride(0) != h.stride(0):
            raise ValueError(f'seg_attn_bwd_scores: {nm} must have the layout of h')
    N = h.shape[0]
    _csr(in_csr[0], in_csr[1], N, 'in_csr'); _chk(dcp, 'dcp')
    if tuple(dcp.shape) != (N, 2) or not dcp.is_contiguous():
        raise ValueError('seg_attn_bwd_scores: dcp must be a contiguous [N, 2] tensor')
    rt, row0, n = _rowspec(rows, N, 'rows')
    dev, st = lib.stream_args(h)
    lib.call('mmft_seg_attn_bwd_scores', DA, h, A, h.stride(0), alpha, key, c12, ATTN_SLOPE, in_csr[0], in_csr[1], rt, row0, n,
             h.shape[1], dcp, dev, st)
    return dcp


def seg_mean_rows_any(src, csr, rows, out):
    """out[rows] = mean over the CSR segments of rows of src (any width; scatter by node id)."""
    _rows2d(src, 'src'); _rows2d(out, 'out'); _idx(rows, 'rows')
    _csr(csr[0], csr[1], src.shape[0], 'csr')
    if out.shape[1] != src.shape[1] or out.shape[0] != src.shape[0]:
        raise ValueError('seg_mean_rows_any: out must have the shape of src')
    dev, st = lib.stream_args(src)
    lib.call('mmft_seg_mean_rows_any', src, src.stride(0), csr[0], csr[1], rows, rows.numel(), src.shape[1], out,
             out.stride(0), 1, dev, st)
    return out


def gather_rows(src, idx):
    _rows2d(src, 'src'); _idx(idx, 'idx')
    out = torch.empty((idx.numel(), src.shape[1]), dtype=torch.float32, device=src.device)
    dev, st = lib.stream_args(src)
    lib.call('mmft_gather_rows', src, src.stride(0), idx, idx.numel(), src.shape[1], out, out.stride(0), dev, st)
    return out


def scatter_add_rows(dst, idx, src):
    _rows2d(dst, 'dst'); _rows2d(src, 'src'); _idx(idx, 'idx', src.shape[0])
    if src.shape[1] != dst.shape[1]:
        raise ValueError('scatter_add_rows: width mismatch')
    dev, st = lib.stream_args(dst)
    lib.call('mmft_scatter_add_rows', dst, dst.stride(0), idx, idx.numel(), dst.shape[1], src, src.stride(0), dev, st)
    return dst


def scatter_add_rows_sorted(dst, idx, order, src):
    """dst[idx[t]] += src[t], bitwise reproducible with duplicates: `order` = stable argsort of idx (int32)."""
    _rows2d(dst, 'dst'); _rows2d(src, 'src'); _idx(idx, 'idx', src.shape[0]); _idx(order, 'order', src.shape[0])
    if src.shape[1] != dst.shape[1]:
        raise ValueError('scatter_add_rows_sorted: width mismatch')
    dev, st = lib.stream_args(dst)
    lib.call('mmft_scatter_add_rows_sorted', dst, dst.stride(0), idx, order, idx.numel(), dst.shape[1], src,
             src.stride(0), dev, st)
    return dst


def scatter_add_targets(dst, idx, src, order=None, unique=None):
    """Endpoint-gradient scatter of the reverse sweep.  `order` given: deterministic sorted form.  `unique` True: the
    caller knows there are no duplicates, one atomic add per element is then exact and order-free.  Otherwise the
    stable order is derived on the device (two torch sort kernels)."""
    if order is None and not unique and idx.numel() > 1:
        order = torch.sort(idx.long(), stable=True)[1].to(torch.int32)
    if order is None:
        return scatter_add_rows(dst, idx, src)
    return scatter_add_rows_sorted(dst, idx, order, src)


# ------------------------------------------------------------------------------------------ CNN (NHWC)
def is_nhwc(t):
    return t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous()


def _nhwc(t, name):
    _chk(t, name)
    if not is_nhwc(t):
        raise ValueError(f'{name}: expected an (N,C,H,W) tensor in channels_last memory format')
    return t


def empty_nhwc(N, C, H, W, device):
    return torch.empty((N, H, W, C), dtype=torch.float32, device=device).permute(0, 3, 1, 2)


def to_nhwc(x):
    """(N,C,H,W) any layout -> channels_last memory, through the HIP layout kernel for NCHW-contiguous input."""
    _chk(x, 'x')
    if is_nhwc(x):
        return x
    N, C, H, W = x.shape
    if not x.is_contiguous():
        x = x.contiguous()
    out = empty_nhwc(N, C, H, W, x.device)
    dev, st = lib.stream_args(x)
    lib.call('mmft_nchw_to_nhwc', x, out, N, C, H, W, C, dev, st)
    return out


def to_nchw(x):
    """channels_last -> NCHW-contiguous copy."""
    _nhwc(x, 'x')
    N, C, H, W = x.shape
    out = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device)
    dev, st = lib.stream_args(x)
    lib.call('mmft_nhwc_to_nchw', x, out, N, C, H, W, C, dev, st)
    return out


def _w_ohwi(w):
    """OIHW parameter -> [Co][KH][KW][Ci] memory (free when the parameter is stored channels_last)."""
    _chk(w, 'weight')
    v = w.detach().permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def conv2d_fwd(x, w, bias, pad, act=ACT_NONE, slope=0.0):
    _nhwc(x, 'x')
    N, Ci, H, W = x.shape
    Co, Ci2, KH, KW = w.shape
    if Ci2 != Ci:
        raise ValueError(f'conv2d: input has {Ci} channels, weight expects {Ci2}')
    wk = _w_ohwi(w)
    if bias is not None:
        _chk(bias, 'bias')
        if bias.numel() != Co:
            raise ValueError('conv2d: bias size')
    y = empty_nhwc(N, Co, H, W, x.device)
    dev, st = lib.stream_args(x)
    lib.call('mmft_conv2d_fwd', x, wk, bias, y, N, H, W, Ci, Co, KH, KW, pad, act, float(slope), dev, st)
    return y


def conv2d_dgrad(gy, w, pad):
    _nhwc(gy, 'gy')
    N, Co, H, W = gy.shape
    Co2, Ci, KH, KW = w.shape
    if Co2 != Co:
        raise ValueError('conv2d_dgrad: channel mismatch')
    wk = _w_ohwi(w)
    dx = empty_nhwc(N, Ci, H, W, gy.device)
    ws = lib.workspace(gy.device, wk.numel() * 4)
    dev, st = lib.stream_args(gy)
    lib.call('mmft_conv2d_dgrad', gy, wk, dx, N, H, W, Ci, Co, KH, KW, pad, ws, ws.numel() * 4, dev, st)
    return dx


def conv2d_wgrad(x, gy, KH, KW, pad, out=None):
    """returns dw as a contiguous [Co][KH][KW][Ci] tensor (`out` when given)."""
    _nhwc(x, 'x'); _nhwc(gy, 'gy')
    N, Ci, H, W = x.shape
    Co = gy.shape[1]
    if gy.shape[0] != N or gy.shape[2:] != x.shape[2:]:
        raise ValueError('conv2d_wgrad: shape mismatch')
    dw = out if out is not None else torch.empty((Co, KH, KW, Ci), dtype=torch.float32, device=x.device)
    _chk(dw, 'dw')
    if tuple(dw.shape) != (Co, KH, KW, Ci) or not dw.is_contiguous():
        raise ValueError(f'conv2d_wgrad: out must be a contiguous {(Co, KH, KW, Ci)} tensor')
    need = lib.query('mmft_conv2d_wgrad_workspace_bytes', N, H, W, Ci, Co, KH, KW)
    ws = lib.workspace(x.device, need)
    dev, st = lib.stream_args(x)
    lib.call('mmft_conv2d_wgrad', x, gy, dw, N, H, W, Ci, Co, KH, KW, pad, ws, ws.numel() * 4, dev, st)
    return dw


def bn_train_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, relu, per_sample=False):
    _nhwc(x, 'x')
    N, C, H, W = x.shape
    for t, nm in ((gamma, 'gamma'), (beta, 'beta')):
        _chk(t, nm)
        if t.numel() != C or not t.is_contiguous():
            raise ValueError(f'bn: {nm} shape')
    groups, rows = (N, H * W) if per_sample else (1, N * H * W)
    y = empty_nhwc(N, C, H, W, x.device)
    mean = torch.empty((groups, C), dtype=torch.float32, device=x.device)
    invstd = torch.empty((groups, C), dtype=torch.float32, device=x.device)
    ws = lib.workspace(x.device, lib.query('mmft_bn_workspace_bytes', groups, rows, C))
    dev, st = lib.stream_args(x)
    lib.call('mmft_bn_train_fwd', x, y, gamma, beta, running_mean, running_var, float(momentum), float(eps), groups,
             rows, C, mean, invstd, int(relu), ws, ws.numel() * 4, dev, st)
    return y, mean, invstd


def bn_train_bwd(gy, x, y, gamma, mean, invstd, relu, dgamma=None, dbeta=None, beta=None):
    _nhwc(gy, 'gy'); _nhwc(x, 'x')
    N, C, H, W = x.shape
    groups = mean.shape[0]
    rows = N * H * W // groups
    dx = empty_nhwc(N, C, H, W, x.device)
    dgamma = torch.empty(C, dtype=torch.float32, device=x.device) if dgamma is None else dgamma
    dbeta = torch.empty(C, dtype=torch.float32, device=x.device) if dbeta is None else dbeta
    for t, nm in ((dgamma, 'dgamma'), (dbeta, 'dbeta')):
        _chk(t, nm)
        if t.numel() != C or not t.is_contiguous():
            raise ValueError(f'bn_train_bwd: {nm} shape')
    ws = lib.workspace(x.device, lib.query('mmft_bn_workspace_bytes', groups, rows, C))
    dev, st = lib.stream_args(x)
    lib.call('mmft_bn_train_bwd', gy, x, y, gamma, beta, mean, invstd, dx, dgamma, dbeta, groups, rows, C, int(relu), ws,
             ws.numel() * 4, dev, st)
    return dx, dgamma, dbeta


def outconv_supported(x, w):
    """True when the fused OutConv kernels take this layer: [N, Ci, H, W] channels_last input, a [1, Ci, 1, 1] weight."""
    if w.dim() != 4 or w.shape[0] != 1 or w.shape[2] != 1 or w.shape[3] != 1 or not w.is_contiguous() or w.data_ptr() % 16:
        return False
    N, Ci, H, W = x.shape
    return Ci == w.shape[1] and bool(lib.query('mmft_outconv_supported', H, W, Ci))


def outconv_fwd(x, w, bias, mode):
    """relu(pool2x2(conv1x1(x) + bias)) -> [N, 1, H/2, W/2]  (mmft_outconv_fwd; src/Unet.py:71-82)."""
    _nhwc(x, 'x'); _chk(w, 'w')
    N, Ci, H, W = x.shape
    if bias is not None:
        _chk(bias, 'bias')
    out = torch.empty((N, 1, H // 2, W // 2), dtype=torch.float32, device=x.device)
    dev, st = lib.stream_args(x)
    lib.call('mmft_outconv_fwd', x, w, bias, out, N, H, W, Ci, mode, dev, st)
    return out


def outconv_bwd(x, w, bias, gout, mode, dw=None, db=None):
    """(dx, dw, db) of the fused OutConv; dw / db are written into the given tensors when present."""
    _nhwc(x, 'x'); _chk(w, 'w'); _chk(gout, 'gout')
    N, Ci, H, W = x.shape
    if gout.numel() != N * (H // 2) * (W // 2) or not gout.is_contiguous():
        raise ValueError('outconv_bwd: gout must be a contiguous [N, 1, H/2, W/2] tensor')
    dx = empty_nhwc(N, Ci, H, W, x.device)
    dw = torch.empty(Ci, dtype=torch.float32, device=x.device) if dw is None else dw
    if bias is not None and db is None:
        db = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = lib.workspace(x.device, lib.query('mmft_outconv_bwd_workspace_bytes', N, H, W, Ci))
    dev, st = lib.stream_args(x)
    lib.call('mmft_outconv_bwd', x, w, bias, gout, dx, dw, db if bias is not None else None, 0, N, H, W, Ci, mode, ws,
             ws.numel() * 4, dev, st)
    return dx, dw, db


def pool2x2_fwd(x, mode):
    _nhwc(x, 'x')
    N, C, H, W = x.shape
    if H < 2 or W < 2:
        raise ValueError('pool2x2: input smaller than the window')
    y = empty_nhwc(N, C, H // 2, W // 2, x.device)
    dev, st = lib.stream_args(x)
    lib.call('mmft_pool2x2_fwd', x, y, N, H, W, C, mode, dev, st)
    return y


def pool2x2_bwd(x, gy, mode):
    _nhwc(x, 'x'); _nhwc(gy, 'gy')
    N, C, H, W = x.shape
    if tuple(gy.shape) != (N, C, H // 2, W // 2):
        raise ValueError('pool2x2_bwd: gradient shape')
    dx = empty_nhwc(N, C, H, W, x.device)
    dev, st = lib.stream_args(x)
    lib.call('mmft_pool2x2_bwd', x, gy, dx, N, H, W, C, mode, dev, st)
    return dx


def upsample_bilinear2x_fwd(x):
    _nhwc(x, 'x')
    N, C, H, W = x.shape
    y = empty_nhwc(N, C, 2 * H, 2 * W, x.device)
    dev, st = lib.stream_args(x)
    lib.call('mmft_upsample_bilinear2x_fwd', x, y, N, H, W, C, dev, st)
    return y


def upsample_bilinear2x_bwd(gy):
    _nhwc(gy, 'gy')
    N, C, H2, W2 = gy.shape
    if H2 % 2 or W2 % 2:
        raise ValueError('upsample_bilinear2x_bwd: gradient of a x2 up-sampling expected')
    dx = empty_nhwc(N, C, H2 // 2, W2 // 2, gy.device)
    dev, st = lib.stream_args(gy)
    lib.call('mmft_upsample_bilinear2x_bwd', gy, dx, N, H2 // 2, W2 // 2, C, dev, st)
    return dx


def pixel_shuffle2(t, bias, N, H, W, Co, out=None, c_off=0):
    """t: [N*H*W, 4*Co] rows -> (N,Co,2H,2W) channels_last, + bias[co]; with `out` given: into its channel slice
    [c_off, c_off + Co)."""
    _rows2d(t, 't')
    if t.shape != (N * H * W, 4 * Co) or not t.is_contiguous():
        raise ValueError('pixel_shuffle2: bad input shape')
    if out is None:
        out = empty_nhwc(N, Co, 2 * H, 2 * W, t.device)
    _nhwc(out, 'out')
    if out.shape[0] != N or tuple(out.shape[2:]) != (2 * H, 2 * W) or c_off < 0 or c_off + Co > out.shape[1]:
        raise ValueError('pixel_shuffle2: destination shape')
    dev, st = lib.stream_args(t)
    lib.call('mmft_pixel_shuffle2_into', t, bias, out, N, H, W, Co, out.shape[1], int(c_off), dev, st)
    return out


def pixel_unshuffle2(g, Co=None, c_off=0):
    """(N,C,2H,2W) channels_last, channel slice [c_off, c_off + Co) (default: all) -> [N*H*W, 4*Co] rows."""
    _nhwc(g, 'g')
    N, C, H2, W2 = g.shape
    Co = C if Co is None else Co
    if c_off < 0 or c_off + Co > C:
        raise ValueError('pixel_unshuffle2: channel slice')
    H, W = H2 // 2, W2 // 2
    out = torch.empty((N * H * W, 4 * Co), dtype=torch.float32, device=g.device)
    dev, st = lib.stream_args(g)
    lib.call('mmft_pixel_unshuffle2_from', g, out, N, H, W, Co, C, int(c_off), dev, st)
    return out


def copy_region(src, dst, c_off, y_off, x_off, reverse=False):
    """dst[:, c_off:c_off+Cs, y_off:y_off+Hs, x_off:x_off+Ws] = src  (or the reverse copy)."""
    _nhwc(src, 'src'); _nhwc(dst, 'dst')
    N, Cs, Hs, Ws = src.shape
    Nd, Cd, Hd, Wd = dst.shape
    if N != Nd or c_off + Cs > Cd or y_off + Hs > Hd or x_off + Ws > Wd or min(c_off, y_off, x_off) < 0:
        raise ValueError('copy_region: region outside destination')
    dev, st = lib.stream_args(src)
    lib.call('mmft_copy_region_nhwc', src, N, Hs, Ws, Cs, dst, Hd, Wd, Cd, c_off, y_off, x_off, int(reverse), dev, st)
    return src if reverse else dst


def rows_view(x):
    """(N,C,H,W) channels_last -> [N*H*W, C] row-major view."""
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])
