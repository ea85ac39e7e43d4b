"""Tensor-level wrappers over the C ABI (no autograd here; see mmft.functional).

Every function checks on the host that operand shapes match what the kernels and their grids assume
before launching, and raises instead of falling back when the inputs are not device fp32 tensors.
"""
import torch
from . import lib

EPI_STORE, EPI_ACCUM, EPI_ADD_ACT = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
POOL_MAX, POOL_AVG = 0, 1


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
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f'{name}: expected a 2-D tensor with unit inner stride, got shape {tuple(t.shape)} '
                         f'strides {t.stride()}')
    return t


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


def linear_wgrad(g, x, dw=None, gidx=None, xidx=None, rows=None, accumulate=False):
    """dw[o][i] (+)= sum_r g[gidx[r]][o] * x[xidx[r]][i]."""
    _rows2d(g, 'g'); _rows2d(x, 'x')
    out, inn = g.shape[1], x.shape[1]
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
    need = lib.query('mmft_linear_wgrad_workspace_bytes', rows, out, inn)
    ws = lib.workspace(g.device, need)
    dev, st = lib.stream_args(g)
    lib.call('mmft_linear_wgrad', g, gidx, g.stride(0), x, xidx, x.stride(0), dw, dw.stride(0), rows, out, inn,
             int(accumulate), ws, ws.numel() * 4, dev, st)
    return dw


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


def seg_softmax_sum_fwd(h, in_csr, rows, A, LSE=None):
    _rows2d(h, 'h'); _rows2d(A, 'A')
    _csr(in_csr[0], in_csr[1], h.shape[0], 'in_csr')
    _idx(rows, 'rows')
    if A.shape != h.shape or (LSE is not None and (LSE.shape != h.shape or LSE.stride(0) != A.stride(0))):
        raise ValueError('seg_softmax_sum_fwd: A/LSE must match h')
    n = rows.numel() if rows is not None else h.shape[0]
    dev, st = lib.stream_args(h)
    lib.call('mmft_seg_softmax_sum_fwd', h, h.stride(0), in_csr[0], in_csr[1], rows, n, h.shape[1], A, LSE,
             A.stride(0), dev, st)
    return A


def seg_mean_add_act_fwd(h, in_csr, rows, relu=True):
    _rows2d(h, 'h')
    _csr(in_csr[0], in_csr[1], h.shape[0], 'in_csr')
    _idx(rows, 'rows')
    n = rows.numel() if rows is not None else h.shape[0]
    dev, st = lib.stream_args(h)
    lib.call('mmft_seg_mean_add_act_fwd', h, h.stride(0), in_csr[0], in_csr[1], rows, n, h.shape[1], int(relu), dev, st)
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


def level_bwd_pull(G, h, rows, out_net, in_net_indptr, out_cell, A, LSE, DA, relu=True):
    for t, nm in ((G, 'G'), (h, 'h'), (A, 'A'), (LSE, 'LSE'), (DA, 'DA')):
        _rows2d(t, nm)
        if t.shape != h.shape or t.stride(0) != h.stride(0):
            raise ValueError(f'level_bwd_pull: {nm} must have the layout of h')
    N = h.shape[0]
    _csr(out_net[0], out_net[1], N, 'out_net'); _csr(out_cell[0], out_cell[1], N, 'out_cell')
    _idx(in_net_indptr, 'in_net_indptr', N + 1)
    _idx(rows, 'rows')
    n = rows.numel() if rows is not None else N
    dev, st = lib.stream_args(h)
    lib.call('mmft_level_bwd_pull', G, h, h.stride(0), rows, n, h.shape[1], out_net[0], out_net[1], in_net_indptr,
             out_cell[0], out_cell[1], A, LSE, DA, int(relu), dev, st)
    return G


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
