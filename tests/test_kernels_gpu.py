"""Kernel-level parity: every C-ABI entry point against plain torch fp64 math on the same inputs."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from mmft import ops
from mmft.detrand import det_uniform, det_ints
from oracle import restatement as R

pytestmark = pytest.mark.gpu
TOL = 2e-5  # fp32 MFMA fmaf chains vs fp64 reference, relative to the largest output magnitude


def T(shape, seed, dev, lo=-1.0, hi=1.0):
    return torch.from_numpy(det_uniform(shape, seed, lo, hi)).to(dev)


@pytest.mark.parametrize('M,N,K', [(1, 32, 1), (9, 7, 5), (300, 256, 36), (1000, 128, 256), (257, 576, 288),
                                   (4096, 16, 144), (130, 1, 576), (64, 130, 2), (2000, 128, 1024)])
def test_linear_fwd(dev, M, N, K):
    x, w, b = T((M, K), 1, dev), T((N, K), 2, dev), T((N,), 3, dev)
    ref = (x.double() @ w.double().t() + b.double())
    y = ops.linear_fwd(x, w, b)
    assert rel_err(y, ref) < TOL
    y = ops.linear_fwd(x, w, b, act=ops.ACT_RELU)
    assert rel_err(y, ref.clamp_min(0)) < TOL
    y = ops.linear_fwd(x, w, None, act=ops.ACT_LEAKY, slope=0.1)
    r2 = x.double() @ w.double().t()
    assert rel_err(y, torch.where(r2 > 0, r2, 0.1 * r2)) < TOL


def test_linear_fwd_gather_scatter_modes(dev):
    R_, M, N, K = 500, 333, 128, 36
    x, w, b = T((R_, K), 1, dev), T((N, K), 2, dev), T((N,), 3, dev)
    xidx = torch.from_numpy(det_ints((M,), 4, 0, R_)).to(torch.int32).to(dev)
    yidx = torch.randperm(R_)[:M].to(torch.int32).to(dev)
    y0 = T((R_, N), 5, dev)
    ref_rows = x.double()[xidx.long()] @ w.double().t() + b.double()
    for epi, fn in ((ops.EPI_STORE, lambda old, v: v), (ops.EPI_ACCUM, lambda old, v: old + v),
                    (ops.EPI_ADD_ACT, lambda old, v: (old + v).clamp_min(0))):
        y = y0.clone()
        ops.linear_fwd(x, w, b, y=y, xidx=xidx, yidx=yidx, epi=epi, act=ops.ACT_RELU if epi == ops.EPI_ADD_ACT else 0)
        ref = y0.double().clone()
        ref[yidx.long()] = fn(y0.double()[yidx.long()], ref_rows)
        assert rel_err(y, ref) < TOL


@pytest.mark.parametrize('M,N,K', [(9, 5, 7), (300, 36, 256), (1000, 256, 128), (513, 288, 576), (77, 130, 1)])
def test_linear_dgrad(dev, M, N, K):
    g, w = T((M, K), 1, dev), T((K, N), 2, dev)
    ref = g.double() @ w.double()
    assert rel_err(ops.linear_dgrad(g, w), ref) < TOL
    mask = T((M, N), 3, dev)
    out = ops.linear_dgrad(g, w, mask=mask)
    assert rel_err(out, ref * (mask.double() > 0)) < TOL
    acc = T((M, N), 4, dev)
    out = ops.linear_dgrad(g, w, dx=acc.clone(), epi=ops.EPI_ACCUM)
    assert rel_err(out, ref + acc.double()) < TOL


@pytest.mark.parametrize('rows,out,inn', [(5, 7, 3), (1000, 128, 256), (40000, 256, 36), (30000, 128, 2),
                                          (2049, 576, 288), (100000, 16, 144), (333, 1, 576)])
def test_linear_wgrad_colsum(dev, rows, out, inn):
    g, x = T((rows, out), 1, dev), T((rows, inn), 2, dev)
    ref = g.double().t() @ x.double()
    dw = ops.linear_wgrad(g, x)
    assert rel_err(dw, ref) < TOL
    dw2 = ops.linear_wgrad(g, x, dw=dw.clone(), accumulate=True)
    assert rel_err(dw2, 2 * ref) < TOL
    # bitwise reproducible (deterministic slab order)
    assert torch.equal(ops.linear_wgrad(g, x), dw)
    cs = ops.colsum(g)
    assert rel_err(cs, g.double().sum(0)) < TOL
    # weight and bias gradient from one pass over g (column sums as a by-product of the n-tile-0 workgroups)
    dw3, db3 = ops.linear_wgrad(g, x, with_bias=True)
    assert torch.equal(dw3, dw)
    assert rel_err(db3, g.double().sum(0)) < TOL
    dw4, db4 = ops.linear_wgrad(g, x, dw=dw3.clone(), db=db3.clone(), accumulate=True)
    assert rel_err(dw4, 2 * ref) < TOL and rel_err(db4, 2 * g.double().sum(0)) < TOL
    assert torch.equal(ops.linear_wgrad(g, x, with_bias=True)[1], db3)


@pytest.mark.parametrize('fin,n,mode', [(36, 5000, 'idx'), (2, 4099, 'range'), (48, 37, 'idx'), (17, 16, 'all'),
                                        (36, 70001, 'range'), (4, 3, 'idx')])
def test_mlp2_first_layer_grads(dev, fin, n, mode):
    """Fused (dW1, db1) of a Linear-ReLU-Linear MLP vs fp64 torch: dH = (G W2) * (H > 0) is never stored."""
    NN, HD, D2 = n + 300, 256, 128
    g, hid, x = T((NN, D2), 1, dev), T((NN, HD), 2, dev), T((NN, fin), 3, dev)
    w2 = T((D2, HD), 4, dev)
    if mode == 'idx':
        rows = torch.randperm(NN)[:n].to(torch.int32).to(dev)
        sel = rows.long()
    elif mode == 'range':
        rows = (123, n)
        sel = torch.arange(123, 123 + n, device=dev)
    else:
        NN = n
        g, hid, x = g[:n].contiguous(), hid[:n].contiguous(), x[:n].contiguous()
        rows, sel = None, torch.arange(n, device=dev)
    dH = (g.double()[sel] @ w2.double()) * (hid.double()[sel] > 0)
    ref_w, ref_b = dH.t() @ x.double()[sel], dH.sum(0)
    dw1, db1 = ops.mlp2_first_layer_grads(g, hid, x, rows, w2)
    assert rel_err(dw1, ref_w) < TOL and rel_err(db1, ref_b) < TOL
    dw1b, db1b = ops.mlp2_first_layer_grads(g, hid, x, rows, w2)
    assert torch.equal(dw1, dw1b) and torch.equal(db1, db1b)          # fixed reduction order


def test_linear_wgrad_indexed(dev):
    R_, rows, out, inn = 5000, 3000, 128, 36
    g, x = T((R_, out), 1, dev), T((R_, inn), 2, dev)
    idx = torch.randperm(R_)[:rows].to(torch.int32).to(dev)
    ref = g.double()[idx.long()].t() @ x.double()[idx.long()]
    assert rel_err(ops.linear_wgrad(g, x, gidx=idx, xidx=idx), ref) < TOL
    assert rel_err(ops.colsum(g, idx=idx), g.double()[idx.long()].sum(0)) < TOL
    dw, db = ops.linear_wgrad(g, x, gidx=idx, xidx=idx, with_bias=True)
    assert rel_err(dw, ref) < TOL and rel_err(db, g.double()[idx.long()].sum(0)) < TOL


@pytest.mark.parametrize('N,H,W,Ci,Co', [(2, 9, 64, 16, 16), (1, 64, 128, 16, 32), (3, 5, 192, 32, 16), (2, 33, 64, 32, 32),
                                        (2, 7, 40, 16, 16), (1, 12, 64, 64, 16), (2, 6, 64, 8, 32)])
def test_conv3x3_forward_dgrad_wgrad(dev, N, H, W, Ci, Co):
    """3x3 / pad 1 convolution through the C ABI vs fp64 torch: the first four shapes take the direct narrow-channel
    kernel (Ci, Co in {16, 32}, W % 64 == 0), the others the implicit-GEMM engine; borders, bias and ReLU included."""
    x = T((N, Ci, H, W), 1, dev).contiguous(memory_format=torch.channels_last)
    w = T((Co, Ci, 3, 3), 2, dev).contiguous(memory_format=torch.channels_last)
    b = T((Co,), 3, dev)
    gy = T((N, Co, H, W), 4, dev).contiguous(memory_format=torch.channels_last)
    F = torch.nn.functional
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, b.double(), padding=1)
    assert rel_err(ops.conv2d_fwd(x, w, b, 1), y64) < TOL
    assert rel_err(ops.conv2d_fwd(x, w, None, 1, act=ops.ACT_RELU), torch.relu(F.conv2d(x64, w64, None, padding=1))) < TOL
    y64.backward(gy.double())
    assert rel_err(ops.conv2d_dgrad(gy, w, 1), x64.grad) < TOL
    dw = ops.conv2d_wgrad(x, gy, 3, 3, 1)                       # [Co][KH][KW][Ci]
    assert rel_err(dw.permute(0, 3, 1, 2), w64.grad) < TOL


@pytest.mark.parametrize('B', [1, 3])
def test_masked_fc_runs_equal_cells_and_dense(dev, B):
    """Masked projection: prefix-sum / run form vs one gather per cell vs the dense fp64 product, forward and backward
    (box masks, duplicated paths, an empty mask row, runs that cross the 64-cell prefix blocks)."""
    from mmft import fusion
    from mmft.fusion import PathMasks, MaskedPathMap, masked_fc
    rng = np.random.default_rng(5 + B)
    m, Dout, npaths = 64, 128, 40
    P = m * m
    masks = []
    for b in range(B):
        ip, cols = [0], []
        for p in range(npaths):
            cells = set()
            for _ in range(int(rng.integers(0, 6)) if p else 0):              # path 0: empty mask
                x0, y0, w, h = rng.integers(0, m - 12), rng.integers(0, m - 12), rng.integers(1, 12), rng.integers(1, 12)
                cells.update(int(x * m + y) for x in range(x0, x0 + w) for y in range(y0, y0 + h))
            c = sorted(cells)
            cols += c
            ip.append(len(cols))
        masks.append(PathMasks(np.array(ip), np.array(cols, dtype=np.int64), P, dev))
    pmk = masks[0] if B == 1 else PathMasks.batch(masks)
    assert pmk.run_block == 64 and 0 < pmk.num_runs < pmk.host_cols.shape[0]
    paths = rng.integers(0, B * npaths, size=70)
    paths[:3] = [0, 5, 5]
    feat = T((B, P), 1, dev).requires_grad_(True)
    w = T((Dout, P), 2, dev, -0.05, 0.05).requires_grad_(True)
    bias = T((Dout,), 3, dev).requires_grad_(True)
    gout = T((70, Dout), 4, dev)
    res = {}
    for use_runs in (True, False):
        fusion.USE_RUNS = use_runs
        try:
            for t in (feat, w, bias):
                t.grad = None
            out = masked_fc(MaskedPathMap(pmk, paths, feat), w, bias)
            out.backward(gout)
            res[use_runs] = (out.detach().clone(), feat.grad.clone(), w.grad.clone(), bias.grad.clone())
        finally:
            fusion.USE_RUNS = True
    dense = torch.zeros((70, P), dtype=torch.float64, device=dev)
    for t, q in enumerate(paths):
        b = int(pmk.row_design[q])
        c = torch.from_numpy(pmk.host_cols[pmk.host_indptr[q]:pmk.host_indptr[q + 1]]).to(dev)
        dense[t, c] = feat.detach().double()[b, c]
    ref = dense @ w.detach().double().t() + bias.detach().double()
    assert rel_err(res[True][0], ref) < TOL and rel_err(res[False][0], ref) < TOL
    assert rel_err(res[True][0], res[False][0]) < 1e-5
    dref = dense.clone().requires_grad_(True)
    w64, b64, f64 = w.detach().double().requires_grad_(True), bias.detach().double().requires_grad_(True), None
    (dref @ w64.t() + b64).backward(gout.double())
    gfeat = torch.zeros((B, P), dtype=torch.float64, device=dev)
    for t, q in enumerate(paths):                                             # d loss / d feat through the mask rows
        b = int(pmk.row_design[q])
        c = torch.from_numpy(pmk.host_cols[pmk.host_indptr[q]:pmk.host_indptr[q + 1]]).to(dev)
        gfeat[b, c] += dref.grad[t, c]
    for r in (res[True], res[False]):
        assert rel_err(r[1], gfeat) < TOL and rel_err(r[2], w64.grad) < TOL and rel_err(r[3], b64.grad) < TOL


def test_act(dev):
    x = T((1000, 7), 1, dev)
    y = ops.act_fwd(x, ops.ACT_LEAKY, 0.1)
    assert torch.equal(y, torch.nn.functional.leaky_relu(x, 0.1))
    dy = T((1000, 7), 2, dev)
    assert torch.equal(ops.act_bwd(dy, y, ops.ACT_LEAKY, 0.1), torch.where(y > 0, dy, 0.1 * dy))
    yr = ops.act_fwd(x, ops.ACT_RELU)
    assert torch.equal(ops.act_bwd(dy, yr, ops.ACT_RELU), torch.where(yr > 0, dy, torch.zeros_like(dy)))


def _rand_graph(N, E, seed):
    rng = np.random.default_rng(seed)
    src = rng.integers(0, N, size=E)
    dst = rng.integers(0, N, size=E)
    dst[:50] = 7                                   # one heavy in-degree row
    src[50:120] = 11                               # one heavy out-degree row
    return src, dst


@pytest.mark.parametrize('D', [16, 128])
def test_seg_reductions(dev, D):
    from mmft.pingraph import PinGraph
    N, E = 700, 2500
    src, dst = _rand_graph(N, E, 3)
    g = PinGraph(N, {'cell': (src, dst), 'net': (src[::2], dst[::2])}).to(dev)
    h = T((N, D), 1, dev, -3, 3)
    rows_np = np.random.default_rng(0).permutation(N)[:400]
    rows = torch.from_numpy(rows_np.astype(np.int32)).to(dev)
    ip, ix = g.csr_host('in', 'cell')
    ref = R.seg_softmax_sum(h.double().cpu(), ip, ix, rows_np)
    A = torch.zeros_like(h); LSE = torch.zeros_like(h)
    ops.seg_softmax_sum_fwd(h, g.csr('in', 'cell'), rows, A, LSE)
    assert rel_err(A[rows.long()], ref) < TOL
    # LSE against torch.logsumexp for a few rows with deg > 0
    for v in rows_np[:40]:
        if ip[v + 1] > ip[v]:
            l = torch.logsumexp(h.double().cpu()[ix[ip[v]:ip[v + 1]]], 0)
            assert rel_err(LSE[v], l) < TOL
    ipn, ixn = g.csr_host('in', 'net')
    refm = R.seg_mean(h.double().cpu(), ipn, ixn, rows_np)
    assert rel_err(ops.seg_mean_fwd(h, g.csr('in', 'net'), rows), refm) < TOL
    # in-place level update: sources must lie outside the updated rows (earlier levels), as in a sweep
    lo = np.arange(0, N // 2)
    ns = lo[np.random.default_rng(1).integers(0, lo.size, 900)]
    nd = np.random.default_rng(2).integers(N // 2, N, 900)
    g2 = PinGraph(N, {'net': (ns, nd), 'cell': ((), ())}).to(dev)
    rows2_np = np.arange(N // 2, N)[::-1].copy()
    rows2 = torch.from_numpy(rows2_np.astype(np.int32)).to(dev)
    h2 = h.clone()
    ops.seg_mean_add_act_fwd(h2, g2.csr('in', 'net'), rows2, relu=True)
    exp = h.double().cpu().clone()
    exp[rows2_np] = (exp[rows2_np] + R.seg_mean(h.double().cpu(), *g2.csr_host('in', 'net'), rows2_np)).clamp_min(0)
    assert rel_err(h2, exp) < TOL


def test_gather_scatter(dev):
    src = T((300, 128), 1, dev)
    idx = torch.from_numpy(det_ints((1000,), 2, 0, 300)).to(torch.int32).to(dev)
    assert torch.equal(ops.gather_rows(src, idx), src[idx.long()])
    dst = torch.zeros((300, 128), device=dev)
    upd = T((1000, 128), 3, dev)
    ops.scatter_add_rows(dst, idx, upd)
    ref = torch.zeros((300, 128), dtype=torch.float64).index_add_(0, idx.long().cpu(), upd.double().cpu())
    assert rel_err(dst, ref) < TOL


def test_level_bwd_pull_matches_autograd(dev):
    """One reverse-sweep level against torch autograd of the restated forward (fp64)."""
    from mmft.pingraph import PinGraph
    N, D = 300, 16
    rng = np.random.default_rng(5)
    # two-layer DAG: sources 0..149 feed sinks 150..299 through net and cell edges
    ns, nd = rng.integers(0, 150, 400), rng.integers(150, 225, 400)
    cs, cd = rng.integers(0, 150, 500), rng.integers(225, 300, 500)
    g = PinGraph(N, {'net': (ns, nd), 'cell': (cs, cd)}).to(dev)
    h64 = torch.from_numpy(det_uniform((N, D), 1, -2, 2)).double().requires_grad_(True)
    net_rows, cell_rows = np.arange(150, 225), np.arange(225, 300)
    a_net = R.seg_mean(h64, *g.csr_host('in', 'net'), net_rows)
    a_cell = R.seg_softmax_sum(h64, *g.csr_host('in', 'cell'), cell_rows)
    gn = torch.from_numpy(det_uniform((75, D), 2)).double()
    gc = torch.from_numpy(det_uniform((75, D), 3)).double()
    gt = torch.from_numpy(det_uniform((N, D), 4)).double()
    ((a_net * gn).sum() + (a_cell * gc).sum() + (h64 * gt).sum()).backward()
    relu_ref = torch.where(h64.detach() > 0, h64.grad, torch.zeros_like(h64.grad))[:150]

    h = h64.detach().float().to(dev)
    G = gt.float().to(dev).clone()            # target-gradient part already in G
    G[150:225] = gn.float().to(dev)           # consumers' d/d(pre-activation)
    A = torch.zeros_like(h); LSE = torch.zeros_like(h); DA = torch.zeros_like(h)
    cr = torch.arange(225, 300, dtype=torch.int32, device=dev)
    ops.seg_softmax_sum_fwd(h, g.csr('in', 'cell'), cr, A, LSE)
    DA[225:300] = gc.float().to(dev)
    rows = torch.arange(0, 150, dtype=torch.int32, device=dev)
    ops.level_bwd_pull(G, h, rows, g.csr('out', 'net'), g.out_net_weight(), g.csr('out', 'cell'), A, LSE, DA, relu=True)
    assert rel_err(G[:150], relu_ref) < 5e-5


@pytest.mark.parametrize('n', [1, 31, 32, 1000, 8064])
def test_mlp2_rows_fused(dev, n):
    """Fused Linear-ReLU-Linear over gathered rows (forward form and transposed-weight backward form)."""
    N = 9000
    A = T((N, 128), 1, dev); h0 = T((N, 128), 2, dev)
    w1, b1 = T((256, 128), 3, dev, -0.1, 0.1), T((256,), 4, dev)
    w2, b2 = T((128, 256), 5, dev, -0.1, 0.1), T((128,), 6, dev)
    rows = torch.randperm(N)[:n].to(torch.int32).to(dev)
    r = rows.long()
    HN = torch.zeros((N, 256), device=dev)
    h = h0.clone()
    ops.mlp2_rows(A, rows, w1, b1, w2, b2, h, kmajor=False, hid_out=HN, add_act=True, relu_out=True)
    hid = (A.double()[r] @ w1.double().t() + b1.double()).clamp_min(0)
    ref = h0.double().clone()
    ref[r] = (h0.double()[r] + hid @ w2.double().t() + b2.double()).clamp_min(0)
    assert rel_err(h, ref) < TOL
    assert rel_err(HN[r], hid) < TOL
    # backward form: DA[rows] = ((G[rows] W2) * (HN[rows] > 0)) W1, weights read transposed
    G = T((N, 128), 7, dev)
    DA = torch.zeros((N, 128), device=dev)
    ops.mlp2_rows(G, rows, w2, None, w1, None, DA, kmajor=True, mask=HN)
    dh = (G.double()[r] @ w2.double()) * (HN.double()[r] > 0)
    refd = torch.zeros((N, 128), dtype=torch.float64)
    refd[r.cpu()] = (dh @ w1.double()).cpu()
    assert rel_err(DA, refd) < TOL


@pytest.mark.parametrize('pooling', ['max', 'avg'])
@pytest.mark.parametrize('N,H,W,Ci', [(2, 8, 32, 16), (1, 64, 256, 16), (3, 6, 64, 32)])
def test_outconv_fused_vs_torch(dev, pooling, N, H, W, Ci):
    """OutConv (src/Unet.py:71-82: 1x1 conv -> pool -> ReLU) as one kernel per direction against torch in fp64:
    output, input gradient, weight and bias gradients; the unfused composition must agree with it too."""
    from mmft import cnn as C
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(1, Ci, 1, 1, generator=g) * 0.3
    b = torch.randn(1, generator=g) * 0.1
    gy = torch.randn(N, 1, H // 2, W // 2, generator=g)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    y = torch.nn.functional.conv2d(xd, wd, bd)
    y = torch.nn.functional.max_pool2d(y, 2) if pooling == 'max' else torch.nn.functional.avg_pool2d(y, 2)
    ref = torch.relu(y)
    ref.backward(gy.double())
    mode = ops.POOL_MAX if pooling == 'max' else ops.POOL_AVG
    xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wg, bg = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    assert ops.outconv_supported(xg, wg)
    out = C.outconv(xg, wg, bg, mode)
    assert out.shape == ref.shape and rel_err(out, ref) < 1e-5
    out.backward(gy.to(dev))
    assert rel_err(xg.grad, xd.grad) < 1e-5
    assert rel_err(wg.grad, wd.grad) < 1e-5 and rel_err(bg.grad, bd.grad) < 1e-5
    # the composition the other shapes take
    x2 = xg.detach().clone().requires_grad_(True)
    out2 = C.relu(C.pool2x2(C.conv2d(x2, wg, bg, pad=0), mode))
    assert rel_err(out2, out) < 1e-5
