"""MMFT_MATH_BF16 (BASELINE.json configs[1]: "bf16"): the MFMA-bound contractions round their operands to bf16 and
accumulate in fp32; tensors in HBM stay fp32.

Two kinds of checks, both against fp64:
  * EXACTNESS of the kernels: with operands that are already bf16-representable the rounding is the identity, so a
    single contraction must agree with fp64 to fp32-accumulation accuracy (2e-6 of the result's scale) - this pins
    the bf16 MFMA fragment layouts, the in-register transposes and the staging of every operand loader;
  * TOLERANCE on arbitrary fp32 data: bf16 keeps 8 significant bits (relative rounding error <= 2^-9 per operand), the
    stated tolerance is 2e-2 of the result's scale for one contraction and for the fused two-layer kernels, and
    5e-2 on the predictions of a whole config-A train step (32 chained levels) with gradient directions within
    cos >= 0.98 (GNN, head) / 0.85 (U-Net, mean over all tensors >= 0.97) of the fp64 oracle's (rounding flips a few ReLU / max-pool decisions, so a
    gradient is compared by direction, not element by element).
fp32 stays the 1e-4 parity mode (every other GPU test)."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from mmft import lib, ops
from oracle import restatement as R

pytestmark = pytest.mark.gpu
EXACT, TOL = 2e-6, 2e-2


def bf(t):
    return t.bfloat16().float()


def rnd(*shape, seed=0, representable=False):
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(*shape, generator=g)
    return bf(t) if representable else t


@pytest.fixture(autouse=True)
def _bf16_mode():
    with lib.math_mode('bf16'):
        yield
    assert lib.get_math_mode() == 'f32'


@pytest.mark.parametrize('M,N,K', [(9, 7, 5), (300, 256, 36), (1000, 128, 256), (257, 576, 288), (4099, 32, 1), (64, 128, 2)])
@pytest.mark.parametrize('representable', [True, False])
def test_linear_fwd_dgrad(dev, M, N, K, representable):
    x, w, b = rnd(M, K, seed=1, representable=representable), rnd(N, K, seed=2, representable=representable), rnd(N, seed=3)
    tol = EXACT if representable else TOL
    y = ops.linear_fwd(x.to(dev), w.to(dev), b.to(dev), act=ops.ACT_RELU)
    ref = torch.relu(x.double() @ w.double().T + b.double())
    assert rel_err(y, ref) < tol
    g = rnd(M, N, seed=4, representable=representable)
    dx = ops.linear_dgrad(g.to(dev), w.to(dev))
    assert rel_err(dx, g.double() @ w.double()) < tol


@pytest.mark.parametrize('rows,out,inn', [(5, 7, 3), (1000, 128, 256), (40000, 256, 36), (30000, 128, 2), (70000, 256, 128)])
@pytest.mark.parametrize('representable', [True, False])
def test_linear_wgrad_with_bias(dev, rows, out, inn, representable):
    g, x = rnd(rows, out, seed=5, representable=representable), rnd(rows, inn, seed=6, representable=representable)
    dw, db = ops.linear_wgrad(g.to(dev), x.to(dev), with_bias=True)
    scale = EXACT * 8 if representable else TOL           # up to 70 000-term fp32 sums
    assert rel_err(dw, g.double().T @ x.double()) < scale
    # the bias gradient is taken from the fp32 staging registers: exact in both cases
    assert rel_err(db, g.double().sum(0)) < 1e-5
    # indexed rows (gather on the reduction axis)
    idx = torch.randperm(rows)[:max(rows // 2, 1)].to(torch.int32)
    dw2 = ops.linear_wgrad(g.to(dev), x.to(dev), gidx=idx.to(dev), xidx=idx.to(dev))
    assert rel_err(dw2, g[idx.long()].double().T @ x[idx.long()].double()) < scale


@pytest.mark.parametrize('N,H,W,Ci,Co,k', [(2, 9, 64, 16, 16, 3), (1, 64, 128, 16, 32, 3), (2, 33, 64, 32, 32, 3),
                                           (2, 16, 16, 64, 128, 3), (1, 12, 20, 3, 16, 3), (1, 24, 24, 2, 32, 9),
                                           (2, 8, 8, 16, 1, 1),
                                           # H % 4 == 0, W % 64 == 0, 16 / 32 channels: the tile-resident weight-gradient kernel
                                           (2, 8, 64, 16, 16, 3), (1, 12, 128, 32, 16, 3), (3, 36, 64, 32, 32, 3),
                                           (1, 128, 256, 16, 16, 3), (2, 16, 64, 3, 16, 3)])
@pytest.mark.parametrize('representable', [True, False])
def test_conv_forward_dgrad_wgrad(dev, N, H, W, Ci, Co, k, representable):
    x = rnd(N, Ci, H, W, seed=7, representable=representable)
    w = rnd(Co, Ci, k, k, seed=8, representable=representable) * (0.25 if not representable else 1.0)
    if representable:
        w = bf(w)
    gy = rnd(N, Co, H, W, seed=9, representable=representable)
    pad = k // 2
    tol = EXACT * 4 if representable else TOL
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(xd, wd, None, padding=pad)
    ref.backward(gy.double())
    xn, gn = ops.to_nhwc(x.to(dev)), ops.to_nhwc(gy.to(dev))
    wg = w.to(dev).contiguous(memory_format=torch.channels_last)
    y = ops.conv2d_fwd(xn, wg, None, pad)
    assert rel_err(y, ref) < tol
    dx = ops.conv2d_dgrad(gn, wg, pad)
    assert rel_err(dx, xd.grad) < tol
    dw = ops.conv2d_wgrad(xn, gn, k, k, pad).permute(0, 3, 1, 2)
    assert rel_err(dw, wd.grad) < tol


@pytest.mark.parametrize('n', [1, 31, 1000, 8064])
def test_mlp2_rows_fused(dev, n):
    """Fused Linear-ReLU-Linear of the level chain: the hidden tile is rounded to bf16 between the layers."""
    N = n + 50
    x1, w1, b1 = rnd(N, 128, seed=10), rnd(256, 128, seed=11) * 0.1, rnd(256, seed=12) * 0.1
    w2, b2 = rnd(128, 256, seed=13) * 0.1, rnd(128, seed=14) * 0.1
    rows = torch.randperm(N)[:n].to(torch.int32)
    old = rnd(N, 128, seed=15)
    out, hid = old.clone().to(dev), torch.zeros(N, 256, device=dev)
    ops.mlp2_rows(x1.to(dev), rows.to(dev), w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev), out, hid_out=hid, add_act=True,
                  relu_out=True)
    r = rows.long()
    h = torch.relu(x1[r].double() @ w1.double().T + b1.double())
    ref = old.double().clone()
    ref[r] = torch.relu(old[r].double() + h @ w2.double().T + b2.double())
    assert rel_err(hid[r.to(dev)], h) < TOL
    assert rel_err(out, ref) < TOL
    # reverse form: weights read transposed, ReLU mask from the saved hidden activations
    g = rnd(N, 128, seed=16)
    da = torch.zeros(N, 128, device=dev)
    ops.mlp2_rows(g.to(dev), rows.to(dev), w2.to(dev), None, w1.to(dev), None, da, kmajor=True, mask=hid)
    # the ReLU mask is the kernel's own saved hidden tile (bf16 rounding flips a few pre-activations near zero)
    dref = ((g[r].double() @ w2.double()) * (hid[r.to(dev)].cpu() > 0)) @ w1.double()
    assert rel_err(da[r.to(dev)], dref) < TOL


@pytest.mark.parametrize('n', [1, 33, 1000, 8064])
@pytest.mark.parametrize('representable', [True, False])
def test_mlp2_rows_prepacked(dev, n, representable):
    """The lean level kernel with pre-packed bf16 weights (mmft_pack_bf16 + mmft_mlp2_rows_bf16), forward and reverse
    form (transposed packs).  With bf16-representable x and weights the FIRST layer is exact (2e-6); the hidden tile is
    rounded to bf16 between the layers, so the second layer carries the stated bf16 tolerance."""
    N = n + 50
    x1, w1, b1 = rnd(N, 128, seed=10, representable=representable), bf(rnd(256, 128, seed=11) * 0.1), rnd(256, seed=12) * 0.1
    w2, b2 = bf(rnd(128, 256, seed=13) * 0.1), rnd(128, seed=14) * 0.1
    rows = torch.randperm(N)[:n].to(torch.int32)
    old = rnd(N, 128, seed=15)
    out, hid = old.clone().to(dev), torch.zeros(N, 256, device=dev)
    w1d, w2d = w1.to(dev), w2.to(dev)
    p1, p2 = ops.pack_bf16(w1d), ops.pack_bf16(w2d)
    assert torch.equal(p1.float().cpu(), w1) and torch.equal(ops.pack_bf16(w2d, transpose=True).float().cpu(), w2.T.contiguous())
    ops.mlp2_rows_bf16(x1.to(dev), rows.to(dev), p1, b1.to(dev), p2, b2.to(dev), out, hid_out=hid, add_act=True, relu_out=True)
    r = rows.long()
    h = torch.relu(x1[r].double() @ w1.double().T + b1.double())
    assert rel_err(hid[r.to(dev)], h) < (EXACT if representable else TOL)
    hk = hid[r.to(dev)].cpu().double()                        # second layer from the kernel's own hidden tile
    ref = old.double().clone()
    ref[r] = torch.relu(old[r].double() + bf(hk.float()).double() @ w2.double().T + b2.double())
    assert rel_err(out, ref) < 1e-5                                      # exact given the bf16-rounded hidden tile
    untouched = torch.ones(N, dtype=torch.bool); untouched[r] = False
    assert torch.equal(out.cpu()[untouched], old[untouched])
    # reverse form
    g = rnd(N, 128, seed=16, representable=representable)
    da = torch.zeros(N, 128, device=dev)
    dh = torch.zeros(N, 256, device=dev)
    ops.mlp2_rows_bf16(g.to(dev), rows.to(dev), ops.pack_bf16(w2d, transpose=True), None, ops.pack_bf16(w1d, transpose=True), None,
                       da, mask=hid, hid_out=dh)
    dhr = (g[r].double() @ w2.double()) * (hid[r.to(dev)].cpu() > 0)
    assert rel_err(dh[r.to(dev)], dhr) < (EXACT if representable else TOL)
    assert rel_err(da[r.to(dev)], bf(dh[r.to(dev)].cpu()).double() @ w1.double()) < 1e-5
    # cone mask: rows outside it are left alone
    active = torch.zeros(N, dtype=torch.uint8); active[r[::2]] = 1
    out2 = old.clone().to(dev)
    ops.mlp2_rows_bf16(x1.to(dev), rows.to(dev), p1, b1.to(dev), p2, b2.to(dev), out2, add_act=True, relu_out=True, active=active.to(dev))
    assert torch.equal(out2.cpu()[r[::2]], out.cpu()[r[::2]]) and torch.equal(out2.cpu()[r[1::2]], old[r[1::2]])


@pytest.mark.parametrize('fin,n', [(36, 5000), (2, 4099), (48, 37)])
def test_first_layer_grads_fused(dev, fin, n):
    g, hid, x, w2 = rnd(n, 128, seed=17), rnd(n, 256, seed=18), rnd(n, fin, seed=19), rnd(128, 256, seed=20) * 0.1
    dw1, db1 = ops.mlp2_first_layer_grads(g.to(dev), hid.to(dev), x.to(dev), None, w2.to(dev))
    dh = (g.double() @ w2.double()) * (hid > 0)
    assert rel_err(dw1, dh.T @ x.double()) < TOL
    assert rel_err(db1, dh.sum(0)) < TOL


def _cos(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def test_config_a_step_vs_oracle_bf16(dev):
    """One full config-A train step (U-Net on its bf16-STORAGE path, 32-level sweep, fusion head, MSE, backward) in bf16 mode
    against the fp64 oracle: predictions within 5e-2 of their scale, loss within 10 %, gradient directions within
    cos >= 0.98 (GNN, fusion head) / >= 0.85 per U-Net tensor with the mean over all tensors >= 0.95 (activations and
    activation gradients rounded to bf16 at each of the 14 BatchNorm layers; 0.967 measured, 0.977 with fp32 storage)."""
    from mmft.synth import config_design
    from mmft.train import build_models, TrainStep
    from mmft.fusion import mse_loss
    d = config_design('A')
    pmodel, cnn = build_models(map_size=d.map_size, device=dev, seed=9294)
    pm_state = {k: v.detach().cpu().clone() for k, v in pmodel.state_dict().items()}
    pc_state = {k: v.detach().cpu().clone() for k, v in cnn.state_dict().items()}
    oracle = R.OracleTrainer(pm_state, pc_state, dtype=torch.float64)
    path_ids = np.random.default_rng(1).permutation(d.num_paths)[:100].tolist()
    hats_o, tl_o, _ = oracle.forward(d, R.design_csr(d), path_ids)
    arr_o = torch.from_numpy(d.arrival_time).double()[torch.tensor(tl_o)].squeeze(-1)
    loss_o = torch.nn.functional.mse_loss(hats_o, arr_o)
    loss_o.backward()
    ts = TrainStep(pmodel, cnn, [d], dev)
    hats, ends_d, ends_h = ts.forward([path_ids])
    loss = mse_loss(hats, ts.batch.arrival[ends_d.long()].squeeze(-1))
    ts.optim.zero_grad()
    loss.backward()
    assert ends_h.tolist() == tl_o
    assert rel_err(hats, hats_o) < 5e-2
    assert abs(float(loss) - float(loss_o)) < 0.1 * float(loss_o)
    cos = {}
    for k, prm in list(pmodel.named_parameters()) + list(cnn.named_parameters()):
        o = oracle.pm.get(k, oracle.pc.get(k))
        if o is None or o.grad is None or float(o.grad.abs().max()) == 0.0:
            continue
        cos[k] = _cos(prm.grad, o.grad)
    head = [c for k, c in cos.items() if k.startswith(('fcn', 'mlp_', 'gnn.'))]
    assert min(head) > 0.98, sorted(cos.items(), key=lambda kv: kv[1])[:5]
    # the U-Net's first layers sit behind 14 BatchNorm + ReLU + max-pool stages whose decisions a rounding can flip
    assert min(cos.values()) > 0.85 and np.mean(list(cos.values())) > 0.95, sorted(cos.items(), key=lambda kv: kv[1])[:5]
    # and the mode really changes the arithmetic: fp32 mode is >100x closer
    with lib.math_mode('f32'):
        hats32, _, _ = ts.forward([path_ids])
    assert rel_err(hats32, hats_o) < 1e-4 < rel_err(hats, hats_o)


def test_training_in_bf16_tracks_fp32(dev):
    """40 optimizer steps from the same initialisation on the same batches: the bf16 run's loss curve stays within 15 % of
    the fp32 run's and the held-out endpoint-slack MAE within 20 % (the drift bench.py reports)."""
    from mmft.synth import synth_design
    from mmft.train import build_models, TrainStep
    from mmft.evaluate import validate
    designs = [synth_design(N=4096, L=16, tile=64, seed=500 + i, end_frac=0.25) for i in range(2)]
    held = synth_design(N=4096, L=16, tile=64, seed=777, end_frac=0.25)
    rng = np.random.default_rng(4)
    batches = [[rng.permutation(d.num_paths)[:64].tolist() for d in designs] for _ in range(40)]
    res = {}
    for mode in ('f32', 'bf16'):
        with lib.math_mode(mode):
            pmodel, cnn = build_models(map_size=designs[0].map_size, device=dev, seed=3)
            ts = TrainStep(pmodel, cnn, designs, dev)
            losses = [float(ts.step(ids)[0]) for ids in batches]
            ev = TrainStep(pmodel, cnn, [held], dev, overlap=False, with_optimizer=False)
            res[mode] = (losses, validate(ev)['endpoint_slack_mae'])
    l32, l16 = np.array(res['f32'][0]), np.array(res['bf16'][0])
    assert l16[-1] < 0.5 * l16[0]                                         # it trains
    assert np.abs(l16[-10:].mean() - l32[-10:].mean()) < 0.15 * l32[-10:].mean() + 1e-4
    assert abs(res['bf16'][1] - res['f32'][1]) < 0.2 * res['f32'][1] + 1e-3


@pytest.mark.parametrize('shape', [(6000, 12), (40000, 8)])
def test_fused_level_kernel_equals_two_kernel_form(dev, shape):
    """bf16 mode: the three forms of the forward level chain - two kernels (mmft_pair_fwd_gather, then mmft_mlp2_rows_bf16),
    the fused kernel (mmft_level_fwd_bf16) and its slot-table form (mmft_level_fwd_slots: static per-row edge slots, net rows
    inside the cell workgroups, early read-modify-write operand) - run the same instruction sequences per row: bitwise
    equal embeddings, saved state and gradients.  Levels of 1 000 rows take the slot kernel's 16-row workgroups, levels of
    10 000 rows its 32-row workgroups (two blocks per workgroup from 6 144 rows on)."""
    from mmft import sweep as S
    from mmft.synth import synth_design
    from mmft.train import build_models, DesignBatch
    designs = [synth_design(N=shape[0], L=shape[1], tile=32, seed=120 + i, end_frac=0.2) for i in range(2)]
    b = DesignBatch(designs, dev)
    pmodel, _ = build_models(map_size=designs[0].map_size, device=dev, seed=8)
    ends = b.select([np.arange(0, d.num_paths, 3) for d in designs])[0]
    res, names = [], []
    for fuse, slots in ((True, True), (True, False), (False, False)):
        S.FUSE_LEVEL_FWD, S.LEVEL_SLOTS = fuse, slots
        try:
            g = b.graph
            g.ndata['h'] = torch.zeros((b.N, 128), dtype=torch.float32, device=dev)
            for p in pmodel.gnn.parameters():
                p.grad = None
            lib.prof_reset()
            lib.prof_enable(True)
            out = S.sweep_forward_all(pmodel.gnn, g, b.level_nodes, ends)
            torch.cuda.synchronize()
            lib.prof_enable(False)
            names.append({r['name'] for r in lib.prof_report()})
            st = g._sweep
            assert st.wpack is not None and st.fold is not None
            (out * out).sum().backward()
            res.append((out.detach().clone(), g.ndata['h'].clone(), st.A.clone(), st.HN.clone(), st.LSE.clone(),
                        {k: p.grad.clone() for k, p in pmodel.gnn.named_parameters() if p.grad is not None}))
        finally:
            S.FUSE_LEVEL_FWD, S.LEVEL_SLOTS = True, True
    assert 'level_fwd_slots_kernel' in names[0] and 'level_fwd_bf16_kernel' not in names[0]
    assert 'level_fwd_bf16_kernel' in names[1] and 'level_fwd_slots_kernel' not in names[1]
    assert 'pair_fwd_gather_kernel' in names[2] and 'level_fwd_bf16_kernel' not in names[2]
    for other in (res[0], res[1]):
        for x, y in zip(other[:5], res[2][:5]):
            assert torch.equal(x, y)
        for k in res[2][5]:
            assert torch.equal(other[5][k], res[2][5][k]), k


@pytest.mark.parametrize('parts', [False, True])
@pytest.mark.parametrize('fanout_skew', [False, True])
def test_paired_reverse_level_kernel_equals_three_kernel_form(dev, fanout_skew, parts):
    """bf16 mode: the reverse sweep with ONE launch per (cell level, net level above it) pair (mmft_level_bwd_pair: both pulls
    and fc_cell_neigh's backward) against the three-launch form (mmft_level_bwd_pull twice, mmft_mlp2_rows_bf16 reversed).
    parts=False: every driver whole inside one tile - the same additions in the same order as the pulls (run without their
    workgroup-per-heavy-row path): bitwise equal G, DA, hidden gradients and parameter gradients.  parts=True (the shipped
    setting): drivers with more than 48 sinks are cut into parts of 32 whose partial sums are added in part order by whichever
    workgroup arrives last - another summation order for those rows (1e-5), and two runs must agree bit for bit.  With
    fanout_skew some pins have more than four cell consumers (the CSR tail of the slot table)."""
    from mmft import sweep as S
    from mmft.pingraph import PinGraph
    from mmft.synth import synth_design
    from mmft.train import build_models, DesignBatch
    kw = dict(fanin='irregular') if fanout_skew else {}
    designs = [synth_design(N=9000, L=12, tile=32, seed=220 + i, end_frac=0.2, **kw) for i in range(2)]
    b = DesignBatch(designs, dev)
    sinks0 = PinGraph.BWD_PAIR_TILE_SINKS
    heavy0 = ops.PAIR_HEAVY_OUT
    try:
        if not parts:
            PinGraph.BWD_PAIR_TILE_SINKS = 1 << 30
        pairs = b.graph.level_bwd_pairs(b.level_nodes)
        assert pairs is not None and all(p is not None for p in pairs[1])
        fan = np.diff(b.graph.csr_host('out', 'net')[0])
        assert fan.max() > 64
        nheavy = sum(int((p['tiles'][:, 3] > 0).sum()) for p in pairs[1])
        assert (nheavy > 0) == parts
        if fanout_skew:
            assert int((pairs[0][:, 3] <= -2).sum()) > 0
        pmodel, _ = build_models(map_size=designs[0].map_size, device=dev, seed=9)
        ends = b.select([np.arange(0, d.num_paths, 3) for d in designs])[0]
        all_rows = torch.as_tensor(np.concatenate([np.asarray(x) for x in b.level_nodes]), device=dev).long()
        rows2 = torch.as_tensor(np.concatenate([np.asarray(x) for x in b.level_nodes[2::2]]), device=dev).long()
        res, names = [], []
        for paired in (True, True, False):
            S.LEVEL_BWD_PAIRS = paired
            ops.PAIR_HEAVY_OUT = 1 << 30
            g = b.graph
            g.__dict__.get('_level_cache', {}).clear()
            g.ndata['h'] = torch.zeros((b.N, 128), dtype=torch.float32, device=dev)
            for p in pmodel.gnn.parameters():
                p.grad = None
            out = S.sweep_forward_all(pmodel.gnn, g, b.level_nodes, ends)
            st = g._sweep
            assert st.wpack is not None and st.fold is not None
            lib.prof_reset()
            lib.prof_enable(True)
            (out * out).sum().backward()
            torch.cuda.synchronize()
            lib.prof_enable(False)
            names.append({r['name'] for r in lib.prof_report()})
            res.append((st.G[all_rows].clone(), st.DA[rows2].clone(), st.DHN[rows2].clone(),
                        {k: p.grad.clone() for k, p in pmodel.gnn.named_parameters() if p.grad is not None}))
    finally:
        S.LEVEL_BWD_PAIRS = True
        ops.PAIR_HEAVY_OUT = heavy0
        PinGraph.BWD_PAIR_TILE_SINKS = sinks0
    assert 'level_bwd_pair_kernel' in names[0] and 'level_bwd_pull_kernel' not in names[0]
    assert 'level_bwd_pull_kernel' in names[2] and 'level_bwd_pair_kernel' not in names[2]
    assert int(pairs[3].abs().sum()) == 0                                    # the part counters are back at zero
    assert float(res[0][0].abs().max()) > 0 and float(res[0][1].abs().max()) > 0
    for i in range(3):                                                       # two paired runs: bit for bit
        assert torch.equal(res[0][i], res[1][i])
    for k in res[0][3]:
        assert torch.equal(res[0][3][k], res[1][3][k]), k
    if not parts:
        for i in range(3):
            assert torch.equal(res[0][i], res[2][i])
        for k in res[2][3]:
            assert torch.equal(res[0][3][k], res[2][3][k]), k
    else:
        for i in range(3):
            assert rel_err(res[0][i], res[2][i]) < 1e-5
        for k in res[2][3]:
            assert rel_err(res[0][3][k], res[2][3][k]) < 1e-4, k


@pytest.mark.parametrize('fin,n,row0', [(36, 1000, 7), (2, 4099, 0), (36, 70000, 128), (5, 33, 3)])
@pytest.mark.parametrize('representable', [True, False])
def test_feature_mlp_without_hidden_tensor(dev, fin, n, row0, representable):
    """mmft_mlp2_feat_fwd_bf16 / _bwd_bf16 (fc_cell_self / fc_net_self with the hidden activations recomputed instead of
    stored) against fp64 math that rounds where the kernels round: operands, the hidden tile and the hidden gradient go
    through bf16, everything accumulates in fp32."""
    N = row0 + n + 5
    x = rnd(N, fin, seed=1, representable=representable)
    w1 = rnd(256, fin, seed=2, representable=representable) * 0.25
    b1 = rnd(256, seed=3, representable=representable) * 0.25
    w2 = rnd(128, 256, seed=4, representable=representable) * 0.125
    b2 = rnd(128, seed=5, representable=representable)
    g = rnd(N, 128, seed=6, representable=representable)
    if representable:
        w1, b1, w2 = bf(w1), bf(b1), bf(w2)
    sl = slice(row0, row0 + n)
    xb, w1b, w2b, gb = bf(x).double(), bf(w1).double(), bf(w2).double(), bf(g).double()
    pre = xb[sl] @ w1b.T + b1.double()
    H = bf(torch.relu(pre).float()).double()                      # the hidden tile is stored as bf16 in LDS
    ref = H @ w2b.T + b2.double()
    out = torch.full((N, 128), 7.0, device=dev)
    ops.mlp2_feat_fwd_bf16(x.to(dev), (row0, n), w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev), out)
    # the kernels round the hidden tile / hidden gradient to bf16 from an fp32 accumulation, this reference from the exact
    # value: among 10^7 hidden values a few land on the other side of a rounding boundary (one bf16 ulp = 2^-8 relative
    # each), so even with representable operands the bar is 1e-3 of the result's scale - a wrong fragment layout is O(1)
    tol = 1e-3 if representable else TOL
    assert rel_err(out[sl], ref) < tol
    assert float((out[:row0] - 7.0).abs().max()) == 0.0 if row0 else True          # rows outside the range are not touched
    assert float((out[row0 + n:] - 7.0).abs().max()) == 0.0
    dH = bf(((gb[sl] @ w2b) * (pre > 0)).float()).double()        # ... and so is the hidden gradient
    dw1, db1, dw2, db2 = ops.mlp2_feat_bwd_bf16(g.to(dev), x.to(dev), (row0, n), w1.to(dev), b1.to(dev), w2.to(dev))
    scale = tol
    assert rel_err(dw2, gb[sl].T @ H) < scale
    assert rel_err(db2, g[sl].double().sum(0)) < 1e-5             # from the fp32 staging registers: exact in both cases
    assert rel_err(dw1, dH.T @ xb[sl]) < scale
    assert rel_err(db1, ((gb[sl] @ w2b) * (pre > 0)).sum(0)) < scale


def test_full_size_config_b_in_bf16_mode(dev):
    """the configuration bench.py times - config B at full size (8 x 65 536 nodes, 64 levels, 256 x 256
    tiles, 1350 endpoints per design) in bf16 math mode - under test: two independent runs of two optimizer steps end
    bitwise equal, the replayed HIP graph follows the eager step, every parameter stays finite, the slot-table level kernel and
    the bf16-storage U-Net convolutions are the kernels that ran, and the first step's predictions are within the stated 5e-2 (of the
    predictions' scale) of the fp32 run of the same step."""
    from mmft.synth import synth_design
    from mmft.train import build_models, TrainStep, GraphedTrainStep
    designs = [synth_design(N=65536, L=64, tile=256, seed=9294 + i) for i in range(8)]
    rng = np.random.default_rng(6)
    batches = [[rng.permutation(d.num_paths)[:1350] for d in designs] for _ in range(2)]
    L = lib.load()

    def run(kind, mode, profile=False):
        with lib.math_mode(mode):
            pmodel, cnn = build_models(map_size=designs[0].map_size, device=dev, seed=9294)
            ts = TrainStep(pmodel, cnn, designs, dev)
            stepper = GraphedTrainStep(ts, batches[0], warmup=0) if kind == 'graph' else ts
            if profile:
                lib.prof_reset()
                lib.prof_enable(True)
            first = stepper.step(batches[0])
            hats0 = first[1].clone()
            out = stepper.step(batches[1])
            torch.cuda.synchronize()
            names = {r['name'] for r in lib.prof_report()} if profile else set()
            if profile:
                lib.prof_enable(False)
            res = (float(out[0]), out[1].clone(), ts.optim.flat_param.clone(), hats0, first[2])
            del ts, stepper, pmodel, cnn
        return res, names

    (a, names), (b, _), (g, _) = run('eager', 'bf16', profile=True), run('eager', 'bf16'), run('graph', 'bf16')
    # the kernels BENCH times are the ones this test ran
    assert any(n.startswith('level_fwd_slots_kernel') for n in names), sorted(names)
    assert any(n.startswith('u16_conv3x3_kernel') for n in names) and any(n.startswith('u16_conv3x3_wgrad_kernel') for n in names), sorted(names)
    assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])               # bitwise run to run
    assert abs(g[0] - a[0]) < 1e-4 * abs(a[0]) and rel_err(g[1], a[1]) < 1e-4                 # graph replay = eager
    assert bool(torch.isfinite(a[2]).all()) and bool(torch.isfinite(a[1]).all())
    (f, _) = run('eager', 'f32')
    assert f[4] == a[4]                                                                       # same endpoints, same order
    assert rel_err(a[3], f[3]) < 5e-2                      # stated bf16 tolerance: first-step predictions vs the fp32 run
    assert rel_err(a[3], f[3]) > 1e-6                      # ... and bf16 mode really was a different arithmetic


@pytest.mark.parametrize('out,inn,rows,g16,x16', [(128, 256, 5000, False, True), (256, 128, 4099, True, False), (128, 256, 777, True, True)])
def test_rows_outer_with_bf16_stored_operands(dev, out, inn, rows, g16, x16):
    """The same contraction with an operand STORED as bf16 (the level MLP's hidden activations / hidden gradients,
    sweep.HIDDEN_BF16): bitwise the result of the fp32-stored operand, which the kernel rounds to the same bf16 while staging -
    except the bias gradient of a bf16-stored g, which sums the rounded values; padded row pitch; and the fallback that widens
    the operand when the row-contraction kernel is switched off."""
    g = rnd(rows, out, seed=41)
    x = rnd(rows, inn, seed=42)
    gd = torch.zeros(rows, out + 8, device=dev)[:, :out].copy_(g.to(dev))
    xd = torch.zeros(rows, inn + 8, device=dev)[:, :inn].copy_(x.to(dev))
    g_in = torch.zeros(rows, out + 8, device=dev, dtype=torch.bfloat16)[:, :out].copy_(gd) if g16 else gd
    x_in = torch.zeros(rows, inn + 8, device=dev, dtype=torch.bfloat16)[:, :inn].copy_(xd) if x16 else xd
    ops.ROWS_OUTER = True
    dw_ref, db_ref = ops.linear_wgrad(bf(g).to(dev) if rows < 4096 else gd, bf(x).to(dev) if rows < 4096 else xd, with_bias=True)
    lib.prof_reset()
    lib.prof_enable(True)
    dw, db = ops.linear_wgrad(g_in, x_in, with_bias=True)
    torch.cuda.synchronize()
    lib.prof_enable(False)
    assert any(r['name'].startswith('rows_outer_kernel') for r in lib.prof_report())
    if rows >= 4096:
        assert torch.equal(dw, dw_ref)
    assert rel_err(dw, bf(g).double().T @ bf(x).double()) < TOL
    assert rel_err(db, (bf(g) if g16 else g).double().sum(0)) < 1e-5
    ops.ROWS_OUTER = False
    try:
        dw3, db3 = ops.linear_wgrad(g_in, x_in, with_bias=True)
    finally:
        ops.ROWS_OUTER = True
    assert rel_err(dw3, dw) < 1e-3 and rel_err(db3, db) < 1e-5


@pytest.mark.parametrize('out,inn,rows,ld_pad', [(128, 256, 5000, 0), (256, 128, 4099, 8), (128, 256, 70001, 0)])
@pytest.mark.parametrize('representable', [True, False])
def test_rows_outer_weight_gradient(dev, out, inn, rows, ld_pad, representable):
    """mmft_rows_outer_bf16 (dw = g^T x, db = column sums of g; the fc_cell_neigh weight gradients of the bf16 mode, operands
    transposed by ds_read_b64_tr_b16): exact layouts on bf16-representable operands, 2e-2 of the result's scale otherwise;
    rows not a multiple of the 32-row step, padded leading dimensions, accumulate = 1; reached through ops.linear_wgrad."""
    g = rnd(rows, out + ld_pad, seed=31, representable=representable)[:, :out]
    x = rnd(rows, inn + ld_pad, seed=32, representable=representable)[:, :inn]
    gd, xd = g.to(dev), x.to(dev)
    if ld_pad:
        gd = torch.zeros(rows, out + ld_pad, device=dev)[:, :out].copy_(g.to(dev))
        xd = torch.zeros(rows, inn + ld_pad, device=dev)[:, :inn].copy_(x.to(dev))
    lib.prof_reset()
    lib.prof_enable(True)
    dw, db = ops.linear_wgrad(gd, xd, with_bias=True)
    torch.cuda.synchronize()
    lib.prof_enable(False)
    assert any(r['name'].startswith('rows_outer_kernel') for r in lib.prof_report())
    gb, xb = bf(g).double(), bf(x).double()
    tol = (1e-5 if rows > 20000 else 2e-6) if representable else TOL      # fp32 accumulation over up to 70 001 rows
    assert rel_err(dw, gb.T @ xb) < tol
    assert rel_err(db, g.double().sum(0)) < 1e-5                      # from the fp32 staging registers
    dw2, db2 = ops.linear_wgrad(gd, xd, dw=dw.clone(), db=db.clone(), accumulate=True)
    assert rel_err(dw2, 2 * (gb.T @ xb)) < tol and rel_err(db2, 2 * g.double().sum(0)) < 1e-5
    ops.ROWS_OUTER = False
    try:
        dw3, _ = ops.linear_wgrad(gd, xd, with_bias=True)            # the generic engine, same math mode
    finally:
        ops.ROWS_OUTER = True
    assert rel_err(dw3, dw) < (tol if representable else 1e-3)
