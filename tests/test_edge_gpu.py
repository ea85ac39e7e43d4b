"""Edge cases and full-size properties of the hot path on the GPU.

Small cases are checked against the fp64 oracle; the BASELINE.json-size cases (65 536-node netlist, 256x256 tile)
through size-independent properties: bitwise run-to-run determinism (reverse sweep = pull, split-K slabs summed in
fixed order), permutation equivariance in the sampled endpoints, and drop-in == whole-sweep."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from mmft.synth import synth_design
from mmft.train import build_models, TrainStep
from mmft.fusion import mse_loss
from oracle import restatement as R

pytestmark = pytest.mark.gpu


def _oracle_forward(d, pmodel, cnn, ids, dtype=torch.float64, cnn_kind='unet', pooling='max'):
    pm = {k: v.detach().cpu().clone() for k, v in pmodel.state_dict().items()}
    pc = {k: v.detach().cpu().clone() for k, v in cnn.state_dict().items()}
    o = R.OracleTrainer(pm, pc, dtype=dtype, cnn_kind=cnn_kind, pooling=pooling)
    hats, tl, _ = o.forward(d, R.design_csr(d), ids)
    return o, hats, tl


def test_irregular_fanin_vs_oracle(dev):
    """Config-E style netlist (Pareto fan-in up to 256 inputs per cell, heavy-tailed driver fan-out)."""
    d = synth_design(N=16384, L=16, tile=32, fanin='irregular', seed=77, end_frac=0.05)
    assert np.bincount(d.cell_dst).max() >= 64
    pmodel, cnn = build_models(map_size=d.map_size, device=dev, seed=3)
    ids = np.random.default_rng(0).permutation(d.num_paths)[:200].tolist()
    o, hats_o, tl_o = _oracle_forward(d, pmodel, cnn, ids)
    loss_o = torch.nn.functional.mse_loss(hats_o, torch.from_numpy(d.arrival_time).double()[torch.tensor(tl_o)].squeeze(-1))
    loss_o.backward()
    ts = TrainStep(pmodel, cnn, [d], dev)
    hats, ends_d, ends_h = ts.forward([ids])
    loss = mse_loss(hats, ts.batch.arrival[ends_d.long()].squeeze(-1))
    ts.optim.zero_grad()
    loss.backward()
    assert ends_h.tolist() == tl_o
    assert rel_err(hats, hats_o) < 1e-4
    for k, p in pmodel.named_parameters():
        if o.pm[k].grad is not None:
            assert rel_err(p.grad, o.pm[k].grad) < 2e-4, k


def test_empty_levels_and_levels_without_targets(dev):
    """A level with zero nodes and levels without sampled endpoints must still advance the sweep (src/model.py:276-283)."""
    import model
    from mmft.pingraph import PinGraph
    d = synth_design(N=512, L=8, tile=16, seed=5, end_frac=0.5)
    pmodel, cnn = build_models(map_size=d.map_size, device=dev, seed=3)
    g = PinGraph.from_synth(d).to(dev)
    levels = [l.tolist() for l in d.levels]
    levels.insert(2, [])                        # an empty level keeps parity (even = cell) only if we add two
    levels.insert(2, [])
    with torch.no_grad():
        g.ndata['h'] = torch.zeros((d.N, 128), device=dev)
        outs = []
        for l, nodes in enumerate(levels):
            y = pmodel(g, nodes, None, [], l, torch.tensor([float(l)], device=dev), None)
            assert y is None
        h_a = g.ndata['h'].clone()
        g.ndata['h'] = torch.zeros((d.N, 128), device=dev)
        for l, nodes in enumerate(d.levels):
            pmodel(g, nodes.tolist(), None, [], l, torch.tensor([float(l)], device=dev), None)
        assert torch.equal(h_a, g.ndata['h'])
        # whole-sweep entry with T = 0
        g.ndata['h'] = torch.zeros((d.N, 128), device=dev)
        assert pmodel.forward_sweep(g, [l.tolist() for l in d.levels], torch.zeros(0, dtype=torch.int32, device=dev),
                                    torch.zeros(0, dtype=torch.int32, device=dev), None) is None
        assert rel_err(g.ndata['h'], h_a) < 1e-6          # other kernels (persistent sweep): same math, not bitwise


def test_sweep_order_is_enforced(dev):
    from mmft.pingraph import PinGraph
    d = synth_design(N=512, L=8, tile=16, seed=5)
    pmodel, _ = build_models(map_size=d.map_size, device=dev, seed=3)
    g = PinGraph.from_synth(d, out_dim=128).to(dev)
    with pytest.raises(RuntimeError, match='level 0'):
        pmodel.gnn(g, d.levels[1].tolist(), None, [], 1)
    pmodel.gnn(g, d.levels[0].tolist(), None, [], 0)
    with pytest.raises(RuntimeError, match='increasing order'):
        pmodel.gnn(g, d.levels[2].tolist(), None, [], 2)


def test_ragged_batch(dev):
    """Designs with different level counts and path counts in one block-diagonal batch."""
    ds = [synth_design(N=1024, L=8, tile=32, seed=21), synth_design(N=2048, L=14, tile=32, seed=22),
          synth_design(N=768, L=6, tile=32, seed=23)]
    rng = np.random.default_rng(2)
    ids = [rng.permutation(d.num_paths)[:n].tolist() for d, n in zip(ds, (10, 25, 3))]
    pmodel, cnn = build_models(map_size=ds[0].map_size, device=dev, seed=4)
    sd_m = {k: v.clone() for k, v in pmodel.state_dict().items()}
    sd_c = {k: v.clone() for k, v in cnn.state_dict().items()}
    ts = TrainStep(pmodel, cnn, ds, dev)
    with torch.no_grad():
        hats, ends_d, ends_h = ts.forward(ids)
    assert hats.numel() == 38
    for i, d in enumerate(ds):
        pm_i, cnn_i = build_models(map_size=d.map_size, device=dev, seed=4)
        pm_i.load_state_dict(sd_m); cnn_i.load_state_dict(sd_c)
        o, hats_o, tl_o = _oracle_forward(d, pm_i, cnn_i, ids[i])
        lo, hi = ts.batch.node_off[i], ts.batch.node_off[i + 1]
        sel = (ends_h >= lo) & (ends_h < hi)
        assert (ends_h[sel] - lo).tolist() == tl_o
        assert rel_err(hats[torch.from_numpy(sel).to(dev)], hats_o) < 1e-4


@pytest.mark.parametrize('kind,pooling', [('layoutnet', 'max'), ('unet', 'avg')])
def test_other_cnn_variants_in_the_step(dev, kind, pooling):
    """LayoutNet (the reference's default CNN without --unet, output H/4) and avg pooling through a full step."""
    unet = kind == 'unet'
    d = synth_design(N=1024, L=8, tile=32, seed=31, channels=3 if unet else 2, map_div=2 if unet else 4)
    pmodel, cnn = build_models(map_size=d.map_size, device=dev, seed=5, unet=unet, pooling=pooling)
    ids = np.random.default_rng(3).permutation(d.num_paths)[:20].tolist()
    o, hats_o, tl_o = _oracle_forward(d, pmodel, cnn, ids, cnn_kind=kind, pooling=pooling)
    loss_o = torch.nn.functional.mse_loss(hats_o, torch.from_numpy(d.arrival_time).double()[torch.tensor(tl_o)].squeeze(-1))
    loss_o.backward()
    ts = TrainStep(pmodel, cnn, [d], dev)
    hats, ends_d, _ = ts.forward([ids])
    loss = mse_loss(hats, ts.batch.arrival[ends_d.long()].squeeze(-1))
    ts.optim.zero_grad()
    loss.backward()
    assert rel_err(hats, hats_o) < 1e-4
    for k, p in cnn.named_parameters():
        assert rel_err(p.grad, o.pc[k].grad) < 3e-4, k


def test_full_size_properties(dev):
    """BASELINE config-B shape (65 536 nodes, 64 levels, 256x256 tile, 1350 endpoints), one design."""
    d = synth_design(N=65536, L=64, tile=256, seed=9294)
    ids = np.random.default_rng(0).permutation(d.num_paths)[:1350]

    def run(mode, order):
        pmodel, cnn = build_models(map_size=d.map_size, device=dev, seed=9294)
        ts = TrainStep(pmodel, cnn, [d], dev, mode=mode, overlap=False)
        hats, ends_d, ends_h = ts.forward([ids[order].tolist()])
        loss = mse_loss(hats, ts.batch.arrival[ends_d.long()].squeeze(-1))
        ts.optim.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        return hats.detach().clone(), ends_h.copy(), ts.optim.flat_grad.clone(), float(loss.detach())

    ident = np.arange(ids.shape[0])
    h1, e1, g1, l1 = run('sweep', ident)
    h2, e2, g2, l2 = run('sweep', ident)
    assert torch.equal(h1, h2) and torch.equal(g1, g2) and l1 == l2          # bitwise reproducible
    perm = np.random.default_rng(1).permutation(ids.shape[0])
    h3, e3, g3, l3 = run('sweep', perm)                                        # same endpoints, other sampling order
    a = dict(zip(e1.tolist(), h1.tolist()))
    b = dict(zip(e3.tolist(), h3.tolist()))
    assert a.keys() == b.keys() and max(abs(a[k] - b[k]) for k in a) < 1e-5
    assert abs(l1 - l3) < 1e-6 * max(1.0, abs(l1)) and rel_err(g3, g1) < 1e-4
    h4, e4, g4, l4 = run('dropin', ident)                                      # per-level API == whole-sweep entry
    assert e4.tolist() == e1.tolist() and rel_err(h4, h1) < 1e-5 and rel_err(g4, g1) < 1e-4
    assert np.isfinite(l1) and float(g1.abs().max()) > 0


@pytest.mark.parametrize('cfg', ['C', 'E'])
def test_large_configs_run_and_are_deterministic(dev, cfg):
    """BASELINE configs[2] / [4] shapes on one GPU: asap7-like 300k-node design and the 1M-node irregular-fan-in
    stress netlist, 512x512 tiles (map 256^2, P = 65 536).  Too large for the CPU oracle inside a test, so:
    finite results, bitwise reproducibility of predictions and gradients, drop-in == whole-sweep predictions."""
    from mmft.synth import config_design
    d = config_design(cfg)
    assert d.tile == 512 and d.map_size == 256
    ids = np.random.default_rng(0).permutation(d.num_paths)[:1350].tolist()
    res = []
    for mode in ('sweep', 'sweep', 'dropin'):
        pmodel, cnn = build_models(map_size=d.map_size, device=dev, seed=9294)
        ts = TrainStep(pmodel, cnn, [d], dev, mode=mode, overlap=False)
        hats, ends_d, ends_h = ts.forward([ids])
        loss = mse_loss(hats, ts.batch.arrival[ends_d.long()].squeeze(-1))
        ts.optim.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        res.append((hats.detach().clone(), ts.optim.flat_grad.clone(), float(loss.detach())))
        del ts, pmodel, cnn
        torch.cuda.empty_cache()
    assert np.isfinite(res[0][2]) and float(res[0][1].abs().max()) > 0
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert rel_err(res[2][0], res[0][0]) < 1e-5 and rel_err(res[2][1], res[0][1]) < 1e-4
