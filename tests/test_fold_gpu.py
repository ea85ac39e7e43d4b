"""Folded forward chain (one gather launch per (net level, cell level) pair) and whole-workgroup reduction of heavy
rows in the forward gather and the reverse pull: equal to the plain per-level kernels and to the fp64 oracle."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import restatement as R

pytestmark = pytest.mark.gpu


def _run(pmodel, b, ends, fold):
    from mmft import sweep as S, ops
    S.FOLD_LEVELS = fold
    keep = ops.PAIR_HEAVY_OUT
    if not fold:                        # reference run: no folded gather, no heavy-row path in the reverse pull either
        ops.PAIR_HEAVY_OUT = 1 << 30
        b.graph._level_cache = {k: v for k, v in b.graph._level_cache.items() if not (isinstance(k, tuple) and k[0] == 'meta')}
    try:
        g = b.graph
        g.ndata['h'] = torch.zeros((b.N, 128), dtype=torch.float32, device=g.device)
        for p in pmodel.gnn.parameters():
            p.grad = None
        h = S.sweep_forward_all(pmodel.gnn, g, b.level_nodes, ends)
        used = g._sweep.fold is not None
        (h * torch.linspace(0.5, 1.5, h.shape[1], device=h.device)).sum().backward()
        return (h.detach().clone(), g.ndata['h'].clone(), g._sweep.G.clone(),
                {k: p.grad.clone() for k, p in pmodel.gnn.named_parameters() if p.grad is not None}, used)
    finally:
        S.FOLD_LEVELS = True
        ops.PAIR_HEAVY_OUT = keep
        if not fold:
            b.graph._level_cache = {k: v for k, v in b.graph._level_cache.items() if not (isinstance(k, tuple) and k[0] == 'meta')}


@pytest.mark.parametrize('fanin,N,L', [('regular', 6000, 12), ('irregular', 30000, 13), ('irregular', 30000, 16)])
def test_folded_chain_equals_per_level_kernels(dev, fanin, N, L):
    from mmft.synth import synth_design
    from mmft.train import build_models, DesignBatch
    designs = [synth_design(N=N, L=L, tile=32, seed=90 + i, fanin=fanin, end_frac=0.2) for i in range(2)]
    b = DesignBatch(designs, dev)
    pmodel, _ = build_models(map_size=designs[0].map_size, device=dev, seed=8)
    ends = b.select([np.arange(0, d.num_paths, 3) for d in designs])[0]
    sched = b.graph.fold_schedule(b.level_nodes)
    assert sched is not None
    if fanin == 'irregular':
        assert any(s['heavy_in'] is not None for s in sched)
        assert any(b.graph.level_meta(l, nodes)['heavy_out'] is not None for l, nodes in enumerate(b.level_nodes))
    out_f, h_f, G_f, grads_f, used_f = _run(pmodel, b, ends, True)
    out_u, h_u, G_u, grads_u, used_u = _run(pmodel, b, ends, False)
    assert used_f and not used_u
    # light rows are bitwise identical; heavy rows sum their partials in another (fixed) order
    assert rel_err(h_f, h_u) < 2e-6 and rel_err(out_f, out_u) < 2e-6
    assert rel_err(G_f, G_u) < 1e-5
    for k in grads_u:
        assert rel_err(grads_f[k], grads_u[k]) < 1e-5, k
    again = _run(pmodel, b, ends, True)
    assert torch.equal(again[0], out_f) and all(torch.equal(again[3][k], grads_f[k]) for k in grads_f)   # reproducible


def test_folded_chain_vs_oracle_irregular(dev):
    """fp64 oracle on a design with Zipf fan-in (up to 256 in-edges) and Zipf driver fan-out: embeddings and parameter
    gradients of the folded chain within 1e-4."""
    from mmft.synth import synth_design
    from mmft.train import build_models, DesignBatch
    d = synth_design(N=12000, L=14, tile=32, seed=31, fanin='irregular', end_frac=0.2)
    b = DesignBatch([d], dev)
    pmodel, _ = build_models(map_size=d.map_size, device=dev, seed=9)
    sel = b.select([np.arange(d.num_paths)])
    ends, ends_old = sel[0], sel[4]
    out, _, _, grads, used = _run(pmodel, b, ends, True)
    assert used
    p = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in pmodel.state_dict().items() if k.startswith('gnn.')}
    csr = R.design_csr(d)
    h = torch.zeros((d.N, 128), dtype=torch.float64)
    cf, nf = torch.from_numpy(d.cell_feat).double(), torch.from_numpy(d.net_feat).double()
    for l in range(d.L):
        h, _ = R.pathconv_level(p, 'gnn.', csr, h, cf, nf, d.levels[l], [], l)
    ref = h[torch.as_tensor(ends_old)]
    (ref * torch.linspace(0.5, 1.5, 128, dtype=torch.float64)).sum().backward()
    assert rel_err(out, ref) < 1e-4
    for k, v in p.items():
        if v.grad is not None:
            assert rel_err(grads[k[4:]], v.grad) < 1e-4, k


def test_fold_preconditions_fall_back(dev):
    """Graphs outside the folded kernels' preconditions (a net node with two drivers; levels that are not contiguous id
    ranges) take the per-level kernels."""
    from mmft.synth import synth_design
    from mmft.train import build_models, DesignBatch
    d = synth_design(N=3000, L=8, tile=32, seed=5, end_frac=0.2)
    b = DesignBatch([d], dev, renumber=False)
    assert b.graph.fold_schedule(b.level_nodes) is None
    # second driver for one net node
    lv1, lv0 = d.levels[1], d.levels[0]
    d.net_src = np.concatenate([d.net_src, lv0[:1]])
    d.net_dst = np.concatenate([d.net_dst, lv1[:1]])
    b2 = DesignBatch([d], dev)
    assert b2.graph.fold_schedule(b2.level_nodes) is None
    pmodel, _ = build_models(map_size=d.map_size, device=dev, seed=8)
    ends = b2.select([np.arange(d.num_paths)])[0]
    out, *_rest, used = _run(pmodel, b2, ends, True)
    assert not used and bool(torch.isfinite(out).all())
