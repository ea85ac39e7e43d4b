"""GPU preprocessing kernels (mmft.prep) against the CPU restatement: integer outputs bit-exact, min-max scaling
bit-exact in fp32; plus the size-independent property at BASELINE sizes that levelizing a synthetic design from its
level-0 nodes reproduces the generator's levels."""
import numpy as np
import pytest
import torch

from oracle import prep_restatement as PR
from test_prep_cpu import random_dag
from conftest import rel_err

pytestmark = pytest.mark.gpu


def csr(n, rows, cols, dev):
    order = np.argsort(rows, kind='stable')
    ip = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=ip[1:])
    return torch.from_numpy(ip.astype(np.int32)).to(dev), torch.from_numpy(cols[order].astype(np.int32)).to(dev)


@pytest.mark.parametrize('n,e,seed', [(50, 120, 0), (3000, 12000, 1), (20000, 50000, 2)])
def test_levelize_trace_masks_vs_restatement(dev, n, e, seed):
    from mmft import prep
    src, dst = random_dag(n, e, seed)
    suc, pre = PR.adjacency(n, src, dst)
    indeg = np.bincount(dst, minlength=n)
    pis = np.array([v for v in range(n) if indeg[v] == 0 and suc[v]][: max(4, n // 20)], dtype=np.int64)
    levels, remaining = PR.cal_topo_level(suc, pis.tolist())
    level, nl = prep.levelize([csr(n, src, dst, dev)], n, torch.from_numpy(pis.astype(np.int32)).to(dev))
    assert nl == len(levels)
    ref = np.full(n, -1, dtype=np.int64)
    for l, s in enumerate(levels):
        ref[list(s)] = l
    assert np.array_equal(level.cpu().numpy(), ref)
    lists = prep.level_lists(level, nl)
    assert [sorted(s) for s in levels] == [t.cpu().tolist() for t in lists]
    # critical paths: every reachable node at level >= 1 as an endpoint, with and without stop flags
    node2level = {v: int(ref[v]) for v in range(n) if ref[v] >= 0}
    ends = np.array([v for v in node2level if node2level[v] >= 1][:4000], dtype=np.int64)
    rng = np.random.default_rng(seed)
    for stop_frac in (0.0, 0.02):
        stop = (rng.random(n) < stop_frac)
        in_csr = csr(n, dst, src, dev)
        paths, lens = prep.trace_critical_paths([in_csr], level, torch.from_numpy(ends.astype(np.int32)).to(dev),
                                                stop=torch.from_numpy(stop.astype(np.uint8)).to(dev) if stop_frac else None)
        want = [PR.find_critical_path(int(e_), pre, node2level, stop if stop_frac else None) for e_ in ends]
        got_l = lens.cpu().tolist()
        got_p = paths.cpu().numpy()
        assert got_l == [len(w) for w in want]
        for row, w in zip(got_p, want):
            assert row[:len(w)].tolist() == w and (row[len(w):] == -1).all()
    # path masks on a 40 x 28 map
    mx, my = 40, 28
    lx, ly = rng.integers(0, mx, size=n), rng.integers(0, my, size=n)
    loc = {v: (int(lx[v]), int(ly[v])) for v in range(n)}
    ip, cols = prep.rasterize_path_masks(paths, lens, torch.from_numpy(lx.astype(np.int32)).to(dev),
                                         torch.from_numpy(ly.astype(np.int32)).to(dev), mx, my)
    rows = PR.path_mask_rows(want, loc, mx, my)
    assert ip.cpu().tolist() == np.concatenate([[0], np.cumsum([len(r) for r in rows])]).tolist()
    assert cols.cpu().tolist() == [c for r in rows for c in r]


def test_prep_edge_cases(dev):
    from mmft import prep
    n = 10
    src, dst = np.array([0, 1, 2, 5]), np.array([1, 2, 3, 6])                       # 5 -> 6 is unreachable from PI 0
    out = csr(n, src, dst, dev)
    level, nl = prep.levelize([out], n, torch.tensor([0], dtype=torch.int32, device=dev))
    assert nl == 4 and level.cpu().tolist() == [0, 1, 2, 3, -1, -1, -1, -1, -1, -1]
    level0, nl0 = prep.levelize([out], n, torch.zeros(0, dtype=torch.int32, device=dev))     # no primary input
    assert nl0 == 1 and (level0 == -1).all()
    # two CSRs (net + cell edges kept apart) give the same levels as the merged graph
    a, b = csr(n, src[:2], dst[:2], dev), csr(n, src[2:], dst[2:], dev)
    level2, nl2 = prep.levelize([a, b], n, torch.tensor([0], dtype=torch.int32, device=dev))
    assert nl2 == nl and torch.equal(level2, level)
    # a cycle reachable from the PI is an error, not a hang
    cyc = csr(3, np.array([0, 1, 2]), np.array([1, 2, 1]), dev)
    with pytest.raises(RuntimeError, match='cycle'):
        prep.levelize([cyc], 3, torch.tensor([0], dtype=torch.int32, device=dev))
    # no endpoints; a level-1 endpoint (path of one pin -> empty mask row); truncation is reported through lens
    inn = csr(n, dst, src, dev)
    p0, l0 = prep.trace_critical_paths([inn], level, torch.zeros(0, dtype=torch.int32, device=dev))
    assert p0.shape[0] == 0 and l0.numel() == 0
    p, l = prep.trace_critical_paths([inn], level, torch.tensor([1, 3], dtype=torch.int32, device=dev), maxlen=2)
    assert l.cpu().tolist() == [1, 3] and p.cpu().tolist() == [[1, -1], [3, 2]]
    loc = torch.arange(n, dtype=torch.int32, device=dev)
    ip, cols = prep.rasterize_path_masks(p, l, loc, loc, 16, 16)
    assert ip.cpu().tolist() == [0, 0, 4] and cols.cpu().tolist() == [2 * 16 + 2, 2 * 16 + 3, 3 * 16 + 2, 3 * 16 + 3]


def test_minmax_normalize_bit_exact(dev):
    from mmft import prep
    torch.manual_seed(0)
    f = torch.randn(70001, 7) * 3 + 1
    f[:, 5] = 2.5                                   # constant column: 0 / 0 -> NaN, as in the reference
    want = PR.norm(f, 2)
    got = prep.minmax_normalize_(f.to(dev).clone(), 2).cpu()
    assert torch.equal(got[:, :5], want[:, :5]) and torch.equal(got[:, 6], want[:, 6])
    assert torch.isnan(got[:, 5]).all() and torch.isnan(want[:, 5]).all()
    g = torch.randn(3, 4)
    g[1, 2] = float('nan')
    w2, g2 = PR.norm(g, 0), prep.minmax_normalize_(g.to(dev).clone(), 0).cpu()
    assert torch.equal(torch.isnan(w2), torch.isnan(g2)) and torch.equal(torch.nan_to_num(w2), torch.nan_to_num(g2))


@pytest.mark.parametrize('cfg', ['B', 'E'])
def test_levelize_reproduces_synthetic_levels_full_size(dev, cfg):
    """BASELINE sizes (64k nodes / 64 levels; 1M nodes): levels from the level-0 nodes == the generator's levels."""
    from mmft import prep
    from mmft.synth import synth_design
    d = synth_design(N=65536, L=64, tile=64, seed=9294) if cfg == 'B' else synth_design(N=1 << 20, L=96, tile=64, seed=9300)
    net = csr(d.N, d.net_src, d.net_dst, dev)
    cell = csr(d.N, d.cell_src, d.cell_dst, dev)
    level, nl = prep.levelize([net, cell], d.N, torch.from_numpy(d.levels[0].astype(np.int32)).to(dev))
    ref = np.full(d.N, -1, dtype=np.int64)
    for l, nodes in enumerate(d.levels):
        ref[nodes] = l
    assert nl == d.L and np.array_equal(level.cpu().numpy(), ref)
    # critical paths of the design's endpoints descend exactly one level per step down to level 1
    ends = torch.from_numpy(d.path2endpoint.astype(np.int32)).to(dev)
    paths, lens = prep.trace_critical_paths([csr(d.N, d.net_dst, d.net_src, dev), csr(d.N, d.cell_dst, d.cell_src, dev)],
                                            level, ends)
    lv = level.long()
    assert torch.equal(lens.long(), torch.clamp(lv[ends.long()], min=1))
    pl = paths.long().clamp(min=0)
    steps = lv[pl[:, :-1]] - lv[pl[:, 1:]]
    valid = torch.arange(paths.shape[1] - 1, device=dev)[None, :] < (lens[:, None] - 1)
    assert bool(((steps == 1) | ~valid).all())


def test_fanin_cone_pruned_sweep_equals_full_sweep(dev):
    """restricting every level to the transitive fan-in of the sampled endpoints leaves their embeddings
    (and the parameter gradients) unchanged; only a fraction of the nodes is touched."""
    from mmft import prep
    from mmft import sweep as S
    from mmft.synth import synth_design
    from mmft.train import build_models, DesignBatch
    d = synth_design(N=20000, L=24, tile=32, seed=77, end_frac=0.2)
    b = DesignBatch([d], dev)
    pmodel, _ = build_models(map_size=d.map_size, device=dev, seed=5)
    g = b.graph
    ends, _, _, _, _, _ = b.select([[3, 11, 11, 40]])                       # a few endpoints (one duplicated)
    in_csrs = [g.csr('in', 'net'), g.csr('in', 'cell')]
    cone = prep.fanin_cone(in_csrs, [torch.tensor(l, dtype=torch.int32, device=dev) for l in b.level_nodes], ends)
    frac = sum(c.numel() for c in cone) / sum(len(l) for l in b.level_nodes)
    assert 0 < frac < 0.5, frac
    # brute-force closure on the host
    ip0, ix0 = [t.cpu().numpy() for t in in_csrs[0]]
    ip1, ix1 = [t.cpu().numpy() for t in in_csrs[1]]
    seen, stack = set(ends.cpu().tolist()), list(set(ends.cpu().tolist()))
    while stack:
        v = stack.pop()
        for u in list(ix0[ip0[v]:ip0[v + 1]]) + list(ix1[ip1[v]:ip1[v + 1]]):
            if u not in seen:
                seen.add(int(u)); stack.append(int(u))
    assert sorted(seen) == sorted(int(v) for c in cone for v in c.cpu().tolist())
    res = []
    for levels in (b.level_nodes, cone):
        g.ndata['h'] = torch.zeros((b.N, 128), dtype=torch.float32, device=dev)
        for p in pmodel.gnn.parameters():
            p.grad = None
        h = S.sweep_forward_all(pmodel.gnn, g, levels, ends)
        (h * h).sum().backward()
        res.append((h.detach().clone(), {k: p.grad.clone() for k, p in pmodel.gnn.named_parameters() if p.grad is not None}))
    assert rel_err(res[1][0], res[0][0]) < 1e-6
    assert res[0][1].keys() == res[1][1].keys()
    for k in res[0][1]:
        assert rel_err(res[1][1][k], res[0][1][k]) < 1e-4, k
