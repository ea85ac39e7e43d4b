"""Synthetic design generator.

Replaces the reference's offline pipeline (src/verilog_parser_asap7.py + src/dataset.py,
which need pyverilog / raw EDA data that are absent) for every BASELINE.json config.
It emits exactly the per-design record of src/generate_data.py:50-54 /
src/train.py:337 and keeps the structural invariants the model relies on:

1. level/edge-type alternation: even levels receive only 'cell' in-edges, odd levels only
   'net' in-edges, level 0 none (src/model.py:180-204);
2. longest-path levels: every node has >=1 predecessor exactly one level below
   (src/verilog_parser_asap7.py:1494-1511);
3. net in-degree 1 (one driver per sink, src/verilog_parser_asap7.py:1183-1188), heavy-tailed
   driver fan-out; cell in-degree = number of input pins;
4. endpoints: targets are a subset of their level, one path id per endpoint
   (src/dataset.py:106-131);
5. masks are 0/1, row = path id, column = x*map + y (src/verilog_parser_asap7.py:1332-1368);
6. topo_levels[l] is a 3-tuple (nodes, targets, path_ids) of python int lists
   (src/dataset.py:124-129);
7. features are emitted pre-trimmed (36 / 2 columns), i.e. feat_reduce=[0,0].
"""
import numpy as np


class SynthDesign:
    """Plain-numpy design record. ``as_record()`` gives the reference 7-tuple layout."""

    def __init__(self):
        self.N = 0
        self.L = 0
        self.map_size = 0
        self.tile = 0
        self.net_src = self.net_dst = None
        self.cell_src = self.cell_dst = None
        self.cell_feat = self.net_feat = None
        self.levels = []            # list[np.int64 array]
        self.level_targets = []     # list[np.int64 array]
        self.level_paths = []       # list[np.int64 array]
        self.path2level = None      # np.int64 (num_paths,)
        self.path2endpoint = None   # np.int64 (num_paths,)
        self.arrival_time = self.required_time = self.label = None
        self.is_end = None
        self.mask_indptr = self.mask_cols = None
        self.critical_paths = None
        self.image = None

    @property
    def num_paths(self):
        return int(self.path2level.shape[0])

    def topo_levels(self):
        return [(self.levels[l].tolist(), self.level_targets[l].tolist(), self.level_paths[l].tolist())
                for l in range(self.L)]


def synth_design(N, L, tile, fanin='regular', seed=9294, permute_ids=True, channels=3,
                 map_div=2, end_frac=1.0 / 16.0):
    """Build one synthetic design.

    N nodes, L levels (cell even / net odd), layout image (channels, tile, tile); mask map is
    (tile // map_div)^2 (UNet output = H/2 -> map_div 2; LayoutNet = H/4 -> map_div 4).
    Seeds follow the reference's fixed seed 9294 (+ design index), src/train.py:596.
    """
    assert L >= 4 and N >= 4 * L
    rng = np.random.default_rng(seed)
    d = SynthDesign()
    d.N, d.L, d.tile = int(N), int(L), int(tile)
    d.map_size = tile // map_div

    # ---- level sizes: level 0 = N/32 (PIs), rest spread uniformly over 1..L-1
    n0 = max(N // 32, 2)
    rest = N - n0
    base = rest // (L - 1)
    sizes = np.full(L, base, dtype=np.int64)
    sizes[0] = n0
    sizes[1:rest - base * (L - 1) + 1] += 1
    assert sizes.sum() == N and sizes.min() >= 1
    starts = np.concatenate([[0], np.cumsum(sizes)])
    perm = rng.permutation(N).astype(np.int64) if permute_ids else np.arange(N, dtype=np.int64)
    d.levels = [perm[starts[l]:starts[l + 1]].copy() for l in range(L)]

    # ---- net edges: every odd-level node has exactly one driver at level l-1 (Zipf-like choice)
    ns, nd = [], []
    for l in range(1, L, 2):
        prev = d.levels[l - 1]
        u = rng.random(sizes[l])
        pick = np.minimum((prev.shape[0] * u ** 3.0).astype(np.int64), prev.shape[0] - 1)
        ns.append(prev[pick])
        nd.append(d.levels[l])
    d.net_src = np.concatenate(ns)
    d.net_dst = np.concatenate(nd)

    # ---- cell edges: every even-level (>=2) node has k in-edges from earlier odd levels, >=1 from l-1
    cs, cd = [], []
    for l in range(2, L, 2):
        n = int(sizes[l])
        if fanin == 'regular':
            k = rng.choice(np.array([1, 2, 3, 4]), size=n, p=[.25, .45, .20, .10])
        elif fanin == 'irregular':
            k = np.minimum(1 + np.floor(rng.pareto(1.3, size=n)).astype(np.int64), 256)
        else:
            raise ValueError(fanin)
        prev = d.levels[l - 1]
        cs.append(prev[rng.integers(0, prev.shape[0], size=n)])
        cd.append(d.levels[l])
        extra = k - 1
        tot = int(extra.sum())
        if tot:
            dst = np.repeat(d.levels[l], extra)
            # source level l-1-2j, j geometric, clipped to the odd levels that exist
            j = rng.geometric(0.5, size=tot) - 1
            src_level = np.maximum(l - 1 - 2 * j, 1)
            off = rng.random(tot)
            src = perm[starts[src_level] + np.minimum((off * sizes[src_level]).astype(np.int64),
                                                       sizes[src_level] - 1)]
            cs.append(src)
            cd.append(dst)
    d.cell_src = np.concatenate(cs)
    d.cell_dst = np.concatenate(cd)

    # ---- features (pre-trimmed: 34 one-hot types + 2 reals; 2 net reals)
    ctype = rng.integers(0, 34, size=N)
    cf = np.zeros((N, 36), dtype=np.float32)
    cf[np.arange(N), ctype] = 1.0
    cf[:, 34:] = rng.random((N, 2), dtype=np.float32)
    d.cell_feat = cf
    d.net_feat = rng.random((N, 2), dtype=np.float32)

    # ---- endpoints: a fraction of odd-level nodes at l>=3
    d.arrival_time = np.zeros((N, 1), dtype=np.float32)
    d.required_time = np.full((N, 1), 0.05 * L / 2, dtype=np.float32)
    d.is_end = np.zeros((N, 1), dtype=np.int64)
    d.level_targets = [np.zeros(0, dtype=np.int64) for _ in range(L)]
    d.level_paths = [np.zeros(0, dtype=np.int64) for _ in range(L)]
    p2l, p2e = [], []
    pid = 0
    for l in range(3, L, 2):
        n = int(sizes[l])
        ne = max(int(round(n * end_frac)), 1)
        sel = d.levels[l][rng.choice(n, size=ne, replace=False)]
        d.level_targets[l] = sel
        d.level_paths[l] = np.arange(pid, pid + ne, dtype=np.int64)
        pid += ne
        p2l.append(np.full(ne, l, dtype=np.int64))
        p2e.append(sel)
        d.arrival_time[sel, 0] = (0.05 * l + rng.normal(0, 0.02, size=ne)).astype(np.float32)
        d.is_end[sel, 0] = 1
    d.path2level = np.concatenate(p2l)
    d.path2endpoint = np.concatenate(p2e)
    slack = d.required_time[:, 0] - d.arrival_time[:, 0]
    d.label = ((slack < 0) & (d.is_end[:, 0] == 1)).astype(np.int64).reshape(N, 1)
    d.critical_paths = np.nonzero(d.label[d.path2endpoint, 0])[0].astype(np.int64)

    # ---- path masks: union of k~U{4..12} boxes with sides U{1..map/8}  -> CSR, col = x*map + y
    m = d.map_size
    smax = max(m // 8, 1)
    P = d.num_paths
    nbox = rng.integers(4, 13, size=P)
    indptr = np.zeros(P + 1, dtype=np.int64)
    cols = []
    for p in range(P):
        kb = int(nbox[p])
        w = rng.integers(1, smax + 1, size=kb)
        h = rng.integers(1, smax + 1, size=kb)
        x0 = rng.integers(0, np.maximum(m - w, 0) + 1)
        y0 = rng.integers(0, np.maximum(m - h, 0) + 1)
        cc = []
        for b in range(kb):
            xs = np.arange(x0[b], min(x0[b] + w[b], m))
            ys = np.arange(y0[b], min(y0[b] + h[b], m))
            cc.append((xs[:, None] * m + ys[None, :]).ravel())
        c = np.unique(np.concatenate(cc))
        cols.append(c)
        indptr[p + 1] = indptr[p] + c.shape[0]
    d.mask_indptr = indptr
    d.mask_cols = np.concatenate(cols).astype(np.int64)

    # ---- layout image
    d.image = rng.random((channels, tile, tile), dtype=np.float32)
    return d


CONFIGS = {
    # BASELINE.json configs[0..4] -> SURVEY.md §8d letters
    'A': dict(N=4096, L=32, tile=64, fanin='regular'),
    'B': dict(N=65536, L=64, tile=256, fanin='regular'),
    'C': dict(N=300000, L=120, tile=512, fanin='regular'),
    'E': dict(N=1048576, L=128, tile=512, fanin='irregular'),
}


def config_design(letter, index=0, **over):
    kw = dict(CONFIGS[letter])
    kw.update(over)
    return synth_design(seed=9294 + index, **kw)
