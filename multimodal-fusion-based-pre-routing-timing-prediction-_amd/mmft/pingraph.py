"""PinGraph: the graph container the hot path consumes.

The reference passes a DGL heterograph with node type 'pin' and edge types 'net' / 'cell'
(src/dataset.py:274-287) and touches only a small accessor surface from the loops
(src/train.py:342-348,467,513-521,559-561; src/model.py:206-213):

    graph.ndata[...]                    graph.nodes['pin'].data[...]
    graph.edges['cell'].data[...]       graph.number_of_nodes()
    graph.number_of_edges(etype=...)    graph.to(device)

DGL has no ROCm build in this image, so PinGraph offers the same surface and additionally
owns what the HIP kernels need: in-edge CSR and out-edge CSR per edge type (int32, device
resident), cached per-level row lists, and the per-sweep scratch state.
"""
import os
import numpy as np
import torch

ETYPES = ('net', 'cell')


class _NodeView:
    def __init__(self, data):
        self.data = data


class _Accessor:
    def __init__(self, table):
        self._t = table

    def __getitem__(self, key):
        if isinstance(key, tuple):
            key = key[1]
        return self._t[key]


def _csr_from_coo(n, rows, cols):
    """CSR over `rows` (stable, keeps insertion order inside a row). Returns indptr, cols, perm."""
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    perm = np.argsort(rows, kind='stable')
    counts = np.bincount(rows, minlength=n)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    return indptr, cols[perm], perm


class PinGraph:
    def __init__(self, num_nodes, edges):
        """edges: {'net': (src, dst), 'cell': (src, dst)} as array-likes of node ids."""
        self._n = int(num_nodes)
        self._coo = {}
        for et in ETYPES:
            s, d = edges.get(et, ((), ()))
            s = np.asarray(torch.as_tensor(s).cpu().numpy() if torch.is_tensor(s) else s, dtype=np.int64)
            d = np.asarray(torch.as_tensor(d).cpu().numpy() if torch.is_tensor(d) else d, dtype=np.int64)
            assert s.shape == d.shape
            if s.size:
                assert s.min() >= 0 and d.min() >= 0 and max(s.max(), d.max()) < self._n, "edge id out of range"
            self._coo[et] = (s, d)
        self.device = torch.device('cpu')
        self._ndata = {}
        self._edata = {et: {} for et in ETYPES}
        self._csr_host = {}
        self._csr_dev = {}
        self._level_cache = {}
        self._sweep = None          # per-sweep state, owned by mmft.sweep
        self._build_csr()

    # ------------------------------------------------------------------ DGL-like surface
    @property
    def ndata(self):
        return self._ndata

    @property
    def nodes(self):
        return _Accessor({'pin': _NodeView(self._ndata)})

    @property
    def edges(self):
        return _Accessor({et: _NodeView(self._edata[et]) for et in ETYPES})

    @property
    def ntypes(self):
        return ['pin']

    @property
    def etypes(self):
        return list(ETYPES)

    def number_of_nodes(self, ntype=None):
        return self._n

    num_nodes = number_of_nodes

    def number_of_edges(self, etype=None):
        if etype is None:
            return sum(self._coo[et][0].shape[0] for et in ETYPES)
        if isinstance(etype, tuple):
            etype = etype[1]
        return int(self._coo[etype][0].shape[0])

    num_edges = number_of_edges

    def to(self, device):
        device = torch.device(device)
        g = PinGraph.__new__(PinGraph)
        g._n = self._n
        g._coo = self._coo
        g.device = device
        g._ndata = {k: v.to(device) for k, v in self._ndata.items()}
        g._edata = {et: {k: v.to(device) for k, v in self._edata[et].items()} for et in ETYPES}
        g._csr_host = self._csr_host
        g._csr_dev = {}
        g._level_cache = {}
        g._sweep = None
        return g

    def in_degrees(self, etype):
        ip = self._csr_host[('in', etype)][0]
        return torch.from_numpy(np.diff(ip))

    # ------------------------------------------------------------------ CSR for the kernels
    def _build_csr(self):
        for et in ETYPES:
            s, d = self._coo[et]
            ip, src_sorted, perm = _csr_from_coo(self._n, d, s)       # in-edges: row = dst, col = src
            self._csr_host[('in', et)] = (ip, src_sorted, perm)
            op, dst_sorted, operm = _csr_from_coo(self._n, s, d)      # out-edges: row = src, col = dst
            self._csr_host[('out', et)] = (op, dst_sorted, operm)

    def csr(self, direction, etype):
        """(indptr int32[N+1], indices int32[E]) on self.device; cached."""
        key = (direction, etype)
        if key not in self._csr_dev:
            ip, idx, _ = self._csr_host[key]
            assert ip[-1] < 2 ** 31, "edge count exceeds int32"
            self._csr_dev[key] = (torch.from_numpy(ip.astype(np.int32)).to(self.device),
                                  torch.from_numpy(idx.astype(np.int32)).to(self.device))
        return self._csr_dev[key]

    def out_net_weight(self):
        """float32[E_net] aligned with csr('out','net'): 1 / (net in-degree of the edge's destination) - the
        derivative of fn.mean (src/model.py:186-187) with respect to each source row."""
        if 'onw' not in self._csr_dev:
            ip = self._csr_host[('in', 'net')][0]
            dst = self._csr_host[('out', 'net')][1]
            indeg = np.diff(ip)[dst]
            self._csr_dev['onw'] = torch.from_numpy((1.0 / np.maximum(indeg, 1)).astype(np.float32)).to(self.device)
        return self._csr_dev['onw']

    def cell_edge_drivers(self):
        """int32[E_cell] on the device, aligned with csr('in','cell')[1]: the single driver of the net behind each cell
        in-edge (the source of that net's one in-edge), -1 where the source has no or several in-edges.  The folded
        gather (mmft_pair_fwd_gather / mmft_level_fwd_bf16) uses it to skip two dependent index loads per edge."""
        if 'icd' not in self._csr_dev:
            nptr, nidx = self._csr_host[('in', 'net')][0], self._csr_host[('in', 'net')][1]
            u = self._csr_host[('in', 'cell')][1]
            drv = np.full(u.shape[0], -1, dtype=np.int32)
            if u.size and nidx.size:
                one = (nptr[u + 1] - nptr[u]) == 1
                drv[one] = nidx[nptr[u[one]]]
            self._csr_dev['icd'] = torch.from_numpy(drv).to(self.device)
        return self._csr_dev['icd']

    def out2in(self, etype):
        """int32[E] on the device: out-CSR edge position -> in-CSR position of the same edge (per-edge data such as the
        attention weights is stored in in-CSR order, the reverse sweep walks out-edges)."""
        key = ('o2i', etype)
        if key not in self._csr_dev:
            perm_in = self._csr_host[('in', etype)][2]
            perm_out = self._csr_host[('out', etype)][2]
            inv_in = np.empty_like(perm_in)
            inv_in[perm_in] = np.arange(perm_in.shape[0])
            self._csr_dev[key] = torch.from_numpy(inv_in[perm_out].astype(np.int32)).to(self.device)
        return self._csr_dev[key]

    def csr_host(self, direction, etype):
        ip, idx, _ = self._csr_host[(direction, etype)]
        return ip, idx

    def level_rows(self, level_id, nodes, tag='nodes'):
        """Device int32 tensor of a level's node list; cached by (tag, level_id).

        `cur_nodes` / `targets` arrive as python int lists on every call
        (src/train.py:491-503); re-uploading them per call is what the cache avoids.
        """
        if torch.is_tensor(nodes) and nodes.dtype == torch.int32 and nodes.device == self.device:
            # already resident (the harness uploads a step's targets in one copy): caller guarantees the range
            return nodes if nodes.is_contiguous() else nodes.contiguous()
        key = (tag, level_id)
        hit = self._level_cache.get(key)
        if hit is not None:
            ref, dev = hit
            if ref is nodes or (len(ref) == len(nodes) and ref == nodes):
                return dev
        if torch.is_tensor(nodes):
            host = nodes.detach().to('cpu', torch.int64)
            ref = host.tolist()
        else:
            ref = nodes
            host = torch.tensor(nodes, dtype=torch.int64) if len(nodes) else torch.zeros(0, dtype=torch.int64)
        if host.numel():
            assert int(host.min()) >= 0 and int(host.max()) < self._n, "node id out of range"
        dev = host.to(torch.int32).to(self.device)
        self._level_cache[key] = (ref, dev)
        return dev

    def level_meta(self, level_id, nodes, D=128):
        """Static facts about a level, cached: `range` = (start, n) when the node list is the contiguous ascending
        id range start..start+n-1 (true after level-major renumbering), else None; algorithmic HBM bytes of the
        level's aggregation kernels at width D (SURVEY.md §8d formulas, fp32)."""
        key = ('meta', level_id, D)
        hit = self._level_cache.get(key)
        if hit is not None and (hit[0] is nodes or (len(hit[0]) == len(nodes) and hit[0] == nodes)):
            return hit[1]
        v = np.asarray(nodes, dtype=np.int64)
        n = int(v.shape[0])
        rng = None
        if n and int(v[0]) + n - 1 == int(v[-1]) and (n == 1 or bool((np.diff(v) == 1).all())):
            rng = (int(v[0]), n)
        deg = lambda d, et: int((self._csr_host[(d, et)][0][v + 1] - self._csr_host[(d, et)][0][v]).sum()) if n else 0
        e_in_net, e_in_cell, e_out_net, e_out_cell = deg('in', 'net'), deg('in', 'cell'), deg('out', 'net'), deg('out', 'cell')
        from . import ops
        odeg = (self._csr_host[('out', 'net')][0][v + 1] - self._csr_host[('out', 'net')][0][v]) + \
               (self._csr_host[('out', 'cell')][0][v + 1] - self._csr_host[('out', 'cell')][0][v]) if n else np.zeros(0, np.int64)
        hv = v[odeg > ops.PAIR_HEAVY_OUT] if n else v
        meta = dict(range=rng, n=n,
                    # rows whose out-degree exceeds ops.PAIR_HEAVY_OUT: reduced by a whole workgroup in the reverse pull
                    heavy_out=torch.from_numpy(hv.astype(np.int32)).to(self.device) if hv.size else None,
                    bytes_mean=4 * D * (e_in_net + 2 * n) + 4 * e_in_net + 8 * n,
                    bytes_softmax=4 * D * (e_in_cell + 3 * n) + 4 * e_in_cell + 8 * n,
                    bytes_pull=4 * D * (e_out_net + 3 * e_out_cell + 3 * n) + 8 * e_out_net + 4 * e_out_cell + 16 * n)
        self._level_cache[key] = (nodes, meta)
        return meta

    # Results derived from a whole level schedule are cached per LIST OBJECTS: the entry keeps the lists alive (an id()
    # can then not be reused by another list while the entry exists), a hit requires `is` on every list plus unchanged
    # lengths and end elements (guards in-place edits that keep the length), and at most LIST_CACHE_MAX schedules are kept
    # (per-step host lists, e.g. fan-in cones, would otherwise grow the cache without bound).
    LIST_CACHE_MAX = 8

    @staticmethod
    def _lists_key(level_nodes):
        return (tuple(id(n) for n in level_nodes), tuple(len(n) for n in level_nodes),
                tuple((n[0], n[-1]) if len(n) else () for n in level_nodes))

    def _lists_hit(self, key, level_nodes):
        hit = self._level_cache.get(key)
        if hit is None or len(hit[0]) != len(level_nodes) or not all(a is b for a, b in zip(hit[0], level_nodes)):
            return None
        return (hit[1],)

    def _lists_store(self, key, level_nodes, value):
        order = self.__dict__.setdefault('_lists_order', [])
        if key not in self._level_cache:
            order.append(key)
            while len(order) > self.LIST_CACHE_MAX:
                self._level_cache.pop(order.pop(0), None)
        self._level_cache[key] = (tuple(level_nodes), value)

    def level_set_is_complete(self, level_nodes):
        """True when the level lists are a complete, well-formed schedule of this graph: every node is in exactly one
        level, every edge runs from a lower to a higher level, net in-edges enter odd levels only and cell in-edges
        even levels >= 2 only (src/model.py:180-204).  Only then may the reverse sweep skip the zero fills of G / DA:
        every row a pull reads is rewritten earlier in the same reverse sweep.  A partial schedule (fan-in cone,
        truncated list) leaves consumers outside it, whose G / DA rows must read as zero.  Cached per list object."""
        if any(torch.is_tensor(n) for n in level_nodes):
            return False                                        # device-resident lists: not inspected, take the safe path
        key = ('complete',) + self._lists_key(level_nodes)
        hit = self._lists_hit(key, level_nodes)
        if hit is not None:
            return hit[0]
        lev = np.full(self._n, -1, dtype=np.int64)
        ok, total = True, 0
        for l, nodes in enumerate(level_nodes):
            v = np.asarray(nodes, dtype=np.int64)
            total += v.shape[0]
            if v.size and (lev[v] >= 0).any():
                ok = False
            lev[v] = l
        ok = ok and total == self._n and bool((lev >= 0).all())
        if ok:
            for et, parity in (('net', 1), ('cell', 0)):
                s_, d_ = self._coo[et]
                if s_.size and not bool(((lev[s_] < lev[d_]) & (lev[d_] % 2 == parity) & (lev[d_] >= 1)).all()):
                    ok = False
        self._lists_store(key, level_nodes, ok)
        return ok

    def fold_schedule(self, level_nodes, heavy_in=None, heavy_out=None):
        """Static facts for the folded level kernels (mmft_pair_fwd_gather / mmft_pair_bwd_pull), or None when the graph
        does not meet their preconditions: a complete schedule, every net in-degree exactly 1, every net edge from an
        even level l to level l + 1, every cell edge from an odd level, and every net level a contiguous id range.
        Per level l: 'range', 'heavy_in' (cell rows with more than `heavy_in` cell in-edges), 'heavy_out' (rows with more
        than `heavy_out` net out-edges) as device int32 tensors or None; the thresholds default to ops.PAIR_HEAVY_IN /
        ops.PAIR_HEAVY_OUT.  Cached per list object."""
        from . import ops
        heavy_in = ops.PAIR_HEAVY_IN if heavy_in is None else heavy_in
        heavy_out = ops.PAIR_HEAVY_OUT if heavy_out is None else heavy_out
        if not self.level_set_is_complete(level_nodes):
            return None
        key = ('fold',) + self._lists_key(level_nodes) + (heavy_in, heavy_out)
        hit = self._lists_hit(key, level_nodes)
        if hit is not None:
            return hit[0]
        lev = np.full(self._n, -1, dtype=np.int64)
        for l, nodes in enumerate(level_nodes):
            lev[np.asarray(nodes, dtype=np.int64)] = l
        ns, nd = self._coo['net']
        cs, _cd = self._coo['cell']
        in_net_deg = np.diff(self._csr_host[('in', 'net')][0])
        ok = bool((in_net_deg[lev % 2 == 1] == 1).all()) and bool((in_net_deg[lev % 2 == 0] == 0).all())
        if ns.size:
            ok = ok and bool(((lev[ns] % 2 == 0) & (lev[nd] == lev[ns] + 1)).all())
        if cs.size:
            ok = ok and bool((lev[cs] % 2 == 1).all())
        sched = None
        if ok:
            in_cell_deg = np.diff(self._csr_host[('in', 'cell')][0])
            out_net_deg = np.diff(self._csr_host[('out', 'net')][0])
            sched = []
            for l, nodes in enumerate(level_nodes):
                v = np.asarray(nodes, dtype=np.int64)
                n = int(v.shape[0])
                rng = (int(v[0]), n) if n and int(v[0]) + n - 1 == int(v[-1]) and bool((np.diff(v) == 1).all()) else None
                if l % 2 == 1 and n and rng is None:
                    sched = None                                  # the range test of the folded gather needs contiguous net levels
                    break
                dev_list = lambda a: torch.from_numpy(a.astype(np.int32)).to(self.device) if a.size else None
                sched.append(dict(range=rng, n=n,
                                  heavy_in=dev_list(v[in_cell_deg[v] > heavy_in]) if (l % 2 == 0 and n) else None,
                                  heavy_out=dev_list(v[out_net_deg[v] > heavy_out]) if (l % 2 == 0 and n) else None))
        self._lists_store(key, level_nodes, sched)
        return sched

    def level_slots(self, level_nodes):
        """Static tables of the slot-form level kernel (mmft_level_fwd_slots), or None when the schedule is not a folded one:
        (slots int32[N][8], net_driver int32[N], per-level maximum cell fan-in).  slots[v] = hrow[0..3], prow[0..3] of v's
        cell in-edges in edge order: an edge from a net u of the level right below v (the folded one) reads h[driver(u)] and
        adds pre[u] (hrow = driver, prow = u), any other edge reads h[u] (prow = -1); unused slots are -1; a row with more
        than four in-edges has hrow[0] = -2 (such levels keep the index-chasing kernels).  Cached per list objects."""
        if self.fold_schedule(level_nodes) is None:
            return None
        key = ('slots',) + self._lists_key(level_nodes)
        hit = self._lists_hit(key, level_nodes)
        if hit is not None:
            return hit[0]
        N = self._n
        lev = np.full(N, -1, dtype=np.int64)
        for l, nodes in enumerate(level_nodes):
            lev[np.asarray(nodes, dtype=np.int64)] = l
        nptr, nidx = self._csr_host[('in', 'net')][0], self._csr_host[('in', 'net')][1]
        cptr, cidx = self._csr_host[('in', 'cell')][0], self._csr_host[('in', 'cell')][1]
        ndeg, cdeg = np.diff(nptr), np.diff(cptr)
        net_drv = np.full(N, -1, dtype=np.int32)
        one = ndeg == 1
        net_drv[one] = nidx[nptr[:-1][one]]
        slots = np.full((N, 8), -1, dtype=np.int32)
        for k in range(4):
            rows = np.nonzero(cdeg > k)[0]
            if rows.size == 0:
                break
            u = cidx[cptr[rows] + k].astype(np.int64)
            folded = (lev[u] == lev[rows] - 1) & (net_drv[u] >= 0)
            slots[rows, k] = np.where(folded, net_drv[u], u)
            slots[rows, 4 + k] = np.where(folded, u, -1)
        slots[cdeg > 4, 0] = -2
        max_in = [int(cdeg[np.asarray(nodes, dtype=np.int64)].max()) if len(nodes) else 0 for nodes in level_nodes]
        out = (torch.from_numpy(slots).to(self.device), torch.from_numpy(net_drv).to(self.device), max_in)
        self._lists_store(key, level_nodes, out)
        return out

    BWD_PAIR_TILE_DRIVERS = 32     # drivers per tile (two 16-row MFMA blocks)
    BWD_PAIR_TILE_SINKS = 32       # sinks a tile of whole drivers may hold (two per thread group); a driver with more is HEAVY:
    BWD_PAIR_PART = 32             # ... cut into parts of this many sinks, one workgroup each

    def level_bwd_pairs(self, level_nodes):
        """Static tables of the paired reverse-sweep kernel (mmft_level_bwd_pair), or None when the schedule is not a folded
        one: (cslots int32[N][4], pairs, scratch fp32[rows][128], counters int32[...]) on the device, with pairs[l // 2] for
        every even level l = None (the pair keeps the per-level kernels) or dict(tiles=int32[ntiles][8], ntiles, sink_shift,
        n_cell, n_net).  A pair qualifies when level l is a contiguous id range whose out-net edge list in CSR order IS the id
        range of level l + 1 (DesignBatch numbers the nets of a level by driver).  Tile rows: (first driver id, count <= 32,
        part, parts, first scratch row, counter index, first / end out-net CSR position of the tile's sinks) - parts = 0: whole drivers with at most BWD_PAIR_TILE_SINKS sinks
        together; a driver with more sinks is cut into parts of BWD_PAIR_PART sinks, one tile each.  cslots[u] = the first four
        cell consumers of u in out-edge order, -1 = none; with more than four, slot 3 = -2 - (out-cell CSR position of the
        fourth).  scratch / counters are shared by the levels (their launches follow each other on one stream).  Cached per
        list objects."""
        fold = self.fold_schedule(level_nodes)
        if fold is None:
            return None
        key = ('bwd_pairs',) + self._lists_key(level_nodes)
        hit = self._lists_hit(key, level_nodes)
        if hit is not None:
            return hit[0]
        N = self._n
        optr, oidx = self._csr_host[('out', 'net')][0], self._csr_host[('out', 'net')][1]
        cptr, cidx = self._csr_host[('out', 'cell')][0], self._csr_host[('out', 'cell')][1]
        cdeg = np.diff(cptr)
        cslots = np.full((N, 4), -1, dtype=np.int32)
        for k in range(4):
            rows = np.nonzero(cdeg > k)[0]
            if rows.size == 0:
                break
            cslots[rows, k] = cidx[cptr[rows] + k]
        many = np.nonzero(cdeg > 4)[0]
        cslots[many, 3] = -2 - (cptr[many] + 3)
        L = len(level_nodes)
        pairs, max_rows, max_cnt = [], 1, 1
        for l in range(0, L, 2):
            rc = fold[l]['range']
            n_net = fold[l + 1]['n'] if l + 1 < L else 0
            rn = fold[l + 1]['range'] if n_net else (0, 0)
            pair = None
            if rc is not None and rn is not None:
                row0, n = rc
                e0, e1 = int(optr[row0]), int(optr[row0 + n])
                fan = np.diff(optr[row0:row0 + n + 1])
                ok = (e1 - e0 == n_net) and (n_net == 0 or bool((oidx[e0:e1] == np.arange(rn[0], rn[0] + n_net)).all()))
                if ok:
                    # greedy tiling: close a tile at 16 drivers, or when the next driver would push it past the sink target
                    tiles, start, sinks, srow, cnt = [], 0, 0, 0, 0
                    fl = fan.tolist()
                    pos = optr[row0:row0 + n + 1].tolist()                # CSR position of every driver's first sink
                    close = lambda a, b: tiles.append((row0 + a, b - a, 0, 0, 0, 0, pos[a], pos[b]))
                    for i, f in enumerate(fl):
                        heavy = f > self.BWD_PAIR_TILE_SINKS
                        if i > start and (i - start == self.BWD_PAIR_TILE_DRIVERS or sinks + f > self.BWD_PAIR_TILE_SINKS or heavy):
                            close(start, i)
                            start, sinks = i, 0
                        if heavy:
                            parts = -(-f // self.BWD_PAIR_PART)
                            tiles += [(row0 + i, 1, p, parts, srow, cnt, pos[i] + p * self.BWD_PAIR_PART,
                                       min(pos[i + 1], pos[i] + (p + 1) * self.BWD_PAIR_PART)) for p in range(parts)]
                            srow, cnt, start, sinks = srow + parts, cnt + 1, i + 1, 0
                        else:
                            sinks += f
                    if n > start:
                        close(start, n)
                    # longest-running tiles first (heavy drivers' parts, then by sink count): the launch ends with short ones
                    tiles.sort(key=lambda t: (0 if t[3] else 1, -(t[7] - t[6])))
                    max_rows, max_cnt = max(max_rows, srow), max(max_cnt, cnt)
                    t = torch.tensor(tiles, dtype=torch.int32).reshape(-1, 8).to(self.device)
                    pair = dict(tiles=t, ntiles=len(tiles), sink_shift=(rn[0] - e0) if n_net else 0, n_cell=n, n_net=n_net)
            pairs.append(pair)
        out = (torch.from_numpy(cslots).to(self.device), pairs,
               torch.zeros((max_rows, 128), dtype=torch.float32, device=self.device),
               torch.zeros(max_cnt, dtype=torch.int32, device=self.device))
        self._lists_store(key, level_nodes, out)
        return out

    # ------------------------------------------------------------------ construction helpers
    @staticmethod
    def from_synth(d, out_dim=None):
        """PinGraph + ndata laid out as src/dataset.py:280-287 / src/train.py:342-343."""
        g = PinGraph(d.N, {'net': (d.net_src, d.net_dst), 'cell': (d.cell_src, d.cell_dst)})
        g.ndata['cell_feat'] = torch.from_numpy(d.cell_feat)
        g.ndata['net_feat'] = torch.from_numpy(d.net_feat)
        g.ndata['arrival_time'] = torch.from_numpy(d.arrival_time)
        g.ndata['required_time'] = torch.from_numpy(d.required_time)
        g.ndata['label'] = torch.from_numpy(d.label)
        g.ndata['end'] = torch.from_numpy(d.is_end)
        if out_dim is not None:
            g.ndata['h'] = torch.zeros((d.N, out_dim), dtype=torch.float32)
        return g

    @staticmethod
    def batch(graphs):
        """Block-diagonal merge (node ids offset by the running node count)."""
        off = 0
        coo = {et: ([], []) for et in ETYPES}
        for g in graphs:
            for et in ETYPES:
                s, d = g._coo[et]
                coo[et][0].append(s + off)
                coo[et][1].append(d + off)
            off += g._n
        merged = PinGraph(off, {et: (np.concatenate(coo[et][0]), np.concatenate(coo[et][1])) for et in ETYPES})
        keys = set(graphs[0].ndata.keys())
        for k in keys:
            if all(k in g.ndata for g in graphs):
                merged.ndata[k] = torch.cat([g.ndata[k] for g in graphs], dim=0)
        return merged
