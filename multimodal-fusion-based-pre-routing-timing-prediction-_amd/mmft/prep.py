"""Design preprocessing on the GPU: levelization, critical-path trace, path-mask rasterisation and
feature min-max scaling - what the reference does in Python over networkx before a design can be trained on
(src/verilog_parser_asap7.py:1452-1517, 1433-1450, 1302-1369; src/train.py:309-318).

Every function takes and returns device tensors (int32 ids / CSR, float32 features) and calls the C ABI in
include/mmft.h; there is no CPU path here (oracle/prep_restatement.py is the checker used by the tests).
"""
import ctypes

import torch

from . import lib, ops


def _csr_pair(csrs):
    """[(indptr, indices)] with one or two entries -> the four pointers the C ABI takes (second pair may be None)."""
    csrs = list(csrs)
    if not 1 <= len(csrs) <= 2:
        raise ValueError('one or two CSRs expected (net edges, cell edges)')
    out = []
    for ip, idx in csrs:
        ops._chk(ip, 'indptr', torch.int32); ops._chk(idx, 'indices', torch.int32)
        if not (ip.is_contiguous() and idx.is_contiguous()):
            raise ValueError('CSR arrays must be contiguous')
        out += [ip, idx]
    return out + [None, None] * (2 - len(csrs))


def levelize(out_csrs, num_nodes, pis):
    """Longest-path level of every node from the primary inputs `pis` (int32 device tensor).
    Returns (level int32[N] with -1 = unreachable, number of levels).  Synchronises the current stream."""
    p = _csr_pair(out_csrs)
    ops._chk(pis, 'pis', torch.int32)
    n = int(num_nodes)
    if p[0].numel() != n + 1 or (p[2] is not None and p[2].numel() != n + 1):
        raise ValueError('levelize: indptr length must be num_nodes + 1')
    if pis.numel() and (int(pis.min()) < 0 or int(pis.max()) >= n):
        raise ValueError('levelize: primary input id outside the node set')
    level = torch.empty(n, dtype=torch.int32, device=pis.device)
    ws = lib.workspace(pis.device, lib.query('mmft_levelize_workspace_bytes', n))
    nl = ctypes.c_int(0)
    dev, st = lib.stream_args(pis)
    lib.call('mmft_levelize', p[0], p[1], p[2], p[3], n, pis.contiguous(), pis.numel(), level, ctypes.byref(nl), ws,
             ws.numel() * 4, dev, st)
    return level, int(nl.value)


def level_lists(level, num_levels):
    """Per-level node lists (ascending ids), as the reference's topo_levels without the unreachable nodes.  The
    grouping is a stable device sort of the level keys (plumbing; the reference's in-level order is a set order)."""
    key = torch.where(level < 0, torch.full_like(level, num_levels), level)
    order = torch.sort(key, stable=True).indices.to(torch.int32)
    counts = torch.bincount(key.long(), minlength=num_levels + 1)[:num_levels].cpu().tolist()
    out, o = [], 0
    for c in counts:
        out.append(order[o:o + c])
        o += c
    return out


def fanin_cone(in_csrs, level_nodes, targets):
    """Per-level node lists restricted to the transitive fan-in of `targets` (int32 device tensor of node ids).

    level_nodes: per level an int32 device tensor (or a (start, n) range) of that level's nodes, as the sweep takes
    them.  The returned lists (int32 device tensors, original order kept) can be passed to PathModel.forward_sweep in
    place of the full levels: predictions for `targets` are unchanged, nodes outside the cone are never touched.
    Synchronises once per level (boolean compaction); meant for queries on few endpoints, not for the training step."""
    p = _csr_pair(in_csrs)
    ops._chk(targets, 'targets', torch.int32)
    n_nodes = p[0].numel() - 1
    mark = torch.zeros(n_nodes, dtype=torch.uint8, device=targets.device)
    if targets.numel():
        if int(targets.min()) < 0 or int(targets.max()) >= n_nodes:
            raise ValueError('fanin_cone: target id outside the node set')
        mark[targets.long()] = 1
    dev, st = lib.stream_args(mark)
    specs = [ops._rowspec(nodes, n_nodes, 'level_nodes') for nodes in level_nodes]
    for idx, row0, n in reversed(specs[1:]):
        lib.call('mmft_fanin_cone_step', idx, row0, n, p[0], p[1], p[2], p[3], mark, dev, st)
    out = []
    for idx, row0, n in specs:
        ids = idx if idx is not None else torch.arange(row0, row0 + n, dtype=torch.int32, device=mark.device)
        out.append(ids[mark[ids.long()].bool()].contiguous())
    return out


def cone_mask(in_csrs, level_nodes, targets, out=None):
    """uint8[N] on the device: 1 for every node in the transitive fan-in of `targets` (the endpoints included).  No host
    synchronisation and only fixed-size launches (one per level), so it can run inside a captured step: the mask is what
    the sweep kernels take as `active`."""
    p = _csr_pair(in_csrs)
    ops._chk(targets, 'targets', torch.int32)
    n_nodes = p[0].numel() - 1
    mark = out if (out is not None and out.numel() == n_nodes and out.device == targets.device) else \
        torch.empty(n_nodes, dtype=torch.uint8, device=targets.device)
    mark.zero_()
    dev, st = lib.stream_args(mark)
    if targets.numel():
        lib.call('mmft_mark_rows', targets, targets.numel(), mark, 1, dev, st)
    for nodes in reversed(list(level_nodes)[1:]):
        idx, row0, n = ops._rowspec(nodes, n_nodes, 'level_nodes')
        lib.call('mmft_fanin_cone_step', idx, row0, n, p[0], p[1], p[2], p[3], mark, dev, st)
    return mark


def trace_critical_paths(in_csrs, level, endpoints, stop=None, maxlen=None):
    """paths int32[P, maxlen] (-1 padded, endpoint first) and lens int32[P]; `stop`: optional uint8 flags per node."""
    p = _csr_pair(in_csrs)
    ops._chk(level, 'level', torch.int32); ops._chk(endpoints, 'endpoints', torch.int32)
    if stop is not None:
        ops._chk(stop, 'stop', torch.uint8)
        if stop.numel() != level.numel():
            raise ValueError('trace_critical_paths: one stop flag per node expected')
    P = endpoints.numel()
    if P and (int(endpoints.min()) < 0 or int(endpoints.max()) >= level.numel()):
        raise ValueError('trace_critical_paths: endpoint id outside the node set')
    if maxlen is None:
        maxlen = max(int(level.max()) + 1, 1) if level.numel() else 1
    paths = torch.empty((P, maxlen), dtype=torch.int32, device=level.device)
    lens = torch.empty(P, dtype=torch.int32, device=level.device)
    dev, st = lib.stream_args(level)
    lib.call('mmft_trace_critical_paths', p[0], p[1], p[2], p[3], level.contiguous(), stop, endpoints.contiguous(), P,
             maxlen, paths, lens, dev, st)
    return paths, lens


def rasterize_path_masks(paths, lens, loc_x, loc_y, map_x, map_y):
    """CSR (indptr int32[P+1], cols int32[nnz], ascending per row) of the path masks over a map_x x map_y map."""
    ops._chk(paths, 'paths', torch.int32); ops._chk(lens, 'lens', torch.int32)
    ops._chk(loc_x, 'loc_x', torch.int32); ops._chk(loc_y, 'loc_y', torch.int32)
    if paths.dim() != 2 or not paths.is_contiguous() or lens.numel() != paths.shape[0]:
        raise ValueError('rasterize_path_masks: paths must be a contiguous [P, maxlen] tensor with one length per row')
    if loc_x.numel() != loc_y.numel():
        raise ValueError('rasterize_path_masks: loc_x / loc_y sizes differ')
    P, maxlen = paths.shape
    valid = paths[paths >= 0]
    if valid.numel() and int(valid.max()) >= loc_x.numel():
        raise ValueError('rasterize_path_masks: path node id outside the location arrays')
    dev, st = lib.stream_args(paths)
    counts = torch.zeros(P, dtype=torch.int32, device=paths.device)
    lib.call('mmft_path_mask_count', paths, lens, P, maxlen, loc_x.contiguous(), loc_y.contiguous(), int(map_x), int(map_y),
             counts, dev, st)
    indptr = torch.zeros(P + 1, dtype=torch.int32, device=paths.device)
    indptr[1:] = torch.cumsum(counts, 0)
    nnz = int(indptr[-1]) if P else 0
    cols = torch.empty(max(nnz, 1), dtype=torch.int32, device=paths.device)
    lib.call('mmft_path_mask_fill', paths, lens, P, maxlen, loc_x.contiguous(), loc_y.contiguous(), int(map_x), int(map_y),
             indptr, cols, dev, st)
    return indptr, cols[:nnz]


def minmax_normalize_(feat, start_col=0):
    """In place: columns [start_col, C) of feat [N, C] -> (a - min) / (max - min), as src/train.py:309-318."""
    ops._rows2d(feat, 'feat')
    n, C = feat.shape
    ws = lib.workspace(feat.device, lib.query('mmft_minmax_workspace_bytes', n, C - start_col))
    dev, st = lib.stream_args(feat)
    lib.call('mmft_minmax_normalize', feat, feat.stride(0), n, C, int(start_col), ws, ws.numel() * 4, dev, st)
    return feat
