"""CPU restatement of the reference's design preprocessing (TEST INFRASTRUCTURE ONLY - nothing in the product path
imports this).  Plain Python over adjacency lists, following the reference statement by statement:

    cal_topo_level        src/verilog_parser_asap7.py:1452-1517   (frontier expansion + reverse de-duplication)
    find_critical_path    src/verilog_parser_asap7.py:1433-1450
    path masks            src/verilog_parser_asap7.py:1302-1369   (masking == 'critical')
    minMax_scalar / norm  src/train.py:309-318

Pinning: the reference functions need networkx graphs built by its Verilog parser (pyverilog and the raw EDA data are
absent here), so they cannot be run; the reference holds no fixture for them either -> **parity unpinned** for this
file.  The restatement is checked instead against properties (every node's level = longest path from a primary
input, brute force) in tests/test_prep_cpu.py.
"""
import numpy as np
import torch


def cal_topo_level(successors, PIs):
    """successors: list (per node) of successor lists; returns a list of sets, one per level."""
    topo_levels = [list(PIs)]
    remaining = set(PIs)
    cur = list(PIs)
    while True:
        suc = []
        for nd in cur:
            suc.extend(successors[nd])
        suc = set(suc)
        cur = list(suc)
        if len(suc) == 0:
            break
        topo_levels.append(cur)
        remaining = remaining.union(cur)
    visited, rev = set(), []
    for rlevel in reversed(topo_levels):
        new = set(rlevel) - visited
        visited = visited.union(new)
        rev.append(new)
    rev.reverse()
    return rev, remaining


def find_critical_path(endpoint, predecessors, node2level, is_clk=None):
    cur, lv = endpoint, node2level[endpoint]
    path, flag = [endpoint], False
    while lv >= 2:
        moved = False
        for nd in predecessors[cur]:
            if nd not in node2level:          # removed from the graph by cal_topo_level (:1511-1513)
                continue
            if is_clk is not None and is_clk[nd]:
                flag = True
                break
            if node2level[nd] == lv - 1:
                path.append(nd)
                lv -= 1
                cur = nd
                moved = True
                break
        if flag or not moved:
            break
    return path


def path_mask_rows(paths, loc, map_x, map_y):
    """paths: list of node lists; loc[node] = (x, y).  Returns a list of sorted column lists."""
    rows = []
    for path in paths:
        idxs = []
        for j in range(len(path) - 1):
            (xa, ya), (xb, yb) = loc[path[j]], loc[path[j + 1]]
            x1, y1, x2, y2 = min(xa, xb), min(ya, yb), max(xa, xb), max(ya, yb)
            for x in range(x1, x2 + 1):
                idxs.extend(range(x * map_y + y1, x * map_y + y2 + 1))
        rows.append(sorted(set(idxs)))
    return rows


def norm(feature, start_idx):
    """torch fp32, as written in the reference."""
    feature = feature.clone()
    for i in range(start_idx, feature.shape[1]):
        a = feature[:, i]
        feature[:, i:i + 1] = ((a - torch.min(a)) / (torch.max(a) - torch.min(a))).reshape(-1, 1)
    return feature


def adjacency(n, src, dst):
    """(successors, predecessors) lists in edge insertion order (networkx DiGraph order for a simple graph)."""
    suc, pre = [[] for _ in range(n)], [[] for _ in range(n)]
    for s, d in zip(np.asarray(src).tolist(), np.asarray(dst).tolist()):
        suc[s].append(d)
        pre[d].append(s)
    return suc, pre
