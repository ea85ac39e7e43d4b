"""On-disk design record.

The reference stores one design as `th.save((graph, topo_levels, path_masks, path2level, path2endpoint,
critical_paths, cnn_inputs))` (src/generate_data.py:50-54) - a pickle that needs DGL to load and is re-read from
disk for every design of every epoch (src/train.py:335-337).  This module keeps the same information as flat
arrays in one `.npz` (no pickle, loads with numpy's safe loader) and converts from the reference tuple wherever
DGL is installed: only duck-typed accessors of the heterograph are used (`edges(etype=...)`, `ndata[...]`,
`number_of_nodes()`), so the converter runs against DGL there and against a stand-in in the tests here.
"""
import numpy as np

from .synth import SynthDesign

_FIELDS = ('net_src', 'net_dst', 'cell_src', 'cell_dst', 'cell_feat', 'net_feat', 'path2level', 'path2endpoint',
           'arrival_time', 'required_time', 'label', 'is_end', 'mask_indptr', 'mask_cols', 'critical_paths', 'image')


def save_design(path, d):
    """Write a design (SynthDesign or anything with the same attributes) to `path` (.npz)."""
    lvl_ptr = np.concatenate([[0], np.cumsum([len(x) for x in d.levels])]).astype(np.int64)
    tgt_ptr = np.concatenate([[0], np.cumsum([len(x) for x in d.level_targets])]).astype(np.int64)
    np.savez_compressed(
        path, version=np.int64(1), N=np.int64(d.N), L=np.int64(d.L), map_size=np.int64(d.map_size), tile=np.int64(d.tile),
        level_ptr=lvl_ptr, level_nodes=np.concatenate([np.asarray(x, dtype=np.int64) for x in d.levels]),
        target_ptr=tgt_ptr,
        level_targets=np.concatenate([np.asarray(x, dtype=np.int64) for x in d.level_targets] + [np.zeros(0, np.int64)]),
        level_paths=np.concatenate([np.asarray(x, dtype=np.int64) for x in d.level_paths] + [np.zeros(0, np.int64)]),
        **{k: np.asarray(getattr(d, k)) for k in _FIELDS})


def load_design(path):
    z = np.load(path, allow_pickle=False)
    if int(z['version']) != 1:
        raise ValueError(f'unsupported design record version {int(z["version"])}')
    d = SynthDesign()
    d.N, d.L, d.map_size, d.tile = int(z['N']), int(z['L']), int(z['map_size']), int(z['tile'])
    lp, tp = z['level_ptr'], z['target_ptr']
    d.levels = [z['level_nodes'][lp[i]:lp[i + 1]] for i in range(d.L)]
    d.level_targets = [z['level_targets'][tp[i]:tp[i + 1]] for i in range(d.L)]
    d.level_paths = [z['level_paths'][tp[i]:tp[i + 1]] for i in range(d.L)]
    for k in _FIELDS:
        setattr(d, k, z[k])
    return d


def from_reference_tuple(record, feat_reduce=(6, 1), map_size=128):
    """Convert the reference's 7-tuple (src/generate_data.py:50-54) to a design record.

    `graph` may be a DGL heterograph (ntype 'pin', etypes 'net'/'cell', src/dataset.py:274-287) or any object with
    the same accessors; feature columns are trimmed as load_single_design does (src/train.py:344-348);
    `path_masks` is the sparse COO tensor (num_paths, map^2) of src/verilog_parser_asap7.py:1368."""
    graph, topo_levels, path_masks, path2level, path2endpoint, critical_paths, cnn_inputs = record
    to_np = lambda t: t.detach().cpu().numpy() if hasattr(t, 'detach') else np.asarray(t)
    d = SynthDesign()
    d.N = int(graph.number_of_nodes())
    d.L = len(topo_levels)
    for et in ('net', 'cell'):
        s, t = graph.edges(etype=et)
        setattr(d, et + '_src', to_np(s).astype(np.int64))
        setattr(d, et + '_dst', to_np(t).astype(np.int64))
    cf, nf = to_np(graph.ndata['cell_feat']).astype(np.float32), to_np(graph.ndata['net_feat']).astype(np.float32)
    d.cell_feat = cf[:, :cf.shape[1] - feat_reduce[0]] if feat_reduce[0] else cf
    d.net_feat = nf[:, :nf.shape[1] - feat_reduce[1]] if feat_reduce[1] else nf
    d.arrival_time = to_np(graph.ndata['arrival_time']).astype(np.float32).reshape(d.N, 1)
    d.required_time = to_np(graph.ndata['required_time']).astype(np.float32).reshape(d.N, 1)
    d.label = to_np(graph.ndata['label']).astype(np.int64).reshape(d.N, 1)
    d.is_end = to_np(graph.ndata['end']).astype(np.int64).reshape(d.N, 1)
    d.levels = [np.asarray(lv[0], dtype=np.int64) for lv in topo_levels]
    d.level_targets = [np.asarray(lv[1], dtype=np.int64) for lv in topo_levels]
    d.level_paths = [np.asarray(lv[2], dtype=np.int64) if len(lv) > 2 else np.zeros(0, np.int64) for lv in topo_levels]
    num_paths = len(path2level)
    d.path2level = np.array([path2level[p] for p in range(num_paths)], dtype=np.int64)
    d.path2endpoint = np.array([path2endpoint[p] for p in range(num_paths)], dtype=np.int64)
    d.critical_paths = np.asarray(list(critical_paths), dtype=np.int64)
    pm = path_masks.coalesce() if hasattr(path_masks, 'coalesce') else path_masks
    idx = to_np(pm.indices())
    order = np.lexsort((idx[1], idx[0]))
    rows, cols = idx[0][order], idx[1][order]
    d.mask_indptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=num_paths))]).astype(np.int64)
    d.mask_cols = cols.astype(np.int64)
    d.map_size = int(map_size)
    d.image = to_np(cnn_inputs).astype(np.float32)
    d.tile = int(d.image.shape[-1])
    return d
