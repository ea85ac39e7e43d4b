"""Drop-in `model` module: MLP, PathConv, LayoutNet, PathModel on the MI355X HIP path.

Same class names, constructor signatures, forward() signatures, attribute names and state_dict keys
as the reference's src/model.py, so `from model import *` in a reference-style train.py / test.py
picks these up.  The arithmetic runs in libmmft_hip.so (hand-written gfx950
kernels) through ctypes; modules hold only torch Parameters, so whole-object pickling
(src/train.py:86-91,583-585) keeps working.  There is no CPU fallback: CPU tensors raise.

`graph` is a mmft.pingraph.PinGraph (DGL has no ROCm build; PinGraph offers the accessor surface
the reference loops use plus the CSR the kernels need).
"""
import torch as th
from torch import nn

from mmft import functional as MF
from mmft import sweep as _sweep
from mmft import cnn as _cnn
from mmft.fusion import MaskedPathMap, masked_fc

__all__ = ['MLP', 'PathConv', 'LayoutNet', 'PathModel', 'MaskedPathMap', 'th', 'nn']


class MLP(th.nn.Module):
    """src/model.py:10-24: Linear -> LeakyReLU(negative_slope) -> ... -> Linear; keys `layers.{0,2,..}`."""

    def __init__(self, *sizes, batchnorm=False, dropout=False, negative_slope=0):
        super().__init__()
        if batchnorm or dropout:
            raise NotImplementedError('MLP(batchnorm/dropout=True) is never used on the reference hot path '
                                      '(src/model.py:48-52,267; src/train.py:77) and has no HIP kernel')
        if negative_slope < 0:
            raise NotImplementedError('negative LeakyReLU slopes are not supported')
        self.negative_slope = negative_slope
        fcs = []
        for i in range(1, len(sizes)):
            fcs.append(th.nn.Linear(sizes[i - 1], sizes[i]))
            if i < len(sizes) - 1:
                fcs.append(th.nn.LeakyReLU(negative_slope=negative_slope))
        self.layers = th.nn.Sequential(*fcs)

    def forward(self, x):
        mods = list(self.layers)
        i = 0
        while i < len(mods):
            lin = mods[i]
            fused_act = i + 1 < len(mods) and isinstance(mods[i + 1], th.nn.LeakyReLU)
            x = MF.linear_act(x, lin.weight, lin.bias, self.negative_slope if fused_act else None)
            i += 2 if fused_act else 1
        return x


class PathConv(nn.Module):
    """src/model.py:27-213. Levelized message passing; one forward() call per topological level."""

    def __init__(self, out_feat_dim, hidden_feat_dim, cell_feat_dim, net_feat_dim, flag_attn=False, num_heads=1,
                 activation=th.nn.functional.relu, bias=True, norm=None):
        super(PathConv, self).__init__()
        self.flag_attn = flag_attn
        self.hidden_feat_dim = hidden_feat_dim
        self.out_feat_dim = out_feat_dim
        self.cell_feat_dim = cell_feat_dim
        self.net_feat_dim = net_feat_dim
        self.num_heads = num_heads
        # creation order matters for seeded initialisation (src/model.py:48-60)
        self.fc_cell_neigh = MLP(self.hidden_feat_dim, 256, self.out_feat_dim)
        self.fc_cell_self = MLP(self.cell_feat_dim, 256, self.out_feat_dim)
        self.fc_net_self = MLP(self.net_feat_dim, 256, self.out_feat_dim)
        self.fc_net_drive = MLP(2, self.out_feat_dim)            # unused in forward (grad stays None)
        self.fc_attn2 = nn.Linear(self.out_feat_dim, 1, bias=False)   # unused in forward
        if flag_attn:
            dim_key = 256
            self.fc_key = nn.Linear(1, dim_key, bias=False)
            self.fc_attn = nn.Linear(2 * dim_key, 1, bias=False)
        if activation is not None and activation not in (th.nn.functional.relu, th.relu):
            raise NotImplementedError('PathConv: only activation=relu or None has a HIP kernel')
        if norm is not None:
            raise NotImplementedError('PathConv: norm is None on the reference path (src/model.py:39)')
        if hidden_feat_dim != out_feat_dim:
            raise ValueError('fc_cell_neigh consumes the aggregated embeddings: hidden_feat_dim must equal out_feat_dim')
        self.activation = activation
        self.norm = norm

    def forward(self, graph, cur_nodes, eids, targets, level_id):
        """Returns h[targets] of shape (len(targets), out_feat_dim); `eids` is ignored.
        flag_attn=True (src/model.py:190-198) reads graph.ndata['key'] (N, 1), which no reference file creates
        (SURVEY D6): the caller has to provide it; the branch is pinned against the reference's own UDFs on a synthetic
        key (tests/golden/attn_reduce.npz, sweep_attn.npz)."""
        return _sweep.level_forward(self, graph, cur_nodes, targets, level_id)


class LayoutNet(nn.Module):
    """src/model.py:216-247: conv9-ReLU-pool-conv7-ReLU-pool-conv9-ReLU-conv7-LeakyReLU(0.1); keys encode.{0,3,6,8}."""

    def __init__(self, pooling):
        super(LayoutNet, self).__init__()
        activation = nn.ReLU()
        activation2 = nn.LeakyReLU(negative_slope=0.1)
        if pooling == 'max':
            pooling_layer = nn.MaxPool2d(2, 2, 0, 1)
        elif pooling == 'avg':
            pooling_layer = nn.AvgPool2d(2, 2, 0)
        else:
            assert False, 'wrong pooling type for layoutnet!'
        self.pooling = pooling
        self.encode = nn.Sequential(
            nn.Conv2d(2, 32, 9, 1, 4), activation, pooling_layer,
            nn.Conv2d(32, 64, 7, 1, 3), activation, pooling_layer,
            nn.Conv2d(64, 32, 9, 1, 4), activation,
            nn.Conv2d(32, 1, 7, 1, 3), activation2)

    def forward(self, x):
        squeeze = x.dim() == 3
        if squeeze:
            x = x.unsqueeze(0)
        e = self.encode
        mode = _cnn.POOL_MAX if self.pooling == 'max' else _cnn.POOL_AVG
        y = _cnn.conv2d(x, e[0].weight, e[0].bias, pad=4, act_slope=0.0)
        y = _cnn.pool2x2(y, mode)
        y = _cnn.conv2d(y, e[3].weight, e[3].bias, pad=3, act_slope=0.0)
        y = _cnn.pool2x2(y, mode)
        y = _cnn.conv2d(y, e[6].weight, e[6].bias, pad=4, act_slope=0.0)
        y = _cnn.conv2d(y, e[8].weight, e[8].bias, pad=3, act_slope=0.1)
        return y[0] if squeeze else y


class PathModel(nn.Module):
    """src/model.py:249-292: fusion head. forward() is called once per level (src/train.py:503)."""

    def __init__(self, gnn, cnn, fcn, mlp_impact, mlp_weight, mlp_fuse, global_dim=32):
        super(PathModel, self).__init__()
        self.global_dim = global_dim
        self.gnn = gnn
        self.cnn = cnn
        self.mlp_impact = mlp_impact
        self.mlp_weight = mlp_weight
        self.fcn = fcn
        self.mlp_fuse = mlp_fuse
        self.mlp_alpha = MLP(1, global_dim * 2, global_dim)

    def _fcn(self, path_map):
        if isinstance(path_map, MaskedPathMap):
            return masked_fc(path_map, self.fcn.weight, self.fcn.bias)
        if isinstance(self.fcn, nn.Linear):
            return MF.linear_act(path_map, self.fcn.weight, self.fcn.bias, None)
        return self.fcn(path_map)

    def forward(self, graph, nodes, eids, target_list, level_id, level_id_th, path_map):
        if self.fcn is not None and len(target_list) != 0:
            h_cnn = self._fcn(path_map)
        else:
            h_cnn = None
        # the GNN level always executes, even without targets: it advances the sweep (src/model.py:276-278)
        h_gnn = self.gnn(graph, nodes, eids, target_list, level_id) if self.gnn is not None else None
        if len(target_list) == 0:
            return None
        h_global = self.mlp_alpha(level_id_th).expand(len(target_list), 32)
        if h_cnn is None:
            h = th.cat([h_gnn, h_global], dim=1)
        elif h_gnn is None:
            h = th.cat([h_cnn, h_global], dim=1)
        else:
            h = th.cat((h_gnn, h_cnn, h_global), 1)
        return self.mlp_fuse(h).squeeze(-1)

    def forward_sweep(self, graph, level_nodes, targets, target_levels, path_map):
        """Whole-sweep entry: every level of one mini-batch in a single call.

        Equivalent to calling forward() for level 0..L-1 and concatenating the non-None results
        (src/train.py:490-511), but level-invariant work is hoisted: fcn once over all sampled paths,
        mlp_alpha once over all level ids, mlp_fuse once over all endpoints, one autograd node for the sweep.
          level_nodes    list over levels of node-id lists (python ints or device int32 tensors)
          targets        device int32 tensor [T]: endpoints ordered by level (the order forward() would emit)
          target_levels  device int32 tensor [T]: level id of each endpoint
          path_map       MaskedPathMap / dense (T,P) tensor over the same T rows, or None
        Returns predictions (T,)."""
        T = len(targets)
        if T == 0:
            if self.gnn is not None:
                _sweep.sweep_forward_all(self.gnn, graph, level_nodes, targets)
            return None
        h_gnn = _sweep.sweep_forward_all(self.gnn, graph, level_nodes, targets) if self.gnn is not None else None
        return self.fuse_heads(h_gnn, path_map, target_levels, len(level_nodes))

    def fuse_heads(self, h_gnn, path_map, target_levels, num_levels, h_cnn=None):
        """Fusion head over all T endpoints at once: fcn(path_map), mlp_alpha(level of each endpoint), mlp_fuse
        (src/model.py:271-292 with the level-invariant work hoisted).  `h_cnn` may be passed pre-computed
        (= self._fcn(path_map)) so that the caller can overlap it with the tail of the sweep."""
        if h_cnn is None:
            h_cnn = self._fcn(path_map) if (self.fcn is not None and path_map is not None) else None
        lv = th.arange(num_levels, dtype=th.float32, device=target_levels.device).unsqueeze(1)
        h_global = MF.gather_rows(self.mlp_alpha(lv), target_levels)          # (T, 32), row = alpha(level of t)
        parts = [p for p in (h_gnn, h_cnn, h_global) if p is not None]
        return self.mlp_fuse(th.cat(parts, dim=1)).squeeze(-1)
