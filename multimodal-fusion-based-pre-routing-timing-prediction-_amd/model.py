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
from mmft import ops as _ops
from mmft.fusion import MaskedPathMap, masked_fc, HeadLevelCtx

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


# Deferred fusion head of the per-level (drop-in) loop.  The reference calls the model once per level and sums the
# losses (src/train.py:490-522); executed literally, every level call carries its own head backward - 31 masked-projection
# gradients over the full 2.3 M-entry weight, ~20 small launches each - and the step takes five times the whole-sweep one.
# Nothing forces that order: the head's parameters and the feature map are the same for every level.  So a level call
# computes its predictions WITHOUT an autograd graph (HeadLevelFn) and hangs them off a per-step root node; its backward
# only stores the incoming gradient.  The engine runs the root's backward after every level node, and there the head of
# ALL levels is recomputed as one batch (the whole-sweep kernels: one masked projection, one mlp_fuse) and differentiated
# once.  The gradient of the endpoint embeddings is scattered into the sweep's gradient buffer directly - the root takes
# the speculative sweep's token as an input, so the reverse sweep (SweepFn.backward) is ordered after it.
LAZY_HEAD = True


class _HeadBatch:
    def __init__(self, model, graph, st, feat_map, masks):
        self.model, self.graph, self.st, self.feat_map, self.masks = model, graph, st, feat_map, masks
        self.tix, self.paths, self.level_th, self.grads = [], [], [], []
        self.links = None               # (first, next) of the whole step when every level call carried them (TrainStep does)
        self.link_rows = []
        self.token = HeadRootFn.apply(st.spec_token, feat_map, self)
        # mlp_alpha of every level in one launch pair: the level-id tensors of the previous step's calls are tried first
        # (the reference loop and TrainStep pass the same objects every step); a call with another tensor computes its own
        self.fast = HeadLevelCtx(st.h, feat_map, masks, model.fcn, model.mlp_fuse, model.global_dim)
        prev = graph.__dict__.get('_head_level_th')
        self.alpha_src, self.alpha = None, None
        if prev and all(t.numel() == 1 for t in prev):
            with th.no_grad():
                self.alpha_src = prev
                self.alpha = model.mlp_alpha(th.cat([t.reshape(1, 1) for t in prev]).to(th.float32))

    def alpha_of(self, slot, level_id_th):
        src = self.alpha_src
        if src is not None and slot < len(src) and src[slot] is level_id_th:
            return self.alpha[slot]
        return self.model.mlp_alpha(level_id_th).reshape(-1)

    def backward(self):
        from mmft import ops
        st, m = self.st, self.model
        slots = [i for i, g in enumerate(self.grads) if g is not None]
        if not slots:
            return None
        counts = [int(self.tix[i].numel()) for i in slots]
        tix = th.cat([self.tix[i] for i in slots])
        paths = th.cat([self.paths[i] for i in slots])
        gr = th.cat([self.grads[i] for i in slots])
        lvt = th.cat([self.level_th[i].reshape(1, -1) for i in slots]).to(th.float32)
        slot_of_row = th.repeat_interleave(th.arange(len(slots), dtype=th.int32), th.tensor(counts)).to(tix.device)
        with th.enable_grad():
            h_all = ops.gather_rows(st.h, tix).requires_grad_(True)
            feat = self.feat_map.detach().requires_grad_(True)
            # the caller's link lists describe ALL rows of the step in level order: usable when every level is in this batch
            links = (None, None)
            if self.links is not None and len(self.link_rows) == len(self.grads) == len(slots) and self.link_rows and \
                    [r[0] for r in self.link_rows] == [sum(counts[:i]) for i in range(len(counts))] and \
                    self.link_rows[0][1] == sum(counts):
                links = self.links
            pm = MaskedPathMap(self.masks, paths, feat, None, *links)
            h_cnn = m._fcn(pm)
            h_global = MF.gather_rows(m.mlp_alpha(lvt), slot_of_row, ascending=True)
            hats = m.mlp_fuse(MF.concat_cols(h_all, h_cnn, h_global)).squeeze(-1)
            params = [p for p in list(m.fcn.parameters()) + list(m.mlp_alpha.parameters()) + list(m.mlp_fuse.parameters())
                      if p.requires_grad]
            th.autograd.backward(hats, gr.reshape(hats.shape), inputs=[h_all, feat] + params)
        # the endpoint rows of the reverse sweep's gradient buffer, as TargetGatherFn.backward fills them
        fast = st.complete and st.active is None
        st.begin_backward(zero_da=not fast, zero_g=not fast)
        if fast:
            ops.target_rows_begin(st.G, tix, st.tflag)
        ops.scatter_add_targets(st.G, tix, h_all.grad, unique=self.graph.__dict__.get('targets_unique'))
        self.graph.__dict__['_head_level_th'] = list(self.level_th)
        # break the reference cycles batch <-> root node and batch <-> sweep state: the batch holds the CNN feature map, whose
        # autograd node keeps the U-Net's activation arena (260 MB at config B) alive until a full garbage collection otherwise
        dfeat = feat.grad
        self.token = None
        if st.__dict__.get('head_batch') is self:
            del st.__dict__['head_batch']
        self.st = self.feat_map = self.masks = self.graph = self.fast = None
        return dfeat


class HeadRootFn(th.autograd.Function):
    @staticmethod
    def forward(ctx, sweep_token, feat_map, hb):
        ctx.hb = hb
        return sweep_token.new_zeros(1)

    @staticmethod
    def backward(ctx, gtoken):
        dfeat = ctx.hb.backward()
        return gtoken.new_zeros(1), dfeat, None


class HeadLevelFn(th.autograd.Function):
    """Predictions of one level, computed without an autograd graph; backward parks the gradient in the batch."""

    @staticmethod
    def forward(ctx, token, hb, slot, tix, path_map, level_id_th):
        m = hb.model
        ctx.hb, ctx.slot = hb, slot
        alpha = hb.alpha_of(slot, level_id_th)
        if hb.fast.ok and alpha.is_contiguous():
            # one C-ABI call per level: gather, masked projection, level embedding, mlp_fuse (mmft_head_level_fwd)
            return hb.fast.run(tix, path_map.paths, path_map.f_off, alpha)
        h_gnn = _ops.gather_rows(hb.st.h, tix)
        k = h_gnn.shape[0]
        h = th.cat((h_gnn, m._fcn(path_map), alpha.expand(k, m.global_dim)), 1)
        return m.mlp_fuse(h).squeeze(-1)

    @staticmethod
    def backward(ctx, g):
        ctx.hb.grads[ctx.slot] = g
        return g.new_zeros(1), None, None, None, None, None


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
        if self._lazy_ok(graph, target_list, path_map):
            graph.__dict__['_head_takes_gradients'] = True                     # speculative sweep: plain gather of h[targets]
            try:
                # advances the (speculative) sweep; the module's forward is called directly when no hooks are registered
                # (64 calls per step: nn.Module.__call__ costs as much as the level bookkeeping itself)
                gnn = self.gnn
                h_gnn = (gnn.forward if not (gnn._forward_hooks or gnn._forward_pre_hooks) else gnn)(graph, nodes, eids, target_list, level_id)
            finally:
                graph.__dict__['_head_takes_gradients'] = False
            st = graph._sweep
            if st is not None and st.spec_token is not None and st.need_grad and not h_gnn.requires_grad:
                hb = st.__dict__.get('head_batch')
                if hb is None or hb.feat_map is not path_map.feat_map or hb.model is not self:
                    hb = st.__dict__['head_batch'] = _HeadBatch(self, graph, st, path_map.feat_map, path_map.masks)
                slot = len(hb.grads)
                hb.tix.append(st.spec_tix[-1])
                hb.paths.append(path_map.paths)
                bl = getattr(path_map, 'batch_links', None)
                if bl is not None:
                    hb.links = bl
                    hb.link_rows.append(path_map.batch_rows)
                hb.level_th.append(level_id_th)
                hb.grads.append(None)
                return HeadLevelFn.apply(hb.token, hb, slot, st.spec_tix[-1], path_map, level_id_th)
            return self._fuse_level(h_gnn, self._fcn(path_map), target_list, level_id_th)
        if self.fcn is not None and len(target_list) != 0:
            h_cnn = self._fcn(path_map)
        else:
            h_cnn = None
        # the GNN level always executes, even without targets: it advances the sweep (src/model.py:276-278)
        h_gnn = self.gnn(graph, nodes, eids, target_list, level_id) if self.gnn is not None else None
        if len(target_list) == 0:
            return None
        return self._fuse_level(h_gnn, h_cnn, target_list, level_id_th)

    def _fuse_level(self, h_gnn, h_cnn, target_list, level_id_th):
        h_global = self.mlp_alpha(level_id_th).expand(len(target_list), 32)
        if h_cnn is None:
            h = th.cat([h_gnn, h_global], dim=1)
        elif h_gnn is None:
            h = th.cat([h_cnn, h_global], dim=1)
        else:
            h = th.cat((h_gnn, h_cnn, h_global), 1)
        return self.mlp_fuse(h).squeeze(-1)

    def _lazy_ok(self, graph, target_list, path_map):
        """The deferred head needs: training (grad mode), all three branches, a lazy path map, and targets."""
        return (LAZY_HEAD and th.is_grad_enabled() and self.gnn is not None and isinstance(self.fcn, nn.Linear)
                and isinstance(path_map, MaskedPathMap) and len(target_list) != 0 and path_map.feat_map.requires_grad)

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
        cache = self.__dict__.setdefault('_level_ids', {})
        key = (num_levels, str(target_levels.device))
        lv = cache.get(key)
        if lv is None:
            lv = cache[key] = th.arange(num_levels, dtype=th.float32, device=target_levels.device).unsqueeze(1)
        # endpoints arrive ordered by level (src/train.py:490-511 emits them level by level): the gradient of the gather is a
        # segmented sum over contiguous runs
        h_global = MF.gather_rows(self.mlp_alpha(lv), target_levels, ascending=True)   # (T, 32), row = alpha(level of t)
        parts = [p for p in (h_gnn, h_cnn, h_global) if p is not None]
        return self.mlp_fuse(MF.concat_cols(*parts) if len(parts) > 1 else parts[0]).squeeze(-1)
