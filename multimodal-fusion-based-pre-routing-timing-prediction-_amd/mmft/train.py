"""Reference-shaped train step on the HIP path (the caller side of the hot path, src/train.py:461-562).

`build_models` follows init_model (src/train.py:34-93) with the snapshot defects D1/D2 of SURVEY.md §0.1
resolved the only way model.py allows: PathModel(gnn, None, fcn, None, None, mlp_fuse) and
mlp_fuse = MLP(D_gnn + D_cnn + 32, 2*(...), nlabels).

`TrainStep` runs one mini-batch exactly as the reference loop does - per level `model(graph, nodes, eids,
targets, level_id, level_id_th, path_map)` through the drop-in modules - for one design or for several
designs merged block-diagonally (same-index levels concatenated, per-image BatchNorm statistics), then MSE
on arrival time, backward and Adam.  Under torch.distributed the flat gradient buffer is all-reduced
(RCCL over xGMI) before the fused Adam kernel.
"""
import os

import numpy as np
import torch
from torch import nn

from .pingraph import PinGraph
from .fusion import PathMasks, MaskedPathMap, FlatAdam, mse_loss, mse_loss_gather, cross_entropy_loss, batch_links, unit_grad


def build_models(map_size=128, out_dim=128, cell_feat_dim=36, net_feat_dim=2, cnn_outdim=128, pooling='max',
                 nlabels=1, device='cuda', seed=9294, unet=True):
    """init_model (src/train.py:34-93) for the defaults of src/options.py with --unet."""
    import model as M
    import Unet as U
    torch.manual_seed(seed)                                                     # src/train.py:596-598
    gnn = M.PathConv(out_feat_dim=out_dim, hidden_feat_dim=out_dim, cell_feat_dim=cell_feat_dim,
                     net_feat_dim=net_feat_dim, flag_attn=False, num_heads=1)
    cnn = U.UNet(pooling) if unet else M.LayoutNet(pooling)
    fcn = nn.Linear(map_size * map_size, cnn_outdim)
    nn.init.xavier_uniform_(fcn.weight, gain=nn.init.calculate_gain('relu'))    # src/train.py:72-73
    mlp_dim = out_dim + cnn_outdim + 32                                          # D2: global_dim is 32
    mlp = M.MLP(mlp_dim, mlp_dim * 2, nlabels)
    pmodel = M.PathModel(gnn, None, fcn, None, None, mlp)                        # D1
    return pmodel.to(device), cnn.to(device)


def trainable_parameters(pmodel, cnn):
    """Parameters that receive gradients (fc_net_drive / fc_attn2 are unused in forward: torch's Adam skips
    them because their .grad stays None, src/model.py:52-54)."""
    return [p for _, ps in parameter_buckets(pmodel, cnn) for p in ps]


def parameter_buckets(pmodel, cnn):
    """The trainable parameters grouped by WHEN their gradients exist in the backward pass: the fusion head (fcn,
    mlp_fuse, mlp_alpha: ready right after the head's backward, 2.27 M of 2.89 M parameters at defaults) and the rest
    (GNN: end of the reverse sweep; CNN: end of the U-Net backward).  Under data parallelism each group is one
    all-reduce (mmft.dist.GradReducer)."""
    skip = ('gnn.fc_net_drive.', 'gnn.fc_attn2.')
    named = [(n, p) for n, p in pmodel.named_parameters() if not n.startswith(skip)]
    head = [p for n, p in named if not n.startswith(('gnn.', 'cnn.'))]
    rest = [p for n, p in named if n.startswith(('gnn.', 'cnn.'))]
    if cnn is not None:
        seen = {id(p) for p in rest}
        rest += [p for p in cnn.parameters() if id(p) not in seen]
    return [('head', head), ('gnn+cnn', rest)]


_JOIN_LATE = os.environ.get('MMFT_JOIN_LATE') == '1'      # see TrainStep.forward

class DesignBatch:
    """B designs resident on the device, merged block-diagonally."""

    def __init__(self, designs, device, out_dim=128, renumber=True):
        """renumber: give the merged nodes level-major ids with the cell levels (0, 2, 4, ...) before the net levels
        (1, 3, 5, ...).  Every level is then one contiguous id range (its rows of h / G / A stream through memory, no
        index array), and so are the three row sets the batched GEMMs of the sweep run over - all cell levels, the
        cell levels after level 0, all net levels - which therefore need no row gather / scatter either.  Purely
        internal: ids handed back to the caller are the original merged ids."""
        self.designs = designs
        self.B = len(designs)
        self.device = torch.device(device)
        self.node_off = np.concatenate([[0], np.cumsum([d.N for d in designs])]).astype(np.int64)
        self.path_off = np.concatenate([[0], np.cumsum([d.num_paths for d in designs])]).astype(np.int64)
        self.N = int(self.node_off[-1])
        self.L = max(d.L for d in designs)
        levels_old = []
        for l in range(self.L):
            parts = [d.levels[l] + self.node_off[i] for i, d in enumerate(designs) if l < d.L]
            levels_old.append(np.concatenate(parts))
        if renumber:
            # inside a net level the nodes are numbered by driver (the cell levels get their ids first): the sinks of one driver
            # are then consecutive ids and the sink runs follow the drivers' order, which is what lets the reverse sweep handle a
            # (cell level, net level) pair in one launch (PinGraph.level_bwd_pairs).  The order inside a level carries no
            # meaning in the reference (graph.pull over a node set, src/model.py:186-204).
            even = np.concatenate(levels_old[0::2]) if levels_old[0::2] else np.zeros(0, np.int64)
            rank = np.full(self.N, -1, dtype=np.int64)
            rank[even] = np.arange(even.shape[0])
            nsrc = np.concatenate([d.net_src + self.node_off[i] for i, d in enumerate(designs)])
            ndst = np.concatenate([d.net_dst + self.node_off[i] for i, d in enumerate(designs)])
            indeg = np.bincount(ndst, minlength=self.N)
            drv = np.full(self.N, -1, dtype=np.int64)
            drv[ndst] = nsrc
            eid = np.zeros(self.N, dtype=np.int64)                 # ... and inside a driver's run by edge order (= out-CSR order)
            eid[ndst] = np.arange(ndst.shape[0])
            for l in range(1, self.L, 2):
                lv = levels_old[l]
                if lv.size and bool((indeg[lv] == 1).all()) and bool((rank[drv[lv]] >= 0).all()):
                    levels_old[l] = lv[np.lexsort((eid[lv], rank[drv[lv]]))]
            order = np.concatenate(levels_old[0::2] + levels_old[1::2])          # new id k <- old id order[k]
            rest = np.setdiff1d(np.arange(self.N), order)             # nodes in no level keep trailing ids
            order = np.concatenate([order, rest])
            new_of_old = np.empty(self.N, dtype=np.int64)
            new_of_old[order] = np.arange(self.N)
        else:
            order = new_of_old = np.arange(self.N)
        self.old_of_new = order
        cat = lambda key: np.concatenate([getattr(d, key) + self.node_off[i] for i, d in enumerate(designs)])
        g = PinGraph(self.N, {'net': (new_of_old[cat('net_src')], new_of_old[cat('net_dst')]),
                              'cell': (new_of_old[cat('cell_src')], new_of_old[cat('cell_dst')])})
        rows = lambda key: torch.from_numpy(np.concatenate([getattr(d, key) for d in designs])[order])
        g.ndata['cell_feat'] = rows('cell_feat')
        g.ndata['net_feat'] = rows('net_feat')
        g.ndata['arrival_time'] = rows('arrival_time')
        g.ndata['required_time'] = rows('required_time')
        g.ndata['label'] = rows('label')
        g.ndata['end'] = rows('is_end')
        self.graph = g.to(device)
        self.out_dim = out_dim
        self.P = designs[0].map_size ** 2
        # same-index levels concatenated; python int lists, as the reference passes them (src/dataset.py:124-129)
        self.level_nodes = [new_of_old[lv].tolist() for lv in levels_old]
        masks = [PathMasks(d.mask_indptr, d.mask_cols, d.map_size ** 2, device) for d in designs]
        self.masks = masks[0] if self.B == 1 else PathMasks.batch(masks)
        self.images = torch.from_numpy(np.stack([d.image for d in designs])).to(device)
        self.arrival = self.graph.ndata['arrival_time']
        self.required = self.graph.ndata['required_time']
        self.level_th = [torch.tensor([float(l)], device=device) for l in range(self.L)]
        self.path2level = np.concatenate([d.path2level for d in designs])
        self.path2endpoint = new_of_old[np.concatenate([d.path2endpoint + self.node_off[i] for i, d in enumerate(designs)])]
        self.path2design = np.concatenate([np.full(d.num_paths, i, dtype=np.int64) for i, d in enumerate(designs)])

    def select(self, path_ids_per_design, static=None):
        """Bucket the sampled paths by level (src/train.py:476-484); order inside a level = design, then
        order of appearance.  One host->device copy for the whole step."""
        gp = np.concatenate([np.asarray(p, dtype=np.int64) + self.path_off[i]
                             for i, p in enumerate(path_ids_per_design)])
        lv = self.path2level[gp]
        order = np.argsort(lv.astype(np.int16) if self.L < 32768 else lv, kind='stable')     # 16-bit keys: numpy's radix sort
        gp, lv = gp[order], lv[order]
        ends = self.path2endpoint[gp]
        counts = np.bincount(lv, minlength=self.L)
        first, nxt = batch_links(gp, self.path2level.shape[0])
        T = gp.shape[0]
        eorder = np.argsort(ends, kind='stable')       # batch rows grouped by endpoint: deterministic gradient scatter
        self.ends_unique = bool(T < 2 or (np.diff(ends[eorder]) != 0).all())
        packed = np.concatenate([ends, gp, self.path2design[gp] * self.P, lv, nxt, eorder, first]).astype(np.int32)
        if static is not None:
            if static.numel() != packed.shape[0]:
                raise ValueError('graph replay needs a constant number of sampled paths per step')
            static.copy_(self._stage(packed), non_blocking=True)
            self._stage_done()
            dev = static
        else:
            dev = torch.from_numpy(packed).to(self.device)
        self.links = (dev[6 * T:], dev[4 * T:5 * T])
        self.end_order = dev[5 * T:6 * T]
        return dev[0:T], dev[T:2 * T], dev[2 * T:3 * T], counts, self.old_of_new[ends], dev[3 * T:4 * T]

    STAGING_SLOTS = 4

    def _stage(self, packed):
        """Copy one step's packed indices into the next of STAGING_SLOTS pinned host buffers.  A slot is reused only
        after the H2D copy that last read it has executed (its event), so the host may run up to STAGING_SLOTS - 1
        steps ahead of the device without overwriting indices that are still to be copied; the copy itself is a true
        asynchronous DMA (pinned source), not a staged pageable copy that stalls the host."""
        n = packed.shape[0]
        st = self.__dict__.get('_staging')
        if st is None or st['n'] != n:
            st = dict(n=n, i=0, bufs=[torch.empty(n, dtype=torch.int32).pin_memory() for _ in range(self.STAGING_SLOTS)],
                      events=[None] * self.STAGING_SLOTS)
            self._staging = st
        k = st['i'] % self.STAGING_SLOTS
        if st['events'][k] is not None:
            st['events'][k].synchronize()
        st['bufs'][k].numpy()[:] = packed
        return st['bufs'][k]

    def _stage_done(self):
        st = self._staging
        k = st['i'] % self.STAGING_SLOTS
        ev = st['events'][k] or torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        st['events'][k] = ev
        st['i'] += 1


class TrainStep:
    def __init__(self, pmodel, cnn, designs, device, lr=1e-3, weight_decay=0.0, fused_optimizer=True, world_size=1,
                 mode='sweep', overlap=True, with_optimizer=True, task='reg', cone=False, dense_path_map=False,
                 keep_grads=True):
        """mode='dropin': per-level model() calls exactly as src/train.py:490-511;
        mode='sweep': PathModel.forward_sweep, same arithmetic with level-invariant work hoisted.
        task='reg': MSE on the arrival time (nlabels = 1); task='cls': CrossEntropy on ndata['label'] with a
        nlabels-wide head (src/train.py:32,513-522; src/options.py:32,49)."""
        assert mode in ('dropin', 'sweep') and task in ('reg', 'cls')
        self.mode, self.task = mode, task
        # dense_path_map (drop-in mode): the reference's literal per-level map, th.index_select(path_masks, 0, paths).to_dense()
        # * feat_map (src/train.py:500-501: sparse COO masks with int64 values, promoted by the multiply), instead of the
        # MaskedPathMap handle - what an UNMODIFIED caller loop passes; kept for measuring that form (bench.py)
        self.dense_path_map = bool(dense_path_map) and mode == 'dropin'
        self._sp_masks = None
        # keep_grads=False: the fused Adam kernel clears the gradients it consumed (no zero_grad fill per step); .grad reads
        # zero after step()
        # cone=True (sweep mode): the level kernels skip every node outside the fan-in cone of the step's endpoints
        # (device-side mask per step; pays when few endpoints of a large design are sampled, see sweep_forward_all)
        self.cone = bool(cone) and mode == 'sweep'
        self.overlap = overlap and mode == 'sweep'
        self.side = torch.cuda.Stream(device=device) if self.overlap else None
        self.pmodel, self.cnn = pmodel, cnn
        self.device = torch.device(device)
        self.batch = DesignBatch(designs, device, pmodel.gnn.out_feat_dim if pmodel.gnn is not None else 128)
        self.world_size = world_size
        if cnn is not None and hasattr(cnn, 'set_per_sample_stats'):
            cnn.set_per_sample_stats(True)      # per-image statistics == the reference's one-image batches, any B
        params = trainable_parameters(pmodel, cnn)
        self.fused = fused_optimizer
        self.reducer = None
        if not with_optimizer:                  # evaluation-only harness (mmft.evaluate.validate): shares the modules
            self.optim = None
        elif fused_optimizer:
            if world_size > 1:
                # data parallel: one flat range per gradient bucket, reduced as soon as its gradients exist
                from .dist import GradReducer
                self.optim = FlatAdam(params, lr=lr, weight_decay=weight_decay, buckets=parameter_buckets(pmodel, cnn),
                                      zero_after_step=not keep_grads)
                self.reducer = GradReducer(self.optim, world_size, device, side_stream=self.side)
            else:
                self.optim = FlatAdam(params, lr=lr, weight_decay=weight_decay, zero_after_step=not keep_grads)
        else:
            self.optim = torch.optim.Adam(params, lr, weight_decay=weight_decay)     # src/train.py:431-435
        pmodel.train()
        if cnn is not None:
            cnn.train()
        self.h = torch.zeros((self.batch.N, self.batch.out_dim), dtype=torch.float32, device=self.device)

    # ---------------------------------------------------------------- forward of one mini-batch
    def forward(self, path_ids_per_design, _sel=None):
        b, g = self.batch, self.batch.graph
        ends_d, paths_d, foff_d, counts, ends_h, lv_d = _sel if _sel is not None else b.select(path_ids_per_design)
        if self.mode != 'sweep':
            self.h.zero_()                                                                # src/train.py:342,559
        # (whole-sweep entry: the *_self MLPs store every level node's row before anything reads it - no 268 MB fill)
        g.ndata['h'] = self.h
        if self.mode == 'sweep':
            # The netlist sweep and the U-Net are independent until the fusion head: the level-serial sweep
            # (many small, latency-bound launches) runs on a side HIP stream underneath the CNN's large kernels.
            from . import sweep as _sweep
            cur = torch.cuda.current_stream(self.device)
            h_gnn = None
            if self.pmodel.gnn is not None:
                if self.overlap:
                    self.side.wait_stream(cur)
                    with torch.cuda.stream(self.side):
                        h_gnn = _sweep.sweep_forward_all(self.pmodel.gnn, g, b.level_nodes, ends_d,
                                                         target_order=b.end_order, cone=self.cone)
                else:
                    h_gnn = _sweep.sweep_forward_all(self.pmodel.gnn, g, b.level_nodes, ends_d, target_order=b.end_order,
                                                     cone=self.cone)
            feat = self.cnn(b.images).reshape(b.B, -1) if self.cnn is not None else None
            pm = MaskedPathMap(b.masks, paths_d, feat, foff_d if b.B > 1 else None, *b.links) \
                if feat is not None else None
            # The sweep stream is joined BEFORE the masked projection.  Issued ahead of the join (it needs only the CNN output)
            # the projection's kernels ran beside the sweep's last launches, and in a replayed graph its prefix kernel then lost
            # one term in ~5 % of the launches - a wrong result of one packed-fma form under that co-residency, not a data race
            # (csrc/common.h MMFT_NO_PACKED_F32).  The kernel no longer contains the form; the join stays first
            # because it is free: the sweep finishes before the U-Net anyway (tools/step_timeline.py).  MMFT_JOIN_LATE=1
            # restores the old order for tools/packed_fma_repro.py.
            if h_gnn is not None and self.overlap and not _JOIN_LATE:
                cur.wait_stream(self.side)
                h_gnn.record_stream(cur)
            h_cnn = self.pmodel._fcn(pm) if (pm is not None and self.pmodel.fcn is not None) else None
            if h_gnn is not None and self.overlap and _JOIN_LATE:
                cur.wait_stream(self.side)
                h_gnn.record_stream(cur)
            return self.pmodel.fuse_heads(h_gnn, pm, lv_d, b.L, h_cnn=h_cnn), ends_d, ends_h
        feat = self.cnn(b.images).reshape(b.B, -1) if self.cnn is not None else None      # src/train.py:465,562
        g.targets_unique = b.ends_unique      # host knowledge: no repeated endpoint -> one exact atomic add per element
        hats, pos = [], 0
        for level_id in range(b.L):                                                       # src/train.py:490-511
            k = int(counts[level_id])
            targets = ends_d[pos:pos + k]
            path_map = None
            if k and feat is not None and self.dense_path_map:
                rows = paths_d[pos:pos + k].long()
                path_mask = torch.index_select(self._sparse_masks(), 0, rows).to_dense()              # src/train.py:500
                path_map = path_mask * (feat[foff_d[pos:pos + k].long() // b.P] if b.B > 1 else feat)  # src/train.py:501
            elif k and feat is not None:
                # per-level rows; the per-row feature-map offsets come from the packed selection (no lookup launch per
                # level), the link lists are derived only if a per-level backward asks for them
                path_map = MaskedPathMap(b.masks, paths_d[pos:pos + k], feat, foff_d[pos:pos + k] if b.B > 1 else None)
                # the step's link lists (all levels, in level order) for the deferred head, which differentiates all levels
                # as one batch: without them it derives the lists from a device-to-host copy of the rows
                path_map.batch_links, path_map.batch_rows = b.links, (pos, paths_d.numel())
            cur = self.pmodel(g, b.level_nodes[level_id], None, targets, level_id, b.level_th[level_id], path_map)
            pos += k
            if cur is not None:
                hats.append(cur)
        return torch.cat(hats, 0), ends_d, ends_h

    def _sparse_masks(self):
        """path_masks as the reference stores them: sparse COO (num_paths, P) with int64 ones
        (src/verilog_parser_asap7.py:1353-1355,1368), on the device."""
        if self._sp_masks is None:
            m = self.batch.masks
            ip, cols = torch.from_numpy(m.host_indptr), torch.from_numpy(m.host_cols)
            rows = torch.repeat_interleave(torch.arange(m.num_paths), ip[1:] - ip[:-1])
            self._sp_masks = torch.sparse_coo_tensor(torch.stack([rows, cols]), torch.ones(cols.numel(), dtype=torch.int64),
                                                     (m.num_paths, m.P)).coalesce().to(self.device)
        return self._sp_masks

    def loss(self, hats, ends_d):
        """src/train.py:513-522: CrossEntropy(label_hats, labels) for task 'cls', MSE(label_hats, arrival_time) for 'reg'."""
        if self.task == 'cls':
            labels = self.batch.graph.ndata['label'][ends_d.long()].squeeze(-1).contiguous()
            return cross_entropy_loss(hats, labels)
        return mse_loss_gather(hats, self.batch.arrival, ends_d)          # arrival_time[target_list], gathered in the kernel

    def step(self, path_ids_per_design):
        """One mini-batch: forward, MSE on arrival time, backward, (all-reduce,) Adam.
        Returns (loss tensor, predictions, target node ids); `self.last_ends` holds the endpoints' device-side
        (renumbered) node ids, the index space of `self.batch.arrival` / `.required`."""
        hats, ends_d, ends_h = self.forward(path_ids_per_design)
        self.last_ends = ends_d
        loss = self.loss(hats, ends_d)
        self.optim.zero_grad()
        if self.reducer is not None:
            # one delivery per parameter and step only in the whole-sweep form; the per-level loop adds into the head's
            # gradients once per level, so its buckets are reduced after backward() has returned
            self.reducer.begin(early=(self.mode == 'sweep'))
            loss.backward(unit_grad(self.device))
            self.reducer.finish()            # buckets not yet reduced; compute stream waits for the last Adam
        elif self.world_size > 1:
            from .dist import allreduce_sum_
            loss.backward()
            grads = [p.grad for p in self.optim.param_groups[0]['params'] if p.grad is not None]
            for g in grads:                  # torch.optim.Adam route: per-tensor all-reduce, mean of the ranks
                allreduce_sum_(g, self.world_size)
                g.mul_(1.0 / self.world_size)
            self.optim.step()
        else:
            loss.backward(unit_grad(self.device))       # cached scalar 1.0: no ones_like fill, no multiply in the loss backward
            self.optim.step()
        return loss.detach(), hats.detach(), ends_h.tolist()


class GraphedTrainStep:
    """The whole mini-batch (forward, loss, backward, fused Adam) captured ONCE and replayed per step: the ~450 kernel
    launches of a step then cost no host time.  Per step the host only packs the sampled endpoints into a pinned int32
    staging slot (one asynchronous H2D copy into a static device buffer); Adam's step counter lives on the device, so
    nothing else is uploaded.  Requires a constant number of sampled paths per step.

    pieces=False (default on one GPU): ONE graph for the whole step, the netlist sweep forked onto a side stream inside
    it.  pieces=True (default under data parallelism): the step is captured as FIVE single-stream HIP graphs that are
    replayed on two streams with ordinary events between them,

        side stream :  [A  netlist sweep forward]                  [bA  reverse sweep + GNN weight gradients]
        main stream :  [B  U-Net forward] [H  fusion head, loss, head backward] [bB  U-Net backward] Adam

    The autograd graph is cut at the two tensors that cross the boundaries (the endpoint embeddings and the CNN feature
    map: detached leaves on the head's side, their .grad fed to the backward piece of the producer), so every piece has
    a backward of its own.  The cut behind H is where a data-parallel step hands the head's finished gradient bucket
    (2.27 M of 2.89 M parameters) to the communication stream: it is all-reduced underneath bA / bB - inside one
    captured graph no point in the middle can be signalled to an outside stream on this stack (mmft.dist.GradReducer).
    Measured on one GPU (rocprofv3 kernel traces, tools/trace_overlap.py, tools/replay_host_time.py) both forms give the
    same step time within 0.2 ms: the sweep and the U-Net overlap only partially in either (A + B in isolation: 1.43 +
    0.92 ms alone, 1.94 ms together) because the level chain's dependent launches wait for CU slots behind the U-Net's
    chip-filling kernels - contention, not a scheduling artefact - so the single graph (one launch, Adam inside) stays
    the one-GPU default.  No collective is ever captured.
    """

    def __init__(self, ts, example_path_ids, warmup=3, pieces=None):
        if not ts.fused or ts.mode != 'sweep':
            raise ValueError('GraphedTrainStep needs mode="sweep" and the fused optimizer')
        self.ts = ts
        b = ts.batch
        if pieces is None:
            pieces = ts.world_size > 1          # see the class docstring: the cuts pay for themselves under data parallelism
        self.pieces = bool(pieces) and ts.overlap and ts.pmodel.gnn is not None and ts.cnn is not None and \
            ts.pmodel.fcn is not None
        for _ in range(warmup):                       # optional eager optimizer steps before the capture
            ts.step(example_path_ids)
        # one eager forward + backward WITHOUT an optimizer step: uploads the cached level rows and allocates
        # every workspace / sweep buffer, none of which may happen while the stream is capturing.  Run on a
        # side stream (as torch's capture recipe asks) and drop every reference to its autograd graph before
        # capturing, otherwise stale AccumulateGrad nodes bound to another stream break the capture.
        warm = torch.cuda.Stream(device=ts.device)
        warm.wait_stream(torch.cuda.current_stream(ts.device))
        with torch.cuda.stream(warm):
            hats0, ends0, _ = ts.forward(example_path_ids)
            ts.optim.zero_grad()
            loss0 = ts.loss(hats0, ends0)
            loss0.backward()
            ts.optim.zero_grad()
            del hats0, ends0, loss0
        torch.cuda.current_stream(ts.device).wait_stream(warm)
        import gc
        gc.collect()
        torch.cuda.synchronize()
        sel = b.select(example_path_ids)
        T = sel[0].numel()
        self.static_idx = torch.zeros(6 * T + b.path2level.shape[0], dtype=torch.int32, device=ts.device)
        self.T = T
        sel = b.select(example_path_ids, static=self.static_idx)
        torch.cuda.synchronize()
        if ts.optim.zero_after_step:
            # the Adam launch of every replay leaves the gradients zero for the next one: start from that state, so that the
            # captured zero_grad() records no fill
            ts.optim.flat_grad.zero_()
            ts.optim._cleared = [True] * len(ts.optim._cleared)
            torch.cuda.synchronize()
        if self.pieces:
            self._capture_pieces(sel)
        else:
            self.graph = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread may query events while this thread captures
            with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
                hats, ends_d, _ = ts.forward(None, _sel=sel)
                loss = ts.loss(hats, ends_d)
                ts.optim.zero_grad()
                loss.backward(unit_grad(ts.device))  # data parallel: the buckets are reduced behind every replay
                if ts.reducer is None:
                    ts.optim.step_captured()
                self.loss, self.hats = loss.detach(), hats.detach()
        torch.cuda.synchronize()

    def _capture_pieces(self, sel):
        from . import sweep as _sweep
        ts, b = self.ts, self.ts.batch
        pm_, cnn, g = ts.pmodel, ts.cnn, ts.batch.graph
        ends_d, paths_d, foff_d, counts, ends_h, lv_d = sel
        skip = ('gnn.fc_net_drive.', 'gnn.fc_attn2.')
        named = [(n, p) for n, p in pm_.named_parameters() if not n.startswith(skip)]
        head_params = [p for n, p in named if not n.startswith(('gnn.', 'cnn.'))]
        gnn_params = [p for n, p in named if n.startswith('gnn.')]
        cnn_params = list(cnn.parameters())
        pool = torch.cuda.graph_pool_handle()
        mk = lambda: torch.cuda.CUDAGraph()
        self.gA, self.gB, self.gH, self.gbA, self.gbB = mk(), mk(), mk(), mk(), mk()
        kw = dict(pool=pool, capture_error_mode='thread_local')
        g.ndata['h'] = ts.h
        with torch.cuda.graph(self.gA, stream=ts.side, **kw):                       # A: netlist sweep forward
            h_out = _sweep.sweep_forward_all(pm_.gnn, g, b.level_nodes, ends_d, target_order=b.end_order, cone=ts.cone)
            h_leaf = h_out.detach().requires_grad_(True)
        with torch.cuda.graph(self.gB, **kw):                                       # B: U-Net forward
            feat_out = cnn(b.images).reshape(b.B, -1)
            feat_leaf = feat_out.detach().requires_grad_(True)
        with torch.cuda.graph(self.gH, **kw):                                       # H: head forward, loss, head backward
            pmap = MaskedPathMap(b.masks, paths_d, feat_leaf, foff_d if b.B > 1 else None, *b.links)
            hats = pm_.fuse_heads(h_leaf, pmap, lv_d, b.L, h_cnn=pm_._fcn(pmap))
            loss = ts.loss(hats, ends_d)
            ts.optim.zero_grad()
            torch.autograd.backward(loss, inputs=head_params + [h_leaf, feat_leaf])
            self.loss, self.hats = loss.detach(), hats.detach()
        with torch.cuda.graph(self.gbA, stream=ts.side, **kw):                      # bA: reverse sweep + GNN weight grads
            torch.autograd.backward(h_out, grad_tensors=h_leaf.grad, inputs=gnn_params)
        with torch.cuda.graph(self.gbB, **kw):                                      # bB: U-Net backward
            torch.autograd.backward(feat_out, grad_tensors=feat_leaf.grad, inputs=cnn_params)
        self._keep = (h_out, h_leaf, feat_out, feat_leaf, pmap)
        self._ev = [torch.cuda.Event() for _ in range(6)]

    def step(self, path_ids_per_design):
        """Host side of a step: pack the sampled endpoints (pinned staging slot -> one async H2D copy), replay.  No
        host synchronisation: the host may run ahead of the device (the staging slots bound it)."""
        ts, b = self.ts, self.ts.batch
        sel = b.select(path_ids_per_design, static=self.static_idx)
        ts.last_ends = sel[0]
        if not self.pieces:
            self.graph.replay()
            if ts.reducer is not None:
                ts.reducer.after_replay()        # per bucket: all-reduce + Adam on the communication stream
            else:
                ts.optim.note_replay()
            return self.loss, self.hats, sel[4].tolist()
        main, side = torch.cuda.current_stream(ts.device), ts.side
        e0, eA, eH, ebA = self._ev[:4]
        e0.record(main)                          # the endpoint indices are in place (and the previous step's Adam is done)
        side.wait_event(e0)
        with torch.cuda.stream(side):
            self.gA.replay()
            eA.record(side)
        self.gB.replay()
        main.wait_event(eA)
        self.gH.replay()
        eH.record(main)
        side.wait_event(eH)
        if ts.reducer is not None:
            ts.reducer.reduce_bucket(0, [eH])    # the head's gradients are final: all-reduce them under bA / bB
        with torch.cuda.stream(side):
            self.gbA.replay()
            ebA.record(side)
        self.gbB.replay()
        main.wait_event(ebA)
        if ts.reducer is not None:
            ts.reducer.reduce_rest_and_join(main)
        else:
            ts.optim.step()
        return self.loss, self.hats, sel[4].tolist()

    def time_pieces(self, reps=10):
        """Diagnostic: every piece replayed ALONE on the stream it is assigned to (ms each), plus A | B and bA | bB
        together.  Only for pieces mode; leaves the parameters untouched (no optimizer step) but overwrites gradients."""
        import time
        ts = self.ts
        main = torch.cuda.current_stream(ts.device)
        qa, qb = ts.side, main

        def run(pairs):
            def once():
                for g, q in pairs:
                    q.wait_stream(main)
                    with torch.cuda.stream(q):
                        g.replay()
                for g, q in pairs:
                    main.wait_stream(q)
            for _ in range(2):
                once()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                once()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / reps * 1e3
        out = {}
        for name, pairs in (('A', [(self.gA, qa)]), ('B', [(self.gB, qb)]), ('A|B', [(self.gA, qa), (self.gB, qb)]),
                            ('H', [(self.gH, main)]), ('bA', [(self.gbA, qa)]), ('bB', [(self.gbB, qb)]),
                            ('bA|bB', [(self.gbA, qa), (self.gbB, qb)]),
                            ('A on main', [(self.gA, main)]), ('B on main', [(self.gB, main)]),
                            ('bA on main', [(self.gbA, main)]), ('bB on main', [(self.gbB, main)])):
            out[name] = round(run(pairs), 3)
        return out
