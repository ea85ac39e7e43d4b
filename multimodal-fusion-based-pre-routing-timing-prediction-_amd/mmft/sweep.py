"""Levelized netlist sweep on the GPU: forward per level, deterministic reverse sweep, batched weight grads.

Mirrors PathConv.forward (reference src/model.py:158-213) one call per topological level, as the
reference loops demand (src/train.py:490-511), but keeps ONE set of [N, .] buffers updated in place
instead of DGL's out-of-place frame update per pull:

    h   [N, D]   node embeddings (= graph.ndata['h'], rows written once per sweep)
    A   [N, D]   softmax-weighted fan-in sums of cell-level nodes      LSE [N, D]  their log-sum-exp
    HS  [N, Hd]  hidden activations of the *_self MLPs                 HN  [N, Hd] of fc_cell_neigh
    G   [N, D]   d loss / d pre-activation (reverse sweep)             DA  [N, D]  d loss / d A

Autograd sees a chain of LevelFn nodes linked by a 1-element token, so the engine runs the levels'
backward in strictly decreasing level order.  Each backward does: scatter the target gradients,
one pull kernel over the level's OUT-edges (no atomics, bitwise reproducible), and for cell levels the
two small GEMMs that turn G into DA.  Weight gradients are NOT computed per level: the level-0 node
(last to run) computes all of them in batched GEMMs over every node of the sweep.
"""
import torch
from . import ops, gradsink, lib

_MLP_KEYS = ('fc_cell_self', 'fc_net_self', 'fc_cell_neigh')


class SweepState:
    def __init__(self, graph, conv):
        self.graph = graph
        h = graph.ndata['h']
        if not (torch.is_tensor(h) and h.is_cuda and h.dtype == torch.float32 and h.dim() == 2 and h.is_contiguous()):
            raise RuntimeError("PathConv: graph.ndata['h'] must be a contiguous float32 CUDA tensor [N, out_dim] "
                               "(the HIP sweep has no CPU fallback)")
        if h.shape[0] != graph.number_of_nodes() or h.shape[1] != conv.out_feat_dim:
            raise ValueError(f"graph.ndata['h'] has shape {tuple(h.shape)}, expected "
                             f"({graph.number_of_nodes()}, {conv.out_feat_dim})")
        self.h = h
        self.N, self.D = h.shape
        self.Hd = conv.fc_cell_self.layers[0].weight.shape[0]
        self.relu = conv.activation is not None
        self.cell_feat = _feat(graph, 'cell_feat', conv.cell_feat_dim)
        self.net_feat = _feat(graph, 'net_feat', conv.net_feat_dim)
        self.params = [p for k in _MLP_KEYS for p in _mlp2_params(getattr(conv, k))]
        self.need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.params)
        bufs = graph.__dict__.setdefault('_sweep_bufs', {})
        key = (self.N, self.D, self.Hd, h.device)
        if bufs.get('key') != key:
            bufs.clear()
            bufs['key'] = key
        self._bufs = bufs
        self.A = self._buf('A', self.D)
        self.LSE = self._buf('LSE', self.D)
        self.HS = self._buf('HS', self.Hd)
        self.HN = self._buf('HN', self.Hd)
        self.G = self.DA = self.DHN = self.tflag = None
        self.level_meta = None      # per-level static facts (contiguous range, algorithmic bytes), whole-sweep entry
        self.levels = []            # (level_id, rows) in forward order
        self.token = None
        self.next_level = 0
        self.bwd_active = False
        self.complete = False               # level lists = a complete schedule of the graph (whole-sweep entry)
        self.fold = None                    # static facts for the folded level kernels (PinGraph.fold_schedule)
        self.level_lists = None
        self.PRE = None
        self.attn = None                    # attention branch (flag_attn): dict(key, c12, alpha, dcp, o2i)
        self.wpack = None                   # bf16 math mode: fc_cell_neigh pre-packed as bf16 (W1, W2, W2^T, W1^T)
        self.active = None                  # uint8 per node: fan-in cone of the step's endpoints (None: every node)
        self.spec_lists = None              # speculative drop-in sweep: the level lists it ran with, its token, target rows
        self.spec_token = None
        self.spec_tix = []
        self.target_order = None
        self.targets_unique = None

    def _buf(self, name, width, dtype=torch.float32):
        b = self._bufs.get(name)
        if b is None:
            # zeros, not empty: with fan-in-cone pruning rows outside the cone are never written, yet batched GEMMs read
            # them next to zero gradients - they must hold finite values
            b = torch.zeros((self.N, width), dtype=dtype, device=self.h.device)
            self._bufs[name] = b
        return b

    def begin_backward(self, zero_da=True, zero_g=True):
        """zero_da=False (whole-sweep entry): every DA row the reverse pull reads belongs to a cell node of a level >= 2,
        and the reverse sweep writes that row (fc_cell_neigh's input gradient) before any lower level pulls from it,
        so the 268 MB fill per step is skipped; the per-level drop-in form keeps it (a caller may stop a sweep early)."""
        if not self.bwd_active:
            if self.G is None:
                self.G = self._buf('G', self.D)
                fresh = 'DA' not in self._bufs
                self.DA = self._buf('DA', self.D)
                if fresh:
                    self.DA.zero_()
            if zero_g:
                self.G.zero_()
            else:
                # whole-sweep entry: only the sampled endpoints start with a gradient of their own; the reverse pull
                # takes zero for every other row (flag per node), so the 268 MB fill of G per step is skipped too
                if 'tflag' not in self._bufs:
                    self._bufs['tflag'] = torch.zeros(self.N, dtype=torch.uint8, device=self.h.device)
                self.tflag = self._bufs['tflag']
            if zero_da:
                self.DA.zero_()
            self.bwd_active = True


def _attn_state(st, graph, c12):
    """Buffers of the attention branch (src/model.py:119-136,190-196): ndata['key'] as a flat vector, the two score
    coefficients on the device, alpha per cell in-edge, per-node score-gradient partials."""
    key = graph.ndata.get('key')
    if key is None:
        raise RuntimeError("PathConv(flag_attn=True) reads graph.ndata['key'] (src/model.py:132-136); no reference file "
                           "creates it (SURVEY D6) - provide a float32 [N, 1] tensor")
    if not (torch.is_tensor(key) and key.is_cuda and key.dtype == torch.float32 and key.numel() == st.N):
        raise RuntimeError("graph.ndata['key'] must be a float32 CUDA tensor with one value per node")
    E = graph.csr('in', 'cell')[1].numel()
    bufs = st._bufs
    if bufs.get('alpha') is None or bufs['alpha'].numel() != max(E, 1):
        bufs['alpha'] = torch.zeros(max(E, 1), dtype=torch.float32, device=st.h.device)
        bufs['dcp'] = torch.zeros((st.N, 2), dtype=torch.float32, device=st.h.device)
    return dict(key=key.reshape(-1).contiguous(), c12=c12.detach().reshape(2).contiguous(), alpha=bufs['alpha'],
                dcp=bufs['dcp'], o2i=graph.out2in('cell'))


def _cell_gather_fwd(st, g, rows):
    """A[rows] (and what the reverse sweep needs) for one cell level: per-channel softmax (src/model.py:113-116) or,
    with flag_attn, one softmax weight per edge (src/model.py:119-136)."""
    if st.attn is not None:
        a = st.attn
        ops.seg_attn_fwd(st.h, a['key'], a['c12'], g.csr('in', 'cell'), rows, st.A, a['alpha'])
    else:
        ops.seg_softmax_sum_fwd(st.h, g.csr('in', 'cell'), rows, st.A, st.LSE)


def _pull_bwd(st, g, rows, own=None, alg_bytes=0, heavy=None):
    if st.attn is not None:
        a = st.attn
        ops.level_bwd_pull_attn(st.G, st.h, rows, g.csr('out', 'net'), g.out_net_weight(), g.csr('out', 'cell'), a['o2i'],
                                a['alpha'], st.DA, relu=st.relu, own=own)
    else:
        ops.level_bwd_pull(st.G, st.h, rows, g.csr('out', 'net'), g.out_net_weight(), g.csr('out', 'cell'),
                           st.A, st.LSE, st.DA, relu=st.relu, alg_bytes=alg_bytes, own=own, heavy=heavy, active=st.active)


def _attn_scores_bwd(st, g, rows):
    a = st.attn
    ops.seg_attn_bwd_scores(st.DA, st.h, st.A, a['alpha'], a['key'], a['c12'], g.csr('in', 'cell'), rows, a['dcp'])


def _attn_c12_grad(st, rows):
    """d loss / d c12 (2, 1) = the column sums of the score-gradient partials of the cell rows (range or index)."""
    dcp = st.attn['dcp']
    if isinstance(rows, tuple):
        return ops.colsum(dcp[rows[0]:rows[0] + rows[1]]).reshape(2, 1)
    return ops.colsum(dcp, idx=rows).reshape(2, 1)


def _feat(graph, name, width):
    t = graph.ndata[name]
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2):
        raise RuntimeError(f"graph.ndata['{name}'] must be a float32 CUDA tensor")
    if t.shape[1] != width:
        raise ValueError(f"graph.ndata['{name}'] has {t.shape[1]} columns, the model expects {width}")
    if t.stride(1) != 1:
        t = t.contiguous()
        graph.ndata[name] = t
    return t


def _mlp2_params(mlp):
    lin = [m for m in mlp.layers if isinstance(m, torch.nn.Linear)]
    if len(lin) != 2 or len(mlp.layers) != 3 or getattr(mlp, 'negative_slope', 0) != 0:
        raise NotImplementedError('PathConv level kernels expect Linear-LeakyReLU(0)-Linear MLPs (src/model.py:48-51)')
    return [lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias]


def _w(p):
    d = p.detach()
    return d if d.is_contiguous() else d.contiguous()


def _cell_neigh_fwd(st, rows, w1g, b1g, w2g, b2g, act):
    """h[rows] = act(h[rows] + fc_cell_neigh(A[rows])), HN[rows] saved: one fused launch when the widths allow."""
    if st.wpack is not None:
        ops.mlp2_rows_bf16(st.A, rows, st.wpack[0], b1g, st.wpack[1], b2g, st.h, hid_out=st.HN, add_act=True,
                           relu_out=(act == ops.ACT_RELU), active=st.active)
    elif ops.mlp2_fusable(st.D, st.Hd, st.D):
        ops.mlp2_rows(st.A, rows, w1g, b1g, w2g, b2g, st.h, kmajor=False, hid_out=st.HN, add_act=True,
                      relu_out=(act == ops.ACT_RELU), active=st.active)
    else:
        ops.linear_fwd(st.A, w1g, b1g, y=st.HN, xidx=rows, yidx=rows, act=ops.ACT_RELU)
        ops.linear_fwd(st.HN, w2g, b2g, y=st.h, xidx=rows, yidx=rows, epi=ops.EPI_ADD_ACT, act=act)


def _cell_neigh_bwd(st, rows, w1g, w2g, keep_dhn=False):
    """DA[rows] = ((G[rows] W2g) * relu'(HN[rows])) W1g.  With keep_dhn the hidden gradient rows are also written to
    st.DHN, so the batched weight-gradient pass does not recompute them."""
    if ops.mlp2_fusable(st.D, st.Hd, st.D):
        dhn_out = None
        if keep_dhn:
            if st.DHN is None:
                st.DHN = st._buf('DHN16', st.Hd, torch.bfloat16) if getattr(st, 'hid16', False) else st._buf('DHN', st.Hd)
            dhn_out = st.DHN
        if st.wpack is not None:
            ops.mlp2_rows_bf16(st.G, rows, st.wpack[2], None, st.wpack[3], None, st.DA, mask=st.HN, hid_out=dhn_out,
                               active=st.active)
            return
        ops.mlp2_rows(st.G, rows, w2g, None, w1g, None, st.DA, kmajor=True, mask=st.HN, hid_out=dhn_out, active=st.active)
    else:
        dhn = ops.linear_dgrad(st.G, w2g, gidx=rows, mask=st.HN, maskidx=rows)
        ops.linear_dgrad(dhn, w1g, dx=st.DA, dxidx=rows)


class LevelFn(torch.autograd.Function):
    """One PathConv.forward call. Inputs: chain token, the attention coefficients c12 (or None) (+ the 12 MLP parameters
    at level 0)."""

    @staticmethod
    def forward(ctx, token, state, level_id, rows, tix, c12, *params):
        st, g = state, state.graph
        if c12 is not None:
            st.attn = _attn_state(st, g, c12)
        P = [_w(p) for p in st.params]
        (w1c, b1c, w2c, b2c, w1n, b1n, w2n, b2n, w1g, b1g, w2g, b2g) = P
        n = rows.numel()
        act = ops.ACT_RELU if st.relu else ops.ACT_NONE
        if n:
            if level_id % 2 == 1:                                                     # net level :186-187
                ops.linear_fwd(st.net_feat, w1n, b1n, y=st.HS, xidx=rows, yidx=rows, act=ops.ACT_RELU)
                ops.linear_fwd(st.HS, w2n, b2n, y=st.h, xidx=rows, yidx=rows)
                ops.seg_mean_add_act_fwd(st.h, g.csr('in', 'net'), rows, relu=st.relu)
            elif level_id == 0:                                                       # :148-153
                ops.linear_fwd(st.cell_feat, w1c, b1c, y=st.HS, xidx=rows, yidx=rows, act=ops.ACT_RELU)
                ops.linear_fwd(st.HS, w2c, b2c, y=st.h, xidx=rows, yidx=rows, act=act)
            else:                                                                     # cell level :113-116,138-146
                _cell_gather_fwd(st, g, rows)
                ops.linear_fwd(st.cell_feat, w1c, b1c, y=st.HS, xidx=rows, yidx=rows, act=ops.ACT_RELU)
                ops.linear_fwd(st.HS, w2c, b2c, y=st.h, xidx=rows, yidx=rows)
                _cell_neigh_fwd(st, rows, w1g, b1g, w2g, b2g, act)
        st.levels.append((level_id, rows))
        out = ops.gather_rows(st.h, tix) if tix.numel() else st.h.new_zeros((0, st.D))       # :213
        ctx.state, ctx.level_id, ctx.rows, ctx.tix = st, level_id, rows, tix
        ctx.nparams = len(params)
        ctx.has_c12 = c12 is not None
        new_token = st.h.new_zeros(1)
        return new_token, out

    @staticmethod
    def backward(ctx, gtoken, gout):
        st, g, level_id, rows = ctx.state, ctx.state.graph, ctx.level_id, ctx.rows
        st.begin_backward()
        if ctx.tix.numel() and gout is not None:
            go = ops.strided_rows(gout)
            ops.scatter_add_targets(st.G, ctx.tix, go, unique=g.__dict__.get('targets_unique'))
        P = [_w(p) for p in st.params]
        (w1c, b1c, w2c, b2c, w1n, b1n, w2n, b2n, w1g, b1g, w2g, b2g) = P
        dc = None
        if rows.numel():
            _pull_bwd(st, g, rows)
            if level_id % 2 == 0 and level_id > 0:
                _cell_neigh_bwd(st, rows, w1g, w2g)
                if ctx.has_c12 and st.attn is not None:
                    _attn_scores_bwd(st, g, rows)
                    dc = _attn_c12_grad(st, rows)
        grads = [None] * ctx.nparams
        if level_id == 0:
            if ctx.nparams:
                grads = _batched_param_grads(st, P)
            st.bwd_active = False
        return (torch.zeros_like(gtoken) if ctx.needs_input_grad[0] else None, None, None, None, None, dc, *grads)


def _cat_rows(st, pred):
    """Rows of the levels selected by `pred`, in level order: a (start, n) range when the levels are contiguous id
    ranges that follow each other (DesignBatch's renumbering makes them so), else one int32 index tensor."""
    sel = [(l, r) for (l, r) in st.levels if pred(l) and r.numel()]
    if not sel:
        return None
    if st.level_meta:
        rng = [st.level_meta[l]['range'] if st.level_meta[l] else None for l, _ in sel]
        if all(rng) and all(rng[i][0] + rng[i][1] == rng[i + 1][0] for i in range(len(rng) - 1)):
            return (rng[0][0], sum(n for _, n in rng))
    return sel[0][1] if len(sel) == 1 else torch.cat([r for _, r in sel])


def _rows_of(t, rows):
    """(tensor the kernel should see, index tensor or None) for a row set that is a range or an index tensor."""
    if isinstance(rows, tuple):
        return t[rows[0]:rows[0] + rows[1]], None
    return t, rows


def _linear_rows(x, w, b, y, rows, **kw):
    """y[rows] = epi(x[rows] @ w.T + b) for a range or an index row set."""
    xv, idx = _rows_of(x, rows)
    yv, _ = _rows_of(y, rows)
    return ops.linear_fwd(xv, w, b, y=yv, xidx=idx, yidx=idx, **kw)


def _mlp_grads(st, sinks, G, gidx, H, X, xidx, w2, dH_rows=None):
    """Gradients of one Linear-ReLU-Linear MLP over the rows `gidx`: (dW1, db1, dW2, db2); entries whose parameter
    has a gradient sink (mmft.gradsink) are written there and returned as None.
    dH_rows: [N, Hd] buffer already holding the hidden gradients of those rows (from the level loop)."""
    s1w, s1b, s2w, s2b = sinks
    Gv, gi = _rows_of(G, gidx)
    Hv, _ = _rows_of(H, gidx)
    Xv, xi = _rows_of(X, xidx)
    dW2, db2 = gradsink.deliver_pair(s2w, s2b, lambda ow, ob: ops.linear_wgrad(Gv, Hv, dw=ow, gidx=gi, xidx=gi, db=ob,
                                                                               with_bias=True))
    if dH_rows is not None:
        Dv, _ = _rows_of(dH_rows, gidx)
        dW1, db1 = gradsink.deliver_pair(s1w, s1b, lambda ow, ob: ops.linear_wgrad(Dv, Xv, dw=ow, gidx=gi, xidx=xi,
                                                                                   db=ob, with_bias=True))
    elif FUSED_FIRST_LAYER_GRADS and xidx is gidx and ops.first_layer_grads_fusable(X.shape[1], H.shape[1], G.shape[1]) \
            and w2.stride(0) % 4 == 0:
        # dH = (G W2) * relu'(H) never leaves the registers: one launch instead of GEMM -> 268 MB -> GEMM
        dW1, db1 = gradsink.deliver_pair(s1w, s1b, lambda ow, ob: ops.mlp2_first_layer_grads(G, H, X, gidx, w2, dw1=ow,
                                                                                             db1=ob))
    else:
        dH = ops.linear_dgrad(Gv, w2, gidx=gi, mask=Hv, maskidx=gi)
        dW1, db1 = gradsink.deliver_pair(s1w, s1b, lambda ow, ob: ops.linear_wgrad(dH, Xv, dw=ow, xidx=xi, db=ob,
                                                                                   with_bias=True))
    return [dW1, db1, dW2, db2]


def _batched_param_grads(st, P, dhn_ready=False):
    (w1c, b1c, w2c, b2c, w1n, b1n, w2n, b2n, w1g, b1g, w2g, b2g) = P
    # a sink is only usable when the kernel-side tensor IS the parameter's storage (contiguous weights)
    S = [gradsink.of(p) if p.is_contiguous() else None for p in st.params]
    rc = _cat_rows(st, lambda l: l % 2 == 0)
    rn = _cat_rows(st, lambda l: l % 2 == 1)
    rc2 = _cat_rows(st, lambda l: l % 2 == 0 and l > 0)
    zeros = lambda ps: [torch.zeros_like(p) for p in ps]
    if getattr(st, 'feat_fused', False):
        feat = lambda sinks, rows, X, w1, b1, w2: gradsink.deliver_many(
            sinks, lambda o1, ob1, o2, ob2: ops.mlp2_feat_bwd_bf16(st.G, X, rows, w1, b1, w2, dw1=o1, db1=ob1, dw2=o2, db2=ob2))
        gc = feat(S[0:4], rc, st.cell_feat, w1c, b1c, w2c) if rc is not None else zeros(P[0:4])
        gn = feat(S[4:8], rn, st.net_feat, w1n, b1n, w2n) if rn is not None else zeros(P[4:8])
    else:
        gc = _mlp_grads(st, S[0:4], st.G, rc, st.HS, st.cell_feat, rc, w2c) if rc is not None else zeros(P[0:4])
        gn = _mlp_grads(st, S[4:8], st.G, rn, st.HS, st.net_feat, rn, w2n) if rn is not None else zeros(P[4:8])
    gg = _mlp_grads(st, S[8:12], st.G, rc2, st.HN, st.A, rc2, w2g,
                    st.DHN if (dhn_ready and st.DHN is not None) else None) if rc2 is not None else zeros(P[8:12])
    return gc + gn + gg


def attention_coefficients(conv):
    """c12 (2, 1): <fc_attn.w[:dk], fc_key.w> and <fc_attn.w[dk:], fc_key.w> - the edge score of message_func_attn
    (src/model.py:132-136) is leaky_relu(c12[0] key_src + c12[1] key_dst) because fc_key / fc_attn have no bias.  One tiny
    GEMM through the autograd-aware dense kernel, so the gradient reaches both parameters."""
    from . import functional as MF
    dk = conv.fc_key.weight.shape[0]
    return MF.linear_act(conv.fc_attn.weight.view(2, dk), conv.fc_key.weight.view(1, dk), None, None)


def level_forward(conv, graph, cur_nodes, targets, level_id):
    """Body of PathConv.forward (both branches)."""
    seen = graph.__dict__.setdefault('_seen_lists', [])
    spec_active = False
    if level_id == 0:
        spec = graph.__dict__.get('_spec_lists')
        if len(seen) >= 2 and (spec is None or len(seen) >= len(spec)):
            spec = graph.__dict__['_spec_lists'] = list(seen)            # the previous sweep's lists, levels 0 .. k
        del seen[:]
        if SPECULATE and spec is not None and not graph.__dict__.get('_spec_disabled') and not torch.is_tensor(cur_nodes) \
                and _same_list(cur_nodes, spec[0]):
            dev_ = graph.ndata['h'].device
            # (only with gradient sinks on every parameter - FlatAdam: parameters that accumulate through autograd's own
            # AccumulateGrad nodes, created under the caller's stream, would be fed from another stream)
            sinks_ = all(gradsink.of(p) is not None for k in _MLP_KEYS for p in _mlp2_params(getattr(conv, k)))
            if SPEC_SIDE_STREAM and sinks_ and not torch.cuda.is_current_stream_capturing():
                # the whole sweep runs on a stream of its own: its autograd node then runs its backward there as well, next
                # to the U-Net's backward on the caller's stream (the caller's loop has one stream; inside a captured step
                # TrainStep forks the streams itself).  The caller's stream waits for the sweep right away - every later level
                # call reads h.
                side = graph.__dict__.get('_spec_stream')
                if side is None or side.device != dev_:
                    side = graph.__dict__['_spec_stream'] = torch.cuda.Stream(device=dev_)
                cur_ = torch.cuda.current_stream(dev_)
                side.wait_stream(cur_)
                with torch.cuda.stream(side):
                    st, token = _run_sweep(conv, graph, spec, None)
                cur_.wait_stream(side)
            else:
                st, token = _run_sweep(conv, graph, spec, None)
            st.spec_lists, st.spec_token, st.spec_tix, st.next_level = spec, token, [], 0
            spec_active = True
    else:
        spec_active = graph._sweep is not None and graph._sweep.spec_lists is not None
    if not torch.is_tensor(cur_nodes):
        seen.append(cur_nodes)
    if spec_active:
        st = graph._sweep
        if st.h is not graph.ndata['h']:
            raise RuntimeError("graph.ndata['h'] was replaced in the middle of a sweep; restart from level 0")
        if level_id != st.next_level:
            raise RuntimeError(f'PathConv: levels must arrive in increasing order (got {level_id}, expected {st.next_level})')
        st.next_level = level_id + 1
        mismatch = level_id >= len(st.spec_lists) or torch.is_tensor(cur_nodes) or not _same_list(cur_nodes, st.spec_lists[level_id])
        if mismatch and not st.need_grad:
            # inference with other lists than last time (e.g. validate() after a truncated sweep): the levels below this
            # one were computed from identical lists, so the sweep simply continues level by level from here
            graph.__dict__['_spec_lists'] = None
            nst = SweepState(graph, conv)
            nst.next_level = level_id
            graph._sweep = st = nst
            spec_active = False
        elif mismatch:
            # training: the levels below were computed from identical lists, but the autograd node of the speculative sweep
            # covers the recorded lists of ALL levels - this step cannot be completed.  Speculation is switched off for the
            # graph, so re-running the step (and every later one) takes the strict per-level path.
            graph.__dict__['_spec_lists'] = None
            graph.__dict__['_spec_disabled'] = True
            graph._sweep = None
            raise RuntimeError(f'PathConv: level {level_id} arrived with a node list that differs from the one this graph was '
                               f'swept with before - the speculative whole-sweep of the level-0 call used the recorded lists. '
                               f'Speculation is now disabled for this graph: zero graph.ndata["h"] and run the step again '
                               f'(mmft.sweep.SPECULATE = False avoids the first failure for loops whose level lists change).')
    if spec_active:
        tix = graph.level_rows(level_id, targets, 'targets')
        if not tix.numel():
            e = st.__dict__.get('_empty_rows')
            if e is None:
                e = st.__dict__['_empty_rows'] = st.h.new_zeros((0, st.D))
            return e
        if st.need_grad:
            st.spec_tix.append(tix)
            if graph.__dict__.get('_head_takes_gradients'):
                # the caller (PathModel's deferred head) gathers h[targets] inside its one-call level head and scatters the
                # endpoint gradients itself from its root node: no launch, no autograd node here
                return st.h.new_empty((tix.numel(), 0))
            return TargetGatherFn.apply(st.spec_token, st, tix, graph.__dict__.get('targets_unique'))
        return ops.gather_rows(st.h, tix)
    if level_id == 0 or graph._sweep is None:
        if level_id != 0:
            raise RuntimeError('PathConv: a sweep must start at level 0 (src/train.py:489-490)')
        graph._sweep = SweepState(graph, conv)
    st = graph._sweep
    if graph.ndata['h'] is not st.h:
        raise RuntimeError("graph.ndata['h'] was replaced in the middle of a sweep; restart from level 0")
    if level_id != st.next_level:
        raise RuntimeError(f'PathConv: levels must arrive in increasing order (got {level_id}, expected {st.next_level})')
    st.next_level = level_id + 1
    rows = graph.level_rows(level_id, cur_nodes, 'nodes')
    tix = graph.level_rows(level_id, targets, 'targets')
    c12 = None
    if getattr(conv, 'flag_attn', False) and level_id % 2 == 0:
        if level_id > 0:
            c12 = attention_coefficients(conv)                                           # src/model.py:132-136
        # second pull of the branch: ndata['h_drive'] = mean of net_feat over the NET in-edges (src/model.py:197-198)
        hd = graph.ndata.get('h_drive')
        if hd is None or hd.shape != st.net_feat.shape or hd.device != st.net_feat.device:
            hd = torch.zeros_like(st.net_feat)
            graph.ndata['h_drive'] = hd
        if rows.numel():
            ops.seg_mean_rows_any(st.net_feat, graph.csr('in', 'net'), rows, hd)
    if level_id == 0:
        token = st.h.new_zeros(1)
        if st.need_grad:
            st.token, out = LevelFn.apply(token, st, level_id, rows, tix, c12, *st.params)
        else:
            with torch.no_grad():
                st.token, out = LevelFn.apply(token, st, level_id, rows, tix, c12)
    else:
        if st.need_grad:
            st.token, out = LevelFn.apply(st.token, st, level_id, rows, tix, c12)
        else:
            with torch.no_grad():
                st.token, out = LevelFn.apply(st.token, st, level_id, rows, tix, c12)
    return out


# ------------------------------------------------------------------------------------------------
# Whole-sweep entry: one autograd node for all L levels.
#   * the *_self MLPs do not depend on h, so they run ONCE over all nodes of a kind (three batched GEMM
#     pairs) and pre-fill h; the level-serial chain is then one gather kernel per net level and
#     gather + two GEMMs per cell level;
#   * targets are gathered once after the last level (rows are final once written);
#   * backward = scatter of all target gradients, the reverse pull sweep, batched weight gradients.
# Same arithmetic as the per-level path (same kernels, same per-row operation order).
# ------------------------------------------------------------------------------------------------
EDGE_DRIVERS = True                 # folded gather: per-edge driver table (3-deep load chain, four edges in flight)
FEAT_MLP_NO_HIDDEN = True           # bf16 mode: fc_cell_self / fc_net_self as one kernel each way, hidden activations recomputed
FUSE_LEVEL_FWD = True               # bf16 mode: folded gather + fused MLP of a level pair in one launch (mmft_level_fwd_bf16)
LEVEL_SLOTS = True                  # ... in its slot-table form where the level allows it (mmft_level_fwd_slots: fan-in <= 4, ranges)
RECORD_LAUNCHES = True              # eager sweeps: the per-level kernels' validated argument lists are kept and re-issued (ops.relaunch)
SPEC_SIDE_STREAM = True             # drop-in loop: the speculative whole sweep (and with it its backward) on a stream of its own
HIDDEN_BF16 = True                  # bf16 mode: fc_cell_neigh's hidden activations / hidden gradients stored as bf16
LEVEL_BWD_PAIRS = True              # reverse sweep: one launch per (cell level, net level above it) pair where the numbering
                                    # allows it (PinGraph.level_bwd_pairs; mmft_level_bwd_pair)
FOLD_LEVELS = True                  # folded forward chain (one gather per (net, cell) level PAIR) when the graph allows it
FUSED_FIRST_LAYER_GRADS = True      # mmft_mlp2_first_layer_grads for the *_self MLPs (False: dgrad GEMM + wgrad GEMM)


SWEEP_REPLAY = True      # drop-in loop: the speculative sweep's forward / reverse launches replayed from captured HIP graphs
# (tools/ab_dropin.py, flags toggled inside one process, config B, ms per drop-in step: plain 7.31, recorded launches
#  6.69, + own stream 6.59, + replay 5.94)

_REPLAY_STATE = ('levels', 'wpack', 'hid16', 'HN', 'DHN', 'prep', 'row_sets', 'feat_fused', 'PRE', 'attn')


class _SweepReplay:
    """Captured HIP graphs of ONE speculative whole sweep (the drop-in loop's level-0 call, launched eagerly ~45 + ~70 times
    per step otherwise): valid while the graph's buffers, the static tables of these level lists, the parameters, their
    gradient sinks and h keep their addresses (`sig`).  The launches are static: features, lists and buffers do not change
    from step to step; what does change - the sampled endpoints - is handled by the per-level nodes outside."""

    def __init__(self, sig, lists):
        self.sig, self.lists = sig, lists
        self.calls = self.bwd_calls = 0
        self.fwd = self.bwd = None
        self.state = None


def _sweep_replay_for(st, tix, c12):
    """The replay record of this sweep, or None: speculative mode only (no target gather inside the node), gradient sinks on
    every parameter, recorded-launch preconditions (static lists / buffers), no profiler, no outer capture."""
    if not SWEEP_REPLAY or tix is not None or c12 is not None or lib.PROF_ON or torch.cuda.is_current_stream_capturing():
        return None
    if getattr(st, 'level_lists', None) is None or st.active is not None or not st.need_grad:
        return None
    recs = [gradsink.of(p) for p in st.params]
    if any(r is None for r in recs) or lib.get_math_mode() != 'bf16':
        return None
    sig = (tuple(id(n) for n in st.level_lists), st.relu, st.h.data_ptr(), tuple(p.data_ptr() for p in st.params),
           tuple(r[0].data_ptr() for r in recs), st.cell_feat.data_ptr(), st.net_feat.data_ptr())
    rp = st._bufs.get('replay')
    if rp is None or rp.sig != sig:
        rp = st._bufs['replay'] = _SweepReplay(sig, list(st.level_lists))
    return rp


class SweepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state, level_rows, tix, c12, anchor, *params):
        st, g = state, state.graph
        rp = _sweep_replay_for(st, tix, c12)
        ctx.replay = rp
        if rp is not None:
            rp.calls += 1
            if rp.fwd is not None:
                st.__dict__.update(rp.state)
                rp.fwd.replay()
                ctx.state, ctx.tix, ctx.nparams, ctx.has_c12 = st, tix, len(params), False
                return st.h.new_zeros(1)
            if rp.calls >= 2:                                   # second sweep with the same addresses: capture, then replay
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                    SweepFn._forward_launches(ctx, st, g, level_rows, tix, c12, params)
                rp.fwd, rp.state = graph, {k: st.__dict__.get(k) for k in _REPLAY_STATE}
                graph.replay()
                return st.h.new_zeros(1)
        SweepFn._forward_launches(ctx, st, g, level_rows, tix, c12, params)
        if tix is None:
            # speculative drop-in sweep: the per-level target gathers (TargetGatherFn) hang off this token
            return st.h.new_zeros(1)
        return ops.gather_rows(st.h, tix) if tix.numel() else st.h.new_zeros((0, st.D))

    @staticmethod
    def _forward_launches(ctx, st, g, level_rows, tix, c12, params):
        """Every launch of the forward sweep (and the python-side state of `st` the backward needs)."""
        st.attn = _attn_state(st, g, c12) if c12 is not None else None
        P = [_w(p) for p in st.params]
        (w1c, b1c, w2c, b2c, w1n, b1n, w2n, b2n, w1g, b1g, w2g, b2g) = P
        act = ops.ACT_RELU if st.relu else ops.ACT_NONE
        st.levels = [(l, r) for l, r in enumerate(level_rows)]
        st.wpack = None
        if lib.get_math_mode() == 'bf16' and ops.mlp2_fusable(st.D, st.Hd, st.D):
            # the level chain's fused MLP takes its weights pre-packed as bf16 (forward: W1, W2; reverse: W2^T, W1^T)
            # (static buffers per graph: their addresses are part of the recorded launches below)
            wp = st._bufs.get('wpack')
            if wp is None or wp[0].device != st.h.device:
                mk = lambda *shape: torch.empty(shape, dtype=torch.bfloat16, device=st.h.device)
                wp = st._bufs['wpack'] = (mk(st.Hd, st.D), mk(st.D, st.Hd), mk(st.Hd, st.D), mk(st.D, st.Hd))
            st.wpack = (ops.pack_bf16(w1g, out=wp[0]), ops.pack_bf16(w2g, out=wp[1]), ops.pack_bf16(w2g, transpose=True, out=wp[2]),
                        ops.pack_bf16(w1g, transpose=True, out=wp[3]))
        r0 = level_rows[0]
        rc2 = _cat_rows(st, lambda l: l % 2 == 0 and l > 0)
        # bf16 mode on range-numbered graphs: fc_cell_neigh's hidden activations and their gradients are STORED as bf16 - every
        # consumer rounds them to bf16 anyway (MFMA operands of the weight gradients) or reads the sign only (ReLU mask); only
        # the first layer's bias gradient now sums the rounded hidden gradients instead of the fp32 ones
        st.hid16 = bool(HIDDEN_BF16 and st.wpack is not None and isinstance(rc2, tuple) and st.attn is None)
        st.HN = st._buf('HN16', st.Hd, torch.bfloat16) if st.hid16 else st._buf('HN', st.Hd)
        st.DHN = None
        # Recorded launches of the two per-level kernels (ops.relaunch): valid while the sweep's buffers (persistent per graph),
        # the static tables of these level lists, the parameters and h are the same objects at the same addresses
        # (the per-level list OBJECTS identify the schedule: the outer list is rebuilt by the drop-in loop's bookkeeping)
        sig = (tuple(id(n) for n in st.level_lists) if getattr(st, 'level_lists', None) is not None else None, st.hid16, st.relu,
               st.h.data_ptr(), st.active is None, tuple(p.data_ptr() for p in P))
        prep = st._bufs.get('prep')
        if not RECORD_LAUNCHES or getattr(st, 'level_lists', None) is None or st.active is not None:
            prep = None
        elif prep is None or prep['sig'] != sig:
            prep = st._bufs['prep'] = dict(sig=sig, lists=list(st.level_lists), calls={})       # (the lists are pinned: ids stay theirs)
        st.prep = prep
        dev_, stream_ = lib.stream_args(st.h)
        rn = _cat_rows(st, lambda l: l % 2 == 1)
        st.row_sets = (_cat_rows(st, lambda l: l % 2 == 0), rn, rc2)
        r0s = _cat_rows(st, lambda l: l == 0) if r0.numel() else None
        # bf16 mode: the feature MLPs run as one kernel each and their hidden activations (268 MB per MLP at config B)
        # are never stored - the backward recomputes them (mmft_mlp2_feat_*); needs contiguous row ranges
        st.feat_fused = (FEAT_MLP_NO_HIDDEN and st.wpack is not None and all(isinstance(r, tuple) for r in (r0s, rc2, rn) if r is not None)
                         and isinstance(st.row_sets[0], tuple) and ops.mlp2_feat_fusable(st.cell_feat.shape[1], st.Hd, st.D)
                         and ops.mlp2_feat_fusable(st.net_feat.shape[1], st.Hd, st.D)
                         and all(p.is_contiguous() for p in P[:8]))
        if r0s is not None:                                                              # level 0, :148-153
            if st.feat_fused:
                ops.mlp2_feat_fwd_bf16(st.cell_feat, r0s, w1c, b1c, w2c, b2c, st.h, relu_out=st.relu)
            else:
                _linear_rows(st.cell_feat, w1c, b1c, st.HS, r0s, act=ops.ACT_RELU)
                _linear_rows(st.HS, w2c, b2c, st.h, r0s, act=act)
        if rc2 is not None:                                                              # fc_cell_self, all cell nodes
            if st.feat_fused:
                ops.mlp2_feat_fwd_bf16(st.cell_feat, rc2, w1c, b1c, w2c, b2c, st.h)
            else:
                _linear_rows(st.cell_feat, w1c, b1c, st.HS, rc2, act=ops.ACT_RELU)
                _linear_rows(st.HS, w2c, b2c, st.h, rc2)
        fold = st.fold if (FOLD_LEVELS and st.attn is None) else None
        if fold is not None:
            st.PRE = st._buf('PRE', st.D)     # fc_net_self outputs live apart from h: the folded gather updates h in place
        if rn is not None:                                                               # fc_net_self, all net nodes
            if st.feat_fused:
                ops.mlp2_feat_fwd_bf16(st.net_feat, rn, w1n, b1n, w2n, b2n, st.PRE if fold is not None else st.h)
            else:
                _linear_rows(st.net_feat, w1n, b1n, st.HS, rn, act=ops.ACT_RELU)
                _linear_rows(st.HS, w2n, b2n, st.PRE if fold is not None else st.h, rn)
        in_net, in_cell = g.csr('in', 'net'), g.csr('in', 'cell')
        if fold is not None:
            # folded chain: one gather launch per (net level l - 1, cell level l) pair + the fused MLP of the cell level
            L = len(level_rows)
            drv = g.cell_edge_drivers() if EDGE_DRIVERS else None
            slot_tabs = g.level_slots(st.level_lists) if (LEVEL_SLOTS and FUSE_LEVEL_FWD and st.wpack is not None
                                                         and getattr(st, 'level_lists', None) is not None) else None
            for level_id in range(2, L + 1, 2):
                net_l = level_id - 1
                has_cell = level_id < L and level_rows[level_id].numel() > 0
                if not has_cell and not fold[net_l]['n']:
                    continue
                meta_n = st.level_meta[net_l] if st.level_meta else None
                meta_c = st.level_meta[level_id] if (st.level_meta and level_id < L) else None
                crow = None
                if has_cell:
                    crow = fold[level_id]['range'] or level_rows[level_id]
                fused = has_cell and st.wpack is not None and fold[level_id]['heavy_in'] is None and FUSE_LEVEL_FWD
                level_bytes = (meta_n['bytes_mean'] if meta_n else 0) + (meta_c['bytes_softmax'] if meta_c else 0) + \
                    (level_rows[level_id].numel() * (8 * st.D + (2 if st.hid16 else 4) * st.Hd) if (meta_c and has_cell) else 0)
                if fused and slot_tabs is not None and fold[level_id]['range'] is not None and slot_tabs[2][level_id] <= 4 and \
                        (fold[net_l]['range'] is not None or not fold[net_l]['n']):
                    # ... with the static slot table instead of the per-edge index chain, net rows inside the cell workgroups
                    rec = prep['calls'].get(('f', level_id)) if prep is not None else None
                    if rec is not None:
                        ops.relaunch('mmft_level_fwd_slots', rec, dev_, stream_)
                        continue
                    rec = ops.level_fwd_slots(st.h, st.PRE, slot_tabs[0], slot_tabs[1], fold[net_l]['range'] or (0, 0), fold[level_id]['range'],
                                              st.A, st.LSE, st.wpack[0], b1g, st.wpack[1], b2g, st.HN, relu=st.relu, active=st.active,
                                              alg_bytes=level_bytes)
                    if prep is not None:
                        prep['calls'][('f', level_id)] = rec
                    continue
                if fused:
                    # bf16 mode: gather + fc_cell_neigh of the pair in ONE launch
                    ops.level_fwd_bf16(st.h, st.PRE, in_net, in_cell, fold[net_l]['range'] or (0, 0), crow, st.A, st.LSE,
                                       st.wpack[0], b1g, st.wpack[1], b2g, st.HN, relu=st.relu, active=st.active, in_cell_driver=drv,
                                       # gather bytes of the pair + what the MLP part must move per cell row: h read and
                                       # written (2 x 4 D) and the hidden row kept for the reverse sweep (4 Hd)
                                       alg_bytes=(meta_n['bytes_mean'] if meta_n else 0) + (meta_c['bytes_softmax'] if meta_c else 0)
                                       + (level_rows[level_id].numel() * (8 * st.D + (2 if st.hid16 else 4) * st.Hd) if meta_c else 0))
                    continue
                ops.pair_fwd_gather(st.h, st.PRE, in_net, in_cell, fold[net_l]['range'] or (0, 0), crow, st.A, st.LSE,
                                    relu=st.relu, heavy=fold[level_id]['heavy_in'] if has_cell else None, active=st.active,
                                    in_cell_driver=drv,
                                    alg_bytes=(meta_n['bytes_mean'] if meta_n else 0) + (meta_c['bytes_softmax'] if (meta_c and has_cell) else 0))
                if has_cell:
                    _cell_neigh_fwd(st, level_rows[level_id], w1g, b1g, w2g, b2g, act)
        for level_id, rows in enumerate(level_rows):
            if fold is not None or level_id == 0 or not rows.numel():
                continue
            meta = st.level_meta[level_id] if st.level_meta else None
            spec = meta['range'] if (meta and meta['range']) else rows       # contiguous levels: no index array
            if level_id % 2 == 1:
                ops.seg_mean_add_act_fwd(st.h, in_net, spec, relu=st.relu, alg_bytes=meta['bytes_mean'] if meta else 0)
            elif st.attn is not None:
                _cell_gather_fwd(st, g, spec)
                _cell_neigh_fwd(st, rows, w1g, b1g, w2g, b2g, act)
            else:
                ops.seg_softmax_sum_fwd(st.h, in_cell, spec, st.A, st.LSE, alg_bytes=meta['bytes_softmax'] if meta else 0)
                _cell_neigh_fwd(st, rows, w1g, b1g, w2g, b2g, act)
        ctx.state, ctx.tix, ctx.nparams = st, tix, len(params)
        ctx.has_c12 = c12 is not None

    @staticmethod
    def backward(ctx, gout):
        st, g = ctx.state, ctx.state.graph
        if ctx.tix is not None:
            st.bwd_active = False
        # The fills of G / DA (2 x 268 MB per step at config B) may only be skipped when the level lists are a complete
        # schedule of the graph: then every row a pull reads was rewritten earlier in this reverse sweep.  With a
        # partial schedule (fan-in cone, truncated lists) consumers outside it keep rows from an earlier step - or
        # uninitialised memory - so both buffers are zero-filled and the own-gradient flags are not used.
        # fan-in-cone pruning: rows outside the cone are skipped, so G / DA are zero-filled (their rows must read as zero)
        fast = st.complete and st.active is None
        st.begin_backward(zero_da=not fast, zero_g=not fast)
        own = st.tflag if fast else None
        tix = ctx.tix
        if tix is None:
            # the TargetGatherFn nodes of this sweep have already zeroed / flagged / filled their rows of G
            tix = torch.cat(st.spec_tix) if st.spec_tix else st.h.new_zeros(0, dtype=torch.int32)
        elif tix.numel():
            if fast:
                ops.target_rows_begin(st.G, tix, st.tflag)
            ops.scatter_add_targets(st.G, tix, ops.strided_rows(gout),
                                    order=st.target_order, unique=st.targets_unique)
        rp = getattr(ctx, 'replay', None)
        recs = [gradsink.of(p) for p in st.params]
        replayable = (rp is not None and fast and ctx.tix is None and not ctx.has_c12 and ctx.nparams == len(st.params)
                      and not torch.cuda.is_current_stream_capturing() and not lib.PROF_ON
                      and all(r is not None and r[1] != gradsink._epoch[0] for r in recs))      # every sink still fresh this step
        if replayable:
            rp.bwd_calls += 1
            if rp.bwd is not None:
                st.DHN = st._bufs.get('DHN16' if getattr(st, 'hid16', False) else 'DHN')
                rp.bwd.replay()
                for r in recs:
                    gradsink._mark(r)
                grads = [None] * ctx.nparams
            else:
                if rp.bwd_calls >= 2:
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                        grads = SweepFn._backward_launches(ctx, st, g, fast, own)
                    rp.bwd = graph
                    graph.replay()
                else:
                    grads = SweepFn._backward_launches(ctx, st, g, fast, own)
        else:
            grads = SweepFn._backward_launches(ctx, st, g, fast, own)
        dc = None
        if ctx.has_c12:
            rc2 = st.row_sets[2]
            dc = _attn_c12_grad(st, rc2) if rc2 is not None else torch.zeros((2, 1), dtype=torch.float32, device=st.h.device)
        if tix.numel() and fast:
            ops.target_rows_end(tix, st.tflag)
        st.bwd_active = False
        # `anchor` (see _run_sweep): a defined gradient, so that its AccumulateGrad node - a function without outputs, living on
        # this node's stream - makes the engine join that stream with the caller's at the end of backward()
        ga = st.h.new_empty(1) if ctx.needs_input_grad[4] else None      # its value is never read: no fill launch
        return (None, None, None, dc, ga, *grads)

    @staticmethod
    def _backward_launches(ctx, st, g, fast, own):
        """The reverse sweep's launches between the endpoint scatter and the flag reset: level chain + batched weight gradients."""
        P = [_w(p) for p in st.params]
        w1g, w2g = P[8], P[10]
        out_net, out_cell, in_net_ptr = g.csr('out', 'net'), g.csr('out', 'cell'), g.out_net_weight()
        pairs = None
        if LEVEL_BWD_PAIRS and fast and st.fold is not None and st.wpack is not None and st.attn is None \
                and getattr(st, 'level_lists', None) is not None:
            pairs = g.level_bwd_pairs(st.level_lists)
        paired = set()
        prep = getattr(st, 'prep', None) if fast else None
        dev_b, stream_b = lib.stream_args(st.h)
        if pairs is not None:
            cslots, plist, pscratch, pcounters = pairs
            if st.DHN is None:
                st.DHN = st._buf('DHN16', st.Hd, torch.bfloat16) if getattr(st, 'hid16', False) else st._buf('DHN', st.Hd)
        for level_id, rows in reversed(st.levels):
            if level_id in paired:
                continue
            if pairs is not None:
                # one launch for the pair (cell level l, net level l + 1): the net level comes first in this reverse order
                cell_l = level_id - (level_id % 2)
                pr = plist[cell_l // 2]
                if pr is not None:
                    paired.add(cell_l)
                    mc = st.level_meta[cell_l] if st.level_meta else None
                    mn = st.level_meta[cell_l + 1] if (st.level_meta and cell_l + 1 < len(st.level_meta)) else None
                    nb = (mc['bytes_pull'] if mc else 0) + (mn['bytes_pull'] if mn else 0) + \
                        (pr['n_cell'] * (8 * st.D + (4 if st.hid16 else 8) * st.Hd) if cell_l > 0 else 0)
                    rec = prep['calls'].get(('b', cell_l)) if prep is not None else None
                    if rec is not None:
                        ops.relaunch('mmft_level_bwd_pair', rec, dev_b, stream_b)
                        continue
                    rec = ops.level_bwd_pair(st.G, st.h, st.A, st.LSE, st.DA, own, pr['tiles'], pr['ntiles'], out_net[0], pr['sink_shift'],
                                             cslots, out_cell, pscratch, pcounters, st.wpack[2], st.wpack[3], st.HN, st.DHN, relu=st.relu,
                                             has_mlp=cell_l > 0, alg_bytes=nb)
                    if prep is not None:
                        prep['calls'][('b', cell_l)] = rec
                    continue
            if not rows.numel():
                continue
            meta = st.level_meta[level_id] if st.level_meta else None
            spec = meta['range'] if (meta and meta['range']) else rows
            _pull_bwd(st, g, spec, own=own, alg_bytes=meta['bytes_pull'] if meta else 0,
                      heavy=meta['heavy_out'] if meta else None)
            if level_id % 2 == 0 and level_id > 0:
                # (cone pruning: the hidden gradients of skipped rows would be stale - recompute them from G = 0 instead)
                _cell_neigh_bwd(st, rows, w1g, w2g, keep_dhn=st.active is None)
                if ctx.has_c12:
                    _attn_scores_bwd(st, g, spec)
        return _batched_param_grads(st, P, dhn_ready=st.active is None) if ctx.nparams else []


class TargetGatherFn(torch.autograd.Function):
    """h[targets] of one level of a SPECULATIVE drop-in sweep (the whole sweep ran at the level-0 call).  Backward: zero,
    flag and fill the endpoint rows of G - the reverse sweep itself is SweepFn.backward, which the engine runs after
    every node that consumes the sweep's token."""

    @staticmethod
    def forward(ctx, token, state, tix, unique):
        ctx.state, ctx.tix, ctx.unique = state, tix, unique
        return ops.gather_rows(state.h, tix)

    @staticmethod
    def backward(ctx, gout):
        st = ctx.state
        fast = st.complete and st.active is None
        st.begin_backward(zero_da=not fast, zero_g=not fast)
        if fast:
            ops.target_rows_begin(st.G, ctx.tix, st.tflag)
        ops.scatter_add_targets(st.G, ctx.tix, ops.strided_rows(gout), unique=ctx.unique)
        return st.h.new_zeros(1), None, None, None


def _run_sweep(conv, graph, level_nodes, tix, target_order=None, targets_unique=None, active=None):
    """Build the sweep state for the given level lists and run SweepFn (tix None: deferred per-level target gathers)."""
    st = SweepState(graph, conv)
    graph._sweep = st
    st.active = active
    st.complete = graph.level_set_is_complete(level_nodes)
    st.fold = graph.fold_schedule(level_nodes) if (FOLD_LEVELS and st.complete and not getattr(conv, 'flag_attn', False)) else None
    st.level_lists = level_nodes if st.fold is not None else None       # key of the static slot tables (PinGraph.level_slots)
    st.target_order, st.targets_unique = target_order, targets_unique
    level_rows = [graph.level_rows(l, nodes, 'nodes') for l, nodes in enumerate(level_nodes)]
    st.level_meta = [graph.level_meta(l, nodes, st.D) if not torch.is_tensor(nodes) else None
                     for l, nodes in enumerate(level_nodes)]
    st.next_level = len(level_rows)
    c12 = None
    if getattr(conv, 'flag_attn', False):
        c12 = attention_coefficients(conv)
        hd = torch.zeros_like(st.net_feat)                     # ndata['h_drive'] of the even levels (src/model.py:197-198)
        for l in range(0, len(level_rows), 2):
            if level_rows[l].numel():
                ops.seg_mean_rows_any(st.net_feat, graph.csr('in', 'net'), level_rows[l], hd)
        graph.ndata['h_drive'] = hd
    if st.need_grad or (c12 is not None and c12.requires_grad):
        # The sweep may run on a stream of its own and, with gradient sinks, returns no gradient to any leaf: nothing would
        # tell the engine to join that stream when backward() ends (it joins the streams of functions WITHOUT outputs, i.e.
        # of AccumulateGrad nodes that ran).  A one-element leaf created here, on the sweep's stream, is that node.
        anchor = st.h.new_empty(1, requires_grad=True)                   # (neither its value nor its gradient is ever read)
        return st, SweepFn.apply(st, level_rows, tix, c12, anchor, *st.params)
    with torch.no_grad():
        return st, SweepFn.apply(st, level_rows, tix, c12, None)


def sweep_forward_all(conv, graph, level_nodes, targets, target_order=None, targets_unique=None, cone=False):
    """All levels of one sweep in one call. level_nodes: list (per level) of python int lists or device int32
    tensors; targets: device int32 tensor (or list) of node ids whose embeddings are returned, in order.
    target_order: optional device int32 stable argsort of `targets` (the host has it for free when it packs a
    step's endpoints); targets_unique: True when the caller knows that no endpoint is repeated.
    cone=True: fan-in-cone pruning (the reference carries the idea unused, src/MyDataloader.py:4-59) -
    the level kernels skip every node that cannot influence `targets`.  The cone is a per-node flag array built ON THE
    DEVICE by L - 1 fixed-size launches (mmft_fanin_cone_step), the launches of the sweep keep their full grids and
    skip flagged-off rows, so a captured HIP graph of the step stays valid while the sampled endpoints - and with them
    the cone - change from replay to replay."""
    tix = graph.level_rows(-1, targets, 'sweep_targets')
    active = None
    if cone:
        from . import prep
        specs = []
        for l, nodes in enumerate(level_nodes):
            meta = graph.level_meta(l, nodes) if not torch.is_tensor(nodes) else None
            specs.append(meta['range'] if (meta and meta['range']) else graph.level_rows(l, nodes, 'nodes'))
        active = prep.cone_mask([graph.csr('in', 'net'), graph.csr('in', 'cell')], specs, tix,
                                out=graph.__dict__.get('_cone_mask'))
        graph.__dict__['_cone_mask'] = active
    return _run_sweep(conv, graph, level_nodes, tix, target_order, targets_unique, active)[1]


# Speculative drop-in sweep.  The reference loop calls the model once per level with the SAME node lists every step
# (topo_levels of the design, src/train.py:490-503) and nothing a later call passes can change what an earlier level
# computes, so once a graph has been swept level by level with some lists, the next level-0 call runs the WHOLE sweep with
# them (the whole-sweep kernels: batched *_self MLPs, folded gathers, one autograd node) and every level call then only
# checks that its list is the recorded one and gathers h[targets].  A call whose list differs raises; set
# mmft.sweep.SPECULATE = False for loops that change their level lists between steps.
SPECULATE = True


def _same_list(a, b):
    return a is b or (len(a) == len(b) and (not len(a) or (a[0] == b[0] and a[-1] == b[-1] and a == b)))
