"""CPU restatement of the reference hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product path (the package next to csrc/) never does and fails loudly without its HIP
library.

Plain PyTorch fp32/fp64 on the CPU, functional style: every function takes a dict of tensors
keyed by the *reference's state_dict names* (SURVEY.md App. A.6), so the same weights load into
the reference modules (fixture generation), into this restatement and into the HIP modules.

What each function follows (paths relative to /root/reference):
  mlp                 src/model.py:10-24      Linear -> LeakyReLU(slope) -> ... -> Linear
  seg_softmax_sum     src/model.py:113-116    per-channel softmax-weighted sum over the mailbox
  seg_mean            src/model.py:186-187    fn.copy_src + fn.mean  (0 for zero in-degree)
  seg_attn_sum        src/model.py:119-136    attention branch: edge score + softmax over in-edges + weighted sum
  pathconv_level      src/model.py:158-213    one PathConv.forward call (net / cell / level 0 / attention)
  pathmodel_level     src/model.py:269-292    one PathModel.forward call
  unet_forward        src/Unet.py:8-119       DoubleConv/Down/Up/OutConv/UNet, BN in train mode
  layoutnet_forward   src/model.py:216-247
  train_step          src/train.py:475-562    one mini-batch step incl. MSE, backward, Adam

Pinning status:
  * unet_forward / layoutnet_forward / mlp / pathmodel fusion / the PathConv UDFs are pinned
    against the reference's own code (tests/golden/make_golden.py runs src/Unet.py as-is and
    src/model.py with only the *names* dgl.function.{copy_src,mean,max} provided) - fixtures in
    tests/golden/*.npz.
  * the DGL `pull` contract itself (in-edge enumeration, degree-bucketed mailbox, zero fill for
    zero in-degree, row write-back) is third-party arithmetic absent from /root/reference (DGL,
    version unpinned, no lock file) and the reference holds no tests or golden vectors for it:
    that part is restated from DGL's documented semantics  ->  **parity unpinned** for `pull`.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- MLP (src/model.py:10-24)
def mlp(p, prefix, x, slope=0.0):
    """Sequential of Linear layers at indices 0,2,4,...; LeakyReLU(slope) between them."""
    idx = sorted({int(k[len(prefix) + len('layers.'):].split('.')[0]) for k in p
                  if k.startswith(prefix + 'layers.') and k.endswith('.weight')})
    for j, i in enumerate(idx):
        x = F.linear(x, p[f'{prefix}layers.{i}.weight'], p[f'{prefix}layers.{i}.bias'])
        if j < len(idx) - 1:
            x = F.leaky_relu(x, negative_slope=slope)
    return x


# --------------------------------------------------------------------------- DGL pull, restated
def _in_edges(indptr, indices, nodes):
    """For each node in `nodes`: (start, degree) into the in-edge CSR (row = dst, col = src)."""
    nodes = np.asarray(nodes, dtype=np.int64)
    return indptr[nodes], indptr[nodes + 1] - indptr[nodes]


def seg_mean(h, indptr, indices, nodes):
    """fn.copy_src('h','m') + fn.mean('m', .) over in-edges of `nodes`; zero rows for degree 0."""
    start, deg = _in_edges(indptr, indices, nodes)
    out = h.new_zeros((len(nodes), h.shape[1]))
    for d in np.unique(deg):
        if d == 0:
            continue
        sel = np.nonzero(deg == d)[0]
        eid = start[sel][:, None] + np.arange(d)[None, :]
        mail = h[torch.from_numpy(indices[eid])]                   # (n_bucket, d, D) degree bucket
        out = out.index_copy(0, torch.from_numpy(sel), mail.mean(1))
    return out


def seg_softmax_sum(h, indptr, indices, nodes):
    """PathConv.cell_msg_reduce over DGL's degree buckets: w = softmax(msg, dim=1); (msg*w).sum(1)."""
    start, deg = _in_edges(indptr, indices, nodes)
    out = h.new_zeros((len(nodes), h.shape[1]))
    for d in np.unique(deg):
        if d == 0:
            continue
        sel = np.nonzero(deg == d)[0]
        eid = start[sel][:, None] + np.arange(d)[None, :]
        mail = h[torch.from_numpy(indices[eid])]
        w = torch.softmax(mail, dim=1)
        out = out.index_copy(0, torch.from_numpy(sel), (mail * w).sum(1))
    return out


def seg_attn_sum(h, key, indptr, indices, nodes, w_key, w_attn):
    """The attention branch's cell pull (src/model.py:119-136,190-196) over DGL's degree buckets:
    message_func_attn: z = cat([fc_key(key_src), fc_key(key_dst)], 1); e = leaky_relu(fc_attn(z))   (slope 0.01)
    cell_msg_reduce_attn: alpha = softmax(e, dim=1) over the in-edges; h_neigh1 = sum(alpha * m, 1), m = h_src.
    w_key = fc_key.weight (dim_key, 1), w_attn = fc_attn.weight (1, 2 * dim_key); zero rows for degree 0."""
    nodes = np.asarray(nodes, dtype=np.int64)
    start, deg = _in_edges(indptr, indices, nodes)
    out = h.new_zeros((len(nodes), h.shape[1]))
    for d in np.unique(deg):
        if d == 0:
            continue
        sel = np.nonzero(deg == d)[0]
        eid = start[sel][:, None] + np.arange(d)[None, :]
        src = torch.from_numpy(indices[eid])                                   # (n_bucket, d)
        dst = torch.from_numpy(np.repeat(nodes[sel][:, None], d, axis=1))
        z = torch.cat([F.linear(key[src], w_key), F.linear(key[dst], w_key)], dim=-1)       # (n, d, 2 * dim_key)
        e = F.leaky_relu(F.linear(z, w_attn))                                  # (n, d, 1)
        alpha = torch.softmax(e, dim=1)
        out = out.index_copy(0, torch.from_numpy(sel), (alpha * h[src]).sum(1))
    return out


# --------------------------------------------------------------------------- PathConv.forward
def pathconv_level(p, prefix, csr, h, cell_feat, net_feat, cur_nodes, targets, level_id, activation=True, key=None):
    """One call of PathConv.forward (src/model.py:158-213). Returns (h_new, h_new[targets]).

    csr = {'net': (indptr, indices), 'cell': (indptr, indices)} numpy int64, in-edges by dst.
    `h` is treated functionally (DGL's frame update is out of place too).
    key (N, 1): ndata['key'] -> the flag_attn=True branch (src/model.py:190-198); its second pull only fills
    ndata['h_drive'] (mean of net_feat over net in-edges, src/model.py:197-198,66-86), which nothing reads - see
    pathconv_h_drive.
    """
    idx = torch.as_tensor(np.asarray(cur_nodes, dtype=np.int64))
    if len(cur_nodes):
        if level_id % 2 == 1:
            a = seg_mean(h, *csr['net'], cur_nodes)                                     # :186-187
            rows = mlp(p, prefix + 'fc_net_self.', net_feat[idx]) + a                   # :103-109
        elif level_id == 0:
            rows = mlp(p, prefix + 'fc_cell_self.', cell_feat[idx])                     # :148-153
        elif key is not None:
            a = seg_attn_sum(h, key, *csr['cell'], cur_nodes, p[prefix + 'fc_key.weight'],
                             p[prefix + 'fc_attn.weight'])                              # :119-136,190-196
            rows = mlp(p, prefix + 'fc_cell_self.', cell_feat[idx]) + \
                mlp(p, prefix + 'fc_cell_neigh.', a)
        else:
            a = seg_softmax_sum(h, *csr['cell'], cur_nodes)                             # :113-116
            rows = mlp(p, prefix + 'fc_cell_self.', cell_feat[idx]) + \
                mlp(p, prefix + 'fc_cell_neigh.', a)                                    # :138-146
        if activation:
            rows = torch.relu(rows)                                                     # :206-208
        h = h.index_copy(0, idx, rows)
    tix = torch.as_tensor(np.asarray(targets, dtype=np.int64))
    return h, h[tix]                                                                    # :213


def pathconv_h_drive(csr, net_feat, cur_nodes):
    """ndata['h_drive'] rows of an even level in the attention branch: fn.copy_src('net_feat') + fn.mean over the NET
    in-edges, passed through apply_netdrive_func unchanged (src/model.py:197-198,66-86)."""
    return seg_mean(net_feat, *csr['net'], cur_nodes)


# --------------------------------------------------------------------------- PathModel.forward
def pathmodel_level(p, csr, h, cell_feat, net_feat, nodes, targets, level_id, level_id_th, path_map,
                    has_gnn=True, has_fcn=True):
    """One call of PathModel.forward (src/model.py:269-292). Returns (h_new, prediction | None)."""
    T = len(targets)
    h_cnn = F.linear(path_map, p['fcn.weight'], p['fcn.bias']) if (has_fcn and T != 0) else None
    h_gnn = None
    if has_gnn:
        h, h_gnn = pathconv_level(p, 'gnn.', csr, h, cell_feat, net_feat, nodes, targets, level_id)
    h_global = mlp(p, 'mlp_alpha.', level_id_th).expand(T, 32)
    if T == 0:
        return h, None
    if h_cnn is None:
        z = torch.cat([h_gnn, h_global], dim=1)
    elif h_gnn is None:
        z = torch.cat([h_cnn, h_global], dim=1)
    else:
        z = torch.cat((h_gnn, h_cnn, h_global), 1)
    return h, mlp(p, 'mlp_fuse.', z).squeeze(-1)


# --------------------------------------------------------------------------- U-Net (src/Unet.py)
def _bn_train(p, prefix, x, update_running=True, momentum=0.1, eps=1e-5):
    rm, rv = p.get(prefix + 'running_mean'), p.get(prefix + 'running_var')
    if not update_running or rm is None:
        rm = rv = None
    y = F.batch_norm(x, rm, rv, p[prefix + 'weight'], p[prefix + 'bias'], True, momentum, eps)
    if rm is not None and (prefix + 'num_batches_tracked') in p:
        p[prefix + 'num_batches_tracked'] += 1
    return y


def _double_conv(p, prefix, x, update_running):
    x = F.conv2d(x, p[prefix + 'double_conv.0.weight'], None, padding=1)
    x = torch.relu(_bn_train(p, prefix + 'double_conv.1.', x, update_running))
    x = F.conv2d(x, p[prefix + 'double_conv.3.weight'], None, padding=1)
    x = torch.relu(_bn_train(p, prefix + 'double_conv.4.', x, update_running))
    return x


def _pool(x, pooling):
    return F.max_pool2d(x, 2) if pooling == 'max' else F.avg_pool2d(x, 2)


def _up(p, prefix, x1, x2, update_running, bilinear=False):
    """Up.forward (src/Unet.py:56-68); bilinear=True: nn.Upsample(scale_factor=2, 'bilinear', align_corners=True) (:50)."""
    if bilinear:
        x1 = F.interpolate(x1, scale_factor=2, mode='bilinear', align_corners=True)
    else:
        x1 = F.conv_transpose2d(x1, p[prefix + 'up.weight'], p[prefix + 'up.bias'], stride=2)
    dy, dx = x2.shape[2] - x1.shape[2], x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return _double_conv(p, prefix + 'conv.', torch.cat([x2, x1], dim=1), update_running)


def up_block(p, x1, x2, bilinear, update_running=True):
    """One Up module on its own (state_dict keys 'up.*' / 'conv.double_conv.*')."""
    return _up(p, '', x1, x2, update_running, bilinear)


def unet_forward(p, x, pooling='max', update_running=True):
    """UNet.forward (src/Unet.py:110-119), BatchNorm in train mode. Accepts (C,H,W) too (D3)."""
    if x.dim() == 3:
        x = x.unsqueeze(0)
    x1 = _double_conv(p, 'inc.', x, update_running)
    x2 = _double_conv(p, 'down1.maxpool_conv.1.', _pool(x1, pooling), update_running)
    x3 = _double_conv(p, 'down2.maxpool_conv.1.', _pool(x2, pooling), update_running)
    x4 = _double_conv(p, 'down3.maxpool_conv.1.', _pool(x3, pooling), update_running)
    y = _up(p, 'up1.', x4, x3, update_running)
    y = _up(p, 'up2.', y, x2, update_running)
    y = _up(p, 'up3.', y, x1, update_running)
    y = F.conv2d(y, p['outc.conv.0.weight'], p['outc.conv.0.bias'])
    return torch.relu(_pool(y, pooling))


def layoutnet_forward(p, x, pooling='max'):
    """LayoutNet.forward (src/model.py:216-247)."""
    y = torch.relu(F.conv2d(x, p['encode.0.weight'], p['encode.0.bias'], padding=4))
    y = _pool(y, pooling)
    y = torch.relu(F.conv2d(y, p['encode.3.weight'], p['encode.3.bias'], padding=3))
    y = _pool(y, pooling)
    y = torch.relu(F.conv2d(y, p['encode.6.weight'], p['encode.6.bias'], padding=4))
    y = F.conv2d(y, p['encode.8.weight'], p['encode.8.bias'], padding=3)
    return F.leaky_relu(y, 0.1)


# --------------------------------------------------------------------------- the train step
def dense_mask_rows(mask_indptr, mask_cols, path_ids, P, dtype=torch.float32):
    """th.index_select(path_masks, 0, paths).to_dense()  (src/train.py:500)."""
    m = torch.zeros((len(path_ids), P), dtype=dtype)
    for i, pid in enumerate(path_ids):
        m[i, torch.from_numpy(mask_cols[mask_indptr[pid]:mask_indptr[pid + 1]])] = 1
    return m


def bucket_paths(path_ids, path2level, path2endpoint):
    """src/train.py:476-484."""
    ends, paths = {}, {}
    for pid in path_ids:
        lv = int(path2level[pid])
        ends.setdefault(lv, []).append(int(path2endpoint[pid]))
        paths.setdefault(lv, []).append(int(pid))
    return ends, paths


def sweep_forward(pm, pc, design, csr, path_ids, pooling='max', update_running=True, dtype=torch.float32,
                  cnn_kind='unet'):
    """U-Net forward + L-level sweep + fusion head for one endpoint batch (src/train.py:465,475-511).

    pm: PathModel params (gnn.*, fcn.*, mlp_fuse.*, mlp_alpha.*), pc: CNN params.
    Returns (label_hats (T,), target_list, feat_map)."""
    img = torch.from_numpy(design.image).to(dtype)
    if cnn_kind == 'unet':
        feat_map = unet_forward(pc, img, pooling, update_running).reshape(1, -1)
    else:
        feat_map = layoutnet_forward(pc, img, pooling).reshape(1, -1)
    D = pm['gnn.fc_cell_self.layers.2.weight'].shape[0]
    h = torch.zeros((design.N, D), dtype=dtype)
    cell_feat = torch.from_numpy(design.cell_feat).to(dtype)
    net_feat = torch.from_numpy(design.net_feat).to(dtype)
    ends, paths = bucket_paths(path_ids, design.path2level, design.path2endpoint)
    P = design.map_size * design.map_size
    outs, target_list = [], []
    for level_id in range(design.L):
        nodes = design.levels[level_id]
        targets = ends.get(level_id, [])
        pids = paths.get(level_id, [])
        target_list.extend(targets)
        path_map = None
        if len(pids):
            path_map = dense_mask_rows(design.mask_indptr, design.mask_cols, pids, P, dtype) * feat_map
        lvl = torch.tensor([float(level_id)], dtype=dtype)
        h, y = pathmodel_level(pm, csr, h, cell_feat, net_feat, nodes, targets, level_id, lvl, path_map)
        if y is not None:
            outs.append(y)
    return torch.cat(outs, dim=0), target_list, feat_map


def design_csr(design):
    """In-edge CSR (row = dst, col = src), stable in edge insertion order, as numpy int64."""
    out = {}
    for et, (s, d) in (('net', (design.net_src, design.net_dst)), ('cell', (design.cell_src, design.cell_dst))):
        perm = np.argsort(d, kind='stable')
        indptr = np.zeros(design.N + 1, dtype=np.int64)
        np.cumsum(np.bincount(d, minlength=design.N), out=indptr[1:])
        out[et] = (indptr, s[perm].astype(np.int64))
    return out


class OracleTrainer:
    """Holds leaf parameters + torch.optim.Adam and runs reference-shaped steps on the CPU
    (src/train.py:431-443,475-562: MSE on arrival time, Adam lr 1e-3, weight_decay 0)."""

    def __init__(self, pm_state, pc_state, lr=1e-3, weight_decay=0.0, pooling='max', dtype=torch.float32,
                 cnn_kind='unet'):
        self.dtype = dtype
        self.pooling = pooling
        self.cnn_kind = cnn_kind
        self.pm = {k: (v.detach().clone().to(dtype).requires_grad_(True) if v.dtype.is_floating_point else v.clone())
                   for k, v in pm_state.items()}
        self.pc = {}
        for k, v in pc_state.items():
            if v.dtype.is_floating_point and not ('running_' in k):
                self.pc[k] = v.detach().clone().to(dtype).requires_grad_(True)
            elif v.dtype.is_floating_point:
                self.pc[k] = v.detach().clone().to(dtype)
            else:
                self.pc[k] = v.clone()
        leaves = [v for v in self.pm.values() if v.requires_grad] + [v for v in self.pc.values() if v.requires_grad]
        self.optim = torch.optim.Adam(leaves, lr, weight_decay=weight_decay)

    def forward(self, design, csr, path_ids):
        return sweep_forward(self.pm, self.pc, design, csr, path_ids, self.pooling, True, self.dtype, self.cnn_kind)

    def step(self, design, csr, path_ids):
        """One mini-batch: forward, MSE, backward, Adam. Returns (loss, label_hats, target_list)."""
        hats, target_list, _ = self.forward(design, csr, path_ids)
        arrival = torch.from_numpy(design.arrival_time).to(self.dtype)[torch.tensor(target_list)].squeeze(-1)
        loss = F.mse_loss(hats, arrival)
        self.optim.zero_grad()
        loss.backward()
        self.optim.step()
        return float(loss.detach()), hats.detach(), target_list


def r2_score(pred, target):
    """torchmetrics.R2Score restated: 1 - SS_res / SS_tot (src/train.py:31,524)."""
    ss_res = ((target - pred) ** 2).sum()
    ss_tot = ((target - target.mean()) ** 2).sum()
    return 1 - ss_res / ss_tot


def judge_critical(pred_arr_time, required_time):
    """src/train.py:391-395."""
    return ((required_time - pred_arr_time) < 0).to(torch.float32)
