"""CPU: the oracle restatement against the committed golden vectors produced by the REFERENCE's own code
(tests/golden/make_golden.py).  This is what pins the oracle on any machine, without /root/reference."""
import os
import numpy as np
import pytest
import torch

from conftest import rel_err
from mmft.detrand import det_uniform, det_state_dict
from oracle import restatement as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def gold(name):
    return np.load(os.path.join(GOLD, name + '.npz'))


def close(a, ref, tol, what=''):
    e = rel_err(a, torch.as_tensor(np.asarray(ref)))
    assert e < tol, f'{what}: {e:.3e}'


class _Shape:
    """state_dict-shaped stand-in so det_state_dict can fill reference-named tensors without the reference."""
    def __init__(self, shapes):
        self._s = shapes

    def state_dict(self):
        return {k: (torch.zeros(v, dtype=torch.long) if k.endswith('num_batches_tracked') else torch.zeros(v))
                for k, v in self._s.items()}


def unet_shapes():
    s = {}

    def dc(prefix, cin, cout):
        s[prefix + 'double_conv.0.weight'] = (cout, cin, 3, 3)
        for i, c in ((1, cout), (4, cout)):
            for n in ('weight', 'bias', 'running_mean', 'running_var'):
                s[f'{prefix}double_conv.{i}.{n}'] = (c,)
            s[f'{prefix}double_conv.{i}.num_batches_tracked'] = ()
            if i == 1:
                s[prefix + 'double_conv.3.weight'] = (cout, cout, 3, 3)
    dc('inc.', 3, 16)
    dc('down1.maxpool_conv.1.', 16, 32)
    dc('down2.maxpool_conv.1.', 32, 64)
    dc('down3.maxpool_conv.1.', 64, 128)
    for name, cin, cout in (('up1.', 128, 64), ('up2.', 64, 32), ('up3.', 32, 16)):
        s[name + 'up.weight'] = (cin, cin // 2, 2, 2)
        s[name + 'up.bias'] = (cin // 2,)
        dc(name + 'conv.', cin, cout)
    s['outc.conv.0.weight'] = (1, 16, 1, 1)
    s['outc.conv.0.bias'] = (1,)
    return s


@pytest.mark.parametrize('name,pooling,hw', [('unet_max_64x64', 'max', (64, 64)), ('unet_avg_64x64', 'avg', (64, 64)),
                                             ('unet_max_37x45', 'max', (37, 45))])
def test_unet(name, pooling, hw):
    g = gold(name)
    seed = int(g['seed'])
    sd = det_state_dict(_Shape(unet_shapes()), seed)
    sd['outc.conv.0.bias'] = torch.full((1,), float(g['outc_bias']))
    p = {k: (v.clone().requires_grad_(True) if (v.dtype.is_floating_point and 'running' not in k) else v.clone())
         for k, v in sd.items()}
    x = torch.from_numpy(det_uniform((1, 3) + hw, seed + 100, 0.0, 1.0)).requires_grad_(True)
    y = R.unet_forward(p, x, pooling)
    wts = torch.from_numpy(det_uniform(tuple(y.shape), seed + 200))
    (y * wts).sum().backward()
    close(y, g['out'], 1e-5, 'out')
    close(x.grad[0, :, ::3, ::3], g['dx'], 1e-4, 'dx')
    close(p['inc.double_conv.0.weight'].grad, g['g_inc0'], 1e-4)
    close(p['up1.up.weight'].grad[::8, ::8], g['g_up1_up_w'], 1e-4)
    close(p['outc.conv.0.weight'].grad, g['g_outc_w'], 1e-4)
    close(p['inc.double_conv.1.running_var'], g['rv_inc1'], 1e-5)
    assert int(p['inc.double_conv.1.num_batches_tracked']) == int(g['nbt'])


@pytest.mark.parametrize('pooling', ['max', 'avg'])
def test_layoutnet(pooling):
    g = gold(f'layoutnet_{pooling}')
    seed = int(g['seed'])
    shapes = {'encode.0.weight': (32, 2, 9, 9), 'encode.0.bias': (32,), 'encode.3.weight': (64, 32, 7, 7),
              'encode.3.bias': (64,), 'encode.6.weight': (32, 64, 9, 9), 'encode.6.bias': (32,),
              'encode.8.weight': (1, 32, 7, 7), 'encode.8.bias': (1,)}
    p = {k: v.requires_grad_(True) for k, v in det_state_dict(_Shape(shapes), seed).items()}
    x = torch.from_numpy(det_uniform((1, 2, 32, 32), seed + 100, 0.0, 1.0)).requires_grad_(True)
    y = R.layoutnet_forward(p, x, pooling)
    wts = torch.from_numpy(det_uniform(tuple(y.shape), seed + 200))
    (y * wts).sum().backward()
    close(y, g['out'], 1e-5)
    close(R.layoutnet_forward(p, x.detach()[0], pooling), g['out3d'], 1e-5)
    close(x.grad, g['dx'], 1e-4)
    close(p['encode.8.weight'].grad, g['g_e8_w'], 1e-4)


@pytest.mark.parametrize('i', [0, 1, 2, 3])
def test_mlp(i):
    g = gold(f'mlp_{i}')
    sizes, slope, seed = [int(s) for s in g['sizes']], float(g['slope']), int(g['seed'])
    shapes = {}
    for j in range(1, len(sizes)):
        shapes[f'layers.{2 * (j - 1)}.weight'] = (sizes[j], sizes[j - 1])
        shapes[f'layers.{2 * (j - 1)}.bias'] = (sizes[j],)
    p = {k: v.requires_grad_(True) for k, v in det_state_dict(_Shape(shapes), seed).items()}
    x = torch.from_numpy(det_uniform((9, sizes[0]), seed + 100)).requires_grad_(True)
    y = R.mlp(p, '', x, slope)
    wts = torch.from_numpy(det_uniform(tuple(y.shape), seed + 200))
    (y * wts).sum().backward()
    close(y, g['out'], 1e-5)
    close(x.grad, g['dx'], 1e-4)
    close(p['layers.0.bias'].grad, g['g_b0'], 1e-4)


def test_cell_msg_reduce():
    g = gold('cell_msg_reduce')
    for deg in range(1, 9):
        mail = torch.from_numpy(det_uniform((6, deg, 16), 400 + deg, -3.0, 3.0))
        h = mail.reshape(6 * deg, 16)
        indptr = np.concatenate([np.arange(0, 6 * deg + 1, deg), np.full(6 * deg - 6, 6 * deg)]).astype(np.int64)
        out = R.seg_softmax_sum(h, indptr, np.arange(6 * deg, dtype=np.int64), np.arange(6))
        close(out, g[f'deg{deg}'], 1e-6)


def test_sweep_small():
    """Full multi-level sweep with the fusion head on the fixture DAG: restatement vs the reference's output."""
    g = gold('sweep_small')
    N, D, P = int(g['cell_feat'].shape[0]), 16, 64
    sizes = g['level_sizes']
    starts = np.concatenate([[0], np.cumsum(sizes)])
    levels = [g['level_nodes'][starts[i]:starts[i + 1]] for i in range(len(sizes))]

    class Dz:
        pass
    d = Dz()
    d.N, d.net_src, d.net_dst, d.cell_src, d.cell_dst = N, g['net_src'], g['net_dst'], g['cell_src'], g['cell_dst']
    csr = R.design_csr(d)
    shapes = {}
    for name, (i, hd, o) in {'gnn.fc_cell_neigh': (D, 256, D), 'gnn.fc_cell_self': (36, 256, D),
                             'gnn.fc_net_self': (2, 256, D), 'mlp_fuse': (D + 24 + 32, 2 * (D + 24 + 32), 1),
                             'mlp_alpha': (1, 64, 32)}.items():
        shapes[name + '.layers.0.weight'], shapes[name + '.layers.0.bias'] = (hd, i), (hd,)
        shapes[name + '.layers.2.weight'], shapes[name + '.layers.2.bias'] = (o, hd), (o,)
    shapes['gnn.fc_net_drive.layers.0.weight'], shapes['gnn.fc_net_drive.layers.0.bias'] = (D, 2), (D,)
    shapes['gnn.fc_attn2.weight'] = (1, D)
    shapes['fcn.weight'], shapes['fcn.bias'] = (24, P), (24,)
    p = {k: v.requires_grad_(True) for k, v in det_state_dict(_Shape(shapes), 51).items()}
    fm = torch.from_numpy(det_uniform((1, P), 52, 0.0, 1.0)).requires_grad_(True)
    path_ids = [int(v) for v in g['path_ids']]
    ends, paths = R.bucket_paths(path_ids, g['path2level'], g['path2endpoint'])
    h = torch.zeros((N, D))
    cf, nf = torch.from_numpy(g['cell_feat']), torch.from_numpy(g['net_feat'])
    outs, tl = [], []
    for level_id in range(len(levels)):
        targets, pids = ends.get(level_id, []), paths.get(level_id, [])
        tl.extend(targets)
        pm = R.dense_mask_rows(g['mask_indptr'], g['mask_cols'], pids, P) * fm if pids else None
        h, y = R.pathmodel_level(p, csr, h, cf, nf, levels[level_id], targets, level_id,
                                 torch.tensor([float(level_id)]), pm)
        if y is not None:
            outs.append(y)
    hats = torch.cat(outs)
    loss = torch.nn.functional.mse_loss(hats, torch.from_numpy(g['arrival'])[torch.tensor(tl)].squeeze(-1))
    loss.backward()
    close(hats, g['hats'], 1e-5, 'hats')
    close(h, g['h_final'], 1e-5, 'h')
    close(fm.grad, g['dfeat'], 1e-4, 'dfeat')
    close(p['gnn.fc_cell_neigh.layers.0.weight'].grad, g['g_gnn_fc_cell_neigh_layers_0_weight'], 1e-4)
    close(p['fcn.weight'].grad, g['g_fcn_weight'], 1e-4)
    assert p['gnn.fc_net_drive.layers.0.weight'].grad is None and p['gnn.fc_attn2.weight'].grad is None


# ------------------------------------------------------------------------------------------------ attention branch
def _attn_shapes(D):
    shapes = {}
    for name, (i, hd, o) in {'fc_cell_neigh': (D, 256, D), 'fc_cell_self': (36, 256, D), 'fc_net_self': (2, 256, D)}.items():
        shapes[name + '.layers.0.weight'], shapes[name + '.layers.0.bias'] = (hd, i), (hd,)
        shapes[name + '.layers.2.weight'], shapes[name + '.layers.2.bias'] = (o, hd), (o,)
    shapes['fc_net_drive.layers.0.weight'], shapes['fc_net_drive.layers.0.bias'] = (D, 2), (D,)
    shapes['fc_attn2.weight'] = (1, D)
    shapes['fc_key.weight'], shapes['fc_attn.weight'] = (256, 1), (1, 512)
    return shapes


def attn_level_targets(g):
    """Per-level target lists of the sweep_attn fixture: the level's endpoints in path order + the first one again."""
    p2l, p2e = g['path2level'], g['path2endpoint']
    out = []
    for l in range(len(g['level_sizes'])):
        t = [int(p2e[p]) for p in range(len(p2l)) if int(p2l[p]) == l]
        out.append(t + t[:1])
    return out


def test_attention_udfs():
    """message_func_attn + cell_msg_reduce_attn (src/model.py:125-136) on the edge batches the reference's own UDFs ran on."""
    g = gold('attn_reduce')
    D, n = 16, 5
    sd = det_state_dict(_Shape(_attn_shapes(D)), 71)
    for deg in range(1, 7):
        ksrc = torch.from_numpy(det_uniform((n * deg, 1), 700 + deg, -2.0, 2.0))
        hsrc = torch.from_numpy(det_uniform((n * deg, D), 710 + deg, -1.0, 3.0))
        kdst = torch.from_numpy(det_uniform((n, 1), 720 + deg, -2.0, 2.0))
        key = torch.cat([kdst, ksrc], 0)
        h = torch.cat([torch.zeros(n, D), hsrc], 0)
        indptr = np.concatenate([np.arange(0, n * deg + 1, deg), np.full(n * deg, n * deg)]).astype(np.int64)
        out = R.seg_attn_sum(h, key, indptr, (np.arange(n * deg) + n).astype(np.int64), np.arange(n), sd['fc_key.weight'],
                             sd['fc_attn.weight'])
        close(out, g[f'deg{deg}'], 2e-6, f'deg{deg}')


def test_sweep_attention():
    """The whole level loop with flag_attn=True on the fixture DAG: restatement vs the reference's unmodified
    PathConv.forward (synthetic ndata['key'])."""
    g, gd = gold('sweep_attn'), gold('sweep_small')
    N, D = int(gd['cell_feat'].shape[0]), 16
    sizes = gd['level_sizes']
    starts = np.concatenate([[0], np.cumsum(sizes)])
    levels = [gd['level_nodes'][starts[i]:starts[i + 1]] for i in range(len(sizes))]

    class Dz:
        pass
    d = Dz()
    d.N, d.net_src, d.net_dst, d.cell_src, d.cell_dst = N, gd['net_src'], gd['net_dst'], gd['cell_src'], gd['cell_dst']
    csr = R.design_csr(d)
    p = {'gnn.' + k: v.requires_grad_(True) for k, v in det_state_dict(_Shape(_attn_shapes(D)), 72).items()}
    key = torch.from_numpy(det_uniform((N, 1), 73, -2.0, 2.0))
    cf, nf = torch.from_numpy(gd['cell_feat']), torch.from_numpy(gd['net_feat'])
    h, outs, tl = torch.zeros((N, D)), [], []
    hd = torch.zeros((N, 2))
    for level_id, t in enumerate(attn_level_targets(gd)):
        tl.extend(t)
        h, y = R.pathconv_level(p, 'gnn.', csr, h, cf, nf, levels[level_id], t, level_id, key=key)
        if level_id % 2 == 0:
            hd = hd.index_copy(0, torch.as_tensor(levels[level_id]), R.pathconv_h_drive(csr, nf, levels[level_id]))
        outs.append(y)
    out = torch.cat(outs, 0)
    assert tl == [int(v) for v in g['targets']]
    wts = torch.from_numpy(det_uniform(tuple(out.shape), 74))
    (out * wts).sum().backward()
    close(out, g['out'], 1e-5, 'out')
    close(h, g['h_final'], 1e-5, 'h')
    assert float((hd - torch.from_numpy(g['h_drive'])).abs().max()) < 1e-6
    for k in ('fc_key.weight', 'fc_attn.weight', 'fc_cell_neigh.layers.0.weight', 'fc_net_self.layers.2.bias'):
        close(p['gnn.' + k].grad, g['g_' + k.replace('.', '_')], 2e-4, k)
    assert p['gnn.fc_net_drive.layers.0.weight'].grad is None and p['gnn.fc_attn2.weight'].grad is None


def _up_shapes():
    s = {'conv.double_conv.0.weight': (16, 32, 3, 3), 'conv.double_conv.3.weight': (16, 16, 3, 3)}
    for i in (1, 4):
        for n in ('weight', 'bias', 'running_mean', 'running_var'):
            s[f'conv.double_conv.{i}.{n}'] = (16,)
        s[f'conv.double_conv.{i}.num_batches_tracked'] = ()
    # state_dict order of the reference module: conv.double_conv.0, .1.*, .3, .4.*
    order = ['conv.double_conv.0.weight'] + [f'conv.double_conv.1.{n}' for n in ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked')] + \
            ['conv.double_conv.3.weight'] + [f'conv.double_conv.4.{n}' for n in ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked')]
    return {k: s[k] for k in order}


@pytest.mark.parametrize('tag,s1,s2,seed', [('even', (1, 16, 8, 8), (1, 16, 16, 16), 81), ('odd', (2, 16, 7, 9), (2, 16, 15, 19), 82)])
def test_up_bilinear(tag, s1, s2, seed):
    """Up(32, 16, bilinear=True) (src/Unet.py:48-51,56-68): restatement vs the reference module's output and gradients."""
    g = gold('up_bilinear_' + tag)
    sd = det_state_dict(_Shape(_up_shapes()), seed)
    p = {k: (v.clone().requires_grad_(True) if (v.dtype.is_floating_point and 'running' not in k) else v.clone()) for k, v in sd.items()}
    x1 = torch.from_numpy(det_uniform(s1, seed + 100)).requires_grad_(True)
    x2 = torch.from_numpy(det_uniform(s2, seed + 200)).requires_grad_(True)
    y = R.up_block(p, x1, x2, bilinear=True)
    (y * torch.from_numpy(det_uniform(tuple(y.shape), seed + 300))).sum().backward()
    close(y, g['out'], 2e-6, 'out')
    close(x1.grad, g['dx1'], 1e-4, 'dx1')
    close(x2.grad, g['dx2'], 1e-4, 'dx2')
    close(p['conv.double_conv.0.weight'].grad, g['g_conv0'], 1e-4, 'g_conv0')
