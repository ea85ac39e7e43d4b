"""CPU, world_size 2, gloo: the data-parallel step = shard designs by rank, sum-all-reduce ONE flat gradient
buffer, scale by 1/world inside the optimizer.  Checked against the oracle: the reduced buffer equals the
gradient of the mean of the per-design losses."""
import os
import sys
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'multimodal-fusion-based-pre-routing-timing-prediction-_amd')


def _worker(rank, world, port, out_dir):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from mmft.dist import design_seeds, allreduce_sum_
    from mmft.synth import synth_design
    from oracle import restatement as R
    seeds = design_seeds(rank, 1)
    d = synth_design(N=512, L=8, tile=16, seed=seeds[0], end_frac=0.5)
    torch.manual_seed(0)                                   # identical parameters on every rank
    p = {}
    for name, (i, hd, o) in {'gnn.fc_cell_neigh': (16, 256, 16), 'gnn.fc_cell_self': (36, 256, 16),
                             'gnn.fc_net_self': (2, 256, 16), 'mlp_fuse': (16 + 32, 96, 1), 'mlp_alpha': (1, 64, 32)}.items():
        p[name + '.layers.0.weight'] = (torch.randn(hd, i) * 0.1).requires_grad_(True)
        p[name + '.layers.0.bias'] = torch.zeros(hd, requires_grad=True)
        p[name + '.layers.2.weight'] = (torch.randn(o, hd) * 0.1).requires_grad_(True)
        p[name + '.layers.2.bias'] = torch.zeros(o, requires_grad=True)
    csr = R.design_csr(d)
    ends, paths = R.bucket_paths(list(range(d.num_paths)), d.path2level, d.path2endpoint)
    h = torch.zeros((d.N, 16))
    cf, nf = torch.from_numpy(d.cell_feat), torch.from_numpy(d.net_feat)
    outs, tl = [], []
    for l in range(d.L):
        t = ends.get(l, [])
        tl.extend(t)
        h, y = R.pathmodel_level(p, csr, h, cf, nf, d.levels[l], t, l, torch.tensor([float(l)]), None, has_fcn=False)
        if y is not None:
            outs.append(y)
    loss = torch.nn.functional.mse_loss(torch.cat(outs), torch.from_numpy(d.arrival_time)[torch.tensor(tl)].squeeze(-1))
    loss.backward()
    keys = sorted(p)
    flat = torch.cat([p[k].grad.reshape(-1) for k in keys])
    local = flat.clone()
    scale = allreduce_sum_(flat, world)
    torch.save({'seeds': seeds, 'local': local, 'reduced': flat * scale, 'scale': scale}, os.path.join(out_dir, f'r{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'r0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'r1.pt'))
    assert r0['seeds'] == [9294] and r1['seeds'] == [9295]            # disjoint shards
    assert r0['scale'] == 0.5
    mean = (r0['local'] + r1['local']) / 2
    assert torch.allclose(r0['reduced'], mean, rtol=1e-6, atol=1e-9)
    assert torch.equal(r0['reduced'], r1['reduced'])
    assert not torch.allclose(r0['local'], r1['local'])
