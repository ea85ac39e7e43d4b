"""Where do the torch-native small kernels (add / copy / fill) of one eager train step come from?

    python tools/trace_small_ops.py [--nodes N --levels L --tile T --designs B]
Prints, per aten op that launches a device kernel outside the library, the python call sites (within this repo)
with launch counts for one step.  Development aid; not part of the product path.
"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'multimodal-fusion-based-pre-routing-timing-prediction-_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--designs', type=int, default=8)
    ap.add_argument('--nodes', type=int, default=65536)
    ap.add_argument('--levels', type=int, default=64)
    ap.add_argument('--tile', type=int, default=256)
    args = ap.parse_args()
    from mmft.synth import synth_design
    from mmft.train import build_models, TrainStep
    dev = torch.device('cuda', 0)
    designs = [synth_design(N=args.nodes, L=args.levels, tile=args.tile, seed=100 + i) for i in range(args.designs)]
    pmodel, cnn = build_models(map_size=designs[0].map_size, device=dev, seed=9294)
    from mmft import lib
    lib.set_math_mode('bf16')
    ts = TrainStep(pmodel, cnn, designs, dev, mode='sweep', overlap=False)
    rng = np.random.default_rng(0)
    pick = lambda: [rng.permutation(d.num_paths)[:1350] for d in designs]
    for _ in range(2):
        ts.step(pick())
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        ts.step(pick())
        torch.cuda.synchronize()
    sites = collections.defaultdict(collections.Counter)
    for ev in prof.events():
        if not (ev.kernels or []):
            continue
        if not ev.name.startswith('aten::'):
            continue
        # attribute the launch to its outermost aten ancestor and the first frame of this repo on any ancestor's stack
        root, site, cur = ev, None, ev
        while cur is not None:
            if cur.name.startswith('aten::'):
                root = cur
            for fr in cur.stack or []:
                if site is None and ROOT in fr and 'tools/trace_small_ops' not in fr:
                    site = fr.replace(ROOT + '/', '')
            cur = cur.cpu_parent
        kname = ','.join(sorted({k.name.split('<')[0].split('(')[0][-40:] for k in ev.kernels}))
        sites[f'{root.name} -> {ev.name} [{kname}]'][site or 'autograd engine / no python frame'] += len(ev.kernels)
    for op, c in sorted(sites.items(), key=lambda kv: -sum(kv[1].values())):
        print(f'{op}: {sum(c.values())} launches/step')
        for site, n in c.most_common(12):
            print(f'    {n:4d}  {site}')


if __name__ == '__main__':
    main()
