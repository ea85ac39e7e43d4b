"""A/B of the drop-in loop's host-side options inside ONE process (the loop is host-bound: box-to-box and run-to-run noise is
larger than the effects): the flags are toggled every 10 steps on the same TrainStep, 5 rounds."""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'multimodal-fusion-based-pre-routing-timing-prediction-_amd')]
import numpy as np, torch
import bench
from mmft import lib, sweep as S
from mmft.synth import synth_design
from mmft.train import build_models, TrainStep
bench.limit_host_threads(1)
lib.set_math_mode('bf16')
dev = torch.device('cuda:0')
designs = [synth_design(N=65536, L=64, tile=256, seed=9294 + i) for i in range(8)]
pm, cnn = build_models(map_size=designs[0].map_size, device=dev, seed=9294)
ts = TrainStep(pm, cnn, designs, dev, mode='dropin', keep_grads=False)
rng = np.random.default_rng(0)
ids = lambda: [rng.permutation(d.num_paths)[:1350] for d in designs]
gc.collect(); gc.freeze()
configs = {'plain': (False, False, False), 'record': (False, False, True), 'record+side': (True, False, True), 'record+side+replay': (True, True, True),
           'record+replay': (False, True, True)}
for name, f in configs.items():          # warm every configuration (captures)
    S.SPEC_SIDE_STREAM, S.SWEEP_REPLAY, S.RECORD_LAUNCHES = f
    for _ in range(4):
        ts.step(ids())
torch.cuda.synchronize()
acc = {k: [] for k in configs}
for rnd in range(5):
    for name, f in configs.items():
        S.SPEC_SIDE_STREAM, S.SWEEP_REPLAY, S.RECORD_LAUNCHES = f
        for _ in range(2):
            ts.step(ids())
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            ts.step(ids())
        torch.cuda.synchronize()
        acc[name].append((time.perf_counter() - t) / 10 * 1e3)
for k, v in acc.items():
    print('%-22s median %.2f ms  (%s)' % (k, float(np.median(v)), ' '.join('%.2f' % x for x in v)))
