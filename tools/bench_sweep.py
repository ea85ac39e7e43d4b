"""Forward level chain: persistent kernel vs per-level launches (no concurrent stream), and barrier-only cost."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multimodal-fusion-based-pre-routing-timing-prediction-_amd'))
import numpy as np, torch
from mmft.synth import synth_design
from mmft.train import build_models, DesignBatch
from mmft import sweep

dev = torch.device('cuda:0')


def run(designs, label):
    pmodel, _ = build_models(map_size=designs[0].map_size, device=dev, seed=1)
    b = DesignBatch(designs, dev)
    g = b.graph
    h = torch.zeros((b.N, 128), device=dev)
    tg = torch.zeros(0, dtype=torch.int32, device=dev)
    for persistent in (True, False):
        sweep.PERSISTENT_FORWARD = persistent
        with torch.no_grad():
            for _ in range(3):
                g.ndata['h'] = h
                sweep.sweep_forward_all(pmodel.gnn, g, b.level_nodes, tg)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.ndata['h'] = h
                sweep.sweep_forward_all(pmodel.gnn, g, b.level_nodes, tg)
            e1.record()
            torch.cuda.synchronize()
        print(f'{label:40s} persistent={persistent!s:5s} {e0.elapsed_time(e1) / 10:8.3f} ms per forward sweep (incl. prefill GEMMs)', flush=True)
        if persistent:
            print('   barrier time-out flag:', int(g._persist['err'].item()))


run([synth_design(N=65536, L=64, tile=32, seed=9294 + i) for i in range(8)], 'config B graph x8 (8k rows/level)')
run([synth_design(N=65536, L=64, tile=32, seed=9294)], 'one design (1k rows/level)')
run([synth_design(N=2048, L=64, tile=32, seed=1)], 'tiny levels (32 rows/level): barrier cost')
