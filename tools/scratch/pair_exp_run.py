"""Times the reverse sweep's pair launches at config B (8 designs) with the launch profiler; prints us per launch."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'tests'))
import conftest  # noqa
import numpy as np, torch
from mmft import lib, ops, sweep as S
from mmft.synth import config_design
from mmft.train import build_models, DesignBatch
dev = torch.device('cuda:0')
lib.set_math_mode('bf16')
designs = [config_design('B', i) for i in range(8)]
b = DesignBatch(designs, dev)
pmodel, _ = build_models(map_size=designs[0].map_size, device=dev, seed=8)
ends = b.select([np.arange(0, d.num_paths, 3)[:1350] for d in designs])[0]
pairs = b.graph.level_bwd_pairs(b.level_nodes)
nt = [p['ntiles'] for p in pairs[1]]
for it in range(4):
    g = b.graph
    g.ndata['h'] = torch.zeros((b.N, 128), dtype=torch.float32, device=dev)
    for p in pmodel.gnn.parameters():
        p.grad = None
    out = S.sweep_forward_all(pmodel.gnn, g, b.level_nodes, ends)
    if it == 3:
        lib.prof_reset(); lib.prof_enable(True)
    (out * out).sum().backward()
    torch.cuda.synchronize()
lib.prof_enable(False)
rep = {r['name']: r for r in lib.prof_report()}
r = rep['level_bwd_pair_kernel']
print('tiles/level min %d max %d; level_bwd_pair_kernel %d launches, %.1f us each' % (min(nt[1:]), max(nt[1:]), r['launches'], 1e3 * r['ms'] / r['launches']))
