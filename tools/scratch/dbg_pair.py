import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'tests'))
import conftest  # noqa
import numpy as np, torch
from mmft import lib, ops, sweep as S
from mmft.synth import synth_design
from mmft.train import build_models, DesignBatch
dev = torch.device('cuda:0')
lib.set_math_mode('bf16')
S.LEVEL_BWD_PAIRS = os.environ.get('PAIRS', '1') == '1'
print('pairs', S.LEVEL_BWD_PAIRS)
designs = [synth_design(N=6000, L=12, tile=32, seed=120 + i, end_frac=0.2) for i in range(2)]
b = DesignBatch(designs, dev)
pmodel, _ = build_models(map_size=designs[0].map_size, device=dev, seed=8)
ends = b.select([np.arange(0, d.num_paths, 3) for d in designs])[0]
res = []
for fuse, slots in ((True, True), (True, False), (False, False), (True, True)):
    S.FUSE_LEVEL_FWD, S.LEVEL_SLOTS = fuse, slots
    g = b.graph
    g.ndata['h'] = torch.zeros((b.N, 128), dtype=torch.float32, device=dev)
    for p in pmodel.gnn.parameters():
        p.grad = None
    out = S.sweep_forward_all(pmodel.gnn, g, b.level_nodes, ends)
    st = g._sweep
    (out * out).sum().backward()
    torch.cuda.synchronize()
    rows = torch.as_tensor(np.concatenate([np.asarray(x) for x in b.level_nodes]), device=dev).long()
    rows2 = torch.as_tensor(np.concatenate([np.asarray(x) for x in b.level_nodes[2::2]]), device=dev).long()
    res.append(dict(Gall=st.G.clone(), G=st.G[rows].clone(), DA=st.DA[rows2].clone(), DHN=st.DHN[rows2].clone(), HN=st.HN[rows2].clone(), A=st.A[rows2].clone(),
                    LSE=st.LSE[rows2].clone(), h=g.ndata['h'].clone(),
                    **{k: p.grad.clone() for k, p in pmodel.gnn.named_parameters() if p.grad is not None}))

for k in ('fc_net_self.layers.0.weight', 'fc_net_self.layers.2.bias'):
    for i, j in ((0, 1), (1, 2), (1, 3), (0, 3)):
        a, c = res[i][k], res[j][k]
        print(k, i, j, float((a - c).abs().max()), float(a.abs().max()))
# rows of G in the net range that are not level nodes?
st = b.graph._sweep
print('row_sets', st.row_sets)
rn = st.row_sets[1]
allr = torch.zeros(b.N, dtype=torch.bool, device=dev); allr[rows] = True
print('net range rows not in levels:', int((~allr[rn[0]:rn[0] + rn[1]]).sum()))

d = (res[0]['Gall'] != res[1]['Gall']).any(1).nonzero().flatten()
print('all-row G diffs run0 vs run1:', d.numel(), d[:10].tolist(), 'N', b.N)
