"""Reproducer of the packed-fma finding (csrc/common.h MMFT_NO_PACKED_F32).

    (cd multimodal-fusion-based-pre-routing-timing-prediction-_amd/csrc && make clean && make -j8 EXTRA=-DMMFT_ALLOW_PACKED_OPSEL)
    MMFT_JOIN_LATE=1 python tools/packed_fma_repro.py 30

Config B at full size, bf16 mode.  Reference: the eager step without stream overlap.  Every run captures the whole-step HIP
graph with the masked projection issued BEFORE the sweep stream is joined (MMFT_JOIN_LATE=1), replays four steps and compares
the projection's transposed weight (before and after the prefix kernel), the prefix table GP and the projection output with
the reference, bit for bit.  With v_pk_fma_f32 ... op_sel:[0,1,0] in fc_prefix_kernel about one run in five shows rows of GP
that lack one f[c] * wT[c] term in 16 columns (columns 4 k + 0 or 4 k + 2 of the channel groups 16..31, cell c = 1 mod 4:
the low result of the instruction, one 16-lane row) from cell c to the end of its 64-cell block, with wT correct in memory
before and after the launch; with the shipped build (no packed fp32 instructions in that kernel) none in 130 runs."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'multimodal-fusion-based-pre-routing-timing-prediction-_amd'))
from mmft import lib, fusion
from mmft.synth import synth_design
from mmft.train import build_models, TrainStep, GraphedTrainStep
dev = torch.device('cuda:0')
designs = [synth_design(N=65536, L=64, tile=256, seed=9294 + i) for i in range(8)]
rng = np.random.default_rng(6)
batches = [[rng.permutation(d.num_paths)[:1350] for d in designs] for _ in range(2)]
STEPS = 4
stash = {}
live = {}
orig_cached = fusion._fc_cached
def cached(kind, deps, build):
    v = orig_cached(kind, deps, build)
    stash.setdefault(kind, []).append(v.detach().clone())
    if kind == 'wT':
        live['wT'] = v
    if kind == 'GP':
        stash.setdefault('wT2', []).append(live['wT'].detach().clone())     # wT as memory holds it AFTER the prefix kernel
    return v
fusion._fc_cached = cached

def run(kind, overlap=True):
    stash.clear()
    with lib.math_mode('bf16'):
        pmodel, cnn = build_models(map_size=designs[0].map_size, device=dev, seed=9294)
        ts = TrainStep(pmodel, cnn, designs, dev, overlap=overlap)
        orig = pmodel._fcn
        def fcn(pm):
            o = orig(pm)
            stash.setdefault('out', []).append(o.detach().clone())
            stash.setdefault('f', []).append(pm.feat_map.detach().clone())
            return o
        pmodel._fcn = fcn
        stepper = GraphedTrainStep(ts, batches[0], warmup=0) if kind == 'graph' else ts
        n0 = {k: len(v) for k, v in stash.items()}
        outs = []
        for i in range(STEPS):
            stepper.step(batches[i % 2])
            torch.cuda.synchronize()
            outs.append({k: (v[-1] if kind == 'graph' else v[n0.get(k, 0) + i]).clone() for k, v in stash.items()})
        info = dict(ptr={k: v[-1].data_ptr() for k, v in stash.items()})
        del ts, stepper, pmodel, cnn
    return outs, info

ref, _ = run('eager', overlap=False)
nbad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    g, info = run('graph')
    line = []
    reported = False
    for i in range(STEPS):
        bad = {k: int(((g[i][k].reshape(ref[i][k].shape[0], -1) != ref[i][k].reshape(ref[i][k].shape[0], -1)).any(1)).sum()) for k in ref[i]}
        bad = {k: v for k, v in bad.items()}
        line.append(str({k: v for k, v in bad.items() if k in ('GP', 'wT2', 'out')}))
        if bad['GP'] and not reported:
            reported = True
            x, y = g[i]['GP'], ref[i]['GP']
            rows = torch.nonzero((x != y).any(1)).flatten().cpu().numpy()
            brk = np.nonzero(np.diff(rows) > 1)[0]
            starts = np.concatenate([[rows[0]], rows[brk + 1]])
            print('    step', i, 'GP bad rows', rows.size, 'segment starts', starts, 'mod 64', starts % 64, flush=True)
            wT = ref[i]['wT']
            P = wT.shape[0]
            for s0 in starts[:4]:
                c = int(s0)
                delta = (x[c] - y[c]).double()
                w = wT[c % P].double()
                alpha = float((delta * w).sum() / (w * w).sum())
                res = float((delta - alpha * w).abs().max())
                ftrue = float(ref[i]['f'].reshape(-1)[c]); fprev = float(ref[i - 1]['f'].reshape(-1)[c]) if i else float('nan')
                ncol = int((x[c] != y[c]).sum())
                cols = torch.nonzero(x[c] != y[c]).flatten().cpu().numpy()
                prev_row = x[c - 1] if c % 64 else torch.zeros_like(x[c])
                w_seen = ((x[c] - prev_row) / ftrue)[cols].cpu().numpy()
                print('        bad cols', cols[[0, -1]], 'contiguous', bool((np.diff(cols) == 1).all()), 'wT the kernel used', np.round(w_seen[:6], 5),
                      'true wT', np.round(wT[c % P][cols][:6].cpu().numpy(), 5), 'wT2 (after prefix) == true:', bool((g[i]['wT2'][c % P] == wT[c % P]).all()), flush=True)
                print('      cell %d (design %d, y %d, x %d): bad cols %d, delta = alpha*wT with alpha %.6g (residual %.2g); f true %.6g, f prev step %.6g, prev - true %.6g'
                      % (c, c // P, (c % P) // 256, c % 256, ncol, alpha, res, ftrue, fprev, fprev - ftrue), flush=True)
    print(it, ' | '.join(line), flush=True)
    nbad += reported
print('runs with a wrong prefix table: %d of %d' % (nbad, it + 1))
