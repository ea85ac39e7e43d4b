"""Host-side duration of the HIP-graph replays of the train step vs the device-side step time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'multimodal-fusion-based-pre-routing-timing-prediction-_amd')]
import numpy as np, torch
from mmft import lib
from mmft.synth import synth_design
from mmft.train import build_models, TrainStep, GraphedTrainStep
lib.set_math_mode(os.environ.get('MMFT_MATH', 'bf16'))
from mmft import sweep as _S, unet16 as _U
_S.LEVEL_BWD_PAIRS = os.environ.get('PAIRS', '1') == '1'
_U.BATCH_REDUCE = os.environ.get('BATCHRED', '1') == '1'
print('pairs', _S.LEVEL_BWD_PAIRS, 'batch reduce', _U.BATCH_REDUCE)
dev = torch.device('cuda:0')
designs = [synth_design(N=65536, L=64, tile=256, seed=9294 + i) for i in range(8)]
pm, cnn = build_models(map_size=designs[0].map_size, device=dev, seed=9294)
ts = TrainStep(pm, cnn, designs, dev)
rng = np.random.default_rng(0)
ids = lambda: [rng.permutation(d.num_paths)[:1350] for d in designs]
gs = GraphedTrainStep(ts, ids(), pieces=os.environ.get('PIECES', '1') == '1')
for _ in range(3):
    gs.step(ids())
torch.cuda.synchronize()
batches = [ids() for _ in range(10)]
if gs.pieces:
    # time every replay call on the host, device idle at the start of each step
    acc = {k: [] for k in ('select', 'gA', 'gB', 'gH', 'gbA', 'gbB', 'adam')}
    for bt in batches:
        torch.cuda.synchronize()
        t = time.perf_counter(); ts.batch.select(bt, static=gs.static_idx); acc['select'].append(time.perf_counter() - t)
        for name in ('gA', 'gB', 'gH', 'gbA', 'gbB'):
            t = time.perf_counter(); getattr(gs, name).replay(); acc[name].append(time.perf_counter() - t)
            torch.cuda.synchronize()
        t = time.perf_counter(); ts.optim.step(); acc['adam'].append(time.perf_counter() - t)
    print('host ms per call (device idle before each):', {k: round(float(np.mean(v)) * 1e3, 3) for k, v in acc.items()})
t_all = time.perf_counter()
for bt in batches:
    gs.step(bt)
t_issue = time.perf_counter() - t_all
torch.cuda.synchronize()
t_tot = time.perf_counter() - t_all
print(f'issue loop {t_issue/10*1e3:.2f} ms/step (host), with final sync {t_tot/10*1e3:.2f} ms/step')

if gs.pieces:
    main, side = torch.cuda.current_stream(), ts.side

    def t(f, reps=10):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def onA():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            gs.gA.replay()
        main.wait_stream(side)

    def onB():
        gs.gB.replay()

    def both():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            gs.gA.replay()
        gs.gB.replay()
        main.wait_stream(side)

    def both_rev():
        side.wait_stream(main)
        gs.gB.replay()
        with torch.cuda.stream(side):
            gs.gA.replay()
        main.wait_stream(side)
    a, b_ = t(onA), t(onB)
    print(f'forward pieces in isolation: A (sweep, side) {a:.3f} ms, B (U-Net, main) {b_:.3f} ms, sum {a + b_:.3f}; '
          f'both (A issued first) {t(both):.3f} ms, both (B issued first) {t(both_rev):.3f} ms')

    def piece(name, stream=None):
        g = getattr(gs, name)

        def run():
            if stream is None:
                g.replay()
            else:
                stream.wait_stream(main)
                with torch.cuda.stream(stream):
                    g.replay()
                main.wait_stream(stream)
        return t(run)

    def bwd_both():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            gs.gbA.replay()
        gs.gbB.replay()
        main.wait_stream(side)
    # the pieces replay on whatever the previous replay left in the buffers: timing only
    res = {n: piece(n, side if n in ('gA', 'gbA') else None) for n in ('gA', 'gB', 'gH', 'gbA', 'gbB')}
    res['adam'] = t(lambda: ts.optim.step())
    res['bA+bB'] = t(bwd_both)
    print('pieces in isolation (ms):', {k: round(v, 3) for k, v in res.items()})
    print('ideal overlap: %.3f ms; serial: %.3f ms' % (max(res['gA'], res['gB']) + res['gH'] + max(res['gbA'], res['gbB']) + res['adam'],
                                                      sum(res[k] for k in ('gA', 'gB', 'gH', 'gbA', 'gbB', 'adam'))))
