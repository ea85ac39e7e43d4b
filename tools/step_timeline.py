"""Where the replayed whole-step HIP graph spends its time: device timestamps (mmft_prof_stamp, wall_clock64) at the joints of
the step - end of the netlist sweep's forward (side stream), of the U-Net's forward, of the masked projection, of the fusion
head, the loss; the moments the gradient reaches the head, the projection, the U-Net and the sweep; the end of the reverse sweep
(side stream) and of the whole backward; Adam.  The kernel trace cannot show this (rocprofv3 serialises the dispatches of a
graph); host timers see one graph launch.

    python tools/step_timeline.py            (config B, bf16 mode; prints the median over 20 replays)
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'multimodal-fusion-based-pre-routing-timing-prediction-_amd')]
import numpy as np, torch
from mmft import lib, sweep as S
from mmft.fusion import unit_grad
from mmft.synth import synth_design
from mmft.train import build_models, TrainStep
lib.set_math_mode(os.environ.get('MMFT_MATH', 'bf16'))
dev = torch.device('cuda:0')
TAGS = ['start', 'A_fwd_end', 'B_fwd_end', 'fcn_fwd_end', 'head_fwd_end', 'loss_end', 'head_bwd_begin', 'fuse_bwd_end@fcn', 'fuse_bwd_end@A',
        'fcn_bwd_end', 'A_bwd_end', 'bwd_end', 'adam_end']
slots = torch.zeros(len(TAGS), dtype=torch.int64, device=dev)


def stamp(tag):
    d, st = lib.stream_args(slots)
    lib.call('mmft_prof_stamp', slots, TAGS.index(tag), d, st)


class Stamp(torch.autograd.Function):
    """Identity; stamps when the forward passes and when the gradient comes back."""

    @staticmethod
    def forward(ctx, x, ftag, btag):
        ctx.btag = btag
        if ftag:
            stamp(ftag)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if ctx.btag:
            stamp(ctx.btag)
        return g, None, None


designs = [synth_design(N=65536, L=64, tile=256, seed=9294 + i) for i in range(8)]
pm, cnn = build_models(map_size=designs[0].map_size, device=dev, seed=9294)
ts = TrainStep(pm, cnn, designs, dev, keep_grads=False)
rng = np.random.default_rng(0)
ids = lambda: [rng.permutation(d.num_paths)[:1350] for d in designs]
for _ in range(3):
    ts.step(ids())
torch.cuda.synchronize()

# the joints
_sfa = S.sweep_forward_all
S.sweep_forward_all = lambda *a, **k: Stamp.apply(_sfa(*a, **k), 'A_fwd_end', 'fuse_bwd_end@A')
_cnn_fwd = cnn.forward
cnn.forward = lambda x: Stamp.apply(_cnn_fwd(x), 'B_fwd_end', 'fcn_bwd_end')
_fcn = pm._fcn
pm._fcn = lambda m: Stamp.apply(_fcn(m), 'fcn_fwd_end', 'fuse_bwd_end@fcn')
_fuse = pm.fuse_heads
pm.fuse_heads = lambda *a, **k: Stamp.apply(_fuse(*a, **k), 'head_fwd_end', 'head_bwd_begin')

b = ts.batch
T = sum(len(p) for p in ids())
static = torch.zeros(6 * T + b.path2level.shape[0], dtype=torch.int32, device=dev)
sel = b.select(ids(), static=static)
torch.cuda.synchronize()
ts.optim.flat_grad.zero_()
ts.optim._cleared = [True] * len(ts.optim._cleared)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, capture_error_mode='thread_local'):
    stamp('start')
    hats, ends_d, _ = ts.forward(None, _sel=sel)
    loss = ts.loss(hats, ends_d)
    stamp('loss_end')
    ts.optim.zero_grad()
    loss.backward(unit_grad(dev))
    with torch.cuda.stream(ts.side):
        stamp('A_bwd_end')
    stamp('bwd_end')
    torch.cuda.current_stream(dev).wait_stream(ts.side)
    ts.optim.step_captured()
    stamp('adam_end')
torch.cuda.synchronize()
rows = []
import time
for _ in range(20):
    b.select(ids(), static=static)
    graph.replay()
    torch.cuda.synchronize()
    v = slots.cpu().numpy().astype(np.float64)
    rows.append((v - v[0]) / 100.0)            # 100 MHz -> microseconds
t0 = time.perf_counter()
for _ in range(20):
    b.select(ids(), static=static)
    graph.replay()
torch.cuda.synchronize()
print('wall per step (20 back-to-back replays): %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
med = np.median(np.array(rows), axis=0)
for tag, us in sorted(zip(TAGS, med), key=lambda x: x[1]):
    print('%-22s %9.1f us' % (tag, us))
