"""Standalone form of the packed-fma reproducer: ONE kernel of this library - mmft_masked_fc_prefix, built with
`make EXTRA=-DMMFT_ALLOW_PACKED_OPSEL` so that it contains v_pk_fma_f32 ... op_sel:[0,1,0] - on one stream, torch bf16
matmuls (hipBLASLt MFMA kernels, nothing of this library) on another; no model, no autograd.

    MMFT_LIB=/path/to/the/packed/build/libmmft_hip.so CORUN=8064,128,256 python tools/packed_fma_standalone.py [rounds]

Three schedules per round, 12 launches of the prefix kernel each, every launch's table copied out on the stream and compared
bit for bit (after the schedule has drained) with the table the same kernel produced alone: (a) prefix kernels only, (b) eager
two-stream overlap, (c) the same two-stream work captured into one HIP graph and replayed.  Measured on MI355X
(profiles/r03_packed_fma_standalone_*.txt): packed build 0 / 0 / 20 wrong tables of 120 launches each - only the replayed
graph - with the signature of the model-level reproducer (segments that start at cells 1 mod 4, 16 columns of the channel
groups 16..31, the lost amount exactly f[c] wT[c]); shipped build 0 / 0 / 0."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'multimodal-fusion-based-pre-routing-timing-prediction-_amd'))
from mmft import lib

dev = torch.device('cuda:0')
B, P, D, S, N = 8, 65536, 128, 64, 12
g = torch.Generator(device='cpu').manual_seed(5)
f = torch.rand(B * P, generator=g).to(dev)
wT = (torch.randn(P, D, generator=g) * 0.02).to(dev)
GP = torch.empty(B * P, D, device=dev)
ref = torch.empty_like(GP)
M_, K_, N_ = [int(x) for x in os.environ.get('CORUN', '8064,128,256').split(',')]      # the co-running matmul: (M, K) @ (K, N), bf16
a16 = torch.randn(M_, K_, generator=g).to(dev).bfloat16()
b16 = torch.randn(K_, N_, generator=g).to(dev).bfloat16()
c16 = torch.empty(M_, N_, device=dev, dtype=torch.bfloat16)
REP = int(os.environ.get('COREP', '8'))
outs = [torch.empty_like(GP) for _ in range(N)]          # every launch's table is kept and compared after the schedule has drained
side = torch.cuda.Stream(device=dev)
import numpy as np


def prefix(out):
    d, st = lib.stream_args(f)
    lib.call('mmft_masked_fc_prefix', f, wT, out, B, P, D, S, d, st)


def work(overlap):
    cur = torch.cuda.current_stream(dev)
    if overlap:
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(REP * N):
                torch.matmul(a16, b16, out=c16)
    for i in range(N):
        prefix(GP)
        outs[i].copy_(GP)
    if overlap:
        cur.wait_stream(side)


shown = [False]


def tally():
    torch.cuda.synchronize()
    n = 0
    for o in outs:
        rows = torch.nonzero((o != ref).any(1)).flatten()
        if rows.numel():
            n += 1
            if not shown[0]:
                shown[0] = True
                r = rows.cpu().numpy()
                starts = np.concatenate([[r[0]], r[1:][np.diff(r) > 1]])
                c0 = int(starts[0])
                cols = torch.nonzero(o[c0] != ref[c0]).flatten().cpu().numpy()
                prev = o[c0 - 1] if c0 % S else torch.zeros_like(o[c0])
                lost = float(((ref[c0] - o[c0])[cols] / (f[c0] * wT[c0 % P][cols])).mean())
                print('  first wrong table: %d wrong rows in %d segments; segment starts mod 4: %s; first segment: cell %d, wrong columns %s,'
                      ' (ref - got) / (f[c] wT[c]) = %.4f' % (r.size, starts.size, dict(zip(*np.unique(starts % 4, return_counts=True))),
                                                             c0, cols.tolist(), lost), flush=True)
    return n


prefix(ref)
torch.cuda.synchronize()
warm = torch.cuda.Stream(device=dev)
with torch.cuda.stream(warm):
    work(True)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    work(True)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = [0, 0, 0]
for r in range(rounds):
    work(False)
    bad[0] += tally()
    work(True)
    bad[1] += tally()
    graph.replay()
    bad[2] += tally()
    if r % 5 == 4 or r == rounds - 1:
        print('round', r + 1, 'wrong tables: alone %d, eager overlap %d, graph replay %d  (of %d launches each)'
              % (*bad, (r + 1) * N), flush=True)
