"""Does a short-K GEMM suffer from lock-step phases (all workgroups loading / multiplying / storing together)?
Split M over S concurrent streams and compare with the single launch.  Development aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multimodal-fusion-based-pre-routing-timing-prediction-_amd'))
import torch
from mmft import ops

dev = torch.device('cuda:0')


def run(M, N, K, S, n=20):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    streams = [torch.cuda.Stream() for _ in range(S)]
    part = M // S

    def go():
        cur = torch.cuda.current_stream()
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                ops.linear_fwd(x[i * part:(i + 1) * part], w, b, y=y[i * part:(i + 1) * part])
        for s in streams:
            cur.wait_stream(s)

    for _ in range(3):
        go()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        go()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    print(f'M={M} N={N} K={K} streams={S}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TF', flush=True)


for S in (1, 2, 4, 8):
    run(245760, 256, 128, S)
for S in (1, 4):
    run(245760, 128, 256, S)
