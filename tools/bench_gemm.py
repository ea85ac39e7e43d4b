"""Micro-benchmark of the GEMM engine through the C ABI (HIP-event timing on the launch stream)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multimodal-fusion-based-pre-routing-timing-prediction-_amd'))
import torch
from mmft import ops

dev = torch.device('cuda:0')


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


def run(tag, M, N, K, **kw):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    us = timeit(lambda: ops.linear_fwd(x, w, b, y=y, **kw))
    print(f'{tag:40s} M={M:7d} N={N:4d} K={K:5d}  {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF', flush=True)


if __name__ == "__main__":
    run('fwd MK,MK plain', 245760, 256, 128)
    run('fwd MK,MK plain', 245760, 128, 256)
    run('fwd MK,MK plain', 245760, 256, 36)
    run('fwd MK,MK big K', 8192, 4096, 4096)
    run('fwd MK,MK big K', 16384, 256, 4096)
    run('level fwd', 8064, 256, 128)
    run('level fwd', 8064, 128, 256)
    run('level fwd relu', 8064, 256, 128, act=ops.ACT_RELU)
    idx = torch.randperm(245760, device=dev).to(torch.int32)
    run('fwd gather+scatter (random rows)', 245760, 256, 128, xidx=idx, yidx=idx)
    g = torch.randn(245760, 128, device=dev); w2 = torch.randn(128, 256, device=dev); m = torch.randn(245760, 256, device=dev)
    us = timeit(lambda: ops.linear_dgrad(g, w2))
    print(f'{"dgrad MK,KM plain":40s} {us:8.1f} us {2.0*245760*256*128/us/1e6:7.1f} TF')
    us = timeit(lambda: ops.linear_dgrad(g, w2, mask=m))
    print(f'{"dgrad MK,KM mask":40s} {us:8.1f} us {2.0*245760*256*128/us/1e6:7.1f} TF')
    us = timeit(lambda: ops.linear_dgrad(g, w2, mask=m, gidx=idx, maskidx=idx))
    print(f'{"dgrad MK,KM mask + gathers":40s} {us:8.1f} us {2.0*245760*256*128/us/1e6:7.1f} TF')
    hh = torch.randn(245760, 256, device=dev)
    us = timeit(lambda: ops.linear_wgrad(g, hh))
    print(f'{"wgrad KM,KM 128x256 over 245k rows":40s} {us:8.1f} us {2.0*245760*256*128/us/1e6:7.1f} TF')
