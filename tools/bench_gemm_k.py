import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multimodal-fusion-based-pre-routing-timing-prediction-_amd'))
import torch
from mmft import ops
from bench_gemm import timeit, dev
M, N = 245760, 256
for K in (16, 32, 64, 128, 256, 512, 1024):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); y = torch.empty(M, N, device=dev)
    us = timeit(lambda: ops.linear_fwd(x, w, None, y=y))
    print(f'K={K:5d} {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF   store-only floor {M*N*4/6.3e6:.0f} us', flush=True)
# pure store bandwidth reference
z = torch.empty(M, N, device=dev)
us = timeit(lambda: z.zero_())
print('torch zero_ of the same output', us, 'us')
