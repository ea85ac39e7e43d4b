"""Tile exploration for the big short-K GEMMs (subprocess per forced tile: the hook is read at launch)."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
code = r'''
import os, sys
sys.path.insert(0, %r)
from bench_gemm import timeit, dev
import torch
from mmft import ops, lib
lib.set_math_mode(os.environ.get('MMFT_MATH', 'f32'))      # MMFT_MATH=bf16: the bf16-operand engine (64-deep steps)
M = 245760
tag = os.environ.get("MMFT_GEMM_FORCE", "default")
for (N, K) in ((256, 128), (128, 256), (256, 36)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); y = torch.empty(M, N, device=dev)
    us = timeit(lambda: ops.linear_fwd(x, w, None, y=y))
    print(f"{tag:12s} fwd   M={M} N={N:4d} K={K:4d} {us:8.1f} us {2.0*M*N*K/us/1e6:6.1f} TF", flush=True)
g = torch.randn(M, 128, device=dev); w2 = torch.randn(128, 256, device=dev); m = torch.randn(M, 256, device=dev)
us = timeit(lambda: ops.linear_dgrad(g, w2, mask=m))
print(f"{tag:12s} dgrad+mask 128->256            {us:8.1f} us {2.0*M*256*128/us/1e6:6.1f} TF", flush=True)
hh = torch.randn(M, 256, device=dev)
us = timeit(lambda: ops.linear_wgrad(g, hh))
print(f"{tag:12s} wgrad 128x256 over 245k rows   {us:8.1f} us {2.0*M*256*128/us/1e6:6.1f} TF", flush=True)
''' % HERE
tiles = (None, '128x128x32', '128x64x32', '64x128x32', '64x64x32') if os.environ.get('MMFT_MATH') == 'bf16' else \
    (None, '128x128x32', '128x64x32', '64x128x32', '64x64x32', '64x64x64', '128x64x16', '64x128x16', '64x64x16')
for force in tiles:
    env = dict(os.environ)
    if force:
        env['MMFT_GEMM_FORCE'] = force
    subprocess.run([sys.executable, '-c', code], env=env, cwd=HERE)
