"""A handful of GEMM launches for a rocprofv3 --pmc pass (one warm-up + one measured launch per shape)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multimodal-fusion-based-pre-routing-timing-prediction-_amd'))
import torch
from mmft import ops
dev = torch.device('cuda:0')
M = 245760
for (N, K) in ((256, 128), (128, 256), (256, 36)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    for _ in range(2):
        ops.linear_fwd(x, w, b, y=y)
x = torch.randn(8192, 4096, device=dev); w = torch.randn(4096, 4096, device=dev); y = torch.empty(8192, 4096, device=dev)
for _ in range(2):
    ops.linear_fwd(x, w, None, y=y)
g = torch.randn(M, 128, device=dev); hh = torch.randn(M, 256, device=dev); w2 = torch.randn(128, 256, device=dev)
for _ in range(2):
    ops.linear_wgrad(g, hh)
for _ in range(2):
    ops.linear_dgrad(g, w2, mask=hh)
torch.cuda.synchronize()
