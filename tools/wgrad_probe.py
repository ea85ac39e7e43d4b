"""Probe: u16 conv weight-gradient kernels alone (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multimodal-fusion-based-pre-routing-timing-prediction-_amd'))
import torch
from mmft import lib
dev = torch.device('cuda:0'); BF = torch.bfloat16
lib.set_math_mode('bf16')
for (Ci, Co, h, w) in ((16, 16, 256, 256), (64, 32, 128, 128), (128, 128, 32, 32)):
    N = 8
    x = torch.randn(N, h, w, Ci, device=dev).to(BF); y = torch.randn(N, h, w, Co, device=dev).to(BF)
    dw = torch.empty(Co, 3, 3, Ci, device=dev)
    ws = lib.workspace(dev, lib.query('mmft_u16_conv3x3_wgrad_workspace_bytes', N, h, w, Ci, Co))
    d, s = lib.stream_args(dw)
    for _ in range(3):
        lib.call('mmft_u16_conv3x3_wgrad', x, 0, y, dw, 0, N, h, w, Ci, Co, ws, ws.numel() * 4, d, s)
    torch.cuda.synchronize()
