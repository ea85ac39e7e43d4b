"""Per-layer device times of the bf16-storage U-Net kernels at config-B size (8 x 256 x 256): every 3x3 convolution forward,
input gradient and weight gradient, the BatchNorm / pooling / transposed-convolution kernels, each launched 5 times alone."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multimodal-fusion-based-pre-routing-timing-prediction-_amd'))
import torch
from mmft import lib, ops, unet16
dev = torch.device('cuda:0')
BF = torch.bfloat16
N, H0, W0 = int(os.environ.get('N', 8)), int(os.environ.get('H', 256)), int(os.environ.get('W', 256))
lib.set_math_mode('bf16')
def timed(label, fn, reps=5):
    fn(); torch.cuda.synchronize()
    lib.prof_reset(); lib.prof_enable(True)
    for _ in range(reps): fn()
    torch.cuda.synchronize(); lib.prof_enable(False)
    rows = lib.prof_report()
    tot = sum(r['ms'] for r in rows) / reps * 1e3
    by = sum(r['bytes'] for r in rows) / reps
    print('%-44s %8.1f us  %7.1f GB/s   %s' % (label, tot, by / (tot * 1e-6) / 1e9 if tot else 0, ' + '.join('%s %.1f' % (r['name'], r['ms'] / reps * 1e3) for r in rows)), flush=True)
    return tot
total = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0, 'bn_fwd': 0.0, 'bn_bwd': 0.0}
for i in range(1, 15):
    Ci, Co, l, inp = unet16._CONV[i]
    h, w = H0 >> l, W0 >> l
    wt = (torch.randn(Co, Ci, 3, 3, device=dev) * 0.1).contiguous(memory_format=torch.channels_last)
    buf, table, offs, lanes = unet16.pack_table([unet16.conv_pack_entries('f', wt), unet16.conv_pack_entries('b', wt, backward=True)], dev)
    unet16.pack_run(buf, table, 2, lanes)
    rgb = Ci == 3
    x = torch.randn(N, h, w, Ci, device=dev) if rgb else torch.randn(N, h, w, Ci, device=dev).to(BF)
    y = torch.empty(N, h, w, Co, dtype=BF, device=dev)
    per = ctypes.c_int(0)
    tiles = lib.load().mmft_u16_conv_tiles(N, h, w, ctypes.byref(per))
    st = torch.zeros(tiles, 2, Co, device=dev)
    d, s = lib.stream_args(y)
    total['fwd'] += timed(f'conv{i} {Ci}->{Co} @{h}x{w} fwd', lambda: lib.call('mmft_u16_conv3x3', x, int(rgb), buf, y, st, N, h, w, Ci, Co, d, s))
    if not rgb:
        dx = torch.empty(N, h, w, Ci, dtype=BF, device=dev)
        total['dgrad'] += timed(f'conv{i} {Ci}->{Co} @{h}x{w} dgrad', lambda: lib.call('mmft_u16_conv3x3', y, 0, buf.data_ptr() + offs['b'] * 2, dx, None, N, h, w, Co, Ci, d, s))
    dw = torch.empty(Co, 3, 3, Ci, device=dev)
    ws = lib.workspace(dev, lib.query('mmft_u16_conv3x3_wgrad_workspace_bytes', N, h, w, Ci, Co))
    total['wgrad'] += timed(f'conv{i} {Ci}->{Co} @{h}x{w} wgrad', lambda: lib.call('mmft_u16_conv3x3_wgrad', x, int(rgb), y, dw, 0, N, h, w, Ci, Co, ws, ws.numel() * 4, d, s))
    bnp = torch.rand(5, N, Co, device=dev) + 0.5
    g, b = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
    rm, rv = torch.zeros(Co, device=dev), torch.ones(Co, device=dev)
    a = torch.empty(N, h, w, Co, dtype=BF, device=dev)
    pooled = torch.empty(N, h // 2, w // 2, Co, dtype=BF, device=dev)
    t1 = timed(f'   bn{i} finalize', lambda: lib.call('mmft_u16_bn_finalize', st, per.value, N, Co, h * w, 1e-5, g, b, bnp, d, s))
    t2 = timed(f'   bn{i} apply' + (' + pool' if i in (2, 4, 6) else ''), lambda: lib.call('mmft_u16_bn_apply', y, bnp, a, Co, pooled if i in (2, 4, 6) else None, N, h, w, Co, ops.POOL_MAX, 0.1, rm, rv, d, s))
    total['bn_fwd'] += t1 + t2
    dz = torch.empty_like(y)
    dg, db = torch.zeros(Co, device=dev), torch.zeros(Co, device=dev)
    ws2 = lib.workspace(dev, lib.query('mmft_u16_bn_bwd_workspace_bytes', N, h * w, Co))
    total['bn_bwd'] += timed(f'   bn{i} backward (3 kernels)', lambda: lib.call('mmft_u16_bn_bwd', a, y, bnp, dz, dg, db, 0, N, h * w, Co, ws2, ws2.numel() * 4, d, s))
print({k: round(v, 1) for k, v in total.items()}, 'us; sum', round(sum(total.values()), 1))
