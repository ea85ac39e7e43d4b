"""Development smoke test of the bf16-storage U-Net path: new path vs the per-operator path (both bf16 math mode) and vs
the fp64 oracle, forward and gradients; then a timing of forward + backward at config-B size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multimodal-fusion-based-pre-routing-timing-prediction-_amd'))
import numpy as np, torch
import Unet
from mmft import lib, unet16
from oracle import restatement as R
dev = torch.device('cuda:0')
def cos(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float(a @ b / (a.norm() * b.norm() + 1e-300))
def re(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))
def run(N, H, W, enabled, pooling='max'):
    torch.manual_seed(3)
    net = Unet.UNet(pooling).to(dev); net.set_per_sample_stats(True); net.train()
    x = torch.rand(N, 3, H, W, generator=torch.Generator().manual_seed(5)).to(dev)
    gy = torch.randn(N, 1, H // 2, W // 2, generator=torch.Generator().manual_seed(6)).to(dev)
    unet16.ENABLED = enabled
    with lib.math_mode('bf16'):
        y = net(x)
        y.backward(gy)
    torch.cuda.synchronize()
    return net, x, gy, y.detach(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}, {k: v.clone() for k, v in net.state_dict().items() if 'running' in k}
for (N, H, W, pooling) in ((2, 64, 64, 'max'), (3, 40, 96, 'avg'), (1, 256, 256, 'max')):
    net, x, gy, y1, g1, rs1 = run(N, H, W, True, pooling)
    _, _, _, y0, g0, rs0 = run(N, H, W, False, pooling)
    pc = {k: (v.detach().cpu().double().requires_grad_(True) if v.dtype.is_floating_point and 'running' not in k else v.detach().cpu().clone()) for k, v in Unet.UNet(pooling).state_dict().items()}
    torch.manual_seed(3); ref = Unet.UNet(pooling)
    pc = {k: (v.detach().double().clone().requires_grad_(True) if (v.dtype.is_floating_point and 'running' not in k) else v.clone()) for k, v in ref.state_dict().items()}
    ys = []
    for i in range(N):
        ys.append(R.unet_forward(pc, x[i:i+1].cpu().double(), pooling, update_running=False))
    yo = torch.cat(ys); yo.backward(gy.cpu().double())
    print(f'--- N={N} {H}x{W} {pooling}: forward new-vs-fp64 {re(y1, yo):.3e}  old-vs-fp64 {re(y0, yo):.3e}  new-vs-old {re(y1, y0):.3e}')
    worst = []
    for k in g1:
        worst.append((cos(g1[k], pc[k].grad), cos(g0[k], pc[k].grad), k))
    worst.sort()
    for c1, c0, k in worst[:6]:
        print(f'   grad cos vs fp64: new {c1:.4f} old {c0:.4f}  {k}')
    print('   mean cos new %.4f old %.4f;  running stats new-vs-old %.2e' % (np.mean([w[0] for w in worst]), np.mean([w[1] for w in worst]), max(re(rs1[k], rs0[k]) for k in rs1)))
# timing at config-B size
for enabled in (True, False):
    unet16.ENABLED = enabled
    torch.manual_seed(3)
    net = Unet.UNet('max').to(dev); net.set_per_sample_stats(True); net.train()
    x = torch.rand(8, 3, 256, 256, device=dev); gy = torch.randn(8, 1, 128, 128, device=dev)
    with lib.math_mode('bf16'):
        for _ in range(3):
            y = net(x); y.backward(gy)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20):
            y = net(x); y.backward(gy)
        torch.cuda.synchronize()
        print(f'8x256x256 fwd+bwd eager, u16 path {enabled}: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms (host-inclusive)')
        lib.prof_reset(); lib.prof_enable(True)
        for _ in range(3):
            y = net(x); y.backward(gy)
        torch.cuda.synchronize(); lib.prof_enable(False)
        rows = sorted(lib.prof_report(), key=lambda r: -r['ms'])
        print('   device time per fwd+bwd (instrumented kernels): %.3f ms' % (sum(r['ms'] for r in rows) / 3))
        for r in rows[:14]:
            print('     %-34s %3d launches %7.3f ms  %6.1f us each %7.1f GB/s' % (r['name'][:34], r['launches'] // 3, r['ms'] / 3, r['ms'] / r['launches'] * 1e3, r['bytes'] / (r['ms'] * 1e-3) / 1e9 if r['ms'] else 0))
