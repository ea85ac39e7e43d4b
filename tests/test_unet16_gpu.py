"""bf16-STORAGE kernels of the layout U-Net (csrc/unet16_*.hip, mmft/unet16.py; BASELINE config B "bf16 storage / fp32
accumulate") against fp64 math, through the C ABI.

Two kinds of checks, as in test_bf16_gpu.py:
  * operands that bf16 represents exactly and whose products / sums fp32 holds exactly: the kernels must reproduce the
    fp64 result ROUNDED TO bf16 bit for bit (pins every fragment layout, the flipped / transposed weight packs, the tile
    masks of partial tiles, the channel-slice pitches);
  * arbitrary data: stated tolerances - one bf16 rounding of the stored result (2^-8 relative) on element-wise kernels,
    2e-2 of the result's scale on contractions over bf16-rounded operands.
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import rel_err
from mmft import lib, ops, unet16

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def grid_vals(shape, seed, step, lim):
    """Multiples of `step` in [-lim, lim]: exact in bf16 when lim / step <= 128."""
    g = torch.Generator().manual_seed(seed)
    n = int(round(lim / step))
    return torch.randint(-n, n + 1, shape, generator=g).double() * step


def nhwc(t):
    """(N,C,H,W) fp64 -> contiguous [N][H][W][C]."""
    return t.permute(0, 2, 3, 1).contiguous()


def conv_weight_dev(w, dev):
    """(Co,Ci,3,3) fp64 -> fp32 parameter stored channels_last, as Unet.DoubleConv keeps it."""
    return w.float().to(dev).contiguous(memory_format=torch.channels_last)


def run_conv(x_nhwc_dev, rgb, w_dev, backward, N, H, W, Ci, Co, stats=True, dev=None):
    """mmft_u16_conv3x3 with a freshly packed weight; returns (bf16 [N,H,W,Co], stats [tiles,2,Co] or None)."""
    buf, table, offs, lanes = unet16.pack_table([unet16.conv_pack_entries('w', w_dev, backward=backward)], dev)
    unet16.pack_run(buf, table, 1, lanes)
    y = torch.empty((N, H, W, Co), dtype=BF, device=dev)
    per = ctypes.c_int(0)
    tiles = lib.load().mmft_u16_conv_tiles(N, H, W, ctypes.byref(per))
    st = torch.zeros((tiles, 2, Co), dtype=torch.float32, device=dev) if stats else None
    d, s = lib.stream_args(y)
    lib.call('mmft_u16_conv3x3', x_nhwc_dev, int(rgb), buf, y, st, N, H, W, Ci, Co, d, s)
    return y, st, per.value


@pytest.mark.parametrize('Ci,Co,N,H,W', [(3, 16, 2, 20, 72), (16, 16, 2, 8, 64), (16, 32, 1, 12, 96), (32, 32, 2, 16, 32), (32, 64, 1, 9, 40),
                                         (64, 64, 2, 8, 32), (64, 128, 1, 8, 32), (128, 128, 2, 4, 32), (128, 64, 1, 16, 64),
                                         (64, 32, 1, 10, 70), (32, 16, 1, 6, 130)])
def test_conv3x3_forward_exact_and_stats(dev, Ci, Co, N, H, W):
    """Forward convolution on representable operands = fp64 result rounded to bf16, bit for bit (all channel counts of the
    network, both tile shapes, partial tiles); per-tile statistics = sums of the stored values."""
    x = grid_vals((N, Ci, H, W), 1, 0.125, 2.0)
    w = grid_vals((Co, Ci, 3, 3), 2, 0.0625, 1.0)
    ref = torch.nn.functional.conv2d(x, w, padding=1)
    rgb = Ci == 3
    xd = nhwc(x).float().to(dev) if rgb else nhwc(x).to(BF).to(dev)
    y, st, per = run_conv(xd, rgb, conv_weight_dev(w, dev), False, N, H, W, Ci, Co, dev=dev)
    want = nhwc(ref).to(BF)
    assert torch.equal(y.cpu(), want)
    yd = y.double().cpu()
    tot = st.double().cpu().reshape(N, per, 2, Co).sum(1)
    assert rel_err(tot[:, 0], yd.sum((1, 2))) < 1e-5
    assert rel_err(tot[:, 1], (yd * yd).sum((1, 2))) < 1e-5


@pytest.mark.parametrize('Ci,Co,N,H,W', [(16, 16, 1, 8, 64), (16, 32, 1, 12, 96), (32, 64, 2, 8, 32), (64, 64, 1, 8, 32), (64, 128, 1, 8, 40),
                                         (128, 128, 1, 4, 32), (128, 64, 1, 8, 64), (32, 16, 1, 6, 66)])
def test_conv3x3_input_gradient_exact(dev, Ci, Co, N, H, W):
    """dx = conv(dy, flipped / transposed pack) of a Ci -> Co layer = autograd's fp64 input gradient rounded to bf16."""
    w = grid_vals((Co, Ci, 3, 3), 3, 0.0625, 1.0)
    dy = grid_vals((N, Co, H, W), 4, 0.125, 2.0)
    x = torch.zeros((N, Ci, H, W), dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x, w, padding=1).backward(dy)
    dx, _, _ = run_conv(nhwc(dy).to(BF).to(dev), False, conv_weight_dev(w, dev), True, N, H, W, Co, Ci, stats=False, dev=dev)
    assert torch.equal(dx.cpu(), nhwc(x.grad).to(BF))


@pytest.mark.parametrize('Ci,Co,N,H,W', [(3, 16, 2, 12, 72), (16, 16, 2, 8, 64), (16, 32, 1, 12, 96), (32, 16, 1, 8, 64), (32, 32, 2, 16, 32),
                                         (32, 64, 1, 9, 40), (64, 64, 2, 8, 32), (64, 128, 1, 8, 32), (128, 128, 2, 4, 32), (128, 64, 1, 8, 64)])
@pytest.mark.parametrize('representable', [True, False])
def test_conv3x3_weight_gradient(dev, Ci, Co, N, H, W, representable):
    if representable:
        x, dy = grid_vals((N, Ci, H, W), 5, 0.25, 1.0), grid_vals((N, Co, H, W), 6, 0.25, 1.0)
    else:
        g = torch.Generator().manual_seed(7)
        x, dy = torch.randn((N, Ci, H, W), generator=g).double(), torch.randn((N, Co, H, W), generator=g).double()
    rgb = Ci == 3
    xb = x if rgb else x.to(BF).double()
    dyb = dy.to(BF).double()
    w = torch.zeros((Co, Ci, 3, 3), dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(xb.float().to(BF).double() if rgb else xb, w, padding=1).backward(dyb)
    xd = nhwc(x).float().to(dev) if rgb else nhwc(x).to(BF).to(dev)
    dw = torch.full((Co, 3, 3, Ci), 3.0, dtype=torch.float32, device=dev)
    nbytes = lib.query('mmft_u16_conv3x3_wgrad_workspace_bytes', N, H, W, Ci, Co)
    ws = lib.workspace(dev, nbytes)
    d, s = lib.stream_args(dw)
    lib.call('mmft_u16_conv3x3_wgrad', xd, int(rgb), nhwc(dy).to(BF).to(dev), dw, 0, N, H, W, Ci, Co, ws, ws.numel() * 4, d, s)
    want = w.grad.permute(0, 2, 3, 1)
    assert rel_err(dw, want) < (2e-6 if representable else 2e-2)
    lib.call('mmft_u16_conv3x3_wgrad', xd, int(rgb), nhwc(dy).to(BF).to(dev), dw, 1, N, H, W, Ci, Co, ws, ws.numel() * 4, d, s)
    assert rel_err(dw, 2 * want) < (2e-6 if representable else 2e-2)            # accumulate = 1 adds into dw


def _bnp(N, C, seed, dev):
    g = torch.Generator().manual_seed(seed)
    mean, var = torch.randn((N, C), generator=g) * 0.5, torch.rand((N, C), generator=g) + 0.2
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = gamma * invstd
    shift = beta - mean * scale
    return torch.stack([mean, invstd, var, scale, shift]).to(dev).contiguous(), gamma, beta


def test_bn_finalize_from_tile_statistics(dev):
    """Tile partial sums -> mean / invstd / scale / shift per (image, channel) and N sequential momentum updates."""
    N, C, tiles, count = 3, 32, 37, 1234
    g = torch.Generator().manual_seed(1)
    st = torch.randn((N * tiles, 2, C), generator=g).double()
    st[:, 1] = st[:, 1].abs() * 40 + 5                                     # sums of squares dominate: positive variances
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    bnp = torch.empty((5, N, C), dtype=torch.float32, device=dev)
    rmd, rvd = rm.clone().to(dev), rv.clone().to(dev)
    d, s = lib.stream_args(bnp)
    lib.call('mmft_u16_bn_finalize', st.float().to(dev), tiles, N, C, count, 1e-5, gamma.to(dev), beta.to(dev), bnp, d, s)
    H, W = 2, count // 2                                                   # an apply launch carries the running-statistics update
    z = torch.zeros((N, H, W, C), dtype=BF, device=dev)
    lib.call('mmft_u16_bn_apply', z, bnp, torch.empty_like(z), C, None, N, H, W, C, ops.POOL_MAX, 0.1, rmd, rvd, d, s)
    st32 = st.float().double().reshape(N, tiles, 2, C).sum(1)
    mean = st32[:, 0] / count
    var = (st32[:, 1] / count - mean * mean).clamp_min(0)
    invstd = 1 / torch.sqrt(var + 1e-5)
    assert rel_err(bnp[0], mean) < 1e-6 and rel_err(bnp[1], invstd) < 1e-6 and rel_err(bnp[2], var) < 1e-6
    assert rel_err(bnp[3], gamma.double() * invstd) < 1e-6 and rel_err(bnp[4], beta.double() - mean * gamma.double() * invstd) < 1e-5
    rmo, rvo = rm.double(), rv.double()
    for i in range(N):
        rmo = 0.9 * rmo + 0.1 * mean[i]
        rvo = 0.9 * rvo + 0.1 * var[i] * count / (count - 1)
    assert rel_err(rmd, rmo) < 1e-6 and rel_err(rvd, rvo) < 1e-6


@pytest.mark.parametrize('C,lda,pool', [(16, 16, None), (16, 32, 'max'), (32, 64, 'avg'), (64, 128, 'max'), (128, 128, None)])
def test_bn_apply_relu_pool(dev, C, lda, pool):
    """a = bf16(relu(z * scale + shift)) written into a channel slice (pitch lda); pooled = 2x2 pool of the STORED values."""
    N, H, W = 2, 12, 40
    g = torch.Generator().manual_seed(2)
    z = torch.randn((N, H, W, C), generator=g).to(BF)
    bnp, _, _ = _bnp(N, C, 3, dev)
    a = torch.full((N, H, W, lda), 7.0, dtype=BF, device=dev)
    pooled = torch.empty((N, H // 2, W // 2, C), dtype=BF, device=dev) if pool else None
    mode = ops.POOL_MAX if pool != 'avg' else ops.POOL_AVG
    d, s = lib.stream_args(a)
    lib.call('mmft_u16_bn_apply', z.to(dev), bnp, a, lda, pooled, N, H, W, C, mode, 0.1, None, None, d, s)
    sc, sh = bnp[3].double().cpu(), bnp[4].double().cpu()
    ref = torch.relu(z.double() * sc[:, None, None, :] + sh[:, None, None, :])
    got = a.cpu()[..., :C].double()
    assert float(((got - ref).abs() - (2.0 ** -8) * ref.abs()).max()) < 1e-6          # one bf16 rounding of the result
    assert bool((a.cpu()[..., C:].float() == 7.0).all())                              # the rest of the pitch is untouched
    if pool:
        win = got.reshape(N, H // 2, 2, W // 2, 2, C)
        want = win.amax((2, 4)) if pool == 'max' else win.mean((2, 4))
        if pool == 'max':
            assert torch.equal(pooled.cpu().double(), want)
        else:
            assert float(((pooled.cpu().double() - want).abs() - (2.0 ** -8) * want.abs()).max()) < 1e-6


@pytest.mark.parametrize('C,P', [(16, 5000), (32, 777), (128, 300)])
def test_bn_backward(dev, C, P):
    """dz, dgamma, dbeta of relu(bn(z)) per image against fp64 (mask from the same fma the forward used)."""
    N = 3
    g = torch.Generator().manual_seed(4)
    z, gy = torch.randn((N, P, C), generator=g).to(BF), torch.randn((N, P, C), generator=g).to(BF)
    # statistics consistent with z (the backward formula assumes mean / invstd are those of z)
    zd = z.double()
    mean, var = zd.mean(1), zd.var(1, unbiased=False)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).double(), (torch.randn(C, generator=g) * 0.3).double()
    invstd = 1 / torch.sqrt(var + 1e-5)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    bnp = torch.stack([mean, invstd, var, scale, shift]).float().to(dev).contiguous()
    dz = torch.empty((N, P, C), dtype=BF, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ws = lib.workspace(dev, lib.query('mmft_u16_bn_bwd_workspace_bytes', N, P, C))
    d, s = lib.stream_args(dz)
    lib.call('mmft_u16_bn_bwd', gy.to(dev), z.to(dev), bnp, dz, dg, db, 0, N, P, C, ws, ws.numel() * 4, d, s)
    pre = torch.addcmul(bnp[4].cpu()[:, None, :], z.float(), bnp[3].cpu()[:, None, :])     # fp32 fma, as the kernels
    gm = gy.double() * (pre > 0)
    xh = (zd - mean[:, None, :]) * invstd[:, None, :]
    ref = scale[:, None, :] * (gm - gm.mean(1, keepdim=True) - xh * (gm * xh).mean(1, keepdim=True))
    assert rel_err(dz, ref) < 1e-2                                        # bf16 rounding of the stored gradient
    assert rel_err(dg, (gm * xh).sum((0, 1))) < 1e-4 and rel_err(db, gm.sum((0, 1))) < 1e-4
    lib.call('mmft_u16_bn_bwd', gy.to(dev), z.to(dev), bnp, dz, dg, db, 1, N, P, C, ws, ws.numel() * 4, d, s)
    assert rel_err(dg, 2 * (gm * xh).sum((0, 1))) < 1e-4                 # accumulate


@pytest.mark.parametrize('C,pool', [(16, 'max'), (32, 'avg'), (64, 'max')])
def test_pool_backward_with_skip_add(dev, C, pool):
    """g = gskip (a slice of the concatenation's gradient) + the pooled gradient routed to torch's argmax (first maximum)."""
    N, H, W = 2, 8, 24
    g = torch.Generator().manual_seed(5)
    a = torch.relu(torch.randn((N, H, W, C), generator=g)).to(BF)         # many ties at zero, as after a ReLU
    cat = torch.zeros((N, H, W, 2 * C), dtype=BF)
    cat[..., :C] = a
    gcat = torch.randn((N, H, W, 2 * C), generator=g).to(BF)
    gp = torch.randn((N, H // 2, W // 2, C), generator=g).to(BF)
    out = torch.empty((N, H, W, C), dtype=BF, device=dev)
    mode = ops.POOL_MAX if pool == 'max' else ops.POOL_AVG
    d, s = lib.stream_args(out)
    lib.call('mmft_u16_pool_bwd', cat.to(dev), 2 * C, gcat.to(dev), 2 * C, gp.to(dev), out, N, H, W, C, mode, d, s)
    an = a.double().permute(0, 3, 1, 2).requires_grad_(True)
    pf = torch.nn.functional.max_pool2d if pool == 'max' else torch.nn.functional.avg_pool2d
    pf(an, 2).backward(gp.double().permute(0, 3, 1, 2))
    want = (gcat[..., :C].double() + an.grad.permute(0, 2, 3, 1)).to(BF)
    assert torch.equal(out.cpu(), want)


@pytest.mark.parametrize('Ci', [32, 64, 128])
def test_conv_transpose_forward_dgrad_wgrad(dev, Ci):
    """ConvTranspose2d(Ci, Ci / 2, 2, 2) into / out of a channel slice: exact on representable operands."""
    Co, N, h, w = Ci // 2, 2, 5, 9
    x = grid_vals((N, Ci, h, w), 8, 0.125, 2.0)
    wt = grid_vals((Ci, Co, 2, 2), 9, 0.0625, 1.0)
    bias = grid_vals((Co,), 10, 0.25, 1.0)
    ref = torch.nn.functional.conv_transpose2d(x, wt, bias, stride=2)
    wd = wt.float().to(dev).permute(2, 3, 1, 0).contiguous().permute(3, 2, 0, 1)      # (a,b,co,ci) memory, as Unet.Up keeps it
    buf, table, offs, lanes = unet16.pack_table([unet16.convt_pack_entries('f', wd), unet16.convt_pack_entries('b', wd, backward=True)], dev)
    unet16.pack_run(buf, table, 2, lanes)
    cat = torch.full((N, 2 * h, 2 * w, 2 * Co), 5.0, dtype=BF, device=dev)
    d, s = lib.stream_args(cat)
    lib.call('mmft_u16_convt_fwd', nhwc(x).to(BF).to(dev), buf.data_ptr() + offs['f'] * 2, bias.float().to(dev), cat.data_ptr() + Co * 2, 2 * Co,
             N, h, w, Ci, d, s)
    assert torch.equal(cat.cpu()[..., Co:], nhwc(ref).to(BF))
    assert bool((cat.cpu()[..., :Co].float() == 5.0).all())
    # backward: gradient read from the slice
    gy = grid_vals((N, Co, 2 * h, 2 * w), 11, 0.125, 1.0)
    gcat = torch.zeros((N, 2 * h, 2 * w, 2 * Co), dtype=BF)
    gcat[..., Co:] = nhwc(gy).to(BF)
    gcat = gcat.to(dev)
    xa = x.clone().requires_grad_(True)
    wa = wt.clone().requires_grad_(True)
    ba = bias.clone().requires_grad_(True)
    torch.nn.functional.conv_transpose2d(xa, wa, ba, stride=2).backward(gy)
    dx = torch.empty((N, h, w, Ci), dtype=BF, device=dev)
    lib.call('mmft_u16_convt_dgrad', gcat.data_ptr() + Co * 2, 2 * Co, buf.data_ptr() + offs['b'] * 2, dx, N, h, w, Ci, d, s)
    assert torch.equal(dx.cpu(), nhwc(xa.grad).to(BF))
    dw, db = torch.zeros((4 * Co, Ci), device=dev), torch.zeros(Co, device=dev)
    ws = lib.workspace(dev, lib.query('mmft_u16_convt_wgrad_workspace_bytes', N, h, w, Ci))
    lib.call('mmft_u16_convt_wgrad', nhwc(x).to(BF).to(dev), gcat.data_ptr() + Co * 2, 2 * Co, dw, db, 0, N, h, w, Ci, ws, ws.numel() * 4, d, s)
    assert rel_err(dw.reshape(2, 2, Co, Ci).permute(3, 2, 0, 1), wa.grad) < 2e-6
    assert rel_err(db, ba.grad) < 2e-6


@pytest.mark.parametrize('pool', ['max', 'avg'])
def test_outconv_bf16_input(dev, pool):
    N, H, W = 2, 8, 64
    g = torch.Generator().manual_seed(12)
    x = torch.randn((N, H, W, 16), generator=g).to(BF)
    w, b = torch.randn(16, generator=g), torch.randn(1, generator=g)
    gout = torch.randn((N, H // 2, W // 2), generator=g)
    mode = ops.POOL_MAX if pool == 'max' else ops.POOL_AVG
    out = torch.empty((N, H // 2, W // 2), device=dev)
    d, s = lib.stream_args(out)
    lib.call('mmft_u16_outconv_fwd', x.to(dev), w.to(dev), b.to(dev), out, N, H, W, mode, d, s)
    xa = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wa, ba = w.double().reshape(1, 16, 1, 1).requires_grad_(True), b.double().requires_grad_(True)
    pf = torch.nn.functional.max_pool2d if pool == 'max' else torch.nn.functional.avg_pool2d
    ref = torch.relu(pf(torch.nn.functional.conv2d(xa, wa, ba), 2))
    assert rel_err(out, ref.squeeze(1)) < 1e-5
    ref.backward(gout.double().unsqueeze(1))
    dx = torch.empty((N, H, W, 16), dtype=BF, device=dev)
    dw, db = torch.zeros(16, device=dev), torch.zeros(1, device=dev)
    ws = lib.workspace(dev, lib.query('mmft_u16_outconv_bwd_workspace_bytes', N, H, W))
    lib.call('mmft_u16_outconv_bwd', x.to(dev), w.to(dev), b.to(dev), gout.to(dev), dx, dw, db, 0, N, H, W, mode, ws, ws.numel() * 4, d, s)
    assert rel_err(dx, xa.grad.permute(0, 2, 3, 1)) < 1e-2
    assert rel_err(dw, wa.grad.reshape(16)) < 1e-5 and rel_err(db, ba.grad) < 1e-5


def _cos(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


@pytest.mark.parametrize('N,H,W,pooling', [(2, 64, 64, 'max'), (3, 40, 96, 'avg')])
def test_unet_module_bf16_storage_vs_fp64_oracle(dev, N, H, W, pooling):
    """UNet.forward + backward on the bf16-storage path (bf16 math mode) against the fp64 oracle run image by image: output
    within 5e-2 of its scale, gradient directions cos >= 0.8 per tensor and >= 0.93 on average (14 chained BatchNorm layers
    on bf16 activations), running statistics within 1e-2; two runs are bitwise equal; the per-operator path of the same
    math mode (unet16.ENABLED = False) is no closer to fp64 than 2.5x."""
    import Unet
    from oracle import restatement as R

    def run(enabled):
        torch.manual_seed(3)
        net = Unet.UNet(pooling).to(dev)
        net.set_per_sample_stats(True)
        net.train()
        x = torch.rand(N, 3, H, W, generator=torch.Generator().manual_seed(5)).to(dev)
        gy = torch.randn(N, 1, H // 2, W // 2, generator=torch.Generator().manual_seed(6)).to(dev)
        unet16.ENABLED = enabled
        try:
            with lib.math_mode('bf16'):
                lib.prof_reset()
                lib.prof_enable(True)
                y = net(x)
                y.backward(gy)
                torch.cuda.synchronize()
                lib.prof_enable(False)
                names = {r['name'].split('<')[0] for r in lib.prof_report()}
        finally:
            unet16.ENABLED = True
        return net, x, gy, y.detach(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}, names
    net, x, gy, y1, g1, names = run(True)
    assert {'u16_conv3x3_kernel', 'u16_conv3x3_wgrad_kernel', 'u16_bn_apply_pool_kernel', 'u16_convt_fwd_kernel'} <= names, names
    _, _, _, y2, g2, _ = run(True)
    assert torch.equal(y1, y2) and all(torch.equal(g1[k], g2[k]) for k in g1)                 # no atomics anywhere
    net0, _, _, y0, g0, names0 = run(False)
    assert not any(n.startswith('u16_') for n in names0)
    torch.manual_seed(3)
    ref = Unet.UNet(pooling)
    pc = {k: (v.detach().double().clone().requires_grad_('running' not in k) if v.dtype.is_floating_point else v.clone())
          for k, v in ref.state_dict().items()}
    yo = torch.cat([R.unet_forward(pc, x[i:i + 1].cpu().double(), pooling, update_running=True) for i in range(N)])
    yo.backward(gy.cpu().double())
    e1, e0 = rel_err(y1, yo.detach()), rel_err(y0, yo.detach())
    assert e1 < 5e-2 and e1 < 2.5 * e0 + 1e-3, (e1, e0)
    cos = {k: _cos(g1[k], pc[k].grad) for k in g1}
    assert min(cos.values()) > 0.8 and np.mean(list(cos.values())) > 0.93, sorted(cos.items(), key=lambda kv: kv[1])[:4]
    for k, v in net.state_dict().items():
        if 'running' in k:
            assert rel_err(v, pc[k]) < 1e-2, k
        if 'num_batches' in k:
            assert int(v) == N


def test_slab_reduce_batch(dev):
    """mmft_slab_reduce_batch: several slab reductions (vector and scalar forms, strided slabs that carry two results, a folded
    bias, accumulate on / off) in one launch; integer-valued floats, so every summation order gives the same bits."""
    gen = torch.Generator().manual_seed(5)
    cases = [dict(splits=512, elems=2304, stride=2304, fold=1, off=0, acc=0),
             dict(splits=37, elems=4 * 64 * 128, stride=4 * 64 * 128 + 4 * 64, fold=1, off=0, acc=1),
             dict(splits=37, elems=64, stride=4 * 64 * 128 + 4 * 64, fold=4, off=4 * 64 * 128, acc=0),
             dict(splits=1024, elems=16, stride=17, fold=1, off=0, acc=1),
             dict(splits=1024, elems=1, stride=17, fold=1, off=16, acc=0),
             dict(splits=3, elems=10, stride=12, fold=1, off=0, acc=0)]
    bufs, rows, want, outs = {}, [], [], []
    for c in cases:
        key = (c['splits'], c['stride'])
        if key not in bufs:
            bufs[key] = torch.randint(-8, 9, (c['splits'], c['stride']), generator=gen).float().to(dev)
        sl = bufs[key]
        out = torch.randint(-8, 9, (c['elems'],), generator=gen).float().to(dev)
        ref = sl[:, c['off']:c['off'] + c['fold'] * c['elems']].reshape(c['splits'], c['fold'], c['elems']).sum((0, 1))
        want.append(ref + out if c['acc'] else ref)
        outs.append(out)
        rows.append([sl.data_ptr() + 4 * c['off'], out.data_ptr(), c['splits'], c['stride'], c['elems'], c['fold'], c['acc']])
    d, st = lib.stream_args(outs[0])
    lib.call('mmft_slab_reduce_batch', torch.tensor(rows, dtype=torch.int64), len(rows), d, st)
    torch.cuda.synchronize()
    for o, w in zip(outs, want):
        assert torch.equal(o, w)
    with pytest.raises(RuntimeError):
        lib.call('mmft_slab_reduce_batch', torch.tensor(rows * 5, dtype=torch.int64), len(rows) * 5, d, st)


def test_batched_reduce_equals_per_layer_reduce(dev):
    """UNet backward with the weight-gradient slabs reduced by one launch at the end against the per-layer reductions: the same
    slabs; the convolution layers add them in the same order (bitwise), the transposed convolutions and OutConv in another."""
    import Unet
    torch.manual_seed(3)
    net = Unet.UNet('max').to(dev)
    net.set_per_sample_stats(True)
    net.train()
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(5)).to(dev)
    res = []
    for batch in (True, False):
        unet16.BATCH_REDUCE = batch
        try:
            for p in net.parameters():
                p.grad = None
            with lib.math_mode('bf16'):
                lib.prof_reset()
                lib.prof_enable(True)
                out = net(x)
                (out * out).sum().backward()
                torch.cuda.synchronize()
                lib.prof_enable(False)
                names = {r['name'].split('<')[0] for r in lib.prof_report()}
            assert ('slab_reduce_batch_kernel' in names) == batch and ('slab_reduce_kernel' in names) != batch, names
            res.append({k: p.grad.clone() for k, p in net.named_parameters()})
        finally:
            unet16.BATCH_REDUCE = True
    for k in res[0]:
        a, b = res[0][k], res[1][k]
        if a.dim() == 4 and a.shape[-1] == 3:
            assert torch.equal(a, b), k
        else:
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-7, k
