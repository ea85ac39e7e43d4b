"""bf16-STORAGE path of the layout U-Net (bf16 math mode): UNet.forward (src/Unet.py:110-119) as ONE autograd node.

Every activation, pre-activation and activation gradient lives in HBM as bf16 (BASELINE config B: "bf16 storage / fp32
accumulate"); BatchNorm statistics, parameters, parameter gradients and Adam state stay fp32.  What the fp32 path does in
~190 launches per step (cnn.py: one autograd Function per operator) runs here as

    forward   per DoubleConv layer (src/Unet.py:8-25):  convolution (+ BatchNorm partial sums in its epilogue) -> finalize
              -> apply (+ ReLU, + the 2x2 pooling of the following Down block, written straight into the skip half of the
              Up block's concatenation buffer);  ConvTranspose2d writes the other half in place;  OutConv in one kernel
    backward  per layer: BatchNorm backward (partial sums, finalize, apply) -> weight gradient -> input gradient; the
              pooling backward adds the skip connection's gradient in the same pass (no autograd add kernels)

with the 64 / 128-channel stages on the same tile-resident convolution kernels as the narrow ones (csrc/unet16_conv.hip)
and all weights re-packed to MFMA fragment order by one launch per step.  Selected by Unet.UNet.forward when
lib.get_math_mode() == 'bf16' and the input qualifies (`supported`); everything else takes the fp32 operators of cnn.py.
"""
import ctypes
import functools
import math
import weakref

import torch
from torch import nn

from . import gradsink, lib, ops

ENABLED = True          # False: UNet.forward keeps the per-operator fp32-storage path in bf16 mode as well (comparison runs)


def supported(net, x):
    """The fused path covers the reference's own configuration: UNet(pooling, bilinear=False) on an fp32 (N,3,H,W) batch
    whose sides survive three 2x2 poolings and the OutConv kernel's 32-pixel row segments."""
    if not ENABLED or net.bilinear or x.dim() != 4 or x.shape[1] != 3 or x.dtype != torch.float32 or not x.is_cuda:
        return False
    if x.shape[0] > 1 and not net.inc.per_sample_stats:
        return False                    # batch statistics over several images: the fp32 operators handle that
    H, W = x.shape[2], x.shape[3]
    return H % 8 == 0 and W % 32 == 0 and H >= 8


def _params(net):
    """list(net.parameters()), cached: walking the module tree costs ~80 us and the eager callers do it several times per
    call.  (The Parameter OBJECTS are stable; optimizers re-home their .data, which the pointer keys below notice.)"""
    ps = net.__dict__.get('_u16_param_list')
    if ps is None:
        ps = net.__dict__['_u16_param_list'] = list(net.parameters())
    return ps


# (name of the DoubleConv, index of the conv inside it) in forward order; conv i (1-based) = CONVS[i - 1]
def _layers(net):
    cached = net.__dict__.get('_u16_layers')
    if cached is not None:
        return cached
    net.__dict__['_u16_layers'] = out = _layers_uncached(net)
    return out


def _layers_uncached(net):
    dcs = [net.inc, net.down1.maxpool_conv[1], net.down2.maxpool_conv[1], net.down3.maxpool_conv[1],
           net.up1.conv, net.up2.conv, net.up3.conv]
    out = []
    for dc in dcs:
        s = dc.double_conv
        out.append((s[0], s[1]))
        out.append((s[3], s[4]))
    return out                                    # 14 x (Conv2d, BatchNorm2d)


class _PackDesc(ctypes.Structure):
    _fields_ = [('w', ctypes.c_void_p), ('out', ctypes.c_void_p), ('rows', ctypes.c_int), ('K', ctypes.c_int), ('taps', ctypes.c_int),
                ('mode', ctypes.c_int), ('Rsrc', ctypes.c_int), ('Ksrc', ctypes.c_int)]


def _pad16(n):
    return (n + 15) // 16 * 16


def pack_table(entries, device):
    """entries: (name, fp32 source tensor, rows, K, taps, mode, Rsrc, Ksrc) - see mmft_u16_pack_weights in include/mmft.h.
    Returns (bf16 buffer, device descriptor table, {name: element offset}, widest entry in fragment lanes)."""
    assert lib.query('mmft_u16_pack_desc_bytes') == ctypes.sizeof(_PackDesc)
    total = sum(rows * K * taps for _, _, rows, K, taps, _, _, _ in entries)
    buf = torch.empty(total, dtype=torch.bfloat16, device=device)
    descs = (_PackDesc * len(entries))()
    offs, off, lanes = {}, 0, 0
    for j, (name, p, rows, K, taps, mode, Rsrc, Ksrc) in enumerate(entries):
        descs[j] = _PackDesc(p.data_ptr(), buf.data_ptr() + off * 2, rows, K, taps, mode, Rsrc, Ksrc)
        offs[name] = off
        off += rows * K * taps
        lanes = max(lanes, rows * K * taps // 4)
    table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(device)
    return buf, table, offs, lanes


def pack_run(buf, table, n, lanes, counters=None, inc=0):
    """counters: int64 vector bumped by `inc` in the same launch (the BatchNorm layers' num_batches_tracked)."""
    dev, st = lib.stream_args(buf)
    lib.call('mmft_u16_pack_weights', table, n, lanes, counters, counters.numel() if counters is not None else 0, int(inc), dev, st)


def conv_pack_entries(name, weight, backward=False):
    """Pack entries of one Conv2d(k=3) weight stored channels_last: forward fragments, or the flipped / transposed ones of
    the input gradient."""
    Co, Ci = weight.shape[0], weight.shape[1]
    if not ops.is_nhwc(weight):
        raise RuntimeError('unet16: Conv2d weights must be stored channels_last ([Co][3][3][Ci] memory)')
    # reduction channels >= 32: fragments of the 16x16x32 MFMA (mode + 4), as mmft_u16_conv3x3 reads them
    if backward:
        return (name, weight, Ci, Co, 9, 1 + (4 if Co >= 32 else 0), Ci, Co)
    return (name, weight, Co, _pad16(Ci), 9, 0 + (4 if Ci >= 32 else 0), Co, Ci)


def convt_pack_entries(name, weight, backward=False):
    Ci, Co = weight.shape[0], weight.shape[1]
    if not weight.permute(2, 3, 1, 0).is_contiguous():
        raise RuntimeError('unet16: ConvTranspose2d weights must be stored in (a,b,co,ci) memory order')
    if backward:
        return (name, weight, Ci, 4 * Co, 1, 2, Ci, 4 * Co)
    return (name, weight, 4 * Co, Ci, 1, 0, 4 * Co, Ci)


class _Packs:
    """Packed bf16 weights of one UNet: forward and flipped (input-gradient) fragments of the 14 convolutions, forward and
    transposed fragments of the 3 transposed convolutions.  The descriptor table is rebuilt when a parameter's storage
    moved (FlatAdam re-homes the parameters once); the pack itself is ONE launch per forward."""

    def __init__(self, net, device):
        self.key = None
        self.device = device

    def _build(self, net):
        entries = []
        for i, (cv, _) in enumerate(_layers(net)):
            entries.append(conv_pack_entries('f%d' % i, cv.weight))
            if i > 0:                                      # the first layer's input needs no gradient
                entries.append(conv_pack_entries('b%d' % i, cv.weight, backward=True))
        for k, up in enumerate([net.up1.up, net.up2.up, net.up3.up]):
            entries.append(convt_pack_entries('tf%d' % k, up.weight))
            entries.append(convt_pack_entries('tb%d' % k, up.weight, backward=True))
        self.buf, self.descs, self.off, self.lanes = pack_table(entries, self.device)
        self.n = len(entries)

    def ensure(self, net):
        """(Re)build the descriptor table when a parameter's storage moved - an H2D copy, so never inside a capture."""
        key = tuple(p.data_ptr() for p in _params(net))
        if key != self.key:
            self._build(net)
            self.key = key

    def refresh(self, net, counters=None, inc=0):
        self.ensure(net)
        pack_run(self.buf, self.descs, self.n, self.lanes, counters, inc)

    def ptr(self, name):
        return self.buf.data_ptr() + self.off[name] * 2


def _arena(sizes, dtype, device, align=128):
    """One allocation carved into named flat tensors (offsets rounded up to `align` elements)."""
    off, table = 0, {}
    for name, n in sizes:
        table[name] = (off, n)
        off += (n + align - 1) // align * align
    buf = torch.empty(max(off, 1), dtype=dtype, device=device)
    return buf, {k: buf[o:o + n] for k, (o, n) in table.items()}


class _Sink:
    """Where a parameter gradient goes: straight into FlatAdam's flat gradient buffer when the parameter has a sink and
    its memory order is the kernels' output order, else a fresh tensor returned through autograd."""

    def __init__(self, param, mem_shape, to_logical):
        self.rec = gradsink.of(param)
        self.to_logical = to_logical
        view = None
        if self.rec is not None:
            mem, _ = gradsink._memory_order(self.rec[0])
            if mem is not None and mem.numel() == math.prod(mem_shape):
                view = mem.reshape(mem_shape)
        self.direct = view is not None
        if self.direct:
            self.out, self.accumulate = view, not gradsink.fresh(self.rec)
        else:
            self.out, self.accumulate = torch.empty(mem_shape, dtype=torch.float32, device=param.device), False

    def done(self):
        """Call once the producing kernel has been issued; returns what backward() must return for the parameter."""
        if self.direct:
            if not self.accumulate:
                gradsink.taken(self.rec)
            return None
        return self.to_logical(self.out)


def _geometry(N, H, W):
    lv = [(H >> k, W >> k) for k in range(4)]
    P = [N * h * w for h, w in lv]
    return lv, P


# conv i (1-based): (Ci, Co, level, input buffer name)
_CONV = {1: (3, 16, 0, 'x'), 2: (16, 16, 0, 'a1'), 3: (16, 32, 1, 'p1'), 4: (32, 32, 1, 'a3'), 5: (32, 64, 2, 'p2'),
         6: (64, 64, 2, 'a5'), 7: (64, 128, 3, 'p3'), 8: (128, 128, 3, 'a7'), 9: (128, 64, 2, 'cat1'), 10: (64, 64, 2, 'a9'),
         11: (64, 32, 1, 'cat2'), 12: (32, 32, 1, 'a11'), 13: (32, 16, 0, 'cat3'), 14: (16, 16, 0, 'a13')}
# where conv i's activation is written: (buffer, pixel pitch, pooled buffer or None)
_ACT = {1: ('a1', 16, None), 2: ('cat3', 32, 'p1'), 3: ('a3', 32, None), 4: ('cat2', 64, 'p2'), 5: ('a5', 64, None),
        6: ('cat1', 128, 'p3'), 7: ('a7', 128, None), 8: ('a8', 128, None), 9: ('a9', 64, None), 10: ('a10', 64, None),
        11: ('a11', 32, None), 12: ('a12', 32, None), 13: ('a13', 16, None), 14: ('a14', 16, None)}
# transposed convolutions: k -> (Ci, input activation, concatenation buffer, channel offset of its slice, level of the INPUT)
_UP = {0: (128, 'a8', 'cat1', 64, 3), 1: (64, 'a10', 'cat2', 32, 2), 2: (32, 'a12', 'cat3', 16, 1)}


def _fwd_sizes(N, H, W):
    lv, P = _geometry(N, H, W)
    sizes = [('z%d' % i, P[_CONV[i][2]] * _CONV[i][1]) for i in range(1, 15)]
    sizes += [('a1', P[0] * 16), ('cat3', P[0] * 32), ('p1', P[1] * 16), ('a3', P[1] * 32), ('cat2', P[1] * 64), ('p2', P[2] * 32),
              ('a5', P[2] * 64), ('cat1', P[2] * 128), ('p3', P[3] * 64), ('a7', P[3] * 128), ('a8', P[3] * 128), ('a9', P[2] * 64),
              ('a10', P[2] * 64), ('a11', P[1] * 32), ('a12', P[1] * 32), ('a13', P[0] * 16), ('a14', P[0] * 16)]
    per_img = ctypes.c_int(0)
    tiles0 = lib.load().mmft_u16_conv_tiles(N, lv[0][0], lv[0][1], ctypes.byref(per_img))
    fsizes = [('stats', tiles0 * 2 * 128)] + [('bnp%d' % i, 5 * N * _CONV[i][1]) for i in range(1, 15)]
    return sizes, fsizes


def _bwd_sizes(N, H, W):
    lv, P = _geometry(N, H, W)
    sizes = [('g%d' % i, P[_CONV[i][2]] * _CONV[i][1]) for i in range(1, 15)] + [('dz%d' % i, P[_CONV[i][2]] * _CONV[i][1]) for i in range(1, 15)]
    sizes += [('gcat3', P[0] * 32), ('gcat2', P[1] * 64), ('gcat1', P[2] * 128), ('gp1', P[1] * 16), ('gp2', P[2] * 32), ('gp3', P[3] * 64)]
    q = lib.query
    ws_bytes = max([q('mmft_u16_outconv_bwd_workspace_bytes', N, H, W)] +
                   [q('mmft_u16_bn_bwd_workspace_bytes', N, lv[_CONV[i][2]][0] * lv[_CONV[i][2]][1], _CONV[i][1]) for i in range(1, 15)] +
                   [q('mmft_u16_conv3x3_wgrad_workspace_bytes', N, lv[_CONV[i][2]][0], lv[_CONV[i][2]][1], _CONV[i][0], _CONV[i][1])
                    for i in range(1, 15)] +
                   [q('mmft_u16_convt_wgrad_workspace_bytes', N, lv[u[4]][0], lv[u[4]][1], u[0]) for u in _UP.values()])
    return sizes, ws_bytes


BATCH_REDUCE = True      # the 18 weight-gradient slab reductions of a backward pass as ONE launch (mmft_slab_reduce_batch)


@functools.lru_cache(maxsize=16)
def _slab_sizes(N, H, W):
    """fp32 regions that keep every layer's weight-gradient slabs until the batched reduction at the end of the backward pass."""
    lv, _ = _geometry(N, H, W)
    q = lib.query
    sizes = [('sw%d' % i, q('mmft_u16_conv3x3_wgrad_workspace_bytes', N, lv[_CONV[i][2]][0], lv[_CONV[i][2]][1], _CONV[i][0], _CONV[i][1]) // 4)
             for i in range(1, 15)]
    sizes += [('st%d' % k, q('mmft_u16_convt_wgrad_workspace_bytes', N, lv[u[4]][0], lv[u[4]][1], u[0]) // 4) for k, u in _UP.items()]
    sizes += [('so', q('mmft_u16_outconv_bwd_workspace_bytes', N, H, W) // 4)]
    return sizes


def _run_forward(net, xn, pool_mode, packs, T, F, out, geom):
    """Every launch of UNet.forward (src/Unet.py:110-119) on the given buffers, in order; the weight pack comes first."""
    N, H, W = geom
    dev, st = lib.stream_args(out)
    lv, _ = _geometry(N, H, W)
    convs = _layers(net)
    ups = [net.up1.up, net.up2.up, net.up3.up]
    packs.refresh(net, counters=net._batch_counters(), inc=N if net.inc.per_sample_stats else 1)
    per_img = ctypes.c_int(0)
    for i in range(1, 15):
        Ci, Co, l, inp = _CONV[i]
        h, w = lv[l]
        cv, bn = convs[i - 1]
        xin = xn if inp == 'x' else T[inp]
        lib.load().mmft_u16_conv_tiles(N, h, w, ctypes.byref(per_img))
        lib.call('mmft_u16_conv3x3', xin, 1 if inp == 'x' else 0, packs.ptr('f%d' % (i - 1)), T['z%d' % i], F['stats'], N, h, w,
                 Ci, Co, dev, st)
        lib.call('mmft_u16_bn_finalize', F['stats'], per_img.value, N, Co, h * w, float(bn.eps), bn.weight.detach(), bn.bias.detach(),
                 F['bnp%d' % i], dev, st)
        abuf_name, lda, pooled = _ACT[i]
        lib.call('mmft_u16_bn_apply', T['z%d' % i], F['bnp%d' % i], T[abuf_name], lda, T[pooled] if pooled else None, N, h, w, Co,
                 pool_mode, float(bn.momentum), bn.running_mean, bn.running_var, dev, st)
        for k, (uCi, uin, cat, coff, ul) in _UP.items():
            if uin == abuf_name:                        # the Up block's transposed convolution follows this layer
                up = ups[k]
                uh, uw = lv[ul]
                lib.call('mmft_u16_convt_fwd', T[uin], packs.ptr('tf%d' % k), up.bias.detach() if up.bias is not None else None,
                         T[cat].data_ptr() + coff * 2, 2 * (uCi // 2), N, uh, uw, uCi, dev, st)
    oc = net.outc.conv[0]
    lib.call('mmft_u16_outconv_fwd', T['a14'], oc.weight.detach(), oc.bias.detach() if oc.bias is not None else None, out, N, H, W,
             pool_mode, dev, st)


def _make_sinks(net):
    """One _Sink per parameter, in a fixed order: decided BEFORE any launch so that the launch sequence itself has no
    data-dependent host decisions (it is captured and replayed for the eager callers)."""
    sinks = {}
    oc = net.outc.conv[0]
    sinks[id(oc.weight)] = _Sink(oc.weight, (16,), lambda t, p=oc.weight: t.reshape(p.shape))
    if oc.bias is not None:
        sinks[id(oc.bias)] = _Sink(oc.bias, (1,), lambda t, p=oc.bias: t.reshape(p.shape))
    for i, (cv, bn) in enumerate(_layers(net), start=1):
        Ci, Co = _CONV[i][0], _CONV[i][1]
        sinks[id(bn.weight)] = _Sink(bn.weight, (Co,), lambda t: t)
        sinks[id(bn.bias)] = _Sink(bn.bias, (Co,), lambda t: t)
        sinks[id(cv.weight)] = _Sink(cv.weight, (Co, 3, 3, Ci), lambda t: t.permute(0, 3, 1, 2))
    for k, up in enumerate([net.up1.up, net.up2.up, net.up3.up]):
        uCi = _UP[k][0]
        Co = uCi // 2
        sinks[id(up.weight)] = _Sink(up.weight, (4 * Co, uCi), lambda t, Co=Co, uCi=uCi: t.reshape(2, 2, Co, uCi).permute(3, 2, 0, 1))
        if up.bias is not None:
            sinks[id(up.bias)] = _Sink(up.bias, (Co,), lambda t: t)
    return sinks


def _run_backward(net, xn, pool_mode, packs, T, F, g, G, ws, geom, sinks, SL=None):
    """Every launch of the U-Net's backward pass on the given buffers, in order.  SL (named fp32 slab regions, _slab_sizes):
    the weight-gradient kernels leave their slabs there and ONE launch at the end adds all of them up."""
    N, H, W = geom
    dev, st = lib.stream_args(g)
    lv, _ = _geometry(N, H, W)
    convs = _layers(net)
    ups = [net.up1.up, net.up2.up, net.up3.up]
    oc = net.outc.conv[0]
    table = []                   # rows of mmft_slab_reduce_batch: slabs, out, splits, stride, elems, fold, accumulate

    def reduce_later(slabs, out, splits, stride, elems, fold, accumulate):
        table.append([slabs, out.data_ptr(), splits, stride, elems, fold, int(accumulate)])

    s_w, s_b = sinks[id(oc.weight)], sinks.get(id(oc.bias)) if oc.bias is not None else None
    if s_b is not None and s_b.accumulate != s_w.accumulate:
        raise RuntimeError('unet16: OutConv weight / bias sinks out of step')
    if SL is None:
        lib.call('mmft_u16_outconv_bwd', T['a14'], oc.weight.detach(), oc.bias.detach() if oc.bias is not None else None, g, G['g14'],
                 s_w.out, s_b.out if s_b is not None else None, int(s_w.accumulate), N, H, W, pool_mode, ws, ws.numel() * 4, dev, st)
    else:
        so = SL['so']
        lib.call('mmft_u16_outconv_bwd', T['a14'], oc.weight.detach(), oc.bias.detach() if oc.bias is not None else None, g, G['g14'],
                 None, None, 0, N, H, W, pool_mode, so, so.numel() * 4, dev, st)
        ns = lib.query('mmft_u16_outconv_bwd_slabs', N, H, W)
        reduce_later(so.data_ptr(), s_w.out, ns, 17, 16, 1, s_w.accumulate)
        if s_b is not None:
            reduce_later(so.data_ptr() + 64, s_b.out, ns, 17, 1, 1, s_b.accumulate)

    def conv_backward(i):
        Ci, Co, l, inp = _CONV[i]
        h, w = lv[l]
        cv, bn = convs[i - 1]
        s_g, s_bt = sinks[id(bn.weight)], sinks[id(bn.bias)]
        if s_g.accumulate != s_bt.accumulate:
            raise RuntimeError('unet16: BatchNorm weight / bias sinks out of step')
        lib.call('mmft_u16_bn_bwd', G['g%d' % i], T['z%d' % i], F['bnp%d' % i], G['dz%d' % i], s_g.out, s_bt.out, int(s_g.accumulate),
                 N, h * w, Co, ws, ws.numel() * 4, dev, st)
        s_cw = sinks[id(cv.weight)]
        xin = xn if inp == 'x' else T[inp]
        if SL is None:
            lib.call('mmft_u16_conv3x3_wgrad', xin, 1 if inp == 'x' else 0, G['dz%d' % i], s_cw.out, int(s_cw.accumulate), N, h, w, Ci, Co,
                     ws, ws.numel() * 4, dev, st)
        else:
            sw = SL['sw%d' % i]
            lib.call('mmft_u16_conv3x3_wgrad', xin, 1 if inp == 'x' else 0, G['dz%d' % i], None, 0, N, h, w, Ci, Co,
                     sw, sw.numel() * 4, dev, st)
            reduce_later(sw.data_ptr(), s_cw.out, lib.query('mmft_u16_conv3x3_wgrad_slabs', N, h, w, Ci, Co), Co * 9 * Ci, Co * 9 * Ci, 1,
                         s_cw.accumulate)
        if inp == 'x':
            return
        tgt = {'cat1': 'gcat1', 'cat2': 'gcat2', 'cat3': 'gcat3', 'p1': 'gp1', 'p2': 'gp2', 'p3': 'gp3'}.get(inp)
        if tgt is None:
            tgt = 'g%d' % (i - 1)                      # plain activation of the previous convolution
        lib.call('mmft_u16_conv3x3', G['dz%d' % i], 0, packs.ptr('b%d' % (i - 1)), G[tgt], None, N, h, w, Co, Ci, dev, st)

    def up_backward(k, g_target):
        uCi, uin, cat, coff, ul = _UP[k]
        up = ups[k]
        uh, uw = lv[ul]
        Co = uCi // 2
        gslice = G['g' + cat].data_ptr() + coff * 2
        s_uw, s_ub = sinks[id(up.weight)], sinks.get(id(up.bias)) if up.bias is not None else None
        if s_ub is not None and s_ub.accumulate != s_uw.accumulate:
            raise RuntimeError('unet16: ConvTranspose2d weight / bias sinks out of step')
        if SL is None:
            lib.call('mmft_u16_convt_wgrad', T[uin], gslice, 2 * Co, s_uw.out, s_ub.out if s_ub is not None else None, int(s_uw.accumulate),
                     N, uh, uw, uCi, ws, ws.numel() * 4, dev, st)
        else:
            stb = SL['st%d' % k]
            lib.call('mmft_u16_convt_wgrad', T[uin], gslice, 2 * Co, None, None, 0, N, uh, uw, uCi, stb, stb.numel() * 4, dev, st)
            ns, wel = lib.query('mmft_u16_convt_wgrad_slabs', N, uh, uw), 4 * Co * uCi
            reduce_later(stb.data_ptr(), s_uw.out, ns, wel + 4 * Co, wel, 1, s_uw.accumulate)
            if s_ub is not None:
                reduce_later(stb.data_ptr() + 4 * wel, s_ub.out, ns, wel + 4 * Co, Co, 4, s_ub.accumulate)
        lib.call('mmft_u16_convt_dgrad', gslice, 2 * Co, packs.ptr('tb%d' % k), G[g_target], N, uh, uw, uCi, dev, st)

    def pool_backward(cat, C, l, gp, g_target):
        h, w = lv[l]
        lib.call('mmft_u16_pool_bwd', T[cat], 2 * C, G['g' + cat], 2 * C, G[gp], G[g_target], N, h, w, C, pool_mode, dev, st)

    conv_backward(14); conv_backward(13); up_backward(2, 'g12')
    conv_backward(12); conv_backward(11); up_backward(1, 'g10')
    conv_backward(10); conv_backward(9); up_backward(0, 'g8')
    conv_backward(8); conv_backward(7); pool_backward('cat1', 64, 2, 'gp3', 'g6')
    conv_backward(6); conv_backward(5); pool_backward('cat2', 32, 1, 'gp2', 'g4')
    conv_backward(4); conv_backward(3); pool_backward('cat3', 16, 0, 'gp1', 'g2')
    conv_backward(2); conv_backward(1)
    if table:
        lib.call('mmft_slab_reduce_batch', torch.tensor(table, dtype=torch.int64), len(table), dev, st)


REPLAY = True           # eager callers (the per-level drop-in loop): second and later calls replay captured HIP graphs


class _Token:
    __slots__ = ('__weakref__',)


class _Replay:
    """Static buffers and captured HIP graphs of one network on one input geometry, for callers that launch eagerly: the 55
    forward and ~120 backward launches of the U-Net become one graph launch each from the second step on (the per-level
    drop-in loop is bound by host launch time).  Never used while an outer capture is running (the whole-sweep
    step is captured as a whole), while an earlier forward's activations are still waiting for their backward pass, or when
    a parameter has no gradient sink (plain torch optimizers)."""

    def __init__(self, key):
        self.key, self.calls, self.bwd_calls = key, 0, 0
        self.fwd = self.bwd = None
        self.pending = None                 # weak reference to the token of the forward whose backward has not run yet

    def busy(self):
        return self.pending is not None and self.pending() is not None


def _replay_for(net, x, pool_mode):
    if not REPLAY or lib.PROF_ON or torch.cuda.is_current_stream_capturing() or not torch.is_grad_enabled():
        return None                         # (the launch profiler wants to see every kernel: no replay while it is on)
    if any(gradsink.of(p) is None for p in _params(net)):
        return None
    key = (tuple(x.shape), pool_mode, x.device, tuple(p.data_ptr() for p in _params(net)),
           tuple(gradsink.of(p)[0].data_ptr() for p in _params(net)))
    rp = net.__dict__.get('_u16_replay')
    if rp is None or rp.key != key:
        rp = net.__dict__['_u16_replay'] = _Replay(key)
    return None if rp.busy() else rp


class UNet16Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, net, pool_mode, rp, *params):
        N, _, H, W = x.shape
        geom = (N, H, W)
        convs = _layers(net)
        if not all(bn.momentum is not None and bn.affine and bn.track_running_stats for _, bn in convs):
            raise NotImplementedError('unet16: BatchNorm2d without momentum / affine / running statistics is not on the reference path')
        packs = net.__dict__.get('_u16_packs')
        if packs is None or packs.device != x.device:
            packs = net.__dict__['_u16_packs'] = _Packs(net, x.device)
        if rp is not None:
            rp.calls += 1
        if rp is not None and rp.calls >= 2:
            if rp.fwd is None:                              # second call: static buffers, capture
                packs.ensure(net)
                fs, ffs = _fwd_sizes(N, H, W)
                rp.x = torch.empty_like(x)
                rp.xn = ops.empty_nhwc(N, 3, H, W, x.device)
                rp.abuf, rp.T = _arena(fs, torch.bfloat16, x.device)
                rp.fbuf, rp.F = _arena(ffs, torch.float32, x.device, align=4)
                rp.out = torch.empty((N, 1, H // 2, W // 2), dtype=torch.float32, device=x.device)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                    dev, st = lib.stream_args(rp.x)
                    lib.call('mmft_nchw_to_nhwc', rp.x, rp.xn, N, 3, H, W, 3, dev, st)
                    _run_forward(net, rp.xn, pool_mode, packs, rp.T, rp.F, rp.out, geom)
                rp.fwd = graph
            # the static input copy is skipped when the caller hands in the same, unmodified tensor as last time (the training
            # loop's resident images): its layout pass then runs on stale-but-identical data
            # (the tensor object is kept: a new tensor could otherwise reuse the freed address with the same version)
            if rp.__dict__.get('x_ref') is not x or rp.x_version != x._version:
                rp.x.copy_(x)
                rp.x_ref, rp.x_version = x, x._version
            rp.fwd.replay()
            tok = _Token()
            rp.pending = weakref.ref(tok)
            ctx.net, ctx.pool_mode, ctx.geom = net, pool_mode, geom
            ctx.keep = (rp.xn, rp.abuf, rp.T, rp.fbuf, rp.F, packs)
            ctx.replay, ctx.token = rp, tok
            return rp.out.clone()
        xn = ops.to_nhwc(x)                                 # fp32 [N][H][W][3]
        fs, ffs = _fwd_sizes(N, H, W)
        abuf, T = _arena(fs, torch.bfloat16, x.device)
        fbuf, F = _arena(ffs, torch.float32, x.device, align=4)
        out = torch.empty((N, 1, H // 2, W // 2), dtype=torch.float32, device=x.device)
        _run_forward(net, xn, pool_mode, packs, T, F, out, geom)
        ctx.net, ctx.pool_mode, ctx.geom = net, pool_mode, geom
        ctx.keep = (xn, abuf, T, fbuf, F, packs)
        ctx.replay, ctx.token = None, None
        return out

    @staticmethod
    def backward(ctx, gout):
        net, pool_mode, geom = ctx.net, ctx.pool_mode, ctx.geom
        N, H, W = geom
        xn, _abuf, T, _fbuf, F, packs = ctx.keep
        g = gout if gout.is_contiguous() else gout.contiguous()
        sinks = _make_sinks(net)
        rp = ctx.replay
        fresh = all(s.direct and not s.accumulate for s in sinks.values())
        if rp is not None and fresh and not torch.cuda.is_current_stream_capturing():
            rp.bwd_calls += 1
            if rp.bwd_calls >= 2:
                if rp.bwd is None:                          # second backward: static gradient buffers, capture
                    gs, ws_bytes = _bwd_sizes(N, H, W)
                    rp.g = torch.empty_like(g)
                    rp.gbuf, rp.G = _arena(gs, torch.bfloat16, g.device)
                    rp.sbuf, rp.SL = _arena(_slab_sizes(N, H, W), torch.float32, g.device, align=4) if BATCH_REDUCE else (None, None)
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                        ws = lib.workspace(g.device, ws_bytes)
                        _run_backward(net, xn, pool_mode, packs, T, F, rp.g, rp.G, ws, geom, sinks, rp.SL)
                    rp.bwd = graph
                rp.g.copy_(g)
                rp.bwd.replay()
                for s in sinks.values():
                    s.done()
                rp.pending = None
                return (None, None, None, None) + (None,) * len(_params(net))
        gs, ws_bytes = _bwd_sizes(N, H, W)
        _gbuf, G = _arena(gs, torch.bfloat16, g.device)
        ws = lib.workspace(g.device, ws_bytes)
        _sbuf, SL = _arena(_slab_sizes(N, H, W), torch.float32, g.device, align=4) if BATCH_REDUCE else (None, None)
        _run_backward(net, xn, pool_mode, packs, T, F, g, G, ws, geom, sinks, SL)
        grads = {k: s.done() for k, s in sinks.items()}
        if rp is not None:
            rp.pending = None
        return (None, None, None, None) + tuple(grads.get(id(p)) for p in _params(net))


def unet_forward(net, x):
    """UNet.forward (src/Unet.py:110-119) on the bf16-storage kernels; x: fp32 (N,3,H,W)."""
    pooling = net.down1.maxpool_conv[0]
    pool_mode = ops.POOL_MAX if isinstance(pooling, nn.MaxPool2d) else ops.POOL_AVG
    # the replay decision needs the caller's grad mode (inside Function.forward it is always off)
    return UNet16Fn.apply(x, net, pool_mode, _replay_for(net, x, pool_mode), *_params(net))
