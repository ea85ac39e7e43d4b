"""Drop-in `Unet` module: DoubleConv, Down, Up, OutConv, UNet on the MI355X HIP path.

Same class names, constructor / forward signatures, attribute names and state_dict keys as the
reference's src/Unet.py (`inc.double_conv.{0,1,3,4}.*`, `down{1,2,3}.maxpool_conv.1.double_conv.*`,
`up{1,2,3}.up.*`, `up{1,2,3}.conv.double_conv.*`, `outc.conv.0.*`).  The torch modules below only HOLD
the parameters and buffers; forward() never calls them - every op goes to libmmft_hip.so
(NHWC implicit-GEMM convolutions on the fp32 MFMA engine, fused train-mode BatchNorm+ReLU, pooling,
ConvTranspose2d as GEMM + pixel shuffle).  Activations travel between ops as channels_last tensors.

Reference behaviours kept on purpose:
  D3  forward accepts (C,H,W) as well as (N,C,H,W);
  D5  BatchNorm always uses batch statistics (the reference never calls eval()); running stats are
      still updated;
  D10 one pooling module instance is shared by down1/2/3 and outc;
  D12 UNet(pooling, bilinear=True) cannot run in the reference either: up3 then has 16 // 2 = 8 output channels while
      OutConv is built for 16 (src/Unet.py:103-104,108) - "expected input ... to have 16 channels, but got 8".  The Up
      module itself works with bilinear=True and is supported here (align_corners=True x2 kernel); the whole UNet with
      bilinear=True raises the same kind of channel-mismatch error as the reference's.
`per_sample_stats` (extra, default False) makes BatchNorm take statistics per image, which is what
batching several designs needs to stay equal to the reference's one-image-at-a-time loop.
"""
import torch
import torch.nn as nn

from mmft import cnn as C
from mmft import lib as _lib
from mmft import unet16 as _u16


def _pool_mode(pooling):
    if isinstance(pooling, nn.MaxPool2d):
        return C.POOL_MAX
    if isinstance(pooling, nn.AvgPool2d):
        return C.POOL_AVG
    raise NotImplementedError(f'unsupported pooling module {type(pooling).__name__}')


def _channels_last_(conv):
    """Keep the OIHW parameter in [Co][KH][KW][Ci] memory so the kernels read it without a re-layout."""
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)


class DoubleConv(nn.Module):
    """(convolution => [BN] => ReLU) * 2   (src/Unet.py:8-25)"""

    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        if not mid_channels:
            mid_channels = out_channels
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(mid_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True))
        _channels_last_(self.double_conv[0])
        _channels_last_(self.double_conv[3])
        self.per_sample_stats = False
        self.count_batches = True          # UNet bumps all num_batches_tracked counters with one launch instead

    def forward(self, x):
        dc = self.double_conv
        x = C.conv2d(x, dc[0].weight, None, pad=1)
        x = C.bn_relu(x, dc[1], relu=True, per_sample=self.per_sample_stats, count=self.count_batches)
        x = C.conv2d(x, dc[3].weight, None, pad=1)
        return C.bn_relu(x, dc[4], relu=True, per_sample=self.per_sample_stats, count=self.count_batches)


class Down(nn.Module):
    """Downscaling with the shared pooling module then double conv   (src/Unet.py:28-39)"""

    def __init__(self, pooling, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(pooling, DoubleConv(in_channels, out_channels))

    def forward(self, x):
        return self.maxpool_conv[1](C.pool2x2(x, _pool_mode(self.maxpool_conv[0])))


class Up(nn.Module):
    """Upscaling then double conv   (src/Unet.py:42-68)"""

    def __init__(self, in_channels, out_channels, bilinear=True):
        super().__init__()
        self.bilinear = bool(bilinear)
        if bilinear:
            # src/Unet.py:48-51; the module only holds the configuration, forward() goes to the HIP kernel
            self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
            self.conv = DoubleConv(in_channels, out_channels, in_channels // 2)
            return
        self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
        # parameter memory ordered (a, b, co, ci): the GEMM's [4*Co][Ci] weight matrix, no re-layout per step
        w = self.up.weight.data
        self.up.weight.data = w.permute(2, 3, 1, 0).contiguous().permute(3, 2, 0, 1)
        self.conv = DoubleConv(in_channels, out_channels)

    def forward(self, x1, x2):
        if getattr(self, 'bilinear', False):
            return self.conv(C.cat_pad(x2, C.upsample_bilinear2x(x1)))            # src/Unet.py:56-68
        return self.conv(C.up_cat(x1, self.up.weight, self.up.bias, x2))


class OutConv(nn.Module):
    """1x1 conv (bias) -> pool -> ReLU   (src/Unet.py:71-82)"""

    def __init__(self, pooling, in_channels, out_channels):
        super(OutConv, self).__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=1), pooling, nn.ReLU(inplace=True))

    def forward(self, x):
        return C.outconv(x, self.conv[0].weight, self.conv[0].bias, _pool_mode(self.conv[1]))


class UNet(nn.Module):
    def __init__(self, pooling, bilinear=False):
        super(UNet, self).__init__()
        if pooling == 'max':
            pooling_layer = nn.MaxPool2d(2)
        elif pooling == 'avg':
            pooling_layer = nn.AvgPool2d(2)
        else:
            assert False, 'wrong pooling type for layoutnet!'
        self.n_channels = 3
        self.bilinear = bilinear
        self.inc = DoubleConv(3, 16)
        self.down1 = Down(pooling_layer, 16, 32)
        self.down2 = Down(pooling_layer, 32, 64)
        factor = 2 if bilinear else 1
        self.down3 = Down(pooling_layer, 64, 128 // factor)
        self.up1 = Up(128, 64 // factor, bilinear)
        self.up2 = Up(64, 32 // factor, bilinear)
        self.up3 = Up(32, 16 // factor, bilinear)
        self.outc = OutConv(pooling_layer, 16, 1)
        for m in self.modules():
            if isinstance(m, DoubleConv):
                m.count_batches = False

    def _batch_counters(self):
        """The 14 num_batches_tracked counters as ONE int64 vector (the buffers are views of it; rebuilt whenever .to() /
        load replaced the buffer objects)."""
        bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d) and m.num_batches_tracked is not None]
        views = self.__dict__.get('_nbt_views')
        if views is None or len(views) != len(bns) or any(b.num_batches_tracked is not v for b, v in zip(bns, views)):
            flat = torch.stack([b.num_batches_tracked.reshape(()) for b in bns])
            views = [flat[i] for i in range(len(bns))]
            for b, v in zip(bns, views):
                b._buffers['num_batches_tracked'] = v
            self.__dict__['_nbt_flat'], self.__dict__['_nbt_views'] = flat, views
        return self.__dict__['_nbt_flat']

    def _count_batches(self, n):
        """num_batches_tracked += n for all 14 BatchNorm layers with ONE launch."""
        self._batch_counters().add_(n)

    def set_per_sample_stats(self, flag=True):
        """BatchNorm statistics per image instead of per batch (used when several designs are batched)."""
        for m in self.modules():
            if isinstance(m, DoubleConv):
                m.per_sample_stats = bool(flag)
        return self

    def forward(self, x):
        if x.dim() == 3:                       # train() feeds (C,H,W), validate()/test() (1,C,H,W): SURVEY D3
            x = x.unsqueeze(0)
        per_sample = self.inc.per_sample_stats
        if _lib.get_math_mode() == 'bf16' and _u16.supported(self, x):
            # bf16 math mode: the whole network as one autograd node on bf16-STORAGE kernels (mmft/unet16.py); the batch
            # counters ride in its first launch
            return _u16.unet_forward(self, x)
        self._count_batches(x.shape[0] if per_sample else 1)
        x1 = self.inc(x)
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        x = self.up1(x4, x3)
        x = self.up2(x, x2)
        x = self.up3(x, x1)
        return self.outc(x)
