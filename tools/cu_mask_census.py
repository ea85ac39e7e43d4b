"""Which CUs does a CU mask select?  Launches a census kernel (mmft_debug_cu_census) on masked streams and prints, per
mask, the (XCC, SE, CU) histogram of where the workgroups ran."""
import collections, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                'multimodal-fusion-based-pre-routing-timing-prediction-_amd'))
from mmft import lib

dev = torch.device('cuda:0')
n = lib.cu_count(dev)


def census(stream, n_wg=1024):
    out = torch.zeros(n_wg, dtype=torch.int32, device=dev)
    with torch.cuda.stream(stream):
        lib.call('mmft_debug_cu_census', out, n_wg, 2000, 0, stream.cuda_stream)
    stream.synchronize()
    v = out.cpu().numpy().astype('uint32')
    xcc, cu, se = v >> 16, (v >> 8) & 0xf, (v >> 13) & 0x7
    per_xcc = collections.Counter(xcc.tolist())
    cus = {(int(a), int(b), int(c)) for a, b, c in zip(xcc, se, cu)}
    return per_xcc, cus


masks = {'all': range(n), 'low half': range(n // 2), 'even': range(0, n, 2), 'stride 8': range(0, n, 8), 'first 32': range(32),
         'first 8': range(8), 'bits 0-15': range(16), 'bit 0': [0], 'bit 1': [1], 'bit 8': [8], 'bit 32': [32],
         'first 64': range(64), '16 per 32': [i for i in range(n) if i % 32 < 16]}
for name, cus in masks.items():
    m = lib.MaskedStream(dev, cus)
    per_xcc, used = census(m.stream)
    print(f'{name:10s} bits {len(list(cus)):3d} -> distinct (xcc, se, cu) {len(used):3d}; workgroups per XCC {dict(sorted(per_xcc.items()))}')
    if len(used) <= 16:
        print('           ', sorted(used))
    m.close()
