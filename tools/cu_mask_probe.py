"""What does a CU-masked stream (mmft_stream_create_cu_mask) do to kernel time on this stack?
A bandwidth-bound kernel (copy of 256 MB), a compute-bound one (fp32 GEMM) and a chain of small kernels, eager and as a
replayed HIP graph, on an ordinary stream and on masked streams of several shapes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                'multimodal-fusion-based-pre-routing-timing-prediction-_amd'))
from mmft import lib

dev = torch.device('cuda:0')
n = lib.cu_count(dev)
print('CUs', n)
big = torch.randn(64 * 1024 * 1024, device=dev)
out = torch.empty_like(big)
ga, gb = torch.randn(4096, 4096, device=dev), torch.randn(4096, 4096, device=dev)
xs = torch.randn(8192, 128, device=dev); ws = torch.randn(128, 128, device=dev) * 0.05


def copy():
    out.copy_(big)


def gemm():
    torch.mm(ga, gb)


def chain():
    y = xs
    for _ in range(40):
        y = torch.relu(y @ ws)


def timeit(f, s, reps=10):
    with torch.cuda.stream(s):
        for _ in range(3):
            f()
        s.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            f()
        s.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def graphed(f, s):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        f()
    torch.cuda.synchronize()
    cap = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=cap):
        f()
    return g.replay


streams = {'plain': torch.cuda.Stream()}
masks = {'low half': range(0, n // 2), 'high half': range(n // 2, n), 'even': range(0, n, 2), 'all': range(n),
         'low quarter': range(0, n // 4), 'first 32': range(32), 'stride 8': range(0, n, 8)}
keep = []
for name, cus in masks.items():
    m = lib.MaskedStream(dev, cus)
    keep.append(m)
    streams[name] = m.stream
for name, s in streams.items():
    r = [timeit(f, s) for f in (copy, gemm, chain)]
    rg = [timeit(graphed(f, s), s) for f in (copy, gemm, chain)]
    print(f'{name:12s} eager: copy {r[0]:.3f} gemm {r[1]:.3f} chain {r[2]:.3f} ms | graph: copy {rg[0]:.3f} gemm {rg[1]:.3f} chain {rg[2]:.3f} ms')

# concurrency: copy on the low half and gemm on the high half, against both on plain streams
def pair(sa, sb, fa, fb):
    def f():
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur); sb.wait_stream(cur)
        with torch.cuda.stream(sa):
            fa()
        with torch.cuda.stream(sb):
            fb()
        cur.wait_stream(sa); cur.wait_stream(sb)
    return f


p2 = torch.cuda.Stream()
for fa, fb, nm in ((copy, gemm, 'copy|gemm'), (chain, copy, 'chain|copy'), (chain, gemm, 'chain|gemm')):
    t_plain = timeit(pair(streams['plain'], p2, fa, fb), torch.cuda.current_stream())
    t_half = timeit(pair(streams['low half'], streams['high half'], fa, fb), torch.cuda.current_stream())
    t_q = timeit(pair(streams['low quarter'], streams['high half'], fa, fb), torch.cuda.current_stream())
    print(f'{nm:12s} two plain streams {t_plain:.3f} ms | low half + high half {t_half:.3f} ms | low quarter + high half {t_q:.3f} ms')
