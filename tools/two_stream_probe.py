"""Do two streams overlap on this stack?  Stream A: a chain of dependent small kernels (the level chain's shape),
stream B: medium chip-filling kernels (the U-Net's shape).  Reports A alone, B alone, both (eager and as HIP graphs)."""
import time, torch
dev = torch.device('cuda:0')
xa = torch.randn(8192, 128, device=dev); wa = torch.randn(128, 128, device=dev) * 0.05
xb = torch.randn(8, 64, 256, 256, device=dev); wb = torch.randn(64, 64, 3, 3, device=dev) * 0.05
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def chain(n=60):
    y = xa
    for _ in range(n):
        y = torch.relu(y @ wa)          # small dependent GEMM + elementwise: ~10-15 us each, far from filling the chip
    return y


def convs(n=12):
    y = xb
    for _ in range(n):
        y = torch.nn.functional.conv2d(y, wb, padding=1)
    return y


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def both_eager():
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    with torch.cuda.stream(sa):
        chain()
    with torch.cuda.stream(sb):
        convs()
    cur.wait_stream(sa); cur.wait_stream(sb)


ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.stream(sa):
    chain(); convs()
torch.cuda.synchronize()
with torch.cuda.graph(ga, stream=sa):
    chain()
with torch.cuda.graph(gb, stream=sb):
    convs()


def both_graphs():
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    with torch.cuda.stream(sa):
        ga.replay()
    with torch.cuda.stream(sb):
        gb.replay()
    cur.wait_stream(sa); cur.wait_stream(sb)


def one(g, s):
    def f():
        with torch.cuda.stream(s):
            g.replay()
        torch.cuda.current_stream().wait_stream(s)
    return f


a, b = timeit(one(ga, sa)), timeit(one(gb, sb))
print(f'graph A (chain) alone {a:.3f} ms, graph B (convs) alone {b:.3f} ms, sum {a + b:.3f}, max {max(a, b):.3f}')
print(f'both as graphs on two streams: {timeit(both_graphs):.3f} ms')
print(f'both eager on two streams:     {timeit(both_eager):.3f} ms')
