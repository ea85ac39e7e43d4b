"""Concurrency analysis of one replayed step from a rocprofv3 --kernel-trace CSV (which kernels overlap, per-queue timeline).

    rocprofv3 --kernel-trace --output-format csv -d out -- python3 bench.py --steps 6 --no-cpu-baseline --no-roofline
    python tools/trace_overlap.py out/*/*_kernel_trace.csv
"""
import collections
import csv
import sys


def main(path, verbose=False):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    rows.sort(key=lambda r: r['s'])
    adam = [i for i, r in enumerate(rows) if 'adam' in r['Kernel_Name']]
    a, b = adam[-3], adam[-2]
    step = rows[a + 1:b + 1]
    t0, t1 = step[0]['s'], step[-1]['e']
    print(f'step span {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels, sum of kernel times {sum(r["e"] - r["s"] for r in step) / 1e6:.3f} ms')
    ev = []
    for r in step:
        ev += [(r['s'], 1), (r['e'], -1)]
    ev.sort()
    c, last, hist = 0, ev[0][0], collections.Counter()
    for t, d in ev:
        hist[c] += t - last
        last = t
        c += d
    print('time with k kernels in flight (ms):', {k: round(v / 1e6, 3) for k, v in sorted(hist.items())})
    for q in sorted({r['Queue_Id'] for r in step}):
        ks = [r for r in step if r['Queue_Id'] == q]
        print(f'queue {q}: {len(ks)} kernels, first start {(ks[0]["s"] - t0) / 1e3:.0f} us, last end {(ks[-1]["e"] - t0) / 1e3:.0f} us, '
              f'busy {sum(r["e"] - r["s"] for r in ks) / 1e3:.0f} us')
    # per-kernel totals INSIDE the replayed step (no per-launch profiling floor), and the idle time in front of each kernel
    short = lambda n: n.replace('void mmft::', '').replace('mmft::', '').split('(')[0][:70]
    agg = collections.defaultdict(lambda: [0, 0, 0])
    busy_until = step[0]['s']
    for r in step:
        a_ = agg[short(r['Kernel_Name'])]
        a_[0] += 1
        a_[1] += r['e'] - r['s']
        if r['s'] > busy_until:
            a_[2] += r['s'] - busy_until
        busy_until = max(busy_until, r['e'])
    print(f'{"kernel":72s} {"n":>4s} {"total us":>9s} {"avg us":>8s} {"idle before, us":>16s}')
    for name, (n, tot, idle) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'{name:72s} {n:4d} {tot / 1e3:9.1f} {tot / 1e3 / n:8.2f} {idle / 1e3:16.1f}')
    if verbose:
        prev, start, cnt = None, None, 0
        for r in step:
            name = r['Kernel_Name'].replace('void mmft::', '').replace('mmft::', '').split('(')[0][:28]
            key = (r['Queue_Id'], name)
            if key != prev:
                if prev:
                    print(f'{start:8.0f} us  q{prev[0]} x{cnt:3d} {prev[1]}')
                prev, start, cnt = key, (r['s'] - t0) / 1e3, 0
            cnt += 1
        print(f'{start:8.0f} us  q{prev[0]} x{cnt:3d} {prev[1]}')


if __name__ == '__main__':
    main(sys.argv[1], verbose=len(sys.argv) > 2)
