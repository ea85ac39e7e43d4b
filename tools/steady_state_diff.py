"""Steady-state per-step kernel counts and time from TWO rocprofv3 kernel-statistics CSVs of the same command that differ
only in the number of steps: everything that happens once per process (uploads, warm-up, capture, the CPU-side set-up
copies that show up as `__amd_rocclr_copyBuffer`) cancels in the difference.

    python tools/steady_state_diff.py short_kernel_stats.csv STEPS_SHORT long_kernel_stats.csv STEPS_LONG [out.csv]

Prints / writes one row per kernel: calls per step, microseconds per step, flagged `torch` for kernels that are not this
library's (at::native, rocprim, rocclr)."""
import csv
import sys


def load(path):
    rows = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            rows[r['Name']] = (int(r['Calls']), float(r['TotalDurationNs']))
    return rows


def main():
    a, na, b, nb = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    out = sys.argv[5] if len(sys.argv) > 5 else None
    ra, rb = load(a), load(b)
    d = float(nb - na)
    rows = []
    for name in sorted(set(ra) | set(rb)):
        ca, ta = ra.get(name, (0, 0.0))
        cb, tb = rb.get(name, (0, 0.0))
        calls, us = (cb - ca) / d, (tb - ta) / d / 1e3
        if abs(calls) < 1e-9 and abs(us) < 1e-3:
            continue
        lib = 'mmft::' in name
        rows.append((us, calls, 'mmft' if lib else 'torch', name))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    other = [(r[1], r[0], r[3]) for r in rows if r[2] == 'torch']
    lines = ['kernel,calls_per_step,us_per_step,origin']
    for us, calls, origin, name in rows:
        lines.append('"%s",%.3f,%.2f,%s' % (name.replace('"', "'"), calls, us, origin))
    text = '\n'.join(lines)
    if out:
        with open(out, 'w') as f:
            f.write(text + '\n')
    print('steady state: %.1f us of kernel time per step over %d kernels; not from this library: %.1f launches / %.1f us per step'
          % (tot, len(rows), sum(c for c, _, _ in other), sum(u for _, u, _ in other)))
    for c, u, n in other:
        print('   torch: %6.2f launches/step %8.2f us/step  %s' % (c, u, n[:110]))


if __name__ == '__main__':
    main()
