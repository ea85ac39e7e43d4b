"""rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate runs, CSV) -> profiles/rNN_pmc_hbm_traffic.json.

    python tools/pmc_to_json.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [steps]
Per kernel name: launches, FETCH_SIZE * 1024 * 2 (KiB units; gfx950 reports half the bytes of wide coalesced reads,
MI355X_MICROARCH.md "HBM"), WRITE_SIZE * 1024, and their sum per launch - what bench.py reports as roofline.traffic.
The table is stamped with the fingerprint of the kernel sources of THIS tree (bench.csrc_fingerprint): run it on the
tree the passes were taken on; bench.py reports traffic = null when the stamp does not match."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(path, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        key = (r['Dispatch_Id'], r['Kernel_Name'])
        tot[r['Kernel_Name']] += float(r['Counter_Value'])
        if key not in seen:
            seen.add(key)
            cnt[r['Kernel_Name']] += 1
    return tot, cnt


def main():
    fpath, wpath, out = sys.argv[1:4]
    ft, fc = per_kernel(fpath, 'FETCH_SIZE')
    wt, wc = per_kernel(wpath, 'WRITE_SIZE')
    table = {}
    for k in ft:
        n = fc[k]
        f = ft[k] * 1024.0 * 2.0 / n
        w = wt.get(k, 0.0) * 1024.0 / max(wc.get(k, n), 1)
        table[k] = dict(launches=n, fetch_bytes_per_launch_x2=f, write_bytes_per_launch=w, hbm_bytes_per_launch=f + w)
    from bench import csrc_fingerprint
    table['__stamp__'] = dict(csrc_sha256=csrc_fingerprint(), note='kernel sources the PMC passes were taken on')
    json.dump(table, open(out, 'w'), indent=0)
    rows = [(k, v) for k, v in table.items() if k != '__stamp__']
    for k, v in sorted(rows, key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:12]:
        print(f"{k[:70]:70s} {v['launches']:5d} launches  {v['hbm_bytes_per_launch'] / 1e6:9.2f} MB/launch")


if __name__ == '__main__':
    main()
