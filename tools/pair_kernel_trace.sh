#!/bin/bash
# per-launch durations of the reverse pair kernel inside one eager step (rocprofv3 kernel trace)
OUT=gpurun_out/pairtrace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 bench.py --no-graph --no-overlap --no-cpu-baseline --no-roofline --no-drift --no-dropin --steps 3 --warmup 2 > $OUT/run.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/pairtrace/tr/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sel = [r for r in rows if 'level_bwd_pair' in r['Kernel_Name'] or 'level_bwd_pull' in r['Kernel_Name'] or 'level_fwd_slots' in r['Kernel_Name']]
last = sel[-63-32:] if len(sel) > 95 else sel
out = []
for r in last:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    out.append('%s grid=%s wg=%s %.1f us' % (r['Kernel_Name'][:40], r.get('Grid_Size_X', r.get('Grid_Size')), r.get('Workgroup_Size_X', r.get('Workgroup_Size')), d))
open('gpurun_out/pairtrace/launches.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out[-40:]))
PY
rm -rf $OUT/tr
