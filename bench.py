"""bench.py - train-step throughput of the multimodal-fusion hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one reference-shaped mini-batch (src/train.py:475-562) over a batch of synthetic designs:
U-Net forward, L-level netlist sweep, fusion head, MSE, full backward, Adam (plus one RCCL all-reduce of the
flat gradient buffer when N > 1).  Workload at N=1 is BASELINE.json configs[1] / SURVEY.md §8d config B:
8 designs per step, each 65 536 nodes / 64 levels / 256x256 layout tile / 1350 endpoints; arithmetic as BASELINE
names it: bf16 operands / fp32 accumulate on the MFMA-bound contractions (--dtype f32: exact fp32 everywhere).
Inputs (graphs, features, masks, images) are resident in HBM before the timed region.  Data parallel =
designs sharded over ranks (weak scaling: every rank steps its own 8 designs).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      dominant kernel by device time: algorithmic flops (or bytes) / launch duration, both from
                HIP events recorded by the library on the launch stream (mmft_prof_*), in a separate
                instrumented region of the same steps so that `value` is not perturbed by the events;
  cpu_baseline  the CPU oracle (oracle/restatement.py, kind "port") timed on this host on a bounded
                sample (one design per step as the reference does; 2 warm-up + 10 timed steps, median, anomaly
                detection off, plus 3 timed steps with torch.autograd.set_detect_anomaly(True) as the reference
                leaves it on at src/train.py:452).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`) BEFORE it touches the
GPU and exits with their status; rank 0 prints the JSON line.  Launched under torchrun already (the driver's form), it
is one of the ranks.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'multimodal-fusion-based-pre-routing-timing-prediction-_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_MFMA_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32-input MFMA, dense
PEAK_MFMA_BF16_TFLOPS = 2500.0   # MI355X_MICROARCH.md: bf16 MFMA, dense (the 5 PF headline includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s achievable)


def log(msg):
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


HISTORY = []        # every endpoint batch of the run, in order (the bf16 run is replayed in fp32 for the MAE drift)


def sample_paths(designs, batch_paths, rng):
    ids = [rng.permutation(d.num_paths)[:min(batch_paths, d.num_paths)] for d in designs]
    HISTORY.append(ids)
    return ids


def _norm_kernel(name):
    name = name.split('(')[0] if not name.startswith('bn_train') else name
    for t in ('void ', 'mmft::', ' '):
        name = name.replace(t, '')
    return name


def csrc_fingerprint():
    """sha256 over the kernel sources (csrc/*.hip, *.h, include/mmft.h): what a PMC table must have been taken on."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(PKG, 'csrc')
    files = sorted(f for f in os.listdir(src) if f.endswith(('.hip', '.h'))) if os.path.isdir(src) else []
    for f in files:
        with open(os.path.join(src, f), 'rb') as fh:
            h.update(f.encode() + b'\0' + fh.read())
    with open(os.path.join(ROOT, 'include', 'mmft.h'), 'rb') as fh:
        h.update(fh.read())
    return h.hexdigest()


PMC_TABLE = os.path.join(ROOT, 'profiles', 'r03_pmc_hbm_traffic.json')


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r03_pmc_hbm_traffic.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this bench, KiB units, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  The table is stamped with the fingerprint of the kernel sources it was
    measured on: (None, reason) when the stamp does not match the sources in this tree or the kernel has no entry -
    a stale table is never reported."""
    if not os.path.exists(PMC_TABLE):
        return None, 'no PMC table committed'
    with open(PMC_TABLE) as f:
        table = json.load(f)
    stamp = table.get('__stamp__', {})
    if stamp.get('csrc_sha256') != csrc_fingerprint():
        return None, 'PMC table was taken on other kernel sources (stamp mismatch)'
    want = _norm_kernel(kernel)
    src = 'rocprofv3 --pmc FETCH_SIZE (x2) + WRITE_SIZE, ' + os.path.basename(PMC_TABLE)
    rows = {_norm_kernel(k): v for k, v in table.items() if k != '__stamp__'}
    if want in rows:
        return rows[want]['hbm_bytes_per_launch'], src
    # the launch profiler names some kernels without their template arguments: accept the one instantiation of that name
    base = [k for k in rows if k.split('<')[0] == want.split('<')[0]]
    if len(base) == 1:
        return rows[base[0]]['hbm_bytes_per_launch'], src + ' (' + base[0] + ')'
    return None, 'kernel not in the PMC table'


def prof_report():
    from mmft import lib
    return lib.prof_report()


SEC8D_FWD = ('level_fwd_bf16_kernel', 'level_fwd_slots_kernel', 'pair_fwd_gather_kernel', 'seg_mean_fwd_kernel',
             'seg_softmax_sum_fwd_kernel', 'seg_attn_fwd_kernel')
SEC8D_BWD = ('level_bwd_pull_kernel', 'level_bwd_pull_attn_kernel', 'level_bwd_pair_kernel')


def sec8d_aggregation(designs, rows, nprof, D=128, s=4):
    """aggregation bytes per design-step, forward `s D (E + N_d) + 4 E + 4 (N_d + L)`, backward
    `s D (3 E + 2 N_d) + 4 E` (E = net + cell edges, N_d = nodes of level >= 1, s = bytes per stored element), summed
    over this rank's designs and divided by the device time of ALL launches of the kernels that carry that traffic (the fused
    level kernels of the bf16 mode run the level MLP inside the same launches: their whole duration counts)."""
    fwd = bwd = 0.0
    for d in designs:
        E = int(d.net_src.shape[0]) + int(d.cell_src.shape[0])
        Nd = int(d.N) - int(len(d.levels[0]))
        fwd += s * D * (E + Nd) + 4 * E + 4 * (Nd + d.L)
        bwd += s * D * (3 * E + 2 * Nd) + 4 * E
    out = {}
    for key, names, nbytes in (('fwd', SEC8D_FWD, fwd), ('bwd', SEC8D_BWD, bwd)):
        sel = [r for r in rows if r['name'].split('<')[0] in names]
        ms = sum(r['ms'] for r in sel) / nprof
        n = sum(r['launches'] for r in sel) / nprof
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out[key] = dict(bytes_per_step=nbytes, launches_per_step=n, device_ms_per_step=ms, kernels=sorted({r['name'] for r in sel}),
                        bytes_per_launch=nbytes / n if n else None, achieved_gbs=gbs, frac_of_hbm_peak=gbs / PEAK_HBM_GBS)
    return out


def cpu_baseline(design, pm_state, pc_state, batch_paths, steps=10, warm=2, anomaly_steps=3):
    """The CPU oracle's train step (reference-shaped: one design per step), timed on this host's cores: `warm` untimed
    + `steps` timed steps with anomaly detection off (median), then `anomaly_steps` timed steps with it on."""
    from oracle import restatement as R
    # the GPU box gives one GPU's share of the host: 16 cores (more threads than that only oversubscribe)
    ncores = host_threads_per_rank(1)
    torch.set_num_threads(ncores)
    log(f'cpu oracle on {ncores} threads (os.cpu_count() = {os.cpu_count()})')
    orc = R.OracleTrainer(pm_state, pc_state)
    csr = R.design_csr(design)
    rng = np.random.default_rng(7)

    def run(n, tag):
        t = []
        for i in range(n):
            ids = rng.permutation(design.num_paths)[:min(batch_paths, design.num_paths)].tolist()
            t0 = time.perf_counter()
            orc.step(design, csr, ids)
            t.append(time.perf_counter() - t0)
            log(f'cpu oracle {tag} step {i}: {t[-1]:.2f} s')
        return t
    run(warm, 'warm-up')
    sec = float(np.median(run(steps, 'timed')))
    sec_an = None
    if anomaly_steps:
        with torch.autograd.set_detect_anomaly(True):
            sec_an = float(np.median(run(anomaly_steps, 'anomaly-on')))
    return dict(value=1.0 / sec, unit='designs/s', cores=torch.get_num_threads(), kind='port',
                sample=f'median of {steps} timed steps (+{warm} warm-up) of ONE config-B design per step (the '
                       f'reference steps one design at a time), fp32 torch CPU, anomaly detection off; {sec:.2f} s/step',
                value_anomaly_on=(1.0 / sec_an) if sec_an else None,
                sample_anomaly_on=(f'median of {anomaly_steps} timed steps with torch.autograd.set_detect_anomaly(True) '
                                   f'(src/train.py:452); {sec_an:.2f} s/step') if sec_an else None)


def self_launch(args):
    """--gpus N > 1 without a torchrun environment: start the N ranks ourselves.  Nothing in this process has touched
    the GPU yet (importing torch does not), and it only waits for the child: no exec of a GPU process."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log('starting %d ranks: %s' % (args.gpus, ' '.join(cmd)))
    return subprocess.call(cmd, env=env)


def host_cpu_quota(cgroup_root='/sys/fs/cgroup'):
    """CPUs this process may use: min(affinity mask, cgroup CPU quota).  cgroup v2 `cpu.max` ("<quota> <period>" or "max
    <period>"), v1 `cpu/cpu.cfs_quota_us` / `cpu.cfs_period_us` (-1 = unlimited); unreadable files count as no quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    quota = None
    try:
        with open(os.path.join(cgroup_root, 'cpu.max')) as f:
            q, per = f.read().split()[:2]
            if q != 'max':
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open(os.path.join(cgroup_root, 'cpu', 'cpu.cfs_quota_us')) as f:
                q = float(f.read())
            with open(os.path.join(cgroup_root, 'cpu', 'cpu.cfs_period_us')) as f:
                per = float(f.read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    return max(1, min(n, int(quota)) if quota is not None and quota >= 1 else n)


def host_threads_per_rank(local_world=1, cap=16, cgroup_root='/sys/fs/cgroup'):
    """Threads one rank gives torch's CPU side.  The GPU box shows 256 CPUs but grants a 16-CPU quota (cgroup cpu.max): a
    single OpenMP region (index packing, link lists) spun up on 256 threads exhausts a 100 ms quota period in a few ms and
    the whole process is then throttled for the rest of it - 50-70 ms stalls in the middle of a step (measured: cpu.stat
    nr_throttled).  All ranks of a node share ONE cgroup, so the quota is divided by the number of local ranks: 8 ranks
    under a 16-CPU quota get 2 threads each (the timed step's host side is one index pack + one graph replay)."""
    return max(1, min(cap, host_cpu_quota(cgroup_root) // max(1, int(local_world))))


def limit_host_threads(local_world=1):
    torch.set_num_threads(host_threads_per_rank(local_world))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--designs', type=int, default=8, help='designs per rank per step (config B: 8)')
    ap.add_argument('--nodes', type=int, default=65536)
    ap.add_argument('--levels', type=int, default=64)
    ap.add_argument('--tile', type=int, default=256)
    ap.add_argument('--batch-paths', type=int, default=1350)
    ap.add_argument('--fanin', default='regular', choices=['regular', 'irregular'],
                    help="cell fan-in distribution of the synthetic netlist ('irregular' = config E's Zipf skew)")
    ap.add_argument('--mode', default='sweep', choices=['sweep', 'dropin'])
    ap.add_argument('--no-overlap', action='store_true', help='run the sweep and the CNN on one stream')
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly instead of replaying HIP graphs')
    ap.add_argument('--graph-pieces', default='auto', choices=['auto', 'one', 'five'],
                    help="'one': the step is ONE HIP graph (in-graph two-stream fork); 'five': five single-stream graphs "
                         "replayed on two streams (gradient buckets handed to the communication stream at the cuts); "
                         "'auto': one on a single GPU, five under data parallelism")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--dtype', default='bf16', choices=['f32', 'bf16'],
                    help='bf16 (default: BASELINE.json configs[1] is quoted as "1xMI355X bf16"): bf16 operands / fp32 '
                         'accumulate on the MFMA-bound contractions, U-Net activations stored bf16; the same schedule is then replayed '
                         'in exact fp32 for the held-out MAE drift and the fp32 step time.  f32: exact fp32 MFMA everywhere '
                         '(the 1e-4 parity mode the GPU tests run in)')
    ap.add_argument('--cpu-steps', type=int, default=10)
    ap.add_argument('--no-drift', action='store_true', help='bf16: skip the fp32 replay of the schedule (MAE drift)')
    ap.add_argument('--no-dropin', action='store_true', help='skip the extra timings of the per-level drop-in API')
    ap.add_argument('--cone', action='store_true',
                    help='fan-in-cone pruning inside the step (device-side mask per step; pays when --batch-paths is small)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    limit_host_threads(int(os.environ.get('LOCAL_WORLD_SIZE', world)))       # the ranks of a node share one CPU quota
    if world != args.gpus:
        log(f'WORLD_SIZE={world} differs from --gpus {args.gpus}: running (and reporting) {world} ranks')
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('MMFT_DIST_BACKEND', 'nccl') != 'nccl':
        local_rank = local_rank % max(torch.cuda.device_count(), 1)      # rehearsal: ranks may share a GPU
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the hot path has no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        # backend "nccl" is RCCL on ROCm; MMFT_DIST_BACKEND=gloo only for rehearsing the N>1 path on one GPU
        backend = os.environ.get('MMFT_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from mmft.synth import synth_design
    from mmft.train import build_models, TrainStep
    from mmft import lib
    lib.set_math_mode(args.dtype)

    from mmft.dist import design_seeds
    designs = [synth_design(N=args.nodes, L=args.levels, tile=args.tile, seed=sd, fanin=args.fanin)
               for sd in design_seeds(rank, args.designs)]
    log(f'rank {rank}: {len(designs)} designs generated')
    pmodel, cnn = build_models(map_size=designs[0].map_size, device=dev, seed=9294)
    pm_state = {k: v.detach().cpu().clone() for k, v in pmodel.state_dict().items()}
    pc_state = {k: v.detach().cpu().clone() for k, v in cnn.state_dict().items()}
    ts = TrainStep(pmodel, cnn, designs, dev, world_size=world, mode=args.mode, overlap=not args.no_overlap, cone=args.cone,
                   keep_grads=False)
    rng = np.random.default_rng(1234 + rank)
    log('resident on device; warm-up')
    stepper, graphed = ts, False
    if args.mode == 'sweep' and not args.no_graph:
        from mmft.train import GraphedTrainStep
        try:
            stepper = GraphedTrainStep(ts, sample_paths(designs, args.batch_paths, rng), pieces={'auto': None, 'one': False, 'five': True}[args.graph_pieces])
            if os.environ.get('MMFT_TIME_PIECES') and stepper.pieces:
                log(f'pieces alone / together (ms): {stepper.time_pieces()}')
            graphed = True
            log('train step captured (%s)' % ('five single-stream HIP graphs on two streams' if stepper.pieces else 'one HIP graph'))
        except Exception as e:                       # noqa: BLE001 - report and keep the eager path
            log(f'HIP-graph capture failed ({type(e).__name__}: {e}); running eagerly')
            stepper = ts

    if args.mode == 'dropin':
        import gc
        gc.collect()
        gc.freeze()     # the eager per-level loop allocates python objects per call: keep the designs' ~4 M list entries out of
                        # the collector's generations (a full collection in the middle of a step is a 50 ms stall)
    for _w in range(args.warmup):
        stepper.step(sample_paths(designs, args.batch_paths, rng))
        torch.cuda.synchronize()
        log(f'warm-up step {_w} done')
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tl = []
    for _it in range(args.steps):
        loss, hats, tl = stepper.step(sample_paths(designs, args.batch_paths, rng))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    log(f'timed region: {elapsed / args.steps * 1e3:.2f} ms/step')
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # endpoint ids in the index space of batch.arrival (renumbered, device side) - NOT the original ids step() returns
    mae = float((hats - ts.batch.arrival[ts.last_ends.long()].squeeze(-1)).abs().mean())

    roofline = None
    if not args.no_roofline:
        # every rank runs the instrumented steps (the all-reduce needs all of them); only rank 0 records events
        L = lib.load()
        if rank == 0:
            lib.prof_reset()
            lib.prof_enable(True)
        nprof = min(args.steps, 3)
        for _i in range(nprof):
            ts.step(sample_paths(designs, args.batch_paths, rng))
        torch.cuda.synchronize()
        if rank == 0:
            lib.prof_enable(False)
    if rank == 0 and not args.no_roofline:
        log('profiled steps done')
        rows = sorted(prof_report(), key=lambda r: -r['ms'])
        L.mmft_prof_reset()
        total_ms = sum(r['ms'] for r in rows)
        for r in rows[:40]:
            log('  %-70s %5d launches/step %8.3f ms/step %7.1f us/launch %7.2f TF %8.1f GB/s' % (
                r['name'][:70], r['launches'] // nprof, r['ms'] / nprof, r['ms'] / r['launches'] * 1e3,
                r['flops'] / (r['ms'] * 1e-3) / 1e12 if r['ms'] else 0, r['bytes'] / (r['ms'] * 1e-3) / 1e9 if r['ms'] else 0))
        top = rows[0]
        per_launch_ms = top['ms'] / top['launches']
        peak_tf = PEAK_MFMA_BF16_TFLOPS if (args.dtype == 'bf16' and 'bf16' in top['name']) else PEAK_MFMA_F32_TFLOPS
        tfs = top['flops'] / (top['ms'] * 1e-3) / 1e12
        gbs = top['bytes'] / (top['ms'] * 1e-3) / 1e9
        # which roof bounds the launch: its arithmetic intensity (algorithmic flops per algorithmic byte) against the ridge
        # of the two peaks.  A bf16 contraction fed from fp32 tensors sits far below the bf16 ridge (312 flop/B): HBM-bound.
        ridge = peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)
        intensity = top['flops'] / top['bytes'] if top['bytes'] > 0 else float('inf')
        if top['flops'] > 0 and intensity >= ridge:
            roofline = dict(bound='mfma', achieved=tfs, peak=peak_tf, unit='TFLOP/s', frac=tfs / peak_tf, traffic=None)
            if top['bytes'] > 0:
                roofline['hbm_view'] = dict(achieved=gbs, peak=PEAK_HBM_GBS, unit='GB/s', frac=gbs / PEAK_HBM_GBS)
        else:
            roofline = dict(bound='hbm', achieved=gbs, peak=PEAK_HBM_GBS, unit='GB/s', frac=gbs / PEAK_HBM_GBS, traffic=None)
            if top['flops'] > 0:
                roofline['mfma_view'] = dict(achieved=tfs, peak=peak_tf, unit='TFLOP/s', frac=tfs / peak_tf)
        roofline['flop_per_byte'] = None if intensity == float('inf') else intensity
        roofline['ridge_flop_per_byte'] = ridge
        default_workload = (args.designs, args.nodes, args.levels, args.tile, args.batch_paths, args.fanin) == \
            (8, 65536, 64, 256, 1350, 'regular')
        if default_workload:
            roofline['traffic'], roofline['traffic_source'] = pmc_traffic(top['name'])
        else:       # the committed counter passes were taken on config B: their bytes per launch say nothing about other sizes
            roofline['traffic'], roofline['traffic_source'] = None, 'PMC table is for config B launches'

        # SURVEY 8(d)'s own count of the aggregation traffic (the MINIMUM a sweep has to move: rows gathered + rows written +
        # indices), next to `frac`, which prices the bytes the kernel really moves (saved-for-backward rows included)
        sec8d = sec8d_aggregation(designs, rows, nprof, D=pmodel.gnn.out_feat_dim)
        roofline['sec8d_aggregation'] = sec8d
        side = 'fwd' if top['name'].split('<')[0] in SEC8D_FWD else ('bwd' if top['name'].split('<')[0] in SEC8D_BWD else None)
        roofline['frac_sec8d'] = sec8d[side]['frac_of_hbm_peak'] if side else None
        zero = [r['name'] for r in rows[:10] if r['bytes'] == 0 and r['flops'] == 0]
        roofline['top10_without_accounting'] = zero
        roofline.update(kernel=top['name'], launches_per_step=top['launches'] / nprof,
                        avg_launch_us=per_launch_ms * 1e3, share_of_device_time=top['ms'] / total_ms,
                        device_ms_per_step=total_ms / nprof,
                        top5=[dict(kernel=r['name'], ms_per_step=r['ms'] / nprof, launches=r['launches'] // nprof,
                                   tflops=(r['flops'] / (r['ms'] * 1e-3) / 1e12) if r['ms'] > 0 else 0.0)
                              for r in rows[:5]])

    # accuracy half of BASELINE.json's metric: endpoint-slack MAE on a held-out synthetic design after the steps above
    heldout_mae = None
    if rank == 0:
        from mmft.evaluate import validate
        held = synth_design(N=args.nodes, L=args.levels, tile=args.tile, seed=9294 + 100003)
        ev = TrainStep(pmodel, cnn, [held], dev, mode=args.mode, overlap=False, with_optimizer=False)
        heldout_mae = validate(ev)
        log(f"held-out design: slack MAE {heldout_mae['endpoint_slack_mae']:.4f}, R2 {heldout_mae['r2']:.4f}")

    # bf16 mode: the same schedule (same init, same endpoint batches in the same order) trained with exact fp32
    # arithmetic, evaluated on the same held-out design -> drift of the accuracy half of the metric
    f32_ref = None
    if rank == 0 and world == 1 and args.dtype == 'bf16' and heldout_mae is not None and graphed and not args.no_drift:
        log('replaying the schedule in fp32 for the held-out MAE drift')
        from mmft.train import GraphedTrainStep
        lib.set_math_mode('f32')
        pm2, cnn2 = build_models(map_size=designs[0].map_size, device=dev, seed=9294)
        pm2.load_state_dict(pm_state)
        cnn2.load_state_dict(pc_state)
        ts2 = TrainStep(pm2, cnn2, designs, dev, world_size=1, mode=args.mode, overlap=not args.no_overlap, keep_grads=False)
        gs2 = GraphedTrainStep(ts2, HISTORY[0], pieces={'auto': None, 'one': False, 'five': True}[args.graph_pieces])
        torch.cuda.synchronize()
        t_f32 = time.perf_counter()
        for ids in HISTORY[1:]:
            gs2.step(ids)
        torch.cuda.synchronize()
        f32_ms = (time.perf_counter() - t_f32) / max(len(HISTORY) - 1, 1) * 1e3
        ev2 = TrainStep(pm2, cnn2, [held], dev, mode=args.mode, overlap=False, with_optimizer=False)
        f32_ref = validate(ev2)
        lib.set_math_mode(args.dtype)
        log(f"fp32 replay ({ts2.optim.step_count} steps): held-out slack MAE {f32_ref['endpoint_slack_mae']:.4f} "
            f"(bf16 run: {heldout_mae['endpoint_slack_mae']:.4f})")

    # the per-level drop-in API (src/train.py:490-511 as written: model(...) once per level) on the same workload, OUTSIDE the
    # timed region: with the MaskedPathMap handle in place of the dense map, and with the reference's literal
    # dense `path_mask.to_dense() * feat_map`
    dropin = {}
    if rank == 0 and world == 1 and args.mode == 'sweep' and not args.no_dropin:
        import gc
        gc.collect()
        gc.freeze()     # the eager loop allocates python objects per level call: keep the designs' ~4 M list entries out of
                        # every generation-2 collection it would otherwise trigger (tens of ms each)
        for key, dense in (('dropin_ms_per_step', False), ('dropin_dense_ms_per_step', True)):
            try:
                pm3, cnn3 = build_models(map_size=designs[0].map_size, device=dev, seed=9294)
                ts3 = TrainStep(pm3, cnn3, designs, dev, world_size=1, mode='dropin', dense_path_map=dense, keep_grads=False)
                gc.collect()
                gc.freeze()      # ... including this step object's own level lists
                # 5 untimed steps (level-by-level first sweep, graph captures of the U-Net, first replays) + 20 timed ones for
                # the handle form; the dense form (10 x slower) 3 + 3
                nwarm, ntimed = (3, 3) if dense else (5, 20)
                sched = (HISTORY * (1 + (nwarm + ntimed) // max(len(HISTORY), 1)))[:nwarm + ntimed]
                for ids in sched[:nwarm]:
                    ts3.step(ids)
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                if os.environ.get('MMFT_DROPIN_TRACE'):
                    per = []
                    for ids in sched[nwarm:]:
                        t4 = time.perf_counter()
                        ts3.step(ids)
                        per.append((time.perf_counter() - t4) * 1e3)
                    log(f'{key}: host ms per step ' + ' '.join(f'{x:.1f}' for x in per))
                else:
                    for ids in sched[nwarm:]:
                        ts3.step(ids)
                torch.cuda.synchronize()
                dropin[key] = (time.perf_counter() - t3) / max(len(sched) - nwarm, 1) * 1e3
                log(f'{key}: {dropin[key]:.2f} ms ({len(sched) - nwarm} steps)')
                del ts3, pm3, cnn3
                torch.cuda.empty_cache()
            except Exception as e:                       # noqa: BLE001 - an extra measurement must not cost the bench line
                dropin[key] = None
                log(f'{key}: failed ({type(e).__name__}: {e})')

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log('timing the CPU oracle (bounded sample)')
        cpu = cpu_baseline(designs[0], pm_state, pc_state, args.batch_paths, steps=args.cpu_steps)

    if rank == 0:
        total_designs = args.designs * world * args.steps
        value = total_designs / elapsed
        out = {
            'metric': 'train-step samples/sec',
            'value': value,
            'unit': 'designs/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': args.dtype,
            'data': 'synthetic',
            'config': {
                'workload': f'{"config B" if (args.designs, args.nodes, args.levels, args.tile) == (8, 65536, 64, 256) else "custom"}: {args.designs} designs/step/GPU, {args.nodes}-node netlist, {args.levels} levels, '
                            f'{args.tile}x{args.tile} tile, {args.batch_paths} endpoints/design, UNet(max), '
                            + ('fp32' if args.dtype == 'f32' else 'bf16 operands / fp32 accumulate on the MFMA-bound contractions; U-Net activations and their gradients stored bf16, netlist tensors, statistics, parameters and gradients fp32'),
                'designs_per_step_per_gpu': args.designs, 'nodes': args.nodes, 'levels': args.levels,
                'tile': args.tile, 'endpoints_per_design': args.batch_paths, 'cone_pruning': bool(args.cone),
                'parallelism': f'dp{world} (designs sharded, one all-reduce of the flat gradient per step)',
                'launch': (('five single-stream HIP graphs replayed on two streams per step' if stepper.pieces else 'one HIP graph replay per step')
                           + ((' (Adam inside)' if not stepper.pieces else ' + eager Adam') if world == 1 else ' + eager bucketed all-reduce + Adam'))
                if graphed else 'eager launches',
                'api': 'PathModel.forward_sweep (whole-sweep entry; per-level drop-in path: --mode dropin)' if args.mode == 'sweep' else 'drop-in per-level model() calls',
            },
            'nodes_per_s': value * args.nodes,
            'pixels_per_s': value * args.tile * args.tile,
            'train_mae_last_step': mae,
            'heldout_eval': dict(steps_trained=ts.optim.step_count, **{k: heldout_mae[k] for k in
                                 ('endpoint_slack_mae', 'r2', 'loss', 'f1', 'n')}) if heldout_mae else None,
            'heldout_eval_f32_same_schedule': dict(steps_trained=ts2.optim.step_count, endpoint_slack_mae=f32_ref['endpoint_slack_mae'],
                                                   r2=f32_ref['r2'],
                                                   mae_drift=heldout_mae['endpoint_slack_mae'] - f32_ref['endpoint_slack_mae'],
                                                   ms_per_step=f32_ms, designs_per_s=args.designs / (f32_ms * 1e-3),
                                                   note='exact fp32 arithmetic (the 1e-4 parity mode), same init / batches / order')
            if f32_ref else None,
            'loss_last_step': float(loss),
            'roofline': roofline,
            'cpu_baseline': cpu,
        }
        out.update(dropin)
        if cpu is not None:
            out['speedup_vs_cpu_baseline'] = value / cpu['value']
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
