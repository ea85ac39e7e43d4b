"""Timing of the preprocessing kernels (mmft.prep) on a synthetic design, with the CPU restatement beside it on a
bounded sample.  python tools/bench_prep.py [--nodes N --levels L]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multimodal-fusion-based-pre-routing-timing-prediction-_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nodes', type=int, default=1 << 20)
    ap.add_argument('--levels', type=int, default=96)
    ap.add_argument('--map', type=int, default=256)
    args = ap.parse_args()
    from mmft import prep
    from mmft.synth import synth_design
    from test_prep_gpu import csr
    from oracle import prep_restatement as PR
    dev = torch.device('cuda:0')
    d = synth_design(N=args.nodes, L=args.levels, tile=64, seed=9300)
    out = [csr(d.N, d.net_src, d.net_dst, dev), csr(d.N, d.cell_src, d.cell_dst, dev)]
    inn = [csr(d.N, d.net_dst, d.net_src, dev), csr(d.N, d.cell_dst, d.cell_src, dev)]
    pis = torch.from_numpy(d.levels[0].astype(np.int32)).to(dev)
    ends = torch.from_numpy(d.path2endpoint.astype(np.int32)).to(dev)
    rng = np.random.default_rng(0)
    lx = torch.from_numpy(rng.integers(0, args.map, size=d.N).astype(np.int32)).to(dev)
    ly = torch.from_numpy(rng.integers(0, args.map, size=d.N).astype(np.int32)).to(dev)
    feat = torch.randn(d.N, 36, device=dev)

    def timed(fn, n=5):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, r

    ms, (level, nl) = timed(lambda: prep.levelize(out, d.N, pis))
    print(f'levelize       {d.N} nodes, {int(out[0][1].numel() + out[1][1].numel())} edges, {nl} levels: {ms:8.2f} ms')
    ms, (paths, lens) = timed(lambda: prep.trace_critical_paths(inn, level, ends))
    print(f'trace paths    {ends.numel()} endpoints, max length {paths.shape[1]}: {ms:8.2f} ms')
    ms, (ip, cols) = timed(lambda: prep.rasterize_path_masks(paths, lens, lx, ly, args.map, args.map))
    print(f'path masks     {ends.numel()} rows over a {args.map}x{args.map} map, nnz {cols.numel()}: {ms:8.2f} ms')
    ms, _ = timed(lambda: prep.minmax_normalize_(feat, 0))
    print(f'min-max        {d.N} x 36 floats: {ms:8.2f} ms  ({2 * d.N * 36 * 4 * 1.5 / ms / 1e6:.0f} GB/s)')
    # CPU restatement on a bounded sample: levelization of the same graph (python sets, as the reference)
    suc, pre = PR.adjacency(d.N, np.concatenate([d.net_src, d.cell_src]), np.concatenate([d.net_dst, d.cell_dst]))
    t0 = time.perf_counter()
    PR.cal_topo_level(suc, d.levels[0].tolist())
    print(f'CPU restatement of cal_topo_level (1 core, python sets): {(time.perf_counter() - t0) * 1e3:8.1f} ms')
    node2level = dict(zip(range(d.N), level.cpu().tolist()))
    sample = d.path2endpoint[:200].tolist()
    t0 = time.perf_counter()
    pp = [PR.find_critical_path(e, pre, node2level) for e in sample]
    loc = dict(zip(range(d.N), zip(lx.cpu().tolist(), ly.cpu().tolist())))
    PR.path_mask_rows(pp, loc, args.map, args.map)
    dt = time.perf_counter() - t0
    print(f'CPU restatement of trace + masks: {dt / len(sample) * 1e3:8.2f} ms per path ({len(sample)} paths sampled)')


if __name__ == '__main__':
    main()
