"""Host-side issue time vs device time of the per-level (drop-in) train step at config B."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'multimodal-fusion-based-pre-routing-timing-prediction-_amd'))
sys.path.insert(0, ROOT)
from mmft import lib
from mmft.synth import synth_design
from mmft.train import build_models, TrainStep

torch.set_num_threads(int(os.environ.get('THREADS', '16')))     # see bench.limit_host_threads
dev = torch.device('cuda:0')
lib.set_math_mode(os.environ.get('MMFT_MATH', 'bf16'))
designs = [synth_design(N=65536, L=64, tile=256, seed=1000 + i) for i in range(8)]
pmodel, cnn = build_models(map_size=designs[0].map_size, device=dev, seed=9294)
ts = TrainStep(pmodel, cnn, designs, dev, mode=os.environ.get('MODE', 'dropin'), overlap=False)
rng = np.random.default_rng(0)
ids = lambda: [rng.permutation(d.num_paths)[:1350].tolist() for d in designs]
import gc
import model as _M
from mmft import sweep as _S, fusion as _F
_T = {}


def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        _T[name] = _T.get(name, 0.0) + time.perf_counter() - t
        return r
    return w


import inspect
from mmft import cnn as _C, functional as _MF
_last = [0.0]


def timed_max(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        gap = t - _last[0]
        if _last[0] and gap > 5e-3:
            _T['GAP before ' + name] = round(gap * 1e3, 1)
        r = fn(*a, **k)
        d = time.perf_counter() - t
        _last[0] = time.perf_counter()
        if d > 5e-3:
            _T['SLOW ' + name] = round(d * 1e3, 1)
        return r
    return w


for mod in (_C, _MF, _F, _S, _M):
    for nm, cls in inspect.getmembers(mod, inspect.isclass):
        if issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function and 'backward' in cls.__dict__:
            cls.backward = staticmethod(timed_max(nm + '.backward', cls.backward))
_M._HeadBatch.backward = timed('head root backward', _M._HeadBatch.backward)
_S.SweepFn.backward = staticmethod(timed('sweep backward', _S.SweepFn.backward))
_F.MaskedFcFn.backward = staticmethod(timed('masked fc backward', _F.MaskedFcFn.backward))
_F.MaskedPathMap._links = timed('path links (host)', _F.MaskedPathMap._links)
gc.callbacks.append(lambda phase, info: _T.__setitem__('gc gen%d' % info['generation'], _T.get('gc gen%d' % info['generation'], 0) + 1) if phase == 'start' else None)
for _ in range(4):
    ts.step(ids())
torch.cuda.synchronize()
if os.environ.get('FREEZE', '1') == '1':
    import gc
    gc.collect()
    gc.freeze()          # the designs' level lists hold ~4 M python ints: keep them out of every later full collection
for it in range(12):
    p = ids()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hats, ends_d, ends_h = ts.forward(p)
    loss = ts.loss(hats, ends_d)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.optim.zero_grad()
    _last[0] = 0.0
    if os.environ.get('STALL_TRACE'):                 # where is the host when a backward takes longer than 22 ms?
        import faulthandler
        faulthandler.dump_traceback_later(0.022, exit=False)
    loss.backward()
    if os.environ.get('STALL_TRACE'):
        faulthandler.cancel_dump_traceback_later()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    ts.optim.step()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    ms = torch.cuda.memory_stats()
    print('device allocs', ms.get('num_device_alloc'), 'frees', ms.get('num_device_free'), 'retries', ms.get('num_alloc_retries'),
          'reserved MB', ms['reserved_bytes.all.current'] >> 20, 'allocated MB', ms['allocated_bytes.all.current'] >> 20)
    print({k: (round(v * 1e3, 2) if isinstance(v, float) else v) for k, v in _T.items()}); _T.clear()
    print(f'forward: host issue {1e3 * (t1 - t0):6.2f} ms, done {1e3 * (t2 - t0):6.2f} ms | backward: host issue {1e3 * (t3 - t2):6.2f} ms, '
          f'done {1e3 * (t4 - t2):6.2f} ms | adam {1e3 * (t5 - t4):5.2f} ms')
if os.environ.get('PROFILE'):
    import cProfile, pstats, io
    for what in ('forward', 'backward'):
        p = ids()
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        if what == 'forward':
            pr.enable()
            hats, ends_d, ends_h = ts.forward(p)
            loss = ts.loss(hats, ends_d)
            pr.disable()
            ts.optim.zero_grad(); loss.backward(); ts.optim.step()
        else:
            hats, ends_d, ends_h = ts.forward(p)
            loss = ts.loss(hats, ends_d)
            ts.optim.zero_grad()
            torch.cuda.synchronize()
            pr.enable()
            loss.backward()
            pr.disable()
            ts.optim.step()
        torch.cuda.synchronize()
        sio = io.StringIO()
        pstats.Stats(pr, stream=sio).strip_dirs().sort_stats('tottime').print_stats(45)
        print('==== cProfile of one', what, '(main thread only)')
        print('\n'.join(l[:150] for l in sio.getvalue().splitlines()[:70]))
