"""ctypes binding of libmmft_hip.so (the C ABI declared in include/mmft.h).

There is NO fallback: if the shared library is missing or an entry point fails, a RuntimeError is
raised.  PyTorch is used only for device memory, streams and torch.distributed.
"""
import ctypes
import os
import re
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.dirname(_HERE)
LIB_PATH = os.environ.get('MMFT_LIB') or os.path.join(PKG_DIR, 'lib', 'libmmft_hip.so')      # MMFT_LIB: another build of the same library (A/B runs)
HEADER_PATH = os.path.join(os.path.dirname(PKG_DIR), 'include', 'mmft.h')

_lib = None


def header_decls():
    """{name: (restype, [argtypes])} parsed from include/mmft.h - the header is the single source of truth."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    decls = {}
    for ret, name, params in re.findall(r'\b(int|long long|const char\*)\s+(mmft_[a-z0-9_]+)\s*\(([^)]*)\)\s*;', text):
        args = []
        params = params.strip()
        if params and params != 'void':
            for prm in params.split(','):
                prm = prm.strip()
                if '*' in prm:
                    args.append(ctypes.c_void_p)
                elif prm.startswith('unsigned'):
                    args.append(ctypes.c_uint)
                elif prm.startswith('long long'):
                    args.append(ctypes.c_longlong)
                elif prm.startswith('int'):
                    args.append(ctypes.c_int)
                elif prm.startswith('float'):
                    args.append(ctypes.c_float)
                elif prm.startswith('double'):
                    args.append(ctypes.c_double)
                else:
                    raise RuntimeError(f'mmft.h: cannot parse parameter "{prm}" of {name}')
        res = {'int': ctypes.c_int, 'long long': ctypes.c_longlong, 'const char*': ctypes.c_char_p}[ret]
        decls[name] = (res, args)
    return decls


def header_symbols():
    """Names of every function declared in include/mmft.h (used by the CPU-side symbol test)."""
    return sorted(header_decls())


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libmmft_hip.so not found at {LIB_PATH}; build it with `python __graft_entry__.py build` "
                f"(make -C {os.path.join(PKG_DIR, 'csrc')}). There is no CPU/eager fallback for the hot path.")
        _lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in header_decls().items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def _conv(a):
    if a is None:
        return None
    if torch.is_tensor(a):
        return a.data_ptr()
    if isinstance(a, bool):
        return int(a)
    return a


def stream_args(t):
    """(device ordinal, hipStream_t) of the current torch stream for tensor t's device."""
    dev = t.device.index if t.device.index is not None else torch.cuda.current_device()
    return dev, torch.cuda.current_stream(dev).cuda_stream


_FN = {}
_Tensor = torch.Tensor


def call(name, *args):
    """Call an `int mmft_*(...)` entry point: tensors are passed as their data pointers, everything else as is (ctypes
    converts ints, floats, bools and None through the argtypes parsed from the header).  Hot on every eager launch: the
    bound function is cached and the argument loop does one isinstance per argument."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(load(), name)
    rc = fn(*[a.data_ptr() if isinstance(a, _Tensor) else a for a in args])
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {load().mmft_last_error().decode()}")


def query(name, *args):
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(load(), name)
    return int(fn(*[a.data_ptr() if isinstance(a, _Tensor) else a for a in args]))


_workspace = {}
_retired = []


def workspace(device, nbytes):
    """Scratch buffer (split-K slabs, BN partials, re-laid-out dgrad weights), grown on demand, never shrunk.
    One buffer per (device, stream): kernels of the netlist sweep and of the CNN may run concurrently on two
    streams and must not share scratch."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    buf = _workspace.get(key)
    need = max(int(nbytes), 1 << 20)
    if buf is None or buf.numel() * 4 < need:
        if buf is not None:
            # a captured HIP graph may hold this buffer's address (the whole-step graph, the U-Net's replay graphs): an
            # outgrown buffer is retired, never handed back to the allocator for somebody else's tensor
            _retired.append(buf)
            need = max(need, 2 * buf.numel() * 4)          # geometric growth bounds what the retired ones add up to
        buf = torch.empty((need + 3) // 4, dtype=torch.float32, device=dev)
        _workspace[key] = buf
    return buf


PROF_ON = False


def prof_enable(on=True):
    """Launch profiler of the library (mmft_prof_*): HIP start / stop events around every instrumented launch."""
    global PROF_ON
    load().mmft_prof_enable(1 if on else 0)
    PROF_ON = bool(on)


def prof_hint(flops, nbytes):
    """Algorithmic work of the calling thread's next launch whose site cannot know it (data-dependent gathers); a no-op
    unless the launch profiler is on."""
    if PROF_ON:
        load().mmft_prof_hint(float(flops), float(nbytes))


def prof_reset():
    load().mmft_prof_reset()


def prof_report():
    """Rows {name, launches, ms, flops, bytes} aggregated per kernel name since the last reset."""
    L = load()
    need = L.mmft_prof_report(None, 0)
    buf = ctypes.create_string_buffer(need + 16)
    L.mmft_prof_report(ctypes.cast(buf, ctypes.c_void_p), need + 16)
    rows = []
    for line in buf.value.decode().splitlines():
        name, n, ms, fl, by = line.split('\t')
        rows.append(dict(name=name, launches=int(n), ms=float(ms), flops=float(fl), bytes=float(by)))
    return rows


MATH_MODES = {'f32': 0, 'bf16': 1}


def set_math_mode(mode):
    """'f32' (default): exact fp32 MFMA everywhere - the 1e-4 parity mode.  'bf16': the MFMA-bound contractions (dense
    layers, convolutions, fused level MLPs) round their operands to bf16 and accumulate in fp32 (mmft_set_math_mode in
    include/mmft.h).  Process-wide: the backward pass runs on another thread."""
    if mode not in MATH_MODES:
        raise ValueError(f'unknown math mode {mode!r} (expected one of {sorted(MATH_MODES)})')
    call('mmft_set_math_mode', MATH_MODES[mode])


def get_math_mode():
    code = query('mmft_get_math_mode')
    return next(k for k, v in MATH_MODES.items() if v == code)


class math_mode:
    """with lib.math_mode('bf16'): ...  (restores the previous mode)."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = get_math_mode()
        set_math_mode(self.mode)
        return self

    def __exit__(self, *exc):
        set_math_mode(self.prev)
        return False
