"""Gradient sinks: let the backward kernels write parameter gradients straight into the flat gradient buffer.

torch's AccumulateGrad adds every returned gradient into an already defined `.grad` with one elementwise kernel
per parameter (71 launches per step on the reference model) after the producing kernel wrote a temporary.  When
FlatAdam owns the parameters (`.grad` = view of one flat buffer) each autograd Function of this package asks
`claim()` for the parameter's slot instead: the first gradient of a step is stored there by the producing
kernel, later ones (shared weights, retain_graph re-runs) are added, and the Function returns None for that
input so that AccumulateGrad is not run.  Parameters without a sink follow the ordinary autograd path, so the
modules keep working under torch.optim.Adam as in the reference (src/train.py:431-435).
"""
import torch

_epoch = [0]


def attach(param, grad_view):
    """Route gradients of `param` (a leaf) into `grad_view`, a dense view with the parameter's shape and strides."""
    param._mmft_sink = [grad_view, -1, None]


class Bucket:
    """A group of sunk parameters whose gradients travel together (one all-reduce under data parallelism).
    `on_complete(bucket)` is called from the backward pass - on whatever thread and stream the last producing
    kernel was issued from - at the moment every member's gradient of the current step has been ISSUED."""

    def __init__(self, name, params, on_complete=None):
        self.name, self.on_complete = name, on_complete
        self.members = [p._mmft_sink for p in params]
        for rec in self.members:
            rec[2] = self
        self.pending, self.epoch = len(self.members), -1

    def _delivered(self):
        if self.epoch != _epoch[0]:
            self.epoch, self.pending = _epoch[0], len(self.members)
        self.pending -= 1
        if self.pending == 0 and self.on_complete is not None:
            self.on_complete(self)

    def complete(self):
        return self.epoch == _epoch[0] and self.pending == 0


def _mark(sink):
    """First gradient of the current step goes into `sink`."""
    sink[1] = _epoch[0]
    if sink[2] is not None:
        sink[2]._delivered()


def detach(param):
    if hasattr(param, '_mmft_sink'):
        del param._mmft_sink


def new_step():
    """Called by the owner of the sinks after it has zeroed them (FlatAdam.zero_grad)."""
    _epoch[0] += 1
    _param_epoch[0] += 1


# Bumped whenever parameters may have been rewritten behind torch's back (the fused Adam kernel writes the flat buffer
# through raw pointers, so Tensor._version does not move): per-step caches derived from parameters key on it.
_param_epoch = [0]


def params_changed():
    _param_epoch[0] += 1


def param_epoch():
    return _param_epoch[0]


def of(param):
    """The sink record to stash in ctx during forward (None when the tensor is not a sunk parameter)."""
    return getattr(param, '_mmft_sink', None) if param is not None else None


def _memory_order(view):
    """`view` permuted so that it is contiguous (its memory order), plus the permutation; None if not dense."""
    if view.is_contiguous():
        return view, None
    perm = sorted(range(view.dim()), key=lambda d: (-view.stride(d), d))
    v = view.permute(perm)
    return (v, perm) if v.is_contiguous() else (None, None)


def deliver(sink, compute, shape=None):
    """Produce one parameter gradient.

    sink     record from `of()` or None
    shape    the kernel's view of the memory-order gradient (same element order, e.g. 2-D for a GEMM output)
    compute  compute(out) -> tensor: writes the gradient into `out` when given (a contiguous tensor in the
             parameter's MEMORY order, e.g. [Co][KH][KW][Ci] for a channels_last conv weight), else allocates;
             returns it in memory order either way.
    Returns what the autograd Function must return for that input: None when the sink took the gradient, else
    the gradient tensor (memory order; the caller permutes it to the logical shape).
    """
    if sink is None:
        return compute(None)
    view, seen = sink[0], sink[1]
    mem, _ = _memory_order(view)
    if mem is None:
        return compute(None)
    if shape is not None:
        mem = mem.reshape(shape)                 # a view: mem is contiguous
    if seen != _epoch[0]:
        compute(mem)
        _mark(sink)
    else:
        mem.add_(compute(None).reshape(mem.shape))
    return None


def deliver_pair(sink_a, sink_b, compute):
    """Two gradients from ONE kernel (weight and bias gradient of a Linear): compute(out_a, out_b) -> (a, b) in
    memory order, writing into the outs that are given.  Returns the pair the autograd Function must return."""
    outs, late, fresh_sinks = [], [], []
    for sink in (sink_a, sink_b):
        mem = _memory_order(sink[0])[0] if sink is not None else None
        if mem is not None and sink[1] != _epoch[0]:
            fresh_sinks.append(sink)
            outs.append(mem); late.append(None)
        else:
            outs.append(None); late.append(mem)
    res = list(compute(outs[0], outs[1]))
    for sink in fresh_sinks:                 # marked after the producing kernel has been issued
        _mark(sink)
    for i in range(2):
        if outs[i] is not None:
            res[i] = None
        elif late[i] is not None:
            late[i].add_(res[i].reshape(late[i].shape))
            res[i] = None
    return res[0], res[1]


def deliver_many(sinks, compute):
    """deliver_pair for any number of gradients produced by ONE kernel: compute(*outs) -> tuple in memory order."""
    outs, late, fresh_sinks = [], [], []
    for sink in sinks:
        mem = _memory_order(sink[0])[0] if sink is not None else None
        if mem is not None and sink[1] != _epoch[0]:
            fresh_sinks.append(sink)
            outs.append(mem); late.append(None)
        else:
            outs.append(None); late.append(mem)
    res = list(compute(*outs))
    for sink in fresh_sinks:                 # marked after the producing kernel has been issued
        _mark(sink)
    for i in range(len(sinks)):
        if outs[i] is not None:
            res[i] = None
        elif late[i] is not None:
            late[i].add_(res[i].reshape(late[i].shape))
            res[i] = None
    return res


def fresh(sink):
    """True when nothing has been written into the sink during the current step."""
    return sink[1] != _epoch[0]


def take(sink):
    """Return the sink's view for the caller's kernel to store into; the caller calls `taken(sink)` once that kernel
    has been issued."""
    return sink[0]


def taken(sink):
    _mark(sink)
