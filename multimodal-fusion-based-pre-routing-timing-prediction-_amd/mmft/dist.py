"""Data-parallel helpers: designs are sharded across ranks (one process per GPU, RCCL over xGMI), the only
collective per step is ONE all-reduce of the flat fp32 gradient buffer.  The reference has no
distributed code at all (two commented-out nn.DataParallel lines, src/train.py:129-130)."""
import os
import torch


def world_info():
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')),
            int(os.environ.get('LOCAL_RANK', '0')))


def design_seeds(rank, designs_per_rank, base=9294):
    """Rank r owns designs r*k .. r*k+k-1; seeds follow the reference's fixed seed 9294 (src/train.py:596)."""
    return [base + rank * designs_per_rank + i for i in range(designs_per_rank)]


def allreduce_sum_(flat, world_size):
    """Sum-all-reduce the flat gradient buffer in place; the 1/world scaling is folded into the fused Adam
    kernel (gscale), so no extra pass over the buffer is made.  Returns the scale to apply."""
    if world_size <= 1:
        return 1.0
    import torch.distributed as dist
    if flat.is_cuda and dist.get_backend() == 'gloo':
        # rehearsal of the N>1 path on a box without RCCL peers: stage through the host
        host = flat.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return 1.0 / world_size


class GradReducer:
    """Bucketed all-reduce of FlatAdam's flat gradient, overlapped with the rest of the backward pass.

    FlatAdam lays the parameters out bucket after bucket; mmft.gradsink tells when the last gradient of a bucket has
    been ISSUED by its producing kernel.  At that moment an event is recorded on each compute stream (the main one and
    the sweep's side stream), and the communication stream waits for those events, sum-all-reduces the bucket's range
    (RCCL over xGMI; backend "nccl") and runs the bucket's own Adam launch - while the remaining backward kernels keep
    running.  With the default buckets of mmft.train.TrainStep the fusion head (fcn + mlp_fuse + mlp_alpha: 2.27 M of
    the 2.89 M parameters, complete ~0.5 ms into the backward) is reduced underneath the U-Net backward and the
    reverse netlist sweep; the GNN + CNN bucket follows at the end.

    Collectives are issued in bucket order on every rank whatever order the buckets complete in (a bucket that
    completes early waits for its predecessors).  No collective is ever captured in a HIP graph.  Under
    GraphedTrainStep the forward + backward replay is ONE graph launch, and a point in the middle of it cannot be
    signalled to a stream outside it on this stack: event-record nodes (hipEventRecordWithFlags +
    hipEventRecordExternal) are refused by the HIP runtime PyTorch-ROCm 2.10 loads ("invalid argument"; torch itself
    raises "External events are disallowed in rocm").  The replayed step therefore reduces every bucket behind the
    whole replay, on the communication stream; the mid-backward overlap is what the eager step does.
    """

    def __init__(self, optim, world_size, device, side_stream=None):
        from . import gradsink
        self.optim, self.world = optim, int(world_size)
        self.device = torch.device(device)
        self.comm = torch.cuda.Stream(device=self.device)
        self.side = side_stream
        self.buckets = []
        for i, params in enumerate(optim.bucket_params):
            b = gradsink.Bucket(optim.bucket_ranges[i][0], params, self._on_complete)
            b.index = i
            self.buckets.append(b)
        self.main, self.early = None, True
        self.ready, self.events, self.issued = [], [], 0

    def begin(self, early=True):
        """Call right before loss.backward() of an EAGER step (after zero_grad).

        early=True hands a bucket to the communication stream at the moment its last member has received its FIRST
        gradient of the step - correct only when every parameter gets exactly one delivery per step (the whole-sweep
        step).  The per-level drop-in loop delivers the fusion head's gradients once per LEVEL (and re-reads the head's
        weights for the next level's input gradient), so there the caller passes early=False and every bucket is reduced
        in finish(), after backward() has returned."""
        self.main = torch.cuda.current_stream(self.device)
        self.early = bool(early)
        nb = len(self.buckets)
        self.ready, self.events, self.issued = [False] * nb, [None] * nb, 0

    def _streams(self):
        return [self.main] + ([self.side] if self.side is not None else [])

    def _on_complete(self, bucket):
        if self.main is None or not self.early:     # not inside an eager step (capture, evaluation, plain backward),
            return                                  # or a step whose parameters receive several deliveries
        evs = []
        for s in self._streams():
            ev = torch.cuda.Event()
            ev.record(s)
            evs.append(ev)
        self.events[bucket.index], self.ready[bucket.index] = evs, True
        while self.issued < len(self.buckets) and self.ready[self.issued]:
            self._reduce(self.issued, self.events[self.issued])
            self.issued += 1

    def _reduce(self, i, evs, streams=None):
        _, lo, hi = self.optim.bucket_ranges[i]
        with torch.cuda.stream(self.comm):
            if evs is None:
                for s in (streams or self._streams()):
                    self.comm.wait_stream(s)
            else:
                for ev in evs:
                    self.comm.wait_event(ev)
            scale = allreduce_sum_(self.optim.flat_grad[lo:hi], self.world)
            self.optim.step_bucket(i, gscale=scale)

    def finish(self):
        """Eager step, after loss.backward() has returned: reduce what has not been issued yet, then make the compute
        stream wait for the communication stream (the next zero_grad / forward touch the same buffers)."""
        for i in range(self.issued, len(self.buckets)):
            self._reduce(i, None)
        self.issued = len(self.buckets)
        self.main.wait_stream(self.comm)
        self.optim.step_count += 1
        self.main = None

    def after_replay(self):
        """Issue every bucket's all-reduce + Adam behind a graph replay that was just enqueued on the current stream
        (a graph launched on `main` is complete when `main` is)."""
        main = torch.cuda.current_stream(self.device)
        for i in range(len(self.buckets)):
            self._reduce(i, None, streams=[main])
        main.wait_stream(self.comm)
        self.optim.step_count += 1

    def reduce_bucket(self, i, events):
        """Piecewise-graph step: bucket i is complete once `events` have fired - all-reduce + Adam on the communication
        stream while the remaining pieces of the backward pass run."""
        self._pending = getattr(self, '_pending', set(range(len(self.buckets))))
        self._reduce(i, events)
        self._pending.discard(i)

    def reduce_rest_and_join(self, main):
        """After the last backward piece was enqueued on `main` (and `main` has waited for the side stream): reduce the
        buckets not handed over yet, make `main` wait for the communication stream."""
        pending = sorted(getattr(self, '_pending', set(range(len(self.buckets)))))
        for i in pending:
            self._reduce(i, None, streams=[main])
        self._pending = set(range(len(self.buckets)))
        main.wait_stream(self.comm)
        self.optim.step_count += 1
