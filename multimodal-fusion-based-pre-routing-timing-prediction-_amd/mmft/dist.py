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
