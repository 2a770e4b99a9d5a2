"""The exchange steps of the data-parallel path (one process per GPU; `nccl` == RCCL over xGMI on ROCm, `gloo` in CPU
tests).  Environments are independent, so ranks only ever exchange:
  * the flat fp32 gradient, averaged, once per optimiser step (the DDP semantics the reference intends: base_agent.py:47-57;
    SURVEY.md section 0 shows its own wiring never fires);
  * the observation-normaliser sums once per iteration (normalizer.py:41-58);
  * the logged scalars (util/logger.py:160-184).
"""
import torch
import torch.distributed as dist


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def all_reduce_mean_(flat):
    """In-place mean over ranks of one flat buffer (one collective per optimiser step: 17.4 MB for the G1 model)."""
    w = world_size()
    if w > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / w)
    return flat


def all_reduce_sum_async(flat):
    """Start an in-place sum over ranks of one gradient bucket; returns the work handle (None on one rank).  The collective
    is ordered after the kernels already enqueued on the current stream (process-group semantics) and runs on the
    backend's own stream, i.e. beside whatever the caller enqueues next."""
    if dist.is_available() and dist.is_initialized():
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
    return None


def wait_all(handles):
    """Make the current stream wait for the collectives started by all_reduce_sum_async (no host block with nccl)."""
    for h in handles:
        if h is not None:
            h.wait()


def all_reduce_sum_(*tensors):
    """In-place sums over ranks (normaliser statistics); returns the world size so callers can scale their counts."""
    w = world_size()
    if w > 1:
        for t in tensors:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return w


def broadcast_(flat, src=0):
    if world_size() > 1:
        dist.broadcast(flat, src)
    return flat
