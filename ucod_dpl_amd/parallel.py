"""Data-parallel plumbing of the hot path: one process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL over
xGMI on ROCm; ``gloo`` in the CPU tests).  The path is pure data parallel over images:

* ``broadcast_state``: rank 0's flat parameter arenas / BN buffers to everyone once (what the DDP constructor did in the
  reference before the wrapper was thrown away, engine/runner/runner.py:357-369);
* ``allreduce_prescaled_``: ONE all-reduce(SUM) of the flat gradient arena per step.  The gradient kernels already scale
  by 1/world (the ``gscale`` / ``gextra`` arguments of ``ucod_apm_bce`` / ``ucod_dba_bwd``), so the sum IS the mean over
  the global batch and no extra pass touches the buffer.  395 KB at C=768: latency-bound, not link-bound;
* ``max_over_ranks``: timing reduction used by bench.py.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_from_env(backend="nccl"):
    """Initialise the default process group from torchrun's environment (no-op for world_size 1)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def broadcast_state(tensors, src=0):
    if world_size() > 1:
        for t in tensors:
            dist.broadcast(t, src=src)


def allreduce_prescaled_(flat):
    """In-place SUM over ranks of a flat buffer whose contents were already scaled by 1/world."""
    if world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def grad_prescale():
    return 1.0 / world_size()


def max_over_ranks(value, device):
    if world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def barrier():
    if world_size() > 1:
        dist.barrier()
