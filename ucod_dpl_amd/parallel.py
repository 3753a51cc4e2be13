"""Data-parallel plumbing of the hot path: one process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL over
xGMI on ROCm; ``gloo`` in the CPU tests).  The path is pure data parallel over images:

* ``broadcast_state``: rank 0's flat parameter arenas / BN buffers to everyone once (what the DDP constructor did in the
  reference before the wrapper was thrown away, engine/runner/runner.py:357-369);
* ``allreduce_prescaled_``: ONE all-reduce(SUM) of the flat gradient arena per step.  The gradient kernels already scale
  by 1/world (the ``gscale`` / ``gextra`` arguments of ``ucod_apm_bce`` / ``ucod_dba_bwd``), so the sum IS the mean over
  the global batch and no extra pass touches the buffer.  395 KB at C=768: latency-bound, not link-bound;
* ``allreduce_prescaled_async``: the same collective issued asynchronously.  With ``nccl`` the collective runs on the
  process group's own RCCL stream behind an event recorded on the launch stream (so it starts when the kernels that
  produced the buffer have finished) and ``handle.wait()`` makes the CURRENT stream wait for it without blocking the
  host: everything enqueued on the launch stream between issue and wait (the backbone backward in LoRA mode) overlaps
  the xGMI transfer.  With ``gloo`` (CPU tests) ``wait()`` blocks the host; same call sequence, same result;
* ``max_over_ranks``: timing reduction used by bench.py.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def device_index():
    """GPU of this rank: LOCAL_RANK (one process per GPU).  UCOD_SINGLE_DEVICE=1 (test rigs with ONE GPU) puts every rank on device 0,
    which together with UCOD_DIST_BACKEND=gloo lets the whole multi-rank path -- broadcast at construction, the asynchronous
    all-reduce inside the step, bench.py's launcher and max-over-ranks timing -- run on a one-GPU box (RCCL itself refuses two ranks
    on one device; the product launch uses neither variable)."""
    return 0 if os.environ.get("UCOD_SINGLE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))


def cap_host_threads(world=None):
    """One process per GPU on one host: every rank otherwise inherits torch's default intra-op pool (= all cores: 8 ranks x 128 threads
    on the MI355X boxes), and the host side of this path is a single-threaded launch loop.  cores // world threads per rank (at least
    one); UCOD_RANK_THREADS overrides.  Returns the count in force."""
    world = world if world is not None else env_world()[2]
    if world > 1 or os.environ.get("UCOD_RANK_THREADS"):
        try:
            avail = len(os.sched_getaffinity(0))                # after pin_rank_cores: this rank's own share
        except AttributeError:
            avail = os.cpu_count() or 1
        # pinned by pin_rank_cores: the affinity IS this rank's share.  Not pinned (UCOD_NO_PIN, too few cores, a cpuset-limited container): the
        # affinity is shared by every rank of this host -- divide it
        share = avail if _PINNED else max(1, avail // _local_world(world))
        n = int(os.environ.get("UCOD_RANK_THREADS") or max(1, share))
        torch.set_num_threads(n)
    return torch.get_num_threads()


_PINNED = False                                                  # set by pin_rank_cores when it has narrowed this process's affinity


def _local_world(world):
    """Ranks on THIS host: LOCAL_WORLD_SIZE as torchrun exports it (two nodes x 8 ranks: 8, not 16), else the global world size."""
    try:
        lw = int(os.environ.get("LOCAL_WORLD_SIZE", "0"))
    except ValueError:
        lw = 0
    return lw if 0 < lw <= world else world


def pin_rank_cores(local_rank=None, world=None):
    """Give this rank its own share of the host's cores (``os.sched_setaffinity``), BEFORE anything initialises the GPU runtime: eight ranks
    on one host otherwise float over all cores -- each rank's single launch thread, its HIP runtime threads and the other ranks' pools
    migrate over each other (scripts/launch_train_first_stage.sh:20-40 leaves this to accelerate / the OS).  Rank r of ``world`` takes the
    r-th contiguous slice of the cores this process may run on; UCOD_NO_PIN=1 disables.  Returns the number of cores in force."""
    rank, lr, w = env_world()
    local_rank = lr if local_rank is None else local_rank
    world = w if world is None else world
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:                                      # not Linux
        return os.cpu_count() or 1
    local_world = _local_world(world)                           # the host's cores are shared by the ranks of THIS host only
    if world <= 1 or os.environ.get("UCOD_NO_PIN") == "1" or len(cores) < local_world or local_rank >= local_world:
        return len(cores)
    per = len(cores) // local_world
    mine = cores[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    global _PINNED
    _PINNED = True
    return len(mine)


def ranks_seen(device):
    """[(rank, local_rank, device index, PCI bus id or device name)] of every rank, identical on all of them: what bench.py's ``--gpus N`` line
    carries so that a reader can verify N distinct devices took part (and RCCL membership: the gather itself is a collective)."""
    rank, local_rank, world = env_world()
    idx = device.index if getattr(device, "index", None) is not None else device_index()
    try:
        props = torch.cuda.get_device_properties(idx)
        where = "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", -1) & 0xFF, getattr(props, "pci_device_id", 0) & 0xFF) \
            if hasattr(props, "pci_bus_id") else str(getattr(props, "uuid", props.name))
    except Exception as e:                                       # noqa: BLE001
        where = f"unknown ({type(e).__name__})"
    mine = [rank, local_rank, int(idx), where]
    if world_size() == 1:
        return [mine]
    out = [None] * world_size()
    dist.all_gather_object(out, mine)
    return out


def init_from_env(backend="nccl", timeout_s=None):
    """Initialise the default process group from torchrun's environment (no-op for world_size 1).  A rendezvous that does not complete
    within ``timeout_s`` (default 300 s, UCOD_DIST_TIMEOUT_S) raises with the addresses it was waiting on instead of hanging: the usual
    causes are a MASTER_ADDR that does not resolve inside the container (use 127.0.0.1 on one node) or fewer ranks started than
    WORLD_SIZE announces.  HSA_ENABLE_IPC_MODE_LEGACY=0 is required for RCCL between processes on this driver (dmabuf IPC)."""
    import datetime
    rank, local_rank, world = env_world()
    backend = os.environ.get("UCOD_DIST_BACKEND", backend)
    if world == 1 and os.environ.get("UCOD_FORCE_DIST") == "1":
        force_single_rank_group(backend)
        return rank, local_rank, world
    if world > 1 and not dist.is_initialized():
        cap_host_threads(world)
        timeout_s = float(os.environ.get("UCOD_DIST_TIMEOUT_S", timeout_s or 300))
        if backend == "nccl":
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if not torch.cuda.is_available():
                raise RuntimeError("backend nccl (= RCCL) needs a GPU per rank; set UCOD_DIST_BACKEND=gloo for CPU / single-GPU rigs")
            if torch.cuda.device_count() <= device_index():
                raise RuntimeError(f"rank {rank}: LOCAL_RANK {local_rank} but this node exposes {torch.cuda.device_count()} GPU(s) "
                                   f"(one process per GPU: launch with --nproc-per-node <= the GPU count)")
        where = f"{os.environ.get('MASTER_ADDR', '?')}:{os.environ.get('MASTER_PORT', '?')}"
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
        except Exception as e:                                  # noqa: BLE001 -- re-raised with the rendezvous it was waiting on
            raise RuntimeError(f"rank {rank}/{world}: torch.distributed ({backend}) did not initialise against {where} within {timeout_s:.0f} s: "
                               f"{type(e).__name__}: {e}") from e
    return rank, local_rank, world


_FORCED = False                                                  # force_single_rank_group(): collectives run although the world is one rank


def force_single_rank_group(backend="nccl"):
    """UCOD_FORCE_DIST=1 on ONE GPU: a real world-size-1 process group (``nccl`` = RCCL: loads librccl, creates the communicator) and NO world-size-1
    short-circuit afterwards -- ``broadcast_state``, ``allreduce_prescaled_async`` / ``handle.wait()``, ``max_over_ranks`` and ``barrier`` issue the
    collectives exactly as they do at 8 ranks (engine/runner/runner.py:349-372 of the reference: accelerator.prepare + DDP).  What a one-GPU box can prove
    of the multi-GPU path: the library loads, the communicator comes up, the asynchronous all-reduce on the group's own stream is ordered against the
    gradient kernels before it and the AdamW launch after it, and what it costs per step (tests/test_gpu_multirank.py::test_nccl_*, bench.py's
    ``collective`` object).  Rendezvous through a file store: no port, no host name."""
    global _FORCED
    import datetime
    import tempfile
    if not dist.is_available():
        raise RuntimeError("torch.distributed is not available in this build")
    if not dist.is_initialized():
        if backend == "nccl":
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if not torch.cuda.is_available():
                raise RuntimeError("backend nccl (= RCCL) needs a GPU; set UCOD_DIST_BACKEND=gloo on a CPU rig")
        store = os.path.join(tempfile.mkdtemp(prefix="ucod_dist_"), "store")
        dist.init_process_group(backend=backend, init_method=f"file://{store}", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120))
    _FORCED = True


def collectives_on():
    """True when the data-path collectives are to be issued: more than one rank, or a forced single-rank group (UCOD_FORCE_DIST=1)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCED)


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def broadcast_state(tensors, src=0):
    if collectives_on():
        for t in tensors:
            dist.broadcast(t, src=src)


def allreduce_prescaled_(flat):
    """In-place SUM over ranks of a flat buffer whose contents were already scaled by 1/world."""
    if collectives_on():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


class _Done:
    """Handle of a collective that did not have to run (world_size 1)."""

    def wait(self):
        return True


def allreduce_prescaled_async(flat):
    """Issue the SUM all-reduce of a pre-scaled flat buffer and return a handle; ``handle.wait()`` before the first consumer
    (the optimiser launch).  The buffer must not be written between issue and wait."""
    if collectives_on():
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
    return _Done()


def grad_prescale():
    return 1.0 / world_size()


def max_over_ranks(value, device):
    if not collectives_on():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def barrier():
    if collectives_on():
        dist.barrier()


def padded_sampler_len(loader):
    """len(dataset) when ``loader`` walks a DistributedSampler that pads its tail with repeats (no drop_last), else None: what
    ``statistics.gather_records`` needs to drop those repeats (the reference's gather_for_metrics does)"""
    sampler = getattr(loader, "sampler", None)
    if getattr(sampler, "num_replicas", None) is None or getattr(sampler, "drop_last", False) or getattr(sampler, "shuffle", False):
        return None
    try:
        return len(loader.dataset)
    except TypeError:
        return None


def shard(loader):
    """This rank's share of a validation loader.  A plain ``DataLoader`` over a map-style dataset with a sequential sampler is REBUILT over
    ``Subset(dataset, range(rank, N, world))`` (same batch size, collate function and workers): a rank then loads and decodes only its own
    images (an ``islice`` over the original loader materialises every batch on every rank and throws (world - 1) / world of them away).
    A loader that is already sharded (its sampler carries ``num_replicas``: a DistributedSampler, what accelerator.prepare makes of it in
    the reference, engine/runner/runner.py:372) is walked as it is -- such a sampler pads its last batches with repeats unless built with
    drop_last, and the reference's gather_for_metrics drops them: ``statistics.gather_records`` truncates to the dataset length
    for the same reason.  Anything else (an iterable, a custom sampler) falls back to batches rank, rank + world, ...  World size 1: the
    loader itself."""
    import itertools
    world = world_size()
    if world == 1 or getattr(getattr(loader, "sampler", None), "num_replicas", None) is not None:
        return loader
    rank = dist.get_rank()
    try:
        from torch.utils.data import BatchSampler, DataLoader, SequentialSampler, Subset
        # only a loader torch built itself from (dataset, batch_size): a custom batch_sampler reports batch_size None and would come back un-batched
        if isinstance(loader, DataLoader) and isinstance(loader.sampler, SequentialSampler) and loader.batch_size is not None \
                and type(loader.batch_sampler) is BatchSampler and hasattr(loader.dataset, "__getitem__") and hasattr(loader.dataset, "__len__"):
            sub = Subset(loader.dataset, range(rank, len(loader.dataset), world))
            extra = {}
            if loader.num_workers > 0:                             # (worker options are only legal with workers)
                extra = dict(persistent_workers=loader.persistent_workers, prefetch_factor=loader.prefetch_factor, worker_init_fn=loader.worker_init_fn)
            return DataLoader(sub, batch_size=loader.batch_size, shuffle=False, num_workers=loader.num_workers, collate_fn=loader.collate_fn,
                              pin_memory=loader.pin_memory, drop_last=loader.drop_last, generator=loader.generator, **extra)
    except ImportError:
        pass
    return itertools.islice(loader, rank, None, world)
