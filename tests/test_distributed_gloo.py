"""CPU, world_size 2 (gloo): the data-parallel plumbing of the product path (ucod_dpl_amd/parallel.py) -- parameter
broadcast from rank 0 and ONE all-reduce of the pre-scaled flat gradient buffer -- reproduces a single process that
sees the global batch as two per-rank BatchNorm groups (the parity definition of SURVEY.md 8e)."""
import os
import socket

import torch
import torch.multiprocessing as mp

from conftest import load_golden, sub, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat(d, keys):
    return torch.cat([d[k].reshape(-1) for k in keys])


KEYS = ["learnable_embedding", "decoupling.weight", "decoupling.bias", "conv_out_fg.weight", "conv_out_bg.weight", "conv_out_fg.bias", "conv_out_bg.bias"]
CFG = dict(feature_size=28, ema_weight=0.99, lr0=6e-4, dis_lr0=1e-3, step_lr_size=2, step_lr_gamma=0.95, dis_step_lr_size=2,
           dis_step_lr_gamma=0.95, max_epoch=25, start_finetune=-5)


def _rank_grads(dec, ema, disc, feats, pl, scale):
    """Per-rank gradients of the (1/world)-scaled loss, as the HIP kernels produce them (gscale = gextra = 1/world)."""
    import sys
    sys.path.insert(0, ROOT)
    from oracle import decoder as OD, train_step as OT
    from oracle.apm import merge_pseudo_label, bce_with_logits_mean
    from oracle.resize import torch_bilinear
    f = torch_bilinear(feats, 28, 28)
    p_l = torch_bilinear(pl, 28, 28)
    with torch.no_grad():
        t, _, _ = OD.rev_decoder_forward(f, ema, ema=True)
    p = {k: v.clone().requires_grad_(True) for k, v in dec.items()}
    fg, bg, extra = OD.rev_decoder_forward(f, p, orth="gram")
    with torch.no_grad():
        merged, _, _, _, _ = merge_pseudo_label(p_l, t, fg.detach(), disc, 0, 25, -5)
    loss = (bce_with_logits_mean(fg, merged) + bce_with_logits_mean(bg, 1 - merged) + extra) * scale
    g = torch.autograd.grad(loss, [p[k] for k in KEYS], allow_unused=True)
    return torch.cat([(torch.zeros_like(p[k]) if gi is None else gi).reshape(-1) for k, gi in zip(KEYS, g)])


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ucod_dpl_amd import parallel
    parallel.init_from_env("gloo")
    assert parallel.world_size() == world and abs(parallel.grad_prescale() - 1.0 / world) < 1e-15
    g = load_golden("g5_process_batch")
    dec, ema, disc = sub(g, "model0.decoder."), sub(g, "model0.decoder_ema."), sub(g, "disc0.")
    # rank 1 starts from garbage: the broadcast must overwrite it with rank 0's parameters
    flat_p = _flat(dec, KEYS).clone()
    if rank != 0:
        flat_p.normal_()
    parallel.broadcast_state([flat_p])
    assert torch.equal(flat_p, _flat(dec, KEYS))
    feats, pl = g["features0"], g["pl0"]
    lo, hi = rank * 2, rank * 2 + 2                     # images [r*B, (r+1)*B)
    grads = _rank_grads(dec, ema, {k: v.clone() for k, v in disc.items()}, feats[lo:hi], pl[lo:hi], parallel.grad_prescale())
    handle = parallel.allreduce_prescaled_async(grads)      # the call sequence of TrainLoop._process_batch: issue, (other work), wait
    assert handle.wait() in (True, None)
    t = parallel.max_over_ranks(float(rank), "cpu")
    assert t == world - 1
    parallel.barrier()
    if rank == 0:
        torch.save(grads, out)
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_global_batch_with_per_rank_bn(tmp_path):
    out = str(tmp_path / "grads.pt")
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, out), nprocs=2, join=True, start_method="spawn")
    got = torch.load(out)
    g = load_golden("g5_process_batch")
    dec, ema, disc = sub(g, "model0.decoder."), sub(g, "model0.decoder_ema."), sub(g, "disc0.")
    # the ranks run with cores // world intra-op threads (parallel.cap_host_threads): the same thread count here, so that the CPU
    # oracle's f32 reductions are summed in the same order and the comparison stays at round-off of the all-reduce alone
    from ucod_dpl_amd import parallel
    before = torch.get_num_threads()
    torch.set_num_threads(max(1, (os.cpu_count() or 1) // 2))
    try:
        ref = sum(_rank_grads(dec, ema, {k: v.clone() for k, v in disc.items()}, g["features0"][r * 2:r * 2 + 2], g["pl0"][r * 2:r * 2 + 2], 0.5)
                  for r in range(2))
    finally:
        torch.set_num_threads(before)
    assert torch.allclose(got, ref, rtol=0, atol=1e-9)
    assert got.abs().max() > 1e-4


def _val_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, ROOT)
    from ucod_dpl_amd import parallel
    from ucod_dpl_amd.engine.utils.metrics import statistics
    parallel.init_from_env("gloo")
    g = torch.Generator().manual_seed(5)
    rec = torch.rand(7, 1032, generator=g, dtype=torch.float64)
    st = statistics()
    shard = rec[:5] if rank == 0 else rec[5:]               # ragged shards: 5 images on rank 0, 2 on rank 1
    st._records = [shard[i:i + 1] for i in range(shard.shape[0])]
    st.gather_records()
    res = st.get_result()
    empty = statistics()                                     # a rank with no image at all must not hang or crash
    if rank == 0:
        empty._records = [rec[:3]]
    empty.gather_records(device="cpu")
    # parallel.shard: an unsharded loader is dealt round-robin (no duplicates, union = the whole set); a pre-sharded one is left alone
    mine = list(parallel.shard(list(range(7))))
    assert mine == list(range(7))[rank::world]

    class _Pre(list):
        sampler = type("S", (), {"num_replicas": world})()
    assert list(parallel.shard(_Pre([1, 2, 3]))) == [1, 2, 3]
    # a plain DataLoader is REBUILT over this rank's indices: the dataset is asked for them only
    from torch.utils.data import DataLoader, Dataset
    from torch.utils.data.distributed import DistributedSampler

    class _DS(Dataset):
        def __init__(self):
            self.asked = []

        def __len__(self):
            return 7

        def __getitem__(self, i):
            self.asked.append(i)
            return torch.tensor([i])
    ds = _DS()
    got = [int(x) for b in parallel.shard(DataLoader(ds, batch_size=2)) for x in b.view(-1)]
    assert got == list(range(7))[rank::world] and sorted(ds.asked) == got
    # a DistributedSampler without drop_last pads 7 images to 8: the repeat (image 0, on rank 1) is dropped by gather_records
    ds2 = _DS()
    dl = DataLoader(ds2, batch_size=2, sampler=DistributedSampler(ds2, num_replicas=world, rank=rank, shuffle=False))
    assert parallel.padded_sampler_len(dl) == 7
    st2 = statistics()
    for b in parallel.shard(dl):
        for i in b.view(-1):
            st2._records.append(rec[int(i):int(i) + 1])
    assert len(st2._records) == 4
    st2.gather_records(dataset_len=parallel.padded_sampler_len(dl))
    p2 = st2.per_image()
    assert p2.shape[0] == 7 and torch.equal(p2, torch.cat([rec[0::2], rec[1::2]]))
    torch.save((res, st.per_image(), empty.per_image()), out + str(rank))
    torch.distributed.destroy_process_group()


def test_validation_records_are_gathered_across_ragged_ranks(tmp_path):
    """Multi-rank validation: each rank steps over its own shard; gather_records() makes get_result() the measure over the WHOLE
    set, identical on every rank (rank order), with unequal and empty shards."""
    from ucod_dpl_amd.engine.utils.metrics import statistics
    out = str(tmp_path / "val")
    mp.start_processes(_val_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    rec = torch.rand(7, 1032, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    whole = statistics()
    whole._records = [rec]
    ref = whole.get_result()
    for r in range(2):
        res, per_image, small = torch.load(out + str(r))
        assert res == ref
        assert torch.equal(per_image, rec) and torch.equal(small, rec[:3])


def _pin_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, ROOT)
    from ucod_dpl_amd import parallel
    before = sorted(os.sched_getaffinity(0))
    n = parallel.pin_rank_cores()
    mine = sorted(os.sched_getaffinity(0))
    threads = parallel.cap_host_threads(world)
    parallel.init_from_env("gloo")
    seen = parallel.ranks_seen(torch.device("cpu"))
    torch.save((before, mine, n, threads, seen), out + str(rank))
    torch.distributed.destroy_process_group()


def test_eight_ranks_pin_disjoint_core_slices_and_see_each_other(tmp_path):
    """parallel.pin_rank_cores / cap_host_threads / ranks_seen with world 8 over gloo: disjoint, covering slices of the cores this process
    may use (when there are at least eight), one intra-op thread count per rank that fits its slice, and every rank reports all eight."""
    out = str(tmp_path / "pin")
    mp.start_processes(_pin_worker, args=(8, _free_port(), out), nprocs=8, join=True, start_method="spawn")
    res = [torch.load(out + str(r)) for r in range(8)]
    cores = res[0][0]
    if len(cores) >= 8:
        per = len(cores) // 8
        for r, (before, mine, n, threads, seen) in enumerate(res):
            assert mine == cores[r * per:(r + 1) * per] and n == per and 1 <= threads <= per
    for before, mine, n, threads, seen in res:
        assert sorted(s[0] for s in seen) == list(range(8))
