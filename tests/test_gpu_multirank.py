"""GPU: the data-parallel path of the HIP step.

* one GPU: the gradient kernels' ``gscale`` / ``gextra`` = 1/world arguments (``ucod_apm_bce`` / ``ucod_dba_bwd`` through
  TrainLoop._process_batch with ``runner.world_size = 2``): the two half-batch gradient arenas SUM to the gradient of the global
  batch seen as two per-rank BatchNorm groups -- the same reference the world-2 gloo test uses (tests/test_distributed_gloo.py);
* two GPUs (skipped on a one-GPU lease): two real ranks over RCCL -- broadcast at construction, asynchronous all-reduce of the flat
  gradient arena inside ``_process_batch`` -- reproduce that gradient and leave both ranks with identical parameters.
"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from conftest import load_golden, sub, maxdiff, ROOT

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from test_distributed_gloo import _rank_grads  # noqa: E402  (oracle-side per-rank gradient of the 1/world-scaled loss)
from test_gpu_train_step import build  # noqa: E402


def _reference(g):
    dec, ema, disc = sub(g, "model0.decoder."), sub(g, "model0.decoder_ema."), sub(g, "disc0.")
    return sum(_rank_grads(dec, ema, {k: v.clone() for k, v in disc.items()}, g["features0"][r * 2:r * 2 + 2], g["pl0"][r * 2:r * 2 + 2], 0.5)
               for r in range(2))


def _check_against_reference(got, ref, A):
    scale = ref[A.o_W:A.o_b].abs().max().item()
    assert maxdiff(got[A.o_W:A.o_b], ref[A.o_W:A.o_b]) < 2e-3 * scale          # same bar as the G5 single-process test
    for name, lo, hi in (("bias", A.o_b, A.o_hw), ("head_w", A.o_hw, A.o_hb), ("head_b", A.o_hb, A.n)):
        d, s = maxdiff(got[lo:hi], ref[lo:hi]), ref[lo:hi].abs().max().item()
        # (the three small vectors are coherent sums over every pixel: a last-bit difference in the APM weight w -- it comes out of
        # the discriminator's train-mode BatchNorm -- shifts all pixels' targets together, and the conv bias gradient is a heavily
        # cancelling sum on top; measured 2.8e-3 / 3.4e-3 of their scale against the f32 oracle, bar 1e-2)
        assert d < 1e-6 + 1e-2 * max(s, scale), (name, d, s, scale)
    assert got[A.o_emb:A.o_W].abs().max().item() == 0.0                          # analytically gradient-free (exact zeros here)


def test_half_batches_with_gscale_half_sum_to_the_global_batch_gradient():
    g = load_golden("g5_process_batch")
    ref = _reference(g)
    total = None
    for r in range(2):
        runner, loop = build(g)                                  # fresh parameters / discriminator for each "rank"
        runner.world_size = 2                                    # -> gscale = gextra = 0.5 in the HIP kernels; no process group exists
        loop._process_batch((g["pl0"][r * 2:r * 2 + 2], g["features0"][r * 2:r * 2 + 2]))
        part = runner.arena.g.detach().cpu().clone()
        total = part if total is None else total + part
        A = runner.arena
    _check_against_reference(total, ref, A)
    # and the scaling is exactly linear: the same half batch with world_size 1 gives twice the arena, bit for bit (x0.5 is exact)
    runner1, loop1 = build(g)
    loop1._process_batch((g["pl0"][2:4], g["features0"][2:4]))
    w1 = runner1.arena.g.cpu()
    assert maxdiff(part[A.o_b:] * 2, w1[A.o_b:]) < 1e-9 + 1e-6 * w1[A.o_b:].abs().max().item()
    assert maxdiff(part[A.o_W:A.o_b] * 2, w1[A.o_W:A.o_b]) < 1e-5 * w1[A.o_W:A.o_b].abs().max().item()   # f32-atomic split-K sums


def test_discriminator_phase_gradient_scales_with_world_size():
    g = load_golden("g6_discriminator_step")
    outs = []
    for world in (1, 2):
        runner, loop = build(g)
        runner.world_size = world
        loop._discriminator_batch((g["pl"], g["features"]))
        outs.append(runner.disc_arena.g.cpu().clone())
    assert outs[0].abs().max() > 0
    assert maxdiff(outs[1] * 2, outs[0]) < 1e-5 * outs[0].abs().max().item()


def test_discriminator_bce_backward_is_finite_at_saturation():
    """torch's BCELoss backward divides by max(p(1-p), 1e-12): a probability that saturates to exactly 1.0 / 0.0 in f32 must give
    a large finite gradient, not inf/NaN in the fused AdamW moments (advisor finding, loop_UCOD_DPL.py discriminator phase)."""
    g = load_golden("g6_discriminator_step")
    runner, loop = build(g)
    with torch.no_grad():
        runner.discriminator.linear.bias.add_(200.0)             # push the head's sigmoid into saturation (the arena view is the storage)
    loss = loop._discriminator_batch((g["pl"], g["features"]))
    assert float(loop.last["probs_student"].max()) == 1.0
    assert torch.isfinite(loss) and torch.isfinite(runner.disc_arena.g).all() and torch.isfinite(runner.disc_arena.p).all()
    assert torch.isfinite(runner.disc_arena.m).all() and torch.isfinite(runner.disc_arena.v).all()


# ------------------------------------------------------------------------------------------------ two real ranks over RCCL
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _nccl_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_train_step import make_cfg
    from ucod_dpl_amd.engine.runner import StandardRunner, TrainLoop
    g = load_golden("g5_process_batch")
    runner = StandardRunner(make_cfg())                          # initialises the RCCL group, broadcasts rank 0's arenas
    dev = runner.device
    if rank == 0:                                                # only rank 0 holds the golden parameters: the broadcast must carry them
        runner.model.load_state_dict({k: v.to(dev) for k, v in sub(g, "model0.").items()}, strict=True)
        runner.discriminator.load_state_dict({k: v.to(dev) for k, v in sub(g, "disc0.").items()}, strict=True)
    from ucod_dpl_amd import parallel
    bn = [b.layers[1] for b in (runner.discriminator.maskConv, runner.discriminator.convs[0], runner.discriminator.convs[1])]
    parallel.broadcast_state([runner.arena.p, runner.arena.ema, runner.disc_arena.p] + [m.running_mean for m in bn] + [m.running_var for m in bn])
    loop = TrainLoop(runner.config, runner)
    loop._process_batch((g["pl0"][rank * 2:rank * 2 + 2], g["features0"][rank * 2:rank * 2 + 2]))
    torch.cuda.synchronize()
    torch.save((runner.arena.g.cpu(), runner.arena.p.cpu()), out + str(rank))
    # every OTHER collective of the product path, so that one two-rank run (RCCL on a 2-GPU lease, gloo on this box) has exercised them all:
    # (a) the discriminator phase: all-reduce of the discriminator's gradient arena, fused AdamW on the reduced arena
    loop._discriminator_batch((g["pl0"][rank * 2:rank * 2 + 2], g["features0"][rank * 2:rank * 2 + 2]))
    torch.cuda.synchronize()
    disc = (runner.disc_arena.g.cpu().clone(), runner.disc_arena.p.cpu().clone())
    # (b) backbone-backward mode: broadcast of the LoRA matrices at attach time, the decoder arena's all-reduce issued BEFORE the backbone
    #     backward, the LoRA arena's after it, both fused optimisers (TrainLoop._process_batch_full)
    from ucod_dpl_amd.vit_engine import ViTLoRAEngine
    from ucod_dpl_amd.data.utils.feature_extractor import random_state_dict, ARCHS
    ARCHS["mr_vit"] = (384, 6, 2, 14, 126, True)                  # 9 x 9 key map -> 28 x 28 features (the resize adjoint takes up to ~3.5x)
    eng = ViTLoRAEngine(random_state_dict("mr_vit", seed=4), heads=6, r=2, lora_alpha=4, device=dev, generator=torch.Generator().manual_seed(100 + rank),
                        lora_dropout=0.0, seed=5)                # different LoRA init per rank: the broadcast must overwrite rank 1's
    loop.attach_lora_backbone(eng)
    gi = torch.Generator().manual_seed(77)
    images = torch.randn(4, 3, 126, 126, generator=gi)[rank * 2:rank * 2 + 2].to(dev)
    loss = loop._process_batch_full(images, g["pl0"][rank * 2:rank * 2 + 2])
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    torch.save((disc, (eng.lora_grad.cpu().clone(), eng.lora.cpu().clone(), loop.lora_engine_ema.lora.cpu().clone(), runner.arena.p.cpu().clone())),
               out + "x" + str(rank))
    torch.distributed.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (a 2-GPU lease); the one-GPU box runs the gscale test above")
def test_two_ranks_over_rccl_match_the_global_batch(tmp_path):
    out = str(tmp_path / "rank")
    mp.start_processes(_nccl_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    g = load_golden("g5_process_batch")
    ref = _reference(g)
    (g0, p0), (g1, p1) = torch.load(out + "0"), torch.load(out + "1")
    assert torch.equal(g0, g1) and torch.equal(p0, p1)           # the all-reduce leaves every rank with the same arena -> same step
    runner, _ = build(g)
    _check_against_reference(g0, ref, runner.arena)
    _check_other_collectives(out)


# ------------------------------------------------------------------------------------------------ first contact with RCCL on ONE GPU
def _nccl1_worker(_, forced, out):
    """Three optimiser steps + a discriminator step + a backbone-backward step in ONE process; `forced`: UCOD_FORCE_DIST=1 -> a real world-size-1 nccl (= RCCL)
    process group and every collective of the path issued for real (ucod_dpl_amd/parallel.py: force_single_rank_group); else the world-size-1 short-circuit."""
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "UCOD_DIST_BACKEND", "UCOD_SINGLE_DEVICE"):
        os.environ.pop(k, None)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if forced:
        os.environ["UCOD_FORCE_DIST"] = "1"
    else:
        os.environ.pop("UCOD_FORCE_DIST", None)
    import sys
    import time
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_train_step import make_cfg
    from ucod_dpl_amd.engine.runner import StandardRunner, TrainLoop
    from ucod_dpl_amd import parallel
    import torch.distributed as dist
    g = load_golden("g5_process_batch")
    runner = StandardRunner(make_cfg())
    assert parallel.collectives_on() == bool(forced) and (dist.is_initialized() == bool(forced))
    if forced:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        maps = open("/proc/self/maps").read()
        assert "librccl" in maps or "libnccl" in maps, "no RCCL library mapped into the process"
    dev = runner.device
    runner.model.load_state_dict({k: v.to(dev) for k, v in sub(g, "model0.").items()}, strict=True)
    runner.discriminator.load_state_dict({k: v.to(dev) for k, v in sub(g, "disc0.").items()}, strict=True)
    bn = [b.layers[1] for b in (runner.discriminator.maskConv, runner.discriminator.convs[0], runner.discriminator.convs[1])]
    parallel.broadcast_state([runner.arena.p, runner.arena.ema, runner.disc_arena.p] + [m.running_mean for m in bn] + [m.running_var for m in bn])
    loop = TrainLoop(runner.config, runner)
    snaps = []
    for _ in range(3):
        loss = loop._process_batch((g["pl0"], g["features0"]))
        snaps.append((float(loss.item()), runner.arena.g.cpu().clone(), runner.arena.p.cpu().clone(), runner.arena.ema.cpu().clone()))
    loop._discriminator_batch((g["pl0"], g["features0"]))
    torch.cuda.synchronize()
    disc = (runner.disc_arena.g.cpu().clone(), runner.disc_arena.p.cpu().clone())
    from ucod_dpl_amd.vit_engine import ViTLoRAEngine
    from ucod_dpl_amd.data.utils.feature_extractor import random_state_dict, ARCHS
    ARCHS["mr_vit"] = (384, 6, 2, 14, 126, True)
    eng = ViTLoRAEngine(random_state_dict("mr_vit", seed=4), heads=6, r=2, lora_alpha=4, device=dev, generator=torch.Generator().manual_seed(100), lora_dropout=0.0, seed=5)
    eng.train_streams = 1                                        # one chunk: the LoRA gradient has ONE summation order (chunks race for nothing here)
    loop.attach_lora_backbone(eng)
    loop.lora_engine_ema.train_streams = 1
    images = torch.randn(4, 3, 126, 126, generator=torch.Generator().manual_seed(77)).to(dev)
    l2 = loop._process_batch_full(images, g["pl0"])
    torch.cuda.synchronize()
    lora = (float(l2.item()), eng.lora.cpu().clone(), loop.lora_engine_ema.lora.cpu().clone(), runner.arena.p.cpu().clone())
    # what the collective costs: host enqueue time of a step and the step time over 200 steps, same process
    for _ in range(20):
        loop._process_batch((g["pl0"], g["features0"]))
    torch.cuda.synchronize()
    t0, host = time.perf_counter(), 0.0
    for _ in range(200):
        h0 = time.perf_counter()
        loop._process_batch((g["pl0"], g["features0"]))
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    torch.save(dict(snaps=snaps, disc=disc, lora=lora, ms_per_step=dt / 200 * 1e3, host_enqueue_ms_per_step=host / 200 * 1e3,
                    max_over_ranks=parallel.max_over_ranks(3.25, dev)), out)
    if forced:
        parallel.barrier()
        dist.destroy_process_group()


def test_nccl_single_rank_group_runs_the_real_collectives(tmp_path):
    """first contact with RCCL on the one GPU of this box.  UCOD_FORCE_DIST=1 -> world-size-1 `nccl` process group, no world-size-1
    short-circuit: broadcast_state at construction, allreduce_prescaled_async + handle.wait() in front of the fused AdamW in every step, the synchronous
    all-reduce of the discriminator phase, the two asynchronous ones of backbone-backward mode, max_over_ranks, barrier.  A SUM over one rank is the identity, so
    losses, gradient arenas, parameters and EMA after each of three steps must equal the short-circuit path's to the run-to-run spread of the f32-atomic
    weight-gradient sums (2e-6; everything else bit for bit).  What this proves on one GPU: librccl loads, the communicator comes up, every collective
    of the three modes is issued on RCCL's stream and completes between the gradient kernels and the optimiser launch without corrupting or stalling the
    step; what it cannot prove is the ordering itself (an identity all-reduce that ran too early would leave the same bytes) -- the two-rank gloo tests and the
    skipped two-GPU test cover that.  The cost per step is recorded."""
    import json
    outs = {}
    for forced in (1, 0):
        out = str(tmp_path / f"nccl1_{forced}")
        mp.start_processes(_nccl1_worker, args=(forced, out), nprocs=1, join=True, start_method="spawn")
        outs[forced] = torch.load(out)
    a, b = outs[1], outs[0]

    def same(x, y):
        return maxdiff(x, y) <= 2e-5 * max(float(y.abs().max()), 1e-30)

    for (la, ga, pa, ea), (lb, gb, pb, eb) in zip(a["snaps"], b["snaps"]):
        assert abs(la - lb) <= 1e-5 * abs(lb) and same(ga, gb) and same(pa, pb) and same(ea, eb)
    assert torch.equal(a["snaps"][0][1][:8], b["snaps"][0][1][:8]) or same(a["snaps"][0][1], b["snaps"][0][1])
    assert all(same(x, y) for x, y in zip(a["disc"], b["disc"]))
    # backbone-backward step: the loss (computed before any update) agrees; the LoRA matrices after ONE AdamW step do not have to -- their first update is
    # lr * g / (|g| + eps), which turns the f32-atomic spread of a near-zero gradient entry into a difference of the size of the step itself
    assert abs(a["lora"][0] - b["lora"][0]) <= 1e-5 * abs(b["lora"][0]) and all(bool(torch.isfinite(x).all()) for x in a["lora"][1:])
    assert maxdiff(a["lora"][1], b["lora"][1]) <= 2.5e-3 and same(a["lora"][3], b["lora"][3])
    assert a["max_over_ranks"] == 3.25
    rec = dict(what="TrainLoop._process_batch on the G5 geometry (4 x 768 x 12 x 12 features), 200 steps, one process, one GPU: real world-size-1 RCCL group vs the short-circuit",
               nccl_ms_per_step=round(a["ms_per_step"], 4), short_circuit_ms_per_step=round(b["ms_per_step"], 4),
               nccl_host_enqueue_ms_per_step=round(a["host_enqueue_ms_per_step"], 4), short_circuit_host_enqueue_ms_per_step=round(b["host_enqueue_ms_per_step"], 4),
               equal_to_f32_atomic_spread=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "nccl_single_rank.json"), "w") as f:
        json.dump(rec, f, indent=1)
    assert a["ms_per_step"] < b["ms_per_step"] + 1.0, rec      # the collective may not cost a millisecond per step


# ------------------------------------------------------------------------------------------------ two ranks on ONE GPU (gloo)
def _check_other_collectives(out):
    """Discriminator phase and backbone-backward mode: after the collectives every rank holds the same reduced gradients and the same
    parameters, and the reduced gradients are not trivially zero."""
    (d0, l0), (d1, l1) = torch.load(out + "x0"), torch.load(out + "x1")
    for a, b in zip(d0 + l0, d1 + l1):
        assert torch.equal(a, b)
    assert d0[0].abs().max() > 0 and l0[0].abs().max() > 0 and torch.isfinite(l0[1]).all() and torch.isfinite(l0[3]).all()


def _gloo_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      UCOD_SINGLE_DEVICE="1", UCOD_DIST_BACKEND="gloo")
    _nccl_worker(rank, world, port, out)                     # same body: StandardRunner -> broadcast -> TrainLoop._process_batch -> save


def test_two_ranks_on_one_gpu_over_gloo_match_the_global_batch(tmp_path):
    """The whole multi-rank product path (StandardRunner's process group + broadcast, the asynchronous all-reduce of the flat gradient
    arena inside TrainLoop._process_batch, the fused AdamW on the reduced arena) with two real processes driving the HIP kernels on
    the ONE GPU of this box; only the transport differs from the RCCL launch (gloo moves the CUDA buffer through the host)."""
    out = str(tmp_path / "rank")
    mp.start_processes(_gloo_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    g = load_golden("g5_process_batch")
    ref = _reference(g)
    (g0, p0), (g1, p1) = torch.load(out + "0"), torch.load(out + "1")
    assert torch.equal(g0, g1) and torch.equal(p0, p1)
    runner, _ = build(g)
    _check_against_reference(g0, ref, runner.arena)
    _check_other_collectives(out)


def test_bench_launches_two_ranks_itself(tmp_path):
    """`python bench.py --gpus 2` with no torchrun environment starts the ranks itself (scripts/launch_train_first_stage.sh:20-40 in
    the reference) and rank 0 prints ONE JSON line with n_gpus = 2 and twice the global batch; here both ranks share the one GPU."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, UCOD_SINGLE_DEVICE="1", UCOD_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--lora-steps", "1",
                        "--no-cpu-baseline", "--image", "224", "--arch", "dino_vits8", "--sustain-s", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2" and d["value"] > 0
    assert d["backbone_backward_mode"]["value"] > 0 and d["discriminator_phase"]["value"] > 0
    assert sorted(r[0] for r in d["ranks_seen"]) == [0, 1] and d["host_cores_pinned"] >= 1
    assert d["sustained"]["windows"][0]["steps"] % 16 == 0 and d["sustained"]["value_sustained"] > 0 and d["sustained"]["held_clock_mhz"] > 100     # the sustained leg with two ranks


def test_eight_ranks_on_one_gpu_first_contact(tmp_path):
    """First-contact insurance for the 8-GPU node: `python bench.py --gpus 8` -- eight processes, each pinned to its slice of
    the host's cores before it touches the GPU, rendezvous on 127.0.0.1, broadcast, the asynchronous all-reduces of all three modes, max-over-
    ranks timing, ONE JSON line with eight entries in ranks_seen -- on a tiny geometry, all ranks sharing this box's one GPU over gloo
    (UCOD_SINGLE_DEVICE=1).  What it cannot show is RCCL / xGMI itself."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, UCOD_SINGLE_DEVICE="1", UCOD_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "2", "--lora-steps", "1",
                        "--no-cpu-baseline", "--image", "224", "--arch", "dino_vits8", "--sustain-s", "2"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp8" and d["value"] > 0
    assert sorted(r_[0] for r_ in d["ranks_seen"]) == list(range(8)) and sorted(r_[1] for r_ in d["ranks_seen"]) == list(range(8))
    assert d["host_cores_pinned"] == max(1, len(os.sched_getaffinity(0)) // 8)
    assert d["backbone_backward_mode"]["value"] > 0 and d["discriminator_phase"]["value"] > 0


# ------------------------------------------------------------------------------------------------ C3 as SURVEY 8(e) defines it, over a schedule
def _schedule_cfg():
    from test_gpu_train_step import _g18_cfg
    cfg = _g18_cfg()
    cfg.train_cfg["max_epoch"], cfg.train_cfg["start_finetune"] = 3, -1          # discriminator phase before epochs 0 and 2, finetune switch at epoch 2
    return cfg


def _schedule_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      UCOD_SINGLE_DEVICE="1", UCOD_DIST_BACKEND="gloo")
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ucod_dpl_amd.engine.runner import StandardRunner, TrainLoop
    from ucod_dpl_amd import parallel
    g = load_golden("g18_train_schedule")
    runner = StandardRunner(_schedule_cfg())
    dev = runner.device
    if rank == 0:
        runner.model.load_state_dict({k: v.to(dev) for k, v in sub(g, "model0.").items()}, strict=True)
        runner.discriminator.load_state_dict({k: v.to(dev) for k, v in sub(g, "disc0.").items()}, strict=True)
    bn = [b.layers[1] for b in (runner.discriminator.maskConv, runner.discriminator.convs[0], runner.discriminator.convs[1])]
    parallel.broadcast_state([runner.arena.p, runner.arena.ema, runner.disc_arena.p] + [m.running_mean for m in bn] + [m.running_var for m in bn])
    per = 4 // world
    runner.train_dataloader = [{"pseudo_label": g[f"pl{i}"][rank * per:(rank + 1) * per], "label_tensor": torch.zeros(1),
                                "features": g[f"features{i}"][rank * per:(rank + 1) * per], "img_path": ["x"]} for i in range(3)]
    loop = TrainLoop(runner.config, runner)
    loop.run()
    torch.cuda.synchronize()
    torch.save(dict(model={k: v.detach().cpu() for k, v in runner.model.state_dict().items()},
                    disc={k: v.detach().cpu() for k, v in runner.discriminator.state_dict().items()},
                    lr=runner.optimizer.param_groups[0]["lr"], dis_lr=runner.dis_optimizer.param_groups[0]["lr"], global_step=loop.global_step), out + str(rank))
    torch.distributed.destroy_process_group()


def _oracle_two_groups(g, world=2):
    """BASELINE configs[2]'s parity definition on the CPU oracle: `world` per-rank states run oracle.train_step.run in lockstep threads,
    each on its shard with its own BatchNorm buffers; before every optimiser step the gradient dicts are replaced by their mean over ranks."""
    import threading
    from oracle import train_step as OT
    cfg = dict(feature_size=12, ema_weight=0.99, lr0=6e-4, dis_lr0=1e-3, step_lr_size=2, step_lr_gamma=0.95, dis_step_lr_size=2, dis_step_lr_gamma=0.95,
               max_epoch=3, start_finetune=-1)
    states = [OT.TrainState(sub(g, "model0.decoder."), sub(g, "model0.decoder_ema."), sub(g, "disc0."), cfg) for _ in range(world)]
    per = 4 // world
    barrier, slots, errors = threading.Barrier(world), [None] * world, []

    def sync_for(r):
        def sync(grads):
            slots[r] = grads
            barrier.wait()
            mean = {k: (None if slots[0][k] is None else sum(s[k] for s in slots) / world) for k in grads}
            barrier.wait()
            return mean
        return sync

    def work(r):
        try:
            loader = [(g[f"features{i}"][r * per:(r + 1) * per], g[f"pl{i}"][r * per:(r + 1) * per]) for i in range(3)]
            OT.run(states[r], loader, dis_intertrain=2, dis_epoch=1, grad_sync=sync_for(r))
        except Exception as e:                                   # noqa: BLE001
            errors.append(e)
            barrier.abort()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    return states


def test_two_ranks_run_the_schedule_like_the_global_batch_in_two_batchnorm_groups(tmp_path):
    """BASELINE configs[2] over a schedule: two real ranks (gloo, one GPU) x 2 images run the build's TrainLoop.run() -- a discriminator epoch, two epochs,
    the finetune switch with rebuilt optimisers, another epoch -- and end with IDENTICAL decoder / EMA / discriminator parameters that equal the oracle's
    "global batch of 4 seen as two per-rank BatchNorm groups" run (gradients averaged before every optimiser step); BatchNorm running statistics stay
    per rank, in the build as in the oracle (the reference has no SyncBN and discards its DDP wrapper, engine/runner/runner.py:357-369)."""
    out = str(tmp_path / "sched")
    mp.start_processes(_schedule_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    r0, r1 = torch.load(out + "0"), torch.load(out + "1")
    for k in r0["model"]:
        assert torch.equal(r0["model"][k], r1["model"][k]), k
    for k in r0["disc"]:
        if "running" not in k:
            assert torch.equal(r0["disc"][k], r1["disc"][k]), k
    assert any(not torch.equal(r0["disc"][k], r1["disc"][k]) for k in r0["disc"] if "running" in k)        # per-rank statistics really are per rank
    g = load_golden("g18_train_schedule")
    states = _oracle_two_groups(g)
    assert r0["global_step"] == states[0].global_step and abs(r0["lr"] - states[0].opt.lr) < 1e-12 and abs(r0["dis_lr"] - states[0].dis_opt.lr) < 1e-12
    from conftest import within
    worst = {"dec": 0.0, "disc": 0.0, "bn": 0.0}
    for k, v in states[0].dec.items():
        if k != "learnable_embedding":
            worst["dec"] = max(worst["dec"], maxdiff(r0["model"]["decoder." + k], v), maxdiff(r0["model"]["decoder_ema." + k], states[0].ema[k]))
    for r, got in enumerate((r0, r1)):
        for k, v in states[r].disc.items():
            if "num_batches" in k:
                assert int(got["disc"][k]) == int(v), k
            else:
                kind = "bn" if "running" in k else "disc"
                worst[kind] = max(worst[kind], maxdiff(got["disc"][k], v))
    within("c3_schedule_decoder", worst["dec"], 5e-6)           # measured 1.2e-7 / 2.5e-7 / 1.8e-7
    within("c3_schedule_disc", worst["disc"], 5e-6)
    within("c3_schedule_bn", worst["bn"], 5e-6)
