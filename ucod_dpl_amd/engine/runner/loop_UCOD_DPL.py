"""Training / discriminator loops -- host-side mirror of engine/runner/loop_UCOD_DPL.py::TrainLoop.

Same class and method names (``_process_batch``, ``merge_pseudo_label``, ``update_ema_decoder``,
``Discriminator_train``, ``Discriminator_epoch``, ``run_epoch``, ``run``, ``decide_to_*``) and the same step
semantics, including the reference's quirks: ``global_step`` advances twice per batch
(:143,182) so the EMA ramp ``alpha = min(1 - 1/(global_step+1), ema_weight)`` (:187) is twice as fast; StepLR is
stepped once per batch (:179); the APM weight denominator is ``max_epoch + start_finetune`` (:266);
``loss -= dis_loss`` has no gradient path (:167-169); the discriminator's BatchNorm always uses batch
statistics and mutates its running buffers on both APM calls (:260-261).

What is different is *how* the step runs: one pass of fused HIP launches over flat parameter arenas, no
autograd graph, no host synchronisation (the reference syncs four times per step to format log strings,
:169,175,263-264,269 -- here scalars stay on the device unless ``log_scalars`` is on), one RCCL all-reduce of
the flat gradient buffer per step when world_size > 1 (the reference's DDP wrapper is discarded and never
reduces anything, engine/runner/runner.py:368-369 -- documented deviation).
"""
import math

import os

import torch

from ... import ops, parallel
from ...models.modules.DBA import EMB


# ------------------------------------------------------------------------------------------------ arenas
class DecoderArena:
    """Flat f32 storage for the student decoder (params, grads, Adam moments) and the EMA teacher.
    Layout: emb[128] | W[128*C] | bias[128] | head_w[128] (fg|bg) | head_b[2] (fg|bg)."""

    def __init__(self, model, device):
        dec, ema = model.decoder, model.decoder_ema
        C = dec.decoupling.weight.shape[1]
        self.C = C
        self.o_emb, self.o_W, self.o_b = 0, 128, 128 + 128 * C
        self.o_hw, self.o_hb = self.o_b + 128, self.o_b + 256
        self.n = self.o_hb + 2
        self.device = device
        mk = lambda: torch.zeros(self.n, dtype=torch.float32, device=device)  # noqa: E731
        self.p, self.g, self.m, self.v, self.ema = mk(), mk(), mk(), mk(), mk()
        self._adopt(dec, self.p)
        self._adopt(ema, self.ema)
        self.Wcat = torch.empty(256, C, dtype=torch.float32, device=device)
        self.bcat = torch.empty(256, dtype=torch.float32, device=device)

    def _views(self, flat):
        C = self.C
        return {"learnable_embedding": flat[self.o_emb:self.o_W].view(2, EMB),
                "decoupling.weight": flat[self.o_W:self.o_b].view(128, C, 1, 1),
                "decoupling.bias": flat[self.o_b:self.o_hw],
                "conv_out_fg.weight": flat[self.o_hw:self.o_hw + 64].view(1, EMB, 1, 1),
                "conv_out_bg.weight": flat[self.o_hw + 64:self.o_hb].view(1, EMB, 1, 1),
                "conv_out_fg.bias": flat[self.o_hb:self.o_hb + 1],
                "conv_out_bg.bias": flat[self.o_hb + 1:self.o_hb + 2]}

    def _adopt(self, decoder, flat):
        """Copy the module's parameters into the arena and re-point them at arena views (state_dict, load_state_dict,
        checkpoint names keep working; the arena is the storage)."""
        params = dict(decoder.named_parameters())
        for name, view in self._views(flat).items():
            view.copy_(params[name].data.to(self.device))
            params[name].data = view

    def slices(self, flat):
        return (flat[self.o_emb:self.o_W], flat[self.o_W:self.o_b].view(128, self.C), flat[self.o_b:self.o_hw],
                flat[self.o_hw:self.o_hb], flat[self.o_hb:self.o_hb + 2])

    def refresh_shared_projection(self):
        """student rows 0..127 | teacher rows 128..255 of ONE projection (the 147 968 x 768 feature read is shared)."""
        _, Ws, bs, _, _ = self.slices(self.p)
        _, Wt, bt, _, _ = self.slices(self.ema)
        ops.copy_segments([(self.Wcat[:128], Ws), (self.Wcat[128:], Wt), (self.bcat[:128], bs), (self.bcat[128:], bt)])


class DiscArena:
    """Flat storage for the discriminator's parameters (11; 14 with the feature branch), their grads and Adam moments (reference order)."""

    def __init__(self, disc, device):
        self.disc = disc
        plist = disc._param_list()
        self.sizes = [p.numel() for p in plist]
        self.n = sum(self.sizes)
        mk = lambda: torch.zeros(self.n, dtype=torch.float32, device=device)  # noqa: E731
        self.p, self.g, self.m, self.v = mk(), mk(), mk(), mk()
        off = 0
        self.grad_views = []
        for p, n in zip(plist, self.sizes):
            view = self.p[off:off + n].view(p.shape)
            view.copy_(p.data.to(device))
            p.data = view
            self.grad_views.append(self.g[off:off + n].view(p.shape))
            off += n
        # buffers to the device too; the three num_batches_tracked counters become views of ONE tensor, so that a forward call bumps them
        # with one launch (discriminator._bump_num_batches) instead of three
        blocks = disc._blocks()
        disc._nbt = torch.stack([b.layers[1].num_batches_tracked.data.to(device) for b in blocks])
        for i, b in enumerate(blocks):
            bn = b.layers[1]
            bn.running_mean.data = bn.running_mean.data.to(device)
            bn.running_var.data = bn.running_var.data.to(device)
            bn.num_batches_tracked.data = disc._nbt[i]


class FusedAdamW:
    """torch.optim.AdamW(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=0.01) over one arena, one launch
    (engine/runner/runner.py:282-298).  ``step(ema=..., alpha=...)`` also applies the EMA update in the same pass."""

    def __init__(self, arena_p, arena_g, arena_m, arena_v, lr):
        self.p, self.g, self.m, self.v = arena_p, arena_g, arena_m, arena_v
        self.param_groups = [{"lr": lr, "initial_lr": lr}]
        self.t = 0

    def zero_grad(self):
        pass                                                  # every gradient kernel overwrites its slice

    def step(self, ema=None, alpha=0.0):
        self.t += 1
        ops.adamw_ema(self.p, self.g, self.m, self.v, ema, self.param_groups[0]["lr"], self.t, ema_alpha=alpha)


class StepLR:
    def __init__(self, optimizer, step_size, gamma):
        self.opt, self.step_size, self.gamma, self.n = optimizer, step_size, gamma, 0

    def step(self):
        self.n += 1
        g = self.opt.param_groups[0]
        g["lr"] = g["initial_lr"] * self.gamma ** (self.n // self.step_size)


# ------------------------------------------------------------------------------------------------ loops
class BaseLoop:
    def __init__(self, config, runner):
        self.cfg = config
        self._runner = runner
        self._dist_train = self.cfg.train_cfg.dist_train

    @property
    def runner(self):
        return self._runner


class TrainLoop(BaseLoop):
    def __init__(self, config, runner):
        super().__init__(config, runner)
        self._mode = "train"
        tc = self.cfg.train_cfg
        self._start_epoch = tc.start_epoch
        self._max_epoch = tc.max_epoch
        self.global_step = 0
        self._cur_epoch = 0
        self._start_finetune = tc.start_finetune
        self.finetune = False
        self.merge_alpha = tc.merge_alpha
        self.ema_alpha = self.cfg.model_cfg.ema_weight
        vc = self.cfg.val_cfg
        self.enable_val = vc.enable_val
        self.val_interval = vc.val_interval
        self.dis_intertrain = tc.dis_intertrain
        self.val_start = (self._max_epoch + vc.start_val) if vc.start_val < 0 else vc.start_val
        sc = tc.save_cfg
        self.save_start = (self._max_epoch + sc.start_save) if sc.start_save < 0 else sc.start_save
        self.save_interval = sc.save_interval
        self.log_interval = self.cfg.log_cfg.log_interval
        self.log_scalars = bool(self.cfg.log_cfg.get("log_scalars", False))   # build-only key: per-step .item() logging
        self.best_mae = 1000.0
        self.best_result = None
        self.last = {}

    # ------------------------------------------------------------------ one optimiser step (loop_UCOD_DPL.py:148-184)
    def _process_batch(self, batch_data):
        if isinstance(batch_data, dict):
            pseudo_labels, _, features, _ = batch_data.values()
        else:
            pseudo_labels, features = batch_data
        r = self.runner
        A = r.arena
        world = r.world_size
        dev = A.device
        fs = self.cfg.model_cfg.feature_size
        features = features.to(dev, torch.float32)
        pseudo_labels = pseudo_labels.to(dev, torch.float32)
        B = features.shape[0]
        pl = ops.bilinear_resize(pseudo_labels, fs, fs)                                                               # :154
        # ONE zero launch for every accumulating output of the step (the diagonal sums of the heads kernel, the loss cell of the APM kernel, the
        # gradient arena behind the embedding slot: weight gradient | bias | head_w | head_b); the discriminator's BatchNorm sums live in its saved
        # buffer and are left zero by every forward call.  Round 4: seven rocclr fills per step.
        scratch = self._step_scratch(B, dev)
        ops.zero_segments([scratch, A.g[A.o_W:]])
        with ops.prezeroed():
            return self._process_batch_zeroed(pseudo_labels, features, pl, scratch)

    def _step_scratch(self, B, dev):
        """[sdiag B | losses 4] f32.  TWO buffers used alternately: the step's loss tensors (``self.last['dis_loss']``, the diagonal sums) are views of the
        buffer of THAT step and stay valid while the next step is enqueued and runs -- only the step after that zeroes the buffer again (with one
        buffer a caller that read ``last`` after the next ``_process_batch`` had been enqueued saw 0 or the next step's value)."""
        ring = getattr(self, "_scratch", None)
        if ring is None or ring[0].numel() != B + 4 or ring[0].device != dev:
            self._scratch = ring = [torch.empty(B + 4, dtype=torch.float32, device=dev) for _ in range(2)]
            self._scratch_i = 0
        self._scratch_i ^= 1
        return ring[self._scratch_i]

    def _process_batch_zeroed(self, pseudo_labels, features, pl, scratch):
        r = self.runner
        A = r.arena
        world = r.world_size
        dev = A.device
        fs = self.cfg.model_cfg.feature_size
        B = features.shape[0]

        # teacher (no grad) and student share one projection (:156-158).  The reference resizes the 768-channel features to
        # fs x fs first (:153) and then applies the 1x1 conv; both are linear and act on different axes, so they commute:
        # project on the native h x w grid (3.4x fewer FLOPs at 37->68) and resize the 256-channel result instead.
        A.refresh_shared_projection()
        emb_s, _, _, hw_s, hb_s = A.slices(A.p)
        emb_t, _, _, hw_t, hb_t = A.slices(A.ema)
        features = features.contiguous()
        fh, fw = features.shape[-2:]
        native_grid = (fh, fw) != (fs, fs)
        d = ops.dba_project(features, A.Wcat, A.bcat)
        if native_grid:
            d = ops.bilinear_resize(d.view(B, 256, fh, fw), fs, fs).view(B, 256, fs * fs)
        norm_s = ops.dba_colnorm(d, 0, emb_s)
        norm_t = ops.dba_colnorm(d, 128, emb_t)
        fg, bg, sdiag = ops.dba_heads(d, 0, emb_s, norm_s, hw_s, hb_s, want_bg=True, want_sdiag=True, sdiag=scratch[:B])
        teacher, _, _ = ops.dba_heads(d, 128, emb_t, norm_t, hw_t, hb_t, want_bg=False)
        extra, gram = ops.orth_gram(d, 0, emb_s, norm_s, sdiag)

        # APM (:257-272) + both BCE losses and their gradients (:161-173)
        p_s, p_p = self._disc_probs(ops.binarize(fg, logits=True).view(B, 1, fs, fs), ops.binarize(pl, logits=False), features, fs)
        epoch_frac = self._cur_epoch / (self._max_epoch + self._start_finetune)
        w, merged, gfg, gbg, losses = ops.apm_bce(pl.view(B, -1), teacher, fg, bg, p_s, p_p, epoch_frac, gscale=1.0 / world, losses=scratch[B:B + 4])

        # backward through heads / gate / normalisation / orthogonality loss, then the 1x1 conv weight gradient
        g_emb, g_W, g_b, g_hw, g_hb = A.slices(A.g)
        gd, _, _, _ = ops.dba_bwd(d, 0, emb_s, norm_s, hw_s, gram, gfg, gbg, 1.0 / world, g_head_w=g_hw, g_head_b=g_hb, g_dec_bias=g_b)
        if native_grid:                                       # pull the gradient back through the resize (its transpose)
            gd = ops.bilinear_resize_adjoint(gd.view(B, 128, fs, fs), fh, fw).view(B, 128, fh * fw)
        ops.dba_wgrad(gd, features, gW=g_W)
        # RCCL: one flat 128C+386-float buffer, pre-scaled by 1/world, issued asynchronously on the process group's stream; the
        # loss scalars below are assembled meanwhile and the optimiser launch is the first consumer.
        reduced = parallel.allreduce_prescaled_async(A.g)
        loss = ops.step_loss(losses, extra, self.finetune)     # losses[0] + losses[1] + extra (- dis loss unless finetune), one launch

        # AdamW + StepLR + EMA (:178-181,186-191)
        alpha = min(1 - 1 / (self.global_step + 1), self.ema_alpha)
        reduced.wait()
        r.optimizer.step(ema=A.ema, alpha=alpha)
        r.lr_scheduler.step()
        self.global_step += 1

        self.last = dict(loss=loss, dis_loss=losses[2], extra=extra[0], w=w, merged=merged, fg=fg, bg=bg, teacher=teacher, p_s=p_s, p_p=p_p)
        if self.log_scalars:
            r.logger.log("train/dis_loss:{:.4f}".format(losses[2].item()))
            r.logger.log("iter{}:loss:{:.4f}".format(self.global_step - 1, loss.item()))
        return loss

    # ------------------------------------------------------------------ training straight from images (no feature cache)
    def run_images(self, batches, feature_extractor, streams=2):
        """Drive ``_process_batch`` from an iterable of ``(pseudo_labels, images)`` batches: the frozen backbone pass of batch k+1
        runs on side HIP streams while the decoder step of batch k runs on the current stream (engine/runner/pipeline.py).
        ``feature_extractor``: the ``backbone`` wrapper or a ``ViTEngine``.  Yields the loss tensor of every step, in order;
        numerically identical to calling ``feature_extractor`` then ``_process_batch`` serially."""
        from .pipeline import FeaturePipeline
        engine = getattr(feature_extractor, "engine", feature_extractor)
        engine.streams = streams
        pipe = FeaturePipeline(engine)
        it = iter(batches)
        try:
            cur = next(it)
        except StopIteration:
            return
        pipe.submit(cur[1].to(engine.device))
        while cur is not None:
            nxt = next(it, None)
            key = pipe.next_features()
            if nxt is not None:
                pipe.submit(nxt[1].to(engine.device))
            loss = self._process_batch((cur[0], key))
            self.global_step += 1                             # run_epoch's increment (loop_UCOD_DPL.py:143)
            yield loss
            cur = nxt
        engine.check_overflow(wait=True)                      # the LAST passes' saturation report (the polls inside forward() see finished passes only)

    # ------------------------------------------------------------------ backbone-backward mode
    def attach_lora_backbone(self, engine, lr=None):
        """Switch the loop to the end-to-end mode models/modules/full_model.py describes: the student's features come from a
        LoRA backbone that is trained with the decoder, the teacher's from its EMA copy (full_model.py:84,108-111).
        ``engine`` is a ``ViTLoRAEngine``; the EMA engine shares its frozen weights.  LoRA matrices get their own fused
        AdamW (same hyper-parameters as the decoder's: the reference ships no loop for this mode) with the EMA update of
        the teacher's copy folded into the same launch."""
        r = self.runner
        if getattr(r.discriminator, "use_features", False):
            # In the reference `loss -= dis_loss` with dis_loss = BCE(D(mask, features), 0) (loop_UCOD_DPL.py:160-169, 257-272): with a TRAINED backbone that term
            # has a gradient path into the features through featureConv, which this loop does not build (it hands the discriminator detached features).
            raise NotImplementedError("backbone-backward mode with dis_use_features=True is not supported: d(dis_loss)/d(features) through the discriminator's "
                                      "featureConv is not implemented (use dis_use_features=False, as every shipped config does)")
        self.lora_engine = engine
        self.lora_engine_ema = engine.clone_for_ema()
        # stream budget of the forward phase: student in `engine.train_streams` image-parallel halves + the teacher on one more
        # stream (more than three concurrent passes lose to cache and CU contention: 602 vs 668 images/s with 2 + 2)
        self.lora_engine_ema.train_streams = int(os.environ.get("UCOD_TEACHER_STREAMS", "1"))
        engine.train_streams = int(os.environ.get("UCOD_STUDENT_STREAMS", str(getattr(engine, "train_streams", 2))))
        n = engine.lora.numel()
        mk = lambda: torch.zeros(n, dtype=torch.float32, device=engine.lora.device)  # noqa: E731
        self.lora_optimizer = FusedAdamW(engine.lora.view(-1), engine.lora_grad.view(-1), mk(), mk(),
                                         lr if lr is not None else r.optimizer.param_groups[0]["initial_lr"])
        self.lora_lr_scheduler = StepLR(self.lora_optimizer, self.cfg.train_cfg.step_lr_size, self.cfg.train_cfg.step_lr_gamma)
        parallel.broadcast_state([engine.lora, self.lora_engine_ema.lora])
        if parallel.world_size() > 1:                         # ranks above 0 just received rank 0's matrices: rebuild the packed columns
            engine.repack()
            self.lora_engine_ema.repack()

    def _process_batch_full(self, images, pseudo_labels):
        """One optimiser step from IMAGES: LoRA backbone forward (student, saved activations) + EMA backbone forward (teacher)
        -> key hooks -> the same DBA / APM / discriminator step as ``_process_batch`` -> decoder backward -> gradient w.r.t.
        the student's key map -> backbone backward -> all-reduce of both flat gradient buffers -> both optimisers."""
        r = self.runner
        A = r.arena
        world = r.world_size
        dev = A.device
        fs = self.cfg.model_cfg.feature_size
        eng, eng_t = self.lora_engine, self.lora_engine_ema
        images = images.to(dev, torch.float32)
        pseudo_labels = pseudo_labels.to(dev, torch.float32)
        B = images.shape[0]
        # the teacher's pass has no consumer until the decoder step: run it on a side stream, concurrently with the student's
        if getattr(self, "_teacher_stream", None) is None:
            self._teacher_stream = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        ev0 = torch.cuda.Event()
        ev0.record(cur)
        tstream = cur if getattr(self, "serial_schedule", False) else self._teacher_stream     # serial_schedule: measurement passes (exclusive kernel times)
        tstream.wait_event(ev0)
        images.record_stream(tstream)
        with torch.cuda.stream(tstream):
            feat_t = eng_t.forward_nograd(images)                               # EMA teacher: no backward -> nothing saved, fp16 residual stream
            ev1 = torch.cuda.Event()
            ev1.record(tstream)
        feat_s = eng.forward_train(images)                                   # [B,C,h,w]; activations kept for backward
        cur.wait_event(ev1)
        feat_t.record_stream(cur)                                            # allocated on the teacher's stream, consumed on this one
        fh, fw = feat_s.shape[-2:]
        pl = ops.bilinear_resize(pseudo_labels, fs, fs)
        emb_s, W_s, b_s, hw_s, hb_s = A.slices(A.p)
        emb_t, W_t, b_t, hw_t, hb_t = A.slices(A.ema)
        native_grid = (fh, fw) != (fs, fs)
        d_s = ops.dba_project(feat_s, W_s, b_s)                              # 1x1 conv on the native grid, resize after (commute)
        d_t = ops.dba_project(feat_t, W_t, b_t)
        if native_grid:
            d_s = ops.bilinear_resize(d_s.view(B, 128, fh, fw), fs, fs).view(B, 128, fs * fs)
            d_t = ops.bilinear_resize(d_t.view(B, 128, fh, fw), fs, fs).view(B, 128, fs * fs)
        norm_s = ops.dba_colnorm(d_s, 0, emb_s)
        norm_t = ops.dba_colnorm(d_t, 0, emb_t)
        fg, bg, sdiag = ops.dba_heads(d_s, 0, emb_s, norm_s, hw_s, hb_s, want_bg=True, want_sdiag=True)
        teacher, _, _ = ops.dba_heads(d_t, 0, emb_t, norm_t, hw_t, hb_t, want_bg=False)
        extra, gram = ops.orth_gram(d_s, 0, emb_s, norm_s, sdiag)
        p_s, p_p = self._disc_probs(ops.binarize(fg, logits=True).view(B, 1, fs, fs), ops.binarize(pl, logits=False), feat_s.detach(), fs)
        epoch_frac = self._cur_epoch / (self._max_epoch + self._start_finetune)
        w, merged, gfg, gbg, losses = ops.apm_bce(pl.view(B, -1), teacher, fg, bg, p_s, p_p, epoch_frac, gscale=1.0 / world)
        g_emb, g_W, g_b, g_hw, g_hb = A.slices(A.g)
        gd, _, _, _ = ops.dba_bwd(d_s, 0, emb_s, norm_s, hw_s, gram, gfg, gbg, 1.0 / world, g_head_w=g_hw, g_head_b=g_hb, g_dec_bias=g_b)
        if native_grid:
            gd = ops.bilinear_resize_adjoint(gd.view(B, 128, fs, fs), fh, fw).view(B, 128, fh * fw)
        ops.dba_wgrad(gd, feat_s, gW=g_W)
        # cotangent of the key hook: the 1x1 conv transposed (same exact-f32 MFMA kernel), then the backbone backward
        if getattr(self, "_zero_c", None) is None or self._zero_c.numel() != A.C:
            self._zero_c = torch.zeros(A.C, device=dev)
        dfeat = ops.dba_project(gd.view(B, 128, fh, fw), W_s.t().contiguous(), self._zero_c).view(B, A.C, fh, fw)
        # The decoder's gradient arena is final here: its all-reduce (128C+386 floats) goes out on the RCCL stream now and
        # travels over xGMI UNDER the backbone backward; the LoRA arena (6*r*D*L floats) exists only after that backward.
        reduced_dec = parallel.allreduce_prescaled_async(A.g)
        eng.backward(dfeat)
        reduced_lora = parallel.allreduce_prescaled_async(eng.lora_grad.view(-1))
        alpha = min(1 - 1 / (self.global_step + 1), self.ema_alpha)
        reduced_dec.wait()
        r.optimizer.step(ema=A.ema, alpha=alpha)
        r.lr_scheduler.step()
        reduced_lora.wait()
        self.lora_optimizer.step(ema=eng_t.lora.view(-1), alpha=alpha)
        self.lora_lr_scheduler.step()
        eng.repack()
        eng_t.repack()
        self.global_step += 1
        loss = ops.step_loss(losses, extra, self.finetune)     # losses[0] + losses[1] + extra (- dis loss unless finetune), one launch
        self.last = dict(loss=loss, dis_loss=losses[2], extra=extra[0], w=w, merged=merged, fg=fg, bg=bg, teacher=teacher, p_s=p_s, p_p=p_p)
        return loss

    def _disc_probs(self, student_mask, pseudo_mask, features, fs):
        """the two discriminator calls of the APM merge (:259-262): the student's binarised prediction first, then the pseudo label (the order
        fixes the BatchNorm running statistics).  dis_use_features: both see the features at fs x fs (the reference resizes them first, :153)."""
        r = self.runner
        disc = r.discriminator
        B = student_mask.shape[0]
        if disc.use_features:
            if features is None:
                raise ValueError("dis_use_features=True: the APM merge needs the feature map")
            feats = features.to(student_mask.device, torch.float32).contiguous()
            if tuple(feats.shape[-2:]) != (fs, fs):
                feats = ops.bilinear_resize(feats, fs, fs)
            p_s, _ = disc.forward_features(student_mask, feats)
            p_p, _ = disc.forward_features(pseudo_mask, feats)
            return p_s, p_p
        disc_t = disc.tensor_table()
        p_s, _ = ops.disc_fwd(student_mask, disc_t, update_running=True, saved=r.disc_saved(B, fs))
        disc._bump_num_batches(disc_t)                        # (a no-op: the kernel call bumps the counters the table carries)
        p_p, _ = ops.disc_fwd(pseudo_mask, disc_t, update_running=True, saved=r.disc_saved(B, fs))
        disc._bump_num_batches(disc_t)
        return p_s, p_p

    def merge_pseudo_label(self, pseudo_labels, p_teachers, p_students, features=None):
        """Stand-alone APM fusion with the reference's signature (:257-272) -> (merged [B,1,H,W], dis_loss)."""
        r = self.runner
        B, _, H, W = pseudo_labels.shape
        p_s, p_p = self._disc_probs(ops.binarize(p_students.contiguous(), logits=True), ops.binarize(pseudo_labels.contiguous(), logits=False), features, H)
        frac = self._cur_epoch / (self._max_epoch + self._start_finetune)
        zeros = torch.zeros(B, H * W, device=pseudo_labels.device)
        _, merged, _, _, losses = ops.apm_bce(pseudo_labels.reshape(B, -1).contiguous(), p_teachers.reshape(B, -1).contiguous(), zeros, zeros, p_s, p_p, frac)
        return merged.view(B, 1, H, W), losses[2]

    def update_ema_decoder(self):
        """:186-191 as a separate call (the fused step applies it inside the optimiser launch)."""
        A = self.runner.arena
        alpha = min(1 - 1 / (self.global_step + 1), self.ema_alpha)
        A.ema.mul_(alpha).add_(A.p, alpha=1 - alpha)

    # ------------------------------------------------------------------ discriminator phase (:215-255)
    def Discriminator_train(self):
        for p in self.runner.discriminator.parameters():
            p.requires_grad = True
        for p in self.runner.model.decoder.parameters():
            p.requires_grad = False
        for _ in range(self.cfg.train_cfg.dis_epoch):
            self.Discriminator_epoch()
        for p in self.runner.discriminator.parameters():
            p.requires_grad = False
        for p in self.runner.model.decoder.parameters():
            p.requires_grad = True

    def Discriminator_epoch(self):
        for batch in self.runner.train_dataloader:
            self._discriminator_batch(batch)

    def _discriminator_batch(self, batch):
        if isinstance(batch, dict):
            pseudo_labels, _, features, _ = batch.values()
        else:
            pseudo_labels, features = batch
        r = self.runner
        A, DA = r.arena, r.disc_arena
        dev = A.device
        fs = self.cfg.model_cfg.feature_size
        features = features.to(dev, torch.float32)
        B = features.shape[0]
        features = features.contiguous()
        emb_s, W_s, b_s, hw_s, hb_s = A.slices(A.p)
        d = ops.dba_project(features, W_s, b_s)               # student only, no grad (:238-240); conv before resize (they commute)
        if features.shape[-2:] != (fs, fs):
            d = ops.bilinear_resize(d.view(B, 128, *features.shape[-2:]), fs, fs).view(B, 128, fs * fs)
        norm = ops.dba_colnorm(d, 0, emb_s)
        fg, _, _ = ops.dba_heads(d, 0, emb_s, norm, hw_s, hb_s, want_bg=False)
        preds = ops.binarize(fg, logits=True).view(B, 1, fs, fs)
        pl = ops.binarize(ops.bilinear_resize(pseudo_labels.to(dev, torch.float32), fs, fs), logits=False)           # :241
        disc = r.discriminator
        world = r.world_size
        if disc.use_features:
            feats = features if tuple(features.shape[-2:]) == (fs, fs) else ops.bilinear_resize(features, fs, fs)           # :236
            probs_pseudo, saved_p = disc.forward_features(pl, feats, save=True)                                       # :244
            probs_student, saved_s = disc.forward_features(preds, feats, save=True)                                   # :245
        else:
            t = disc.tensor_table()
            probs_pseudo, saved_p = ops.disc_fwd(pl, t, update_running=True)                                          # :244
            disc._bump_num_batches(t)
            probs_student, saved_s = ops.disc_fwd(preds, t, update_running=True)                                      # :245
            disc._bump_num_batches(t)
        # BCELoss(cat(student, pseudo), [0..0, 1..1]) mean over 2B (:246-247) and its gradient, one launch (ucod_disc_bce: torch's clamp of the
        # logarithms at -100 and its max(p (1 - p), 1e-12) denominator); the gradient carries 1 / (2 B world) for the pre-scaled all-reduce
        g_student, g_pseudo, loss = ops.disc_bce(probs_student, probs_pseudo, 1.0 / (2 * B * world))
        if disc.use_features:
            disc.backward_features(saved_p, g_pseudo, DA.grad_views, accumulate=False)
            disc.backward_features(saved_s, g_student, DA.grad_views, accumulate=True)
        else:
            ops.disc_bwd(pl, t, saved_p, g_pseudo, grads=DA.grad_views, accumulate=False)
            ops.disc_bwd(preds, t, saved_s, g_student, grads=DA.grad_views, accumulate=True)
        parallel.allreduce_prescaled_(DA.g)
        r.dis_optimizer.step()
        r.dis_lr_scheduler.step()
        self.last = dict(dis_phase_loss=loss, probs_student=probs_student, probs_pseudo=probs_pseudo)
        if self.log_scalars:
            r.logger.log("dis:loss:{:.4f}".format(loss.item()))
        return loss

    # ------------------------------------------------------------------ epoch / schedule (:94-146,193-213)
    def run_epoch(self):
        self.runner.model.train()
        for batch_data in self.runner.train_dataloader:
            loss = self._process_batch(batch_data)
            if self.log_scalars and self._cur_epoch % self.log_interval == 0:
                self.runner.logger.log(f"iter{self.global_step}:loss:{loss.item():.4f}")
            self.global_step += 1                             # second increment per batch (:143)
        for eng in (getattr(self, "lora_engine_ema", None), getattr(self, "lora_engine", None)):
            if eng is not None:                               # backbone-backward mode: the teacher's fp16-stream passes report at the epoch boundary
                eng.check_overflow(wait=True)

    def run(self):
        self.runner.logger.log(self.cfg)
        while self._cur_epoch < self._max_epoch:
            if self.decide_to_finetune():
                self.runner.start_finetune()
                self.global_step = 0
            if self.decide_to_train_dis():
                self.Discriminator_train()
            self.run_epoch()
            self._cur_epoch += 1
            if self.decide_to_save():
                self.runner.save_checkpoint(self._cur_epoch)
            if self.decide_to_val():
                self._update_best_result(self.runner.launch_val_look_twice())

    def _update_best_result(self, result):
        if result is not None and result["MAE"] < self.best_mae:
            self.best_mae, self.best_result = result["MAE"], result
            self.runner.logger.log("best result:")
            self.runner.logger.log_table({k: [round(v, 4)] for k, v in result.items()})

    def decide_to_train_dis(self):
        return self.cfg.train_cfg.merge_method == "dis" and self._cur_epoch % self.dis_intertrain == 0 and not self.finetune

    def decide_to_finetune(self):
        if self._cur_epoch == self._max_epoch + self._start_finetune:
            self.finetune = True
            return True
        return False

    def decide_to_save(self):
        return self._cur_epoch >= self.save_start and self._cur_epoch % self.save_interval == 0

    def decide_to_val(self):
        return self.enable_val and self._cur_epoch >= self.val_start and self._cur_epoch % self.val_interval == 0
