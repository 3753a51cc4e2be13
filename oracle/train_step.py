"""Oracle: one optimiser step of the UCOD-DPL training loop.  TEST INFRASTRUCTURE ONLY.

Restates engine/runner/loop_UCOD_DPL.py:148-191 (TrainLoop._process_batch +
update_ema_decoder), :230-255 (Discriminator_epoch, one batch) and the optimiser /
scheduler the runner builds (engine/runner/runner.py:276-308: AdamW default betas
(0.9,0.999) eps 1e-8 weight_decay 0.01; StepLR stepped once per batch).

Gradients of the student decoder come from torch autograd applied to the oracle's own
explicit forward (oracle/decoder.py) -- an independent check of the closed-form backward
the HIP kernels implement.
"""
import math
import torch

from . import decoder as D
from . import discriminator as DISC
from .apm import merge_pseudo_label, bce_with_logits_mean, bce_mean
from .resize import torch_bilinear

BETA1, BETA2, ADAM_EPS, WEIGHT_DECAY = 0.9, 0.999, 1e-8, 0.01


class AdamW:
    """torch.optim.AdamW single-tensor update (torch/optim/adamw.py, amsgrad=False)."""

    def __init__(self, params, lr):
        self.lr = lr
        self.t = 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    def step(self, params, grads):
        self.t += 1
        bc1 = 1 - BETA1 ** self.t
        bc2 = 1 - BETA2 ** self.t
        for k, g in grads.items():
            if g is None:
                continue
            p = params[k]
            p.mul_(1 - self.lr * WEIGHT_DECAY)
            self.m[k].lerp_(g, 1 - BETA1)
            self.v[k].mul_(BETA2).addcmul_(g, g, value=1 - BETA2)
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(ADAM_EPS)
            p.addcdiv_(self.m[k], denom, value=-(self.lr / bc1))


class StepLR:
    """torch.optim.lr_scheduler.StepLR: lr = lr0 * gamma ** (n_steps // step_size)."""

    def __init__(self, opt, step_size, gamma):
        self.opt, self.step_size, self.gamma, self.n, self.lr0 = opt, step_size, gamma, 0, opt.lr

    def step(self):
        self.n += 1
        self.opt.lr = self.lr0 * self.gamma ** (self.n // self.step_size)


class TrainState:
    """Everything the reference loop mutates."""

    def __init__(self, dec, ema, disc, cfg):
        self.dec = {k: v.clone() for k, v in dec.items()}
        self.ema = {k: v.clone() for k, v in ema.items()}
        self.disc = {k: v.clone() for k, v in disc.items()}
        self.cfg = cfg                                   # dict: feature_size, ema_weight, lr0, dis_lr0, step sizes, gammas, max_epoch, start_finetune
        self.global_step = 0
        self.cur_epoch = 0
        self.finetune = False
        self.build_optimizers()

    def build_optimizers(self):                          # runner.py:276-308, re-run by start_finetune (:378-379)
        c = self.cfg
        self.opt = AdamW(self.dec, c["lr0"])
        self.sched = StepLR(self.opt, c["step_lr_size"], c["step_lr_gamma"])
        self.disc_trainable = {k: v for k, v in self.disc.items() if v.is_floating_point() and "running" not in k}
        self.dis_opt = AdamW(self.disc_trainable, c["dis_lr0"])
        self.dis_sched = StepLR(self.dis_opt, c["dis_step_lr_size"], c["dis_step_lr_gamma"])


def update_ema(state):
    """loop_UCOD_DPL.py:186-191."""
    alpha = min(1 - 1 / (state.global_step + 1), state.cfg["ema_weight"])
    for k in state.ema:
        state.ema[k].mul_(alpha).add_(state.dec[k], alpha=1 - alpha)
    return alpha


def process_batch(state, features, pseudo_labels, orth="gram", extra_leaves=None, grad_sync=None):
    """loop_UCOD_DPL.py:148-184.  features [B,C,h,w], pseudo_labels [B,1,ph,pw].
    Returns dict(loss, dis_loss, extra, w, merged, fg, bg, teacher, grads, alpha, lr).
    ``extra_leaves``: tensors upstream of ``features`` (backbone-backward mode, SURVEY.md 8a row B9: the LoRA matrices,
    or ``features`` itself) whose gradients of the same loss are returned as ``extra_grads``."""
    c = state.cfg
    fs = c["feature_size"]
    feats = torch_bilinear(features, fs, fs)
    pl = torch_bilinear(pseudo_labels, fs, fs).float()
    with torch.no_grad():
        teacher, _, _ = D.rev_decoder_forward(feats, state.ema, ema=True)
    p = {k: v.clone().requires_grad_(True) for k, v in state.dec.items()}
    fg, bg, extra = D.rev_decoder_forward(feats, p, ema=False, orth=orth)
    with torch.no_grad():
        merged, dis_loss, w, p_s, p_p = merge_pseudo_label(
            pl, teacher, fg.detach(), state.disc, state.cur_epoch, c["max_epoch"], c["start_finetune"])
    loss = bce_with_logits_mean(fg.permute(0, 2, 3, 1).reshape(-1, 1), merged.permute(0, 2, 3, 1).reshape(-1, 1))
    if not state.finetune:
        loss = loss - dis_loss
    loss = loss + bce_with_logits_mean(bg.permute(0, 2, 3, 1).reshape(-1, 1), 1 - merged.permute(0, 2, 3, 1).reshape(-1, 1))
    loss = loss + extra
    extra_leaves = list(extra_leaves or [])
    allg = torch.autograd.grad(loss, list(p.values()) + extra_leaves, allow_unused=True)
    grads = dict(zip(p.keys(), allg[:len(p)]))
    extra_grads = [torch.zeros_like(t) if g is None else g for t, g in zip(extra_leaves, allg[len(p):])]
    lr_used = state.opt.lr
    if grad_sync is not None:                            # data parallel: the mean over ranks of the per-rank gradients, see run()
        grads = grad_sync(grads)
    with torch.no_grad():
        state.opt.step(state.dec, grads)
        state.sched.step()
        alpha = update_ema(state)
    state.global_step += 1                               # :182 (the caller run_epoch adds one more, :143)
    return dict(loss=loss.detach(), dis_loss=dis_loss, extra=extra.detach(), w=w, merged=merged, fg=fg.detach(),
                bg=bg.detach(), teacher=teacher, grads=grads, alpha=alpha, lr=lr_used, p_s=p_s, p_p=p_p, extra_grads=extra_grads)


def discriminator_batch(state, features, pseudo_labels, grad_sync=None):
    """loop_UCOD_DPL.py:232-252 for one batch (discriminator trainable, student frozen)."""
    c = state.cfg
    fs = c["feature_size"]
    feats = torch_bilinear(features, fs, fs)
    with torch.no_grad():
        fg, _, _ = D.rev_decoder_forward(feats, state.dec, ema=False, orth="gram")
        preds = (torch.sigmoid(fg) > 0.5).float()
    pl = (torch_bilinear(pseudo_labels, fs, fs) > 0.5).float()
    B = preds.shape[0]
    label = torch.cat((torch.zeros(B), torch.ones(B))).unsqueeze(-1)
    names = list(state.disc_trainable.keys())
    leaf = {k: state.disc[k].clone().requires_grad_(True) for k in names}
    sd = dict(state.disc)
    sd.update(leaf)
    probs_pseudo = DISC.discriminator_forward(pl, sd)
    probs_student = DISC.discriminator_forward(preds, sd)
    loss = bce_mean(torch.cat((probs_student, probs_pseudo), 0), label)
    gl = torch.autograd.grad(loss, [leaf[k] for k in names])
    grads = dict(zip(names, gl))
    if grad_sync is not None:
        grads = grad_sync(grads)
    with torch.no_grad():
        for k in sd:                                     # carry the mutated running statistics back
            if k not in leaf:
                state.disc[k] = sd[k]
        state.dis_opt.step(state.disc_trainable, grads)
        state.dis_sched.step()
    return dict(loss=loss.detach(), grads=grads, probs_student=probs_student.detach(), probs_pseudo=probs_pseudo.detach())


def run(state, loader, dis_intertrain, dis_epoch=1, merge_method="dis", on_event=None, orth="gram", grad_sync=None):
    """loop_UCOD_DPL.py:94-118 (TrainLoop.run) with :120-146 (run_epoch), :193-213 (decide_to_train_dis / decide_to_finetune) and :215-227
    (Discriminator_train) -- validation and saving left out.  ``loader``: list of (features, pseudo_labels).  ``state.cfg`` carries max_epoch and
    start_finetune.  At the finetune epoch the runner REBUILDS both optimisers and schedulers (engine/runner/runner.py:378-379 -> :276-311: fresh
    moments, step counts and learning rates) and the loop resets ``global_step`` (:101-103); the discriminator phase runs before every
    ``dis_intertrain``-th epoch while not finetuning.  ``on_event(tag)`` is called after every discriminator phase ("dis<epoch>") and after every
    epoch ("epoch<epoch>", the index of the epoch just run).  Returns the per-batch losses.
    ``grad_sync(grads) -> grads``: the data-parallel form (BASELINE configs[2] is "N ranks x B == the global batch seen as N per-rank
    BatchNorm groups"): every rank runs this function on ITS shard with its own discriminator buffers, and ``grad_sync`` replaces each gradient dict by
    the mean over ranks before the optimiser step -- what the build's one all-reduce of the pre-scaled gradient arena does."""
    c = state.cfg
    losses = []
    while state.cur_epoch < c["max_epoch"]:
        if state.cur_epoch == c["max_epoch"] + c["start_finetune"]:        # decide_to_finetune
            state.finetune = True
            state.build_optimizers()                                       # runner.start_finetune()
            state.global_step = 0
        if merge_method == "dis" and state.cur_epoch % dis_intertrain == 0 and not state.finetune:    # decide_to_train_dis
            for _ in range(dis_epoch):
                for feats, pl in loader:
                    discriminator_batch(state, feats, pl, grad_sync=grad_sync)
            if on_event:
                on_event(f"dis{state.cur_epoch}")
        for feats, pl in loader:                                           # run_epoch
            losses.append(float(process_batch(state, feats, pl, orth=orth, grad_sync=grad_sync)["loss"]))
            state.global_step += 1                                         # :143
        if on_event:
            on_event(f"epoch{state.cur_epoch}")
        state.cur_epoch += 1
    return losses
