"""Thin runner -- host-side mirror of engine/runner/runner.py::StandardRunner (:242-397) for the hot path.

Builds what the loops need (``model``, ``discriminator``, ``optimizer``/``lr_scheduler``, ``dis_optimizer``/
``dis_lr_scheduler``, ``train_dataloader``, ``logger``) and the MI355X-specific state (flat arenas, reusable
buffers, the process group).  One process per GPU: ``torch.distributed`` with backend ``nccl`` (= RCCL over xGMI
on ROCm) is initialised from the torchrun environment; parameters are broadcast from rank 0 once (what the DDP
constructor did in the reference, runner.py:357-365) and the loop all-reduces the flat gradient buffer.
``accelerate`` is not used: the reference unwraps the DDP module right after ``prepare`` (:368-369), so the only
behaviour it contributed on this path was device placement and that initial broadcast.
Checkpoints keep the reference's format: a directory ``epochN.pth/`` containing ``model.safetensors`` with the
``decoder.*`` / ``decoder_ema.*`` names (:165-185); ``load_checkpoint`` raises on failure instead of swallowing it
(:201-207, documented deviation).
"""
import os

import torch

from ... import ops, native, parallel
from ...models.uscod import baseline
from ...models.discriminator import Discriminator
from .loop_UCOD_DPL import DecoderArena, DiscArena, FusedAdamW, StepLR, TrainLoop


class Logger:
    """log / info / error / log_table on the ranks listed in ``log_cfg.multi_rank`` (engine/utils/logger.py:85-171 surface)."""

    def __init__(self, rank=0, ranks=(0,)):
        self.enabled = rank in tuple(ranks)

    def log(self, msg, *a, **k):
        if self.enabled:
            print(msg, flush=True)

    info = log
    warning = log
    error = log

    def log_table(self, table, *a, **k):
        if self.enabled:
            print(" | ".join(f"{k}={v}" for k, v in table.items()), flush=True)


class StandardRunner:
    def __init__(self, config, train_dataloader=None, val_dataloader=None, device=None):
        self.config = config
        if not torch.cuda.is_available():
            raise RuntimeError("StandardRunner needs a GPU: the hot path is HIP-only (no CPU fallback)")
        native.load()
        self.rank, self.local_rank, self.world_size = parallel.env_world()
        torch.cuda.set_device(parallel.device_index())
        self.device = torch.device("cuda", parallel.device_index()) if device is None else torch.device(device)
        parallel.init_from_env("nccl")                         # RCCL over xGMI; no-op on one GPU
        self.logger = Logger(self.rank, config.log_cfg.get("multi_rank", [0]))
        self.train_dataloader = train_dataloader if train_dataloader is not None else []
        self.val_dataloader = val_dataloader if val_dataloader is not None else []
        self._saved = {}
        self._build_model()
        self._build_optimizer()

    # ------------------------------------------------------------------ builders (runner.py:266-308)
    def _build_model(self):
        self.model = baseline(self.config.model_cfg)
        # dis_use_features=True (no shipped config): the same step with the discriminator's feature branch (models/discriminator.py:77-90) on the
        # generic convolution kernels of csrc/disc_features.hip instead of the fused small-channel ones
        self.discriminator = Discriminator(self.config.model_cfg)
        ckpt = self.config.train_cfg.get("checkpoint", None)
        if ckpt:
            self.load_checkpoint(ckpt)
        self.arena = DecoderArena(self.model, self.device)
        self.disc_arena = DiscArena(self.discriminator, self.device)
        bn = [b.layers[1] for b in self.discriminator._blocks()]
        parallel.broadcast_state([self.arena.p, self.arena.ema, self.disc_arena.p] + [m.running_mean for m in bn] + [m.running_var for m in bn])

    def _build_optimizer(self):
        tc = self.config.train_cfg
        A, DA = self.arena, self.disc_arena
        A.m.zero_(); A.v.zero_(); DA.m.zero_(); DA.v.zero_()
        self.optimizer = FusedAdamW(A.p, A.g, A.m, A.v, tc.lr0)
        self.dis_optimizer = FusedAdamW(DA.p, DA.g, DA.m, DA.v, tc.dis_lr0)
        self.lr_scheduler = StepLR(self.optimizer, tc.step_lr_size, tc.step_lr_gamma)
        self.dis_lr_scheduler = StepLR(self.dis_optimizer, tc.dis_step_lr_size, tc.dis_step_lr_gamma)

    def start_finetune(self):                                 # runner.py:378-379
        self._build_optimizer()

    def disc_saved(self, B, fs):
        key = (B, fs)
        if key not in self._saved:
            # zeroed: its tail holds the BatchNorm sums, which every ucod_disc_fwd leaves zero again (ucod_accumulators_prezeroed)
            self._saved[key] = torch.zeros(native.load().ucod_disc_saved_bytes(B, fs), dtype=torch.uint8, device=self.device)
        return self._saved[key]

    # ------------------------------------------------------------------ checkpoints (runner.py:165-240)
    def save_checkpoint(self, epoch, save_mode="model"):
        from safetensors.torch import save_file
        parallel.barrier()
        if self.rank == 0:
            path = os.path.join(self.config.log_cfg.log_path, "ckp", f"epoch{epoch}.pth")
            os.makedirs(path, exist_ok=True)
            save_file({k: v.detach().cpu().contiguous() for k, v in self.model.state_dict().items()}, os.path.join(path, "model.safetensors"))

    def load_checkpoint(self, checkpoint_path):
        from safetensors.torch import load_file
        if os.path.isdir(checkpoint_path):
            checkpoint_path = os.path.join(checkpoint_path, "model.safetensors")
        self.model.load_state_dict(load_file(checkpoint_path), strict=True)
        self.logger.info("Successfully loaded checkpoint weights from {}".format(checkpoint_path))

    # ------------------------------------------------------------------ launchers (runner.py:381-397)
    def launch_train(self):
        self.trainloop = TrainLoop(self.config, self)
        self.trainloop.run()

    def launch_val_look_twice(self):
        from .loop_look_twice import ValLoop_Look_Twice
        return ValLoop_Look_Twice(self.config, self).run()


class LocalRefineRunner(StandardRunner):
    """CORAL second stage -- host-side mirror of engine/runner/runner.py::LocalRefineRunner (:400-590): the frozen first-stage
    ``baseline`` (checkpoint from ``train_cfg.checkpoint``), the ``SparseRefiner`` built by ``SparseRefiner.from_config(model_cfg)``
    (weights from ``train_cfg.refiner_path``), ``launch_val`` -> ``LocalRefineValidationLoop`` on the HIP decoder / refiner.
    The reference's second-stage TRAINING loop is empty (``LocalRefineTrainLoop: pass``, loop_CORAL.py:38-39) and so is this one:
    the AdamW/StepLR pair over the refiner's parameters is built with the reference's hyper-parameters for parity of the object,
    ``launch_train`` raises.  No first-stage discriminator, arena or dataloader factory is needed here."""

    def __init__(self, config, train_dataloader=None, val_dataloader=None, device=None, window_features=None):
        self.refiner = None
        self.window_features = window_features
        super().__init__(config, train_dataloader, val_dataloader, device)

    def _build_model(self):
        from ...models.UDLR import SparseRefiner
        self.model = baseline(self.config.model_cfg)
        self.refiner = SparseRefiner.from_config(self.config.model_cfg)
        ckpt = self.config.train_cfg.get("checkpoint", None)
        if ckpt:
            self.load_checkpoint(ckpt)
        self._freeze_model(self.model)
        self.model.to(self.device)
        self.refiner.to(self.device)
        self.load_refiner_checkpoint()
        parallel.broadcast_state([p.data for p in self.model.parameters()] + [p.data for p in self.refiner.parameters()])

    def _freeze_model(self, model):                            # :439-444
        model.eval()
        for p in model.parameters():
            p.requires_grad_(False)

    def _build_optimizer(self):                                # :446-468 (refiner parameters only)
        tc = self.config.train_cfg
        self.optimizer = torch.optim.AdamW(self.refiner.parameters(), lr=tc.lr0)
        self.lr_scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=tc.step_lr_size, gamma=tc.step_lr_gamma)

    def save_checkpoint(self, epoch, save_mode="model"):       # :532-551: <log_path>/refiner_ckp/epochN.pth/model.safetensors
        from safetensors.torch import save_file
        parallel.barrier()
        if self.rank == 0:
            path = os.path.join(self.config.log_cfg.log_path, "refiner_ckp", f"epoch{epoch}.pth")
            os.makedirs(path, exist_ok=True)
            save_file({k: v.detach().cpu().contiguous() for k, v in self.refiner.state_dict().items()}, os.path.join(path, "model.safetensors"))

    def load_refiner_checkpoint(self, refiner_path=None):      # :553-573; raises instead of logging the failure (documented deviation)
        from safetensors.torch import load_file
        if refiner_path is None:
            refiner_path = self.config.train_cfg.get("refiner_path", None)
        if refiner_path is None:
            return
        if os.path.isdir(refiner_path):
            refiner_path = os.path.join(refiner_path, "model.safetensors")
        self.refiner.load_state_dict(load_file(refiner_path), strict=True)
        self.refiner._prepared = None
        self.logger.info("Successfully loaded refiner weights")

    def launch_train(self):
        raise NotImplementedError("the reference ships no second-stage training loop (LocalRefineTrainLoop: pass, loop_CORAL.py:38-39)")

    def launch_val(self):                                      # :584-590
        from .loop_CORAL import LocalRefineValidationLoop
        return LocalRefineValidationLoop(self.config, self, window_features=self.window_features).run()


Runner_local_refine = LocalRefineRunner                       # legacy alias (runner.py:681)


class RunnerFactory:
    """engine/runner/runner.py:597-655: runner type by name, or detected from the configuration."""
    _RUNNER_TYPES = {"standard": StandardRunner, "local_refine": LocalRefineRunner, "lr": LocalRefineRunner}

    @classmethod
    def create_runner(cls, config, runner_type=None, **kw):
        if runner_type is None:
            runner_type = cls._detect_runner_type(config)
        if runner_type not in cls._RUNNER_TYPES:
            raise ValueError(f"Unknown runner type '{runner_type}'. Available: {list(cls._RUNNER_TYPES.keys())}")
        return cls._RUNNER_TYPES[runner_type](config, **kw)

    @classmethod
    def _detect_runner_type(cls, config):
        if hasattr(config, "model_cfg") and hasattr(config.model_cfg, "window_size"):
            return "local_refine"
        if hasattr(config, "train_cfg") and config.train_cfg.get("refiner_path"):
            return "local_refine"
        return "standard"

    @classmethod
    def get_available_runners(cls):
        return list(cls._RUNNER_TYPES.keys())


def create_runner(config, runner_type=None, **kw):
    """engine/runner/runner.py:688-699."""
    return RunnerFactory.create_runner(config, runner_type, **kw)


def get_available_runner_types():
    return RunnerFactory.get_available_runners()
