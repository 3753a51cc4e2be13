# Launcher-level switches.  The reference lifts these keys into `accelerate.Accelerator(**kwargs)`
# (engine/runner/runner.py:118-120); this build launches one process per GPU with torchrun/RCCL instead and
# keeps the keys so configs written for the reference still load.  Every value is the reference default.
_UNSET = None

cfg = {
    "cpu": False,
    "device_placement": True,
    "gradient_accumulation_steps": 1,
    "mixed_precision": _UNSET,
    "dataloader_config": _UNSET,
    "deepspeed_plugin": _UNSET,
    "fsdp_plugin": _UNSET,
    "megatron_lm_plugin": _UNSET,
    "log_with": _UNSET,
    "project_config": _UNSET,
    "project_dir": _UNSET,
    "rng_types": _UNSET,
}
