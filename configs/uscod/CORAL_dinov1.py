# CORAL second stage on DINO ViT-B/8 features -- the overrides of the reference's configs/uscod/CORAL_dinov1.py on top of the
# first-stage file, so `CfgNode.load_with_base` yields an identical tree.
_WINDOWS = {"window_size": 3, "window_length": 56, "threshold": 0.0015}

cfg = {
    "_BASE_": ["./UCOD-DPL_dinov1.py"],
    "start_ema": 1,
    "enable_plabel_cache": True,
    "model_cfg": {**_WINDOWS, "ema_weight": 0.70},
    "train_cfg": {"max_epoch": 8, "lr0": 2e-4, "step_lr_size": 2, "step_lr_gamma": 0.95},
    "val_cfg": {"val_interval": 4, "val_start": 4},
    "dataset_cfg": {
        "trainset_cfg": {"image_size": (296, 296), "require_label": True, "look_twice": False, "look_twice_th": 0.15, "bkg_th": 0.6,
                         "use_cache": True, "require_m_patches": True},
        "valset_cfg": {"use_cache": True, "require_m_patches": True},
        "trainloader_cfg": {"batch_size": 2, "num_workers": 0, "shuffle": True},
    },
}
