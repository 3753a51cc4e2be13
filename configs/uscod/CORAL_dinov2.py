# CORAL second stage (SparseRefiner over the frozen first-stage decoder) on DINOv2 ViT-B/14 features -- the overrides of the
# reference's configs/uscod/CORAL_dinov2.py on top of the first-stage file, so `CfgNode.load_with_base` yields an identical tree.
_WINDOWS = {"window_size": 3, "window_length": 56, "threshold": 0.0015}      # 3x3 windows, entropy selector threshold

cfg = {
    "_BASE_": ["./UCOD-DPL_dinov2.py"],
    "start_ema": 1,
    "enable_plabel_cache": True,
    "model_cfg": {**_WINDOWS, "ema_weight": 0.70},
    "train_cfg": {"max_epoch": 8, "lr0": 1e-4, "step_lr_size": 2, "step_lr_gamma": 0.95},
    "val_cfg": {"val_interval": 4, "val_start": 4},
    "dataset_cfg": {
        "trainset_cfg": {"image_size": (518, 518), "require_label": True, "look_twice": False, "look_twice_th": 0.15, "bkg_th": 0.6,
                         "use_cache": True, "require_m_patches": True},
        "valset_cfg": {"DATASET": "TE-CAMO", "use_cache": True, "require_m_patches": False},
        "trainloader_cfg": {"batch_size": 2, "num_workers": 0, "shuffle": True},
    },
}
