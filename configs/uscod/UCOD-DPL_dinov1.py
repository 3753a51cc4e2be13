# UCOD-DPL first stage on DINO ViT-B/8 features -- same keys and values as the reference's
# configs/uscod/UCOD-DPL_dinov1.py, so `CfgNode.load_with_base` yields an identical tree.
_BACKBONE = {
    "type": "dinov1",
    "backbone": "facebook/dino-vitb8",
    "backbone_type": "huggingface",
    "backbone_weights": "./weights",
    "backbone_weight_base": "~/workspace/weights/huggingface",
    "backbone_feat_dim": [768],
}
_IMAGE = (296, 296)          # 296 / 8 = 37 x 37 patch grid

cfg = {
    "_BASE_": ["../__base__/accelerate.py", "../__base__/newbase.py", "../dataset/cod4040.py"],
    "exp_name": "UCOD-DPL_dinov1",
    "model_cfg": {"dim": 768, "feature_size": 68, "ema_weight": 0.99, "dis_use_features": False},
    "train_cfg": {
        "start_epoch": 0,
        "max_epoch": 25,
        "lr0": 6e-4,
        "step_lr_size": 25,
        "step_lr_gamma": 0.95,
        # discriminator phase: one epoch every second epoch
        "dis_epoch": 1,
        "dis_intertrain": 2,
        "dis_lr0": 1e-3,
        "dis_step_lr_size": 25,
        "dis_step_lr_gamma": 0.95,
    },
    "val_cfg": {"look_twice": True, "look_twice_th": 0.05, "expand_type": "dynamic", "val_interval": 5, "val_start": 5},
    "log_cfg": {"log_interval": 50},
    "dataset_cfg": {
        "cache_dir": "./datasets/cache",
        "trainset_cfg": {"DATASET": "TR-CAMO+TR-COD10K", "image_size": _IMAGE, "require_label": False, "bkg_th": 0.3},
        "valset_cfg": {"DATASET": "TE-CAMO", "image_size": _IMAGE, "require_label": True},
        "trainloader_cfg": {"batch_size": 16, "num_workers": 0, "shuffle": True},
        "val_loader_cfg": {"batch_size": 1, "num_workers": 0, "shuffle": False},
        "feature_extractor_cfg": _BACKBONE,
    },
}
