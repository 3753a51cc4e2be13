# Project-wide defaults (key set and values of the reference's configs/__base__/newbase.py), grouped by consumer.
_loader_stub = {}
_set_stub = {"type": "USCODDataset"}

cfg = {
    "work_dir": "./work",
    # --- optimisation schedule: TrainLoop (engine/runner/loop_UCOD_DPL.py) --------------------------------
    "train_cfg": {
        "dist_train": True,
        "start_epoch": 0,
        "max_epoch": 25,
        "start_finetune": -5,          # finetune begins at max_epoch + start_finetune
        "merge_method": "dis",         # APM driven by the discriminator
        "merge_alpha": 0.5,
        "add_noise": False,
        "grad_norm": 1.0,
        "save_cfg": {"save_mode": "model", "save_interval": 5, "start_save": -50},
    },
    # --- decoder / discriminator constructors -------------------------------------------------------------
    "model_cfg": {
        "decoder": "BGDecoder",
        "dim": 768,
        "feature_size": 16,
        "ema_weight": 0.999,
        "dis_use_features": True,
        "up_sample": False,
        "use_attention": False,
        "conv_num": 1,
    },
    "val_cfg": {"enable_val": True, "val_interval": 5, "start_val": -50},
    "log_cfg": {"name": "Ablation 1", "log_path": "/home/yanweiq/storage/trainlog.log", "multi_rank": [0]},
    "dataset_cfg": {
        "trainset_cfg": dict(_set_stub),
        "trainloader_cfg": dict(_loader_stub),
        "valset_cfg": dict(_set_stub),
        "val_loader_cfg": dict(_loader_stub),
    },
    "feature_extractor_cfg": {},
}
