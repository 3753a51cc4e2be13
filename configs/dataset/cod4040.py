# COD training mix (4 040 images) / COD10K test split, as in the reference's configs/dataset/cod4040.py.
cfg = {
    "dataset_cfg": {
        "dataset_dir": "./datasets/RefCOD",
        "cache_dir": "./datasets/cache/look_twice",
        "trainset_cfg": {"DATASET": "TR-CAMO+TR-COD10K", "require_label": False},
        "valset_cfg": {"DATASET": "TE-COD10K", "require_label": True},
    }
}
