"""Feature cache: the on-disk format either side of the hot path.

Host-side mirror of data/datasets/cache_manager.py::{CacheManager, MultiCacheManager} and of the
``MetaListPickleIO`` they sit on (engine/utils/fileio/backend/ioctl/pickleio.py:54-142): a directory
``<cache_dir>/features_cache/<feature_extractor_type>/<mode>/<DATASET>/`` (``pseudo_label_cache/<DATASET>/`` for pseudo
labels, cache_manager.py:54-79) holding ``index.json`` = ``{"<i>": "data_<i>.pkl"}`` (plain ``json.dump``) and one pickle
per item -- a CPU f32 tensor ``[C, h, w]`` for features (base_dataset.py:124-145).  Caches written here are read by the
reference and vice versa (tests/golden/cache_ref was written by the reference's own class).

``build_feature_cache`` is the pass the reference runs one image at a time (base_dataset.py:131-141): here the backbone
sees ``batch_size`` images per launch, and each key map is copied to the host and pickled exactly as the reference does.
"""
import json
import os
import pickle
from pathlib import Path

import torch


class MetaListPickleIO:
    def __init__(self, index_path=None, base_path=None, file_prefix="data", logger_in=None):
        if index_path is not None:
            self.index_path = Path(index_path)
            self.base_path = self.index_path.parent
        elif base_path is not None:
            self.base_path = Path(base_path)
            self.index_path = self.base_path / "index.json"
        else:
            raise ValueError("Either index_path or base_path must be specified.")
        self.file_prefix = file_prefix
        self.prefix_counter = {}
        self.logger = logger_in
        ok, _ = self.check_integrity(self.index_path)
        self.mode = "r" if ok else "w"
        self.index_map = {}
        if self.mode == "r":
            self._prepare_reading()

    @staticmethod
    def check_integrity(index_file_path):
        index_file_path = Path(index_file_path)
        if not index_file_path.exists():
            return False, "Index file does not exist."
        with open(index_file_path, "r") as f:
            index_map = json.load(f)
        for index, file in index_map.items():
            if not (index_file_path.parent / file).exists():
                return False, "File with index {} does not exist.".format(index)
        return True, "_"

    def reload_path(self):
        ok, _ = self.check_integrity(self.index_path)
        if not ok:
            self.mode, self.index_map = "w", {}
        else:
            self.mode = "r"
            self._prepare_reading()

    def _prepare_reading(self):
        with open(self.index_path, "r") as f:
            self.index_map = json.load(f)
        for index, file in self.index_map.items():
            self.index_map[index] = self.base_path / file

    def read_file(self, index):
        assert self.mode == "r", "Not working on read mode!"
        with open(self.index_map[str(index)], "rb") as f:
            return pickle.load(f)

    def len(self):
        return len(self.index_map)

    def write_file(self, index, obj, file_name=None):
        assert self.mode == "w", "Not working on write mode!"
        if file_name:
            self.index_map[index] = "{}_{}.pkl".format(file_name, self.prefix_counter.get(file_name, 0))
            self.prefix_counter[file_name] = self.prefix_counter.get(file_name, 0) + 1
        else:
            self.index_map[index] = "{}_{}.pkl".format(self.file_prefix, index)
        path = self.base_path / self.index_map[index]
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "wb") as f:
            pickle.dump(obj, f)

    def write_index(self):
        with open(self.index_path, "w") as f:
            json.dump(self.index_map, f)

    def dump_list(self, obj_list, file_name_list=None):
        for index, obj in enumerate(obj_list):
            self.write_file(index, obj, file_name_list[index] if file_name_list else None)
        self.write_index()


class CacheManager:
    def __init__(self, base_path, logger=None):
        self.base_path = base_path
        self.logger = logger
        self._io = None

    @property
    def io(self):
        if self._io is None:
            self._io = MetaListPickleIO(base_path=self.base_path, logger_in=self.logger)
        return self._io

    @property
    def mode(self):
        return self.io.mode

    def dump_list(self, data_list):
        self.io.dump_list(data_list)
        self.io.reload_path()

    def read_file(self, index):
        return self.io.read_file(index)

    def length(self):
        return self.io.len()


class MultiCacheManager:
    def __init__(self, cache_dir, feature_extractor_type, mode, dataset_name, logger=None):
        self.cache_dir = cache_dir
        self.feature_extractor_type = feature_extractor_type
        self.mode = mode
        self.dataset_name = dataset_name
        self.logger = logger
        self._caches = {}

    def get_cache(self, cache_type):
        if cache_type not in self._caches:
            cache_name = "features_cache" if cache_type == "features" else f"{cache_type}_cache"
            if cache_type == "pseudo_label":
                path = os.path.join(self.cache_dir, cache_name, self.dataset_name)
            else:
                path = os.path.join(self.cache_dir, cache_name, self.feature_extractor_type, self.mode, self.dataset_name)
            self._caches[cache_type] = CacheManager(path, self.logger)
        return self._caches[cache_type]

    def get_features_cache(self):
        return self.get_cache("features")

    def get_pseudo_label_cache(self):
        return self.get_cache("pseudo_label") if self.mode == "train" else None

    def get_patch_cache(self):
        return self.get_cache("patch")

    def get_m_patch_cache(self):
        return self.get_cache("m_patch")


def build_feature_cache(images, feature_extractor, features_cache, batch_size=32, device="cuda"):
    """Run ``feature_extractor`` (``backbone``-like: ``(img) -> (outputs, key [B,C,h,w])``) over ``images`` -- an iterable of
    ``[3,H,W]`` f32 tensors, already transformed as base_dataset.py:133 does -- in batches, and write the features cache in
    the reference's format.  Items are streamed to disk as they are produced (the reference first collects the whole list
    in host memory, base_dataset.py:128-143).  Returns the number of items written."""
    io = features_cache.io
    if io.mode != "w":
        raise RuntimeError(f"cache at {io.base_path} already exists and is valid; remove it to rebuild")
    n = 0
    batch = []

    def flush():
        nonlocal n
        if not batch:
            return
        x = torch.stack(batch).to(device)
        _, key = feature_extractor(x)
        key = key.to("cpu")                                       # base_dataset.py:138: features.squeeze(0).to('cpu')
        for i in range(key.shape[0]):
            io.write_file(n, key[i].clone())
            n += 1
        batch.clear()

    for img in images:
        if batch and tuple(img.shape) != tuple(batch[0].shape):
            flush()                                               # the reference's transform can yield ragged sizes: one launch per shape
        batch.append(img)
        if len(batch) == batch_size:
            flush()
    flush()
    io.write_index()
    io.reload_path()
    return n
