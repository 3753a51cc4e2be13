"""Feature cache: the on-disk format either side of the hot path.

What the reference keeps on disk (data/datasets/cache_manager.py:54-79 on top of ``MetaListPickleIO``,
engine/utils/fileio/backend/ioctl/pickleio.py:54-142):

    <cache_dir>/features_cache/<extractor>/<mode>/<DATASET>/index.json      {"0": "data_0.pkl", "1": "data_1.pkl", ...}
    <cache_dir>/features_cache/<extractor>/<mode>/<DATASET>/data_<i>.pkl    pickle of a CPU f32 tensor [C, h, w]
    <cache_dir>/pseudo_label_cache/<DATASET>/...                            same layout, train mode only

``IndexedPickleDir`` implements that directory format once; ``MetaListPickleIO`` / ``CacheManager`` / ``MultiCacheManager``
expose it under the reference's names and call shapes, so caches are interchangeable in both directions (a cache written
by the reference's own classes is a test fixture, tests/golden/cache_ref).  ``build_feature_cache`` is the cache-building
pass (base_dataset.py:124-145) with the backbone fed ``batch_size`` images per launch instead of one.
"""
import json
import os
import pickle
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import torch

INDEX_NAME = "index.json"


class IndexedPickleDir:
    """A directory of pickles addressed through ``index.json`` (string keys -> file names relative to the directory)."""

    def __init__(self, directory, stem="data"):
        self.dir = Path(directory)
        self.stem = stem
        self.names = {}                       # key (as written) -> file name
        self._named_counts = {}
        self.refresh()

    # ---- state
    @property
    def index_file(self):
        return self.dir / INDEX_NAME

    @staticmethod
    def probe(index_file):
        """(usable, reason): the index exists and every file it names exists (pickleio.py:93-103)."""
        index_file = Path(index_file)
        if not index_file.is_file():
            return False, "Index file does not exist."
        listed = json.loads(index_file.read_text())
        missing = [k for k, name in listed.items() if not (index_file.parent / name).exists()]
        if missing:
            return False, "File with index {} does not exist.".format(missing[0])
        return True, "_"

    def refresh(self):
        usable, _ = self.probe(self.index_file)
        self.readable = usable
        self.names = json.loads(self.index_file.read_text()) if usable else {}
        return usable

    def __len__(self):
        return len(self.names)

    # ---- items
    def load(self, key):
        if not self.readable:
            raise AssertionError("Not working on read mode!")
        with open(self.dir / self.names[str(key)], "rb") as fh:
            return pickle.load(fh)

    def reserve(self, key, named=None):
        """Enter ``key`` into the index (in call order) and return the file name its pickle will get."""
        if self.readable:
            raise AssertionError("Not working on write mode!")
        if named:                                                    # pickleio.py:126-128: <name>_<running count>.pkl
            n = self._named_counts.get(named, 0)
            self._named_counts[named] = n + 1
            fname = f"{named}_{n}.pkl"
        else:
            fname = f"{self.stem}_{key}.pkl"
        self.names[key] = fname
        self.dir.mkdir(parents=True, exist_ok=True)
        return fname

    def write(self, fname, obj):
        """The file half of ``store``: safe to run on a worker thread (one file per call, nothing shared)."""
        with open(self.dir / fname, "wb") as fh:
            pickle.dump(obj, fh)

    def store(self, key, obj, named=None):
        self.write(self.reserve(key, named), obj)

    def commit(self):
        """Write index.json (plain json.dump: integer keys become the strings the readers look up)."""
        self.dir.mkdir(parents=True, exist_ok=True)
        with open(self.index_file, "w") as fh:
            json.dump(self.names, fh)


# ------------------------------------------------------------------------------------------- reference-named adapters
class MetaListPickleIO:
    """engine/utils/fileio/backend/ioctl/pickleio.py::MetaListPickleIO (same constructor, attributes and methods)."""

    def __init__(self, index_path=None, base_path=None, file_prefix="data", logger_in=None):
        if index_path is None and base_path is None:
            raise ValueError("Either index_path or base_path must be specified.")
        directory = Path(index_path).parent if index_path is not None else Path(base_path)
        self._store = IndexedPickleDir(directory, stem=file_prefix)
        self.base_path, self.index_path, self.file_prefix, self.logger = self._store.dir, self._store.index_file, file_prefix, logger_in

    check_integrity = staticmethod(IndexedPickleDir.probe)

    @property
    def mode(self):
        return "r" if self._store.readable else "w"

    @property
    def index_map(self):
        if self._store.readable:
            return {k: self.base_path / v for k, v in self._store.names.items()}
        return self._store.names

    def reload_path(self):
        self._store.refresh()

    def read_file(self, index):
        return self._store.load(index)

    def len(self):
        return len(self._store)

    def write_file(self, index, obj, file_name=None):
        self._store.store(index, obj, named=file_name)

    def dump_list(self, obj_list, file_name_list=None):
        for i, obj in enumerate(obj_list):
            self._store.store(i, obj, named=file_name_list[i] if file_name_list else None)
        self._store.commit()


class CacheManager:
    """data/datasets/cache_manager.py::CacheManager."""

    def __init__(self, base_path, logger=None):
        self.base_path, self.logger = base_path, logger
        self._io = None

    @property
    def io(self):
        if self._io is None:
            self._io = MetaListPickleIO(base_path=self.base_path, logger_in=self.logger)
        return self._io

    mode = property(lambda self: self.io.mode)

    def dump_list(self, data_list):
        self.io.dump_list(data_list)
        self.io.reload_path()

    def read_file(self, index):
        return self.io.read_file(index)

    def length(self):
        return self.io.len()


class MultiCacheManager:
    """data/datasets/cache_manager.py::MultiCacheManager: one CacheManager per cache type, paths as cache_manager.py:54-79."""

    def __init__(self, cache_dir, feature_extractor_type, mode, dataset_name, logger=None):
        self.cache_dir, self.feature_extractor_type = cache_dir, feature_extractor_type
        self.mode, self.dataset_name, self.logger = mode, dataset_name, logger
        self._caches = {}

    def _path(self, cache_type):
        top = "features_cache" if cache_type == "features" else f"{cache_type}_cache"
        if cache_type == "pseudo_label":                             # not keyed by extractor / mode
            return os.path.join(self.cache_dir, top, self.dataset_name)
        return os.path.join(self.cache_dir, top, self.feature_extractor_type, self.mode, self.dataset_name)

    def get_cache(self, cache_type):
        if cache_type not in self._caches:
            self._caches[cache_type] = CacheManager(self._path(cache_type), self.logger)
        return self._caches[cache_type]

    def get_features_cache(self):
        return self.get_cache("features")

    def get_pseudo_label_cache(self):
        return self.get_cache("pseudo_label") if self.mode == "train" else None

    def get_patch_cache(self):
        return self.get_cache("patch")

    def get_m_patch_cache(self):
        return self.get_cache("m_patch")


# ------------------------------------------------------------------------------------------- the cache-building pass
def build_feature_cache(images, feature_extractor, features_cache, batch_size=32, device="cuda", writers=8, precision="f32eq"):
    """Run ``feature_extractor`` (``backbone``-like: ``(img) -> (outputs, key [B,C,h,w])``) over ``images`` -- an iterable of
    ``[3,H,W]`` f32 tensors, already transformed as base_dataset.py:133 does -- in batches, and write the features cache in
    the reference's format.  Items are streamed to disk as they are produced (the reference first collects the whole list
    in host memory, base_dataset.py:128-143).  Returns the number of items written.
    ``precision``: the reference runs THIS pass with the backbone in plain fp32 (base_dataset.py:124-138: no autocast; the features it caches are what every
    training epoch then reads), so by default a ``backbone`` wrapper is asked for its f32-equivalent sibling (``with_precision("f32eq")``: split-operand MFMA,
    f32 residual stream; ~6x the matrix work of the fp16 engine, which a pass bound by pickle writes does not notice).  ``precision=None`` keeps the extractor
    as given (any callable without ``with_precision`` -- a test double -- is used as it is)."""
    if precision is not None and hasattr(feature_extractor, "with_precision"):
        feature_extractor = feature_extractor.with_precision(precision)
    store = features_cache.io._store
    if store.readable:
        raise RuntimeError(f"cache at {store.dir} already exists and is valid; remove it to rebuild")
    written = 0
    pending = []
    # The format is one pickle per image (4.2 MB at ViT-B/14, 518x518): pickling and file writes, not the device, set the pace of this
    # pass (3 k images/s of backbone against ~60 images/s of serial pickle.dump).  Index entries are made in order on this thread; the
    # files themselves are written by a small pool while the next batch is on the device.
    pool = ThreadPoolExecutor(max_workers=writers) if writers > 0 else None
    jobs = []

    def flush():
        nonlocal written
        if not pending:
            return
        _, key = feature_extractor(torch.stack(pending).to(device))
        key = key.to("cpu")                                       # base_dataset.py:138: features.squeeze(0).to('cpu')
        eng = getattr(feature_extractor, "engine", None)
        if eng is not None and hasattr(eng, "check_overflow"):
            eng.check_overflow(wait=True)                         # these key maps go to disk: a saturated fp16 residual stream must fail HERE
        for row in key:
            fname = store.reserve(written)
            if pool is None:
                store.write(fname, row.clone())
            else:
                jobs.append(pool.submit(store.write, fname, row.clone()))
            written += 1
        pending.clear()

    try:
        for img in images:
            if pending and tuple(img.shape) != tuple(pending[0].shape):
                flush()                                           # ragged sizes: one launch per shape
            pending.append(img)
            if len(pending) == batch_size:
                flush()
        flush()
        for j in jobs:
            j.result()                                            # surface a failed write before the index is committed
    finally:
        if pool is not None:
            pool.shutdown(wait=True)
    store.commit()
    features_cache.io.reload_path()
    return written
