"""Seeding (engine/utils/seed.py of the reference: ``set_random_seed(seed)`` before the runner is built).

One call seeds every generator the hot path can draw from -- Python, NumPy, torch CPU and every visible GPU -- and pins
``PYTHONHASHSEED`` for child processes.  Returns the seed so launch scripts can log it."""
import os
import random

import numpy
import torch

_SEEDERS = (random.seed, numpy.random.seed, torch.manual_seed)


def set_random_seed(seed: int = 42) -> int:
    seed = int(seed)
    for seeder in _SEEDERS:
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)          # includes the current device
    os.environ["PYTHONHASHSEED"] = str(seed)
    return seed
