"""engine/utils/seed.py:6-14 of the reference: one call seeds python, numpy and torch (all devices)."""
import random

import numpy as np
import torch


def set_random_seed(seed=42):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
