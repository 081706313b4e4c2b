"""Helpers to read the committed golden fixtures (tests/golden/*.npz)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def state_from(npz, prefix):
    """Rebuild a reference-style state dict (complex z included) from `prefix/...` entries."""
    P = {}
    for k, v in npz.items():
        if not k.startswith(prefix):
            continue
        key = k[len(prefix):]
        if key.endswith("@re"):
            base = key[:-3]
            P[base] = torch.complex(torch.from_numpy(v), torch.from_numpy(npz[prefix + base + "@im"]))
        elif key.endswith("@im"):
            continue
        else:
            P[key] = torch.from_numpy(np.asarray(v))
    return P
