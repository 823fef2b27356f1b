"""Loader for the committed golden fixtures (tests/golden/*.npz)."""

import glob
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name: str, device="cpu") -> dict:
    """Return {key: tensor-or-python-scalar}; ``__bf16`` keys become bfloat16 tensors."""
    data = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    out = {}
    for key in data.files:
        arr = data[key]
        if key.endswith("__bf16"):
            out[key[: -len("__bf16")]] = torch.from_numpy(arr.view(np.int16).copy()).view(torch.bfloat16).to(device)
        elif arr.ndim == 0:
            out[key] = arr.item()
        else:
            out[key] = torch.from_numpy(arr.copy()).to(device)
    return out


def names(prefix: str):
    return sorted(
        os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz"))
    )
