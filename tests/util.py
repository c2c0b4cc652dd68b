"""Shared helpers for the test-suite (fixture loading, parameter dicts)."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    return meta, {k: z[k] for k in z.files if k != "meta"}


def t(a, dtype=None, device=None):
    x = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None and x.is_floating_point():
        x = x.to(dtype)
    return x if device is None else x.to(device)


def tparams(p, dtype=torch.float32, device=None):
    return {k: t(v, dtype, device) for k, v in p.items()}


def maxabs(a, b):
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a)).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.from_numpy(np.asarray(b)).double()
    return float((a - b).abs().max()) if a.numel() else 0.0
