"""Helpers shared by the golden-vector tests (fixture layout: see oracle/make_golden.py)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SAMPLE = 4096


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def sample_idx(n):
    stride = max(1, n // SAMPLE)
    return np.arange(0, n, stride)[:SAMPLE]


def state_from_golden(g):
    return {k[2:]: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in g.items() if k.startswith("w/")}


def grad_keys(g):
    return sorted(k[len("gnorm/"):] for k in g if k.startswith("gnorm/"))


def check_grad(g, key, grad, rtol, atol_frac=0.0):
    """Compares ``grad`` (tensor) for state-dict key ``key`` with the stored golden form.
    Error metric: ||got - ref|| / ||ref|| over the stored elements (rel-L2), plus the full-tensor norm."""
    a = grad.detach().float().cpu().reshape(-1).numpy().astype(np.float64)
    ref_norm = float(g["gnorm/" + key])
    got_norm = float(np.sqrt((a ** 2).sum()))
    if "grad/" + key in g:
        ref = g["grad/" + key].reshape(-1).astype(np.float64)
        got = a
    else:
        ref = g["gsample/" + key].astype(np.float64)
        got = a[sample_idx(a.size)]
    denom = max(float(np.sqrt((ref ** 2).sum())), 1e-30)
    rel = float(np.sqrt(((got - ref) ** 2).sum())) / denom
    norm_rel = abs(got_norm - ref_norm) / max(ref_norm, 1e-30)
    return rel, norm_rel
