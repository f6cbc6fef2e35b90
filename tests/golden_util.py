"""Helpers to read the golden fixtures captured from the reference
(tools/make_golden.py)."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

DT = {"cartpole": 0.1, "pendulum": 0.1, "double_cartpole": 0.05,
      "rendezvous": 0.1}
HORIZONS = {"cartpole": (5, 100), "pendulum": (5, 50),
            "double_cartpole": (5, 60), "rendezvous": (5, 40)}
FWD_NAMES = ("Z", "F_z", "F_u", "L", "L_z", "L_u", "L_zz", "L_uz", "L_uu")


def load(problem, encoding="ignore", dtype="f64"):
    path = os.path.join(GOLDEN_DIR, "%s_%s_%s.npz" % (problem, encoding, dtype))
    return np.load(path)


def load_dc150():
    """BASELINE configs[3]'s horizon: the double cartpole under
    IGNORE_UNCERTAINTY at N = 5 and N = 150, fp64, with a three-iteration
    bounded fit at N = 150 (tools/make_golden.py --dc-default; at N = 150
    only the bounded forward pass, the backward branches and the fit
    schedule's line search are stored)."""
    return np.load(os.path.join(GOLDEN_DIR,
                                "double_cartpole_ignore150_f64.npz"))


TAGS_DC150 = ["N5_cos", "N5_seeded", "N150_cos"]


def tags(problem):
    N0, N1 = HORIZONS[problem]
    return ["N%d_cos" % N0, "N%d_seeded" % N0, "N%d_cos" % N1]


def np_dtype(dtype):
    return np.float64 if dtype == "f64" else np.float32


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-300) if b.size else 1.0
    return np.abs(a - b).max() / scale if b.size else 0.0


def elementwise_err(a, b, floor=1e-6):
    """Largest |a - b| / max(|b|, floor * max|b| per last-axis row): every entry
    against its OWN size (entries below `floor` of the largest entry of their
    row are measured against that floor - they are sums that cancel to that
    level).  rel_err measures everything against the global maximum and never
    looks at a small gain next to a large one."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if not b.size:
        return 0.0
    row = np.abs(b).max(axis=-1, keepdims=True)
    den = np.maximum(np.abs(b), floor * np.maximum(row, 1e-300))
    return float((np.abs(a - b) / den).max())
