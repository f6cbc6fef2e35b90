#!/usr/bin/env python3
"""HIP BNN kernels against tests/golden/bnn_cartpole_real_size.npz (reference
outputs at [200, 200] x 100 particles in float32 and float64): prints the
relative error of every quantity against both, next to the reference's own
float32-vs-float64 difference."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import _bnn_real_size_run  # noqa: E402

if __name__ == "__main__":
    rows = _bnn_real_size_run()
    for r in rows:
        print(json.dumps(r))
