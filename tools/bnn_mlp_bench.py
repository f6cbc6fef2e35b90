"""Fused BNN network kernel (pddp_bnn_mlp_f32 / _f64) against the same network
on library GEMMs: time per call and achieved TFLOP/s against the f32 matrix
peak (157.3 TFLOP/s on MI355X: the exact-f32 MFMA runs at the vector rate) or
the f64 matrix peak (78.6 TFLOP/s).

    python tools/bnn_mlp_bench.py [--states 40960] [--particles 100] [--dtype f64]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pddp_amd.models.bnn import BayesianMLP  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--states", type=int, default=40960)  # B x A candidates
    ap.add_argument("--particles", type=int, default=100)
    ap.add_argument("--hidden", type=int, default=200)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    a = ap.parse_args()
    dt = torch.float32 if a.dtype == "f32" else torch.float64
    peak = 157.3 if a.dtype == "f32" else 78.6
    in_dim, out_dim, H, P = 6, 8, a.hidden, a.particles
    torch.manual_seed(0)
    net = BayesianMLP(in_dim, out_dim, [H, H]).cuda().to(dt).eval()
    x = torch.randn(a.states, P, in_dim, device="cuda", dtype=dt)
    flop = 2.0 * a.states * P * (in_dim * H + H * H + H * out_dim)
    out = {"rows": a.states * P, "H": H, "dtype": a.dtype,
           "GFLOP_per_call": flop * 1e-9}
    with torch.no_grad():
        t = timed(lambda: net(x), a.reps)
        out["fused_ms"] = t * 1e3
        out["fused_TFLOPs"] = flop / t * 1e-12
        out["fused_frac_of_%s_matrix_peak" % a.dtype] = flop / t * 1e-12 / peak
        net.use_native = False
        t = timed(lambda: net(x), max(2, a.reps // 3))
        out["library_gemm_ms"] = t * 1e3
        out["library_gemm_TFLOPs"] = flop / t * 1e-12
    print(json.dumps(out))


if __name__ == "__main__":
    main()
