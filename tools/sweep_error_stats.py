#!/usr/bin/env python3
"""Error distribution of every fp32 backward-sweep kernel `auto` can select.

For B trajectories of the bench's distribution (cartpole n=4: seed 5, N=100,
bounds +-10; DEFAULT-encoding shapes through random well-conditioned records is
tests' business) it runs each variant and reports, per (variant, branch, reg):

  * flips: trajectories whose success / failure status differs from the fp32
    oracle's, and trajectories whose clamped pattern (K row == 0) differs;
  * max / median / p99 relative error of k and K against the fp32 oracle and
    against the fp64 oracle (the oracle is pinned to the reference at 1e-9);
  * the same figures for the fp32 ORACLE against the fp64 oracle: what plain
    IEEE fp32 arithmetic in the reference's operation order costs.

The tests' fp32 bounds (tests/test_gpu_parity.py) are set from this output
(profiles/r02_sweep_error_stats.json).
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as orc  # noqa: E402  (the checker)
from test_gpu_parity import _setup  # noqa: E402


def rel(a, b):
    b = np.asarray(b, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - b).max() /
                 max(np.abs(b).max(), 1e-300))


def stats(x):
    x = np.asarray(x)
    if x.size == 0:
        return None
    return {"n": int(x.size), "median": float(np.median(x)),
            "p90": float(np.percentile(x, 90)),
            "p99": float(np.percentile(x, 99)), "max": float(x.max())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=384)
    ap.add_argument("--N", type=int, default=100)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    B, N = args.B, args.N
    out = {"B": B, "N": N, "problem": "cartpole", "cases": []}
    s, op, z0, U, u_min, u_max = _setup("cartpole", "f32", B, N, seed=5)
    s.set_nominal(torch.from_numpy(z0).cuda(), torch.from_numpy(U).cuda())
    s.derivs(mask=s.fresh)
    o32, o64 = orc.load(np.float32), orc.load(np.float64)
    f32 = [o32.forward(op, z0[b], U[b], u_min, u_max) for b in range(B)]
    z064, U64 = z0.astype(np.float64), U.astype(np.float64)
    um64, uM64 = u_min.astype(np.float64), u_max.astype(np.float64)
    f64 = [o64.forward(op, z064[b], U64[b], um64, uM64) for b in range(B)]
    names = ("F_z", "F_u", "L_z", "L_u", "L_zz", "L_uz", "L_uu")
    for branch, bounded, variants in (
            (0, True, (1, 6, 7, 15, 17)),
            (1, True, (1, 6, 7, 15, 17)),
            (0, False, (1, 6, 7, 15, 17)),
            (1, False, (1, 6, 7, 15, 17))):
        for reg in (1e-3, 1.0):
            kw = dict(reg=reg, V_zz_reg=bool(branch))
            r32, r64 = [], []
            for b in range(B):
                k32 = dict(kw)
                k64 = dict(kw)
                if bounded:
                    k32.update(u_min=u_min, u_max=u_max, U=U[b])
                    k64.update(u_min=um64, u_max=uM64, U=U64[b])
                r32.append(o32.backward(*[f32[b][n] for n in names], **k32))
                r64.append(o64.backward(*[f64[b][n] for n in names], **k64))
            # the fp32 oracle against the fp64 oracle
            e_k, e_K, fl = [], [], 0
            for b in range(B):
                if (r32[b][2] == 0) != (r64[b][2] == 0):
                    fl += 1
                if r32[b][2] == 0 and r64[b][2] == 0:
                    e_k.append(rel(r32[b][0], r64[b][0]))
                    e_K.append(rel(r32[b][1], r64[b][1]))
            out["cases"].append({
                "what": "fp32 oracle vs fp64 oracle", "branch": branch,
                "bounded": bounded, "reg": reg, "status_flips": fl,
                "k": stats(e_k), "K": stats(e_K)})
            regv = torch.full((B,), reg, dtype=torch.float64, device="cuda")
            for variant in variants:
                s.gains.zero_()
                s.backward(reg=regv, branch=branch, bounded=bounded,
                           variant=variant)
                k, K = s.gain_views()
                k, K = k.cpu().numpy(), K.cpu().numpy()
                st = s.bwd_status.cpu().numpy()
                c = {"what": "hip f32 variant %d" % variant, "variant": variant,
                     "branch": branch, "bounded": bounded, "reg": reg}
                for tag, ref in (("vs_f32_oracle", r32), ("vs_f64_oracle", r64)):
                    ek, eK, fl, pat = [], [], 0, 0
                    for b in range(B):
                        if (ref[b][2] == 0) != (st[b] == 0):
                            fl += 1
                            continue
                        if st[b] != 0:
                            continue
                        za = np.all(K[b] == 0, axis=(-1, -2))
                        zb = np.all(np.asarray(ref[b][1]) == 0, axis=(-1, -2))
                        if not np.array_equal(za, zb):
                            pat += 1
                            continue
                        ek.append(rel(k[b], ref[b][0]))
                        eK.append(rel(K[b], ref[b][1]))
                    c[tag] = {"status_flips": fl, "clamp_pattern_flips": pat,
                              "k": stats(ek), "K": stats(eK)}
                out["cases"].append(c)
                print(json.dumps(c), flush=True)
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
