import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import test_gpu_parity as T
from pddp_amd import _native
lib = _native.lib()
for deal in (1, 2):
    prev = lib.pddp_bnn_mlp_deal(deal)
    rows = T._bnn_real_size_run()
    lib.pddp_bnn_mlp_deal(prev)
    print("deal", deal)
    for r in rows[:-1]:
        if r["what"] in ("F_z", "F_u"):
            print("   %s r=%s hip_vs_f64=%.2e ref32_vs_f64=%.2e" % (r["what"], r["r"], r["hip_vs_f64"], r["ref32_vs_f64"]))
