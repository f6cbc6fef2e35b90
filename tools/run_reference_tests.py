#!/usr/bin/env python3
"""Runs the REFERENCE's own unit tests against pddp_amd (build container only:
needs /root/reference; nothing of it is stored in the repo).

The reference's test files are copied to a temporary directory, `pddp` is
aliased to `pddp_amd` in sys.modules, and the files that exercise host-side
code (torch ops on CPU tensors) are run: utils (encoding, angular,
gaussian_variable, autodiff, trajectory, evaluation), costs (aggregate,
quadratic), examples (costs, models, envs), models (bnn).  The controller and
boxqp tests need CUDA tensors in pddp_amd (no CPU fallback) and are covered by
tests/test_gpu_parity.py instead.  `benchmark` cases (pytest-benchmark is not
installed) are deselected.

    python tools/run_reference_tests.py            # summary on stdout
"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/tests"
FILES = ["utils/test_encoding.py", "utils/test_angular.py",
         "utils/test_gaussian_variable.py", "utils/test_autodiff.py",
         "utils/test_trajectory.py", "utils/test_evaluation.py",
         "costs/test_aggregate.py", "costs/test_quadratic.py",
         "examples/test_costs.py", "examples/test_models.py",
         "examples/test_envs.py", "models/test_bnn.py"]
CONFTEST = '''
import importlib, sys
import pddp_amd
sys.modules["pddp"] = pddp_amd
for sub in ("utils", "utils.encoding", "utils.angular", "utils.gaussian_variable",
            "utils.autodiff", "utils.evaluation", "utils.constraint",
            "utils.trajectory", "utils.particles", "utils.classproperty",
            "costs", "costs.quadratic", "costs.base", "models", "models.base",
            "models.bnn", "controllers", "envs", "envs.base", "examples",
            "examples.cartpole", "examples.pendulum", "examples.double_cartpole",
            "examples.rendezvous"):
    sys.modules["pddp." + sub] = importlib.import_module("pddp_amd." + sub)
'''


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, "conftest.py"), "w") as f:
            f.write(CONFTEST)
        names = []
        for rel in FILES:
            dst = os.path.join(tmp, rel.replace("/", "_"))
            shutil.copy(os.path.join(REF, rel), dst)
            names.append(os.path.basename(dst))
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=ROOT,
                   CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
        out = subprocess.run(
            [sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider",
             "-k", "not benchmark", "-rf"] + names, cwd=tmp, env=env,
            capture_output=True, text=True).stdout
    keep = [l for l in out.splitlines()
            if l.startswith("FAILED") or " passed" in l or " failed" in l]
    print("reference test files run against pddp_amd:", ", ".join(FILES))
    print("\n".join(keep))


if __name__ == "__main__":
    main()
