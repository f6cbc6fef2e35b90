"""The reference's OWN host-side unit tests, run against pddp_amd under the
module alias `pddp` (tools/run_reference_tests.py).  Build container only: the
reference does not travel to the GPU box, where this module is skipped."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/tests"),
                    reason="needs /root/reference (build container only)")
def test_reference_host_side_tests_pass_against_pddp_amd():
    """utils (encoding, angular, gaussian_variable, autodiff, trajectory,
    evaluation), costs (aggregate, quadratic), examples (costs, models, envs)
    and models/bnn of the reference's suite: everything passes except the four
    FULL_COVARIANCE_MATRIX gradchecks of tests/models/test_bnn.py that the
    reference itself fails under torch 2.x."""
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, "tools", "run_reference_tests.py")],
        capture_output=True, text=True, timeout=1500).stdout
    failed = [l for l in out.splitlines() if l.startswith("FAILED")]
    summary = [l for l in out.splitlines() if " passed" in l]
    assert summary, out[-2000:]
    n_passed = int(summary[-1].split(" passed")[0].split()[-1])
    assert n_passed >= 1100, summary
    known = "models_test_bnn.py::test_gradcheck["
    unexpected = [l for l in failed
                  if not (known in l and "FULL_COVARIANCE_MATRIX" in l)]
    assert not unexpected, unexpected
    assert len(failed) <= 4, failed
