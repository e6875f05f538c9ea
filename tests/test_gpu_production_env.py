"""The GPU parity suite once more, the way PRODUCTION runs the library: without PIGO_TUNING.

tests/conftest.py sets PIGO_TUNING=1 for the whole suite because many tests force schedule variants, queue sizes and code paths
through the library's tuning switches (inert without it).  That left exactly one test running the default configuration the way
a host program does.  This module re-runs the parity files in a child pytest with PIGO_TEST_NO_TUNING=1: every tuning variable a
test sets is then ignored by the library, so each of those tests checks the DEFAULT path against the oracle instead -- same
inputs, same expected lists.  Tests whose point is a switch's effect (they assert the forced behaviour itself) are deselected."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# tests that assert what a tuning switch DOES (not just the results): meaningless with the switches inert
NEEDS_SWITCHES = [
    "test_one_launch_plan_reports_a_queue_overflow",   # PIGO_ONE_QCAP forces the overflow it is about
    "test_region_deep_list_spill_goes_through_the_tail",
    "test_tile_geometry_rules",
    "test_chunked_pipeline_large_batches",
    "test_queue_overflow_falls_back_to_monolithic",    # PIGO_QUEUE_DIV shrinks the queue it overflows
]
# (test_big_scales_side_chain_and_its_switches asserts which kernels a switch brings in: only its default case runs here)
EXPR = " and ".join("not " + t for t in NEEDS_SWITCHES) + " and not (test_big_scales_side_chain_and_its_switches and not env0)"


@pytest.mark.gpu
def test_parity_suite_without_pigo_tuning():
    env = dict(os.environ)
    env.pop("PIGO_TUNING", None)
    env["PIGO_TEST_NO_TUNING"] = "1"
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_pipeline.py"),
           "-m", "gpu", "-q", "-k", EXPR, "-p", "no:cacheprovider", "--tb=short"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=3000)
    tail = "\n".join(r.stdout.splitlines()[-40:])
    assert r.returncode == 0, tail + "\n" + r.stderr[-1500:]
    assert " passed" in tail and "failed" not in tail, tail
