"""Deterministic form of the "wrong result on the first launches of a process" class of bug
(VERDICT r1 item 1): every CU's LDS is scribbled immediately before the first launch of each
MFMA kernel family and the result is compared with the oracle.  Runs in a fresh process so the
launches really are the first ones; one pass."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_first_launch_after_lds_scribble():
    p = subprocess.run([sys.executable, os.path.join(HERE, "first_launch_worker.py")],
                       capture_output=True, text=True, timeout=900)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert line, "worker printed no result:\n" + p.stdout[-2000:] + p.stderr[-2000:]
    res = json.loads(line[-1])
    kernels = {r["kernel"] for r in res["results"] if r["first_launch_of_kernel"]}
    assert len(kernels) >= 10, kernels
    wrong = [r for r in res["results"] if r["mismatches"]]
    assert p.returncode == 0 and not wrong, wrong
