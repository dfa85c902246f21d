"""The oracle against the REFERENCE ITSELF on configurations drawn at random -- tests/golden/fuzz_reference.py as a test (CPU only).

Runs where /root/reference exists (the build container) and is skipped elsewhere (the GPU box has no reference; there the committed
fixtures of tests/golden/ carry the pin).  Each arm is a separate process with the reference first on its path: this process may already
hold the repository's own `pydream` alias package, and the vectors must come from the reference, never from the implementation under test
(make_golden.py asserts it).  What is compared, per case: snooker / CR / selected-try / accept sequences exactly, log densities to 1e-10,
states to 1e-9 relative, the archive, delta_m / ncr_updates, and -- in lockstep cases with an adaptation -- the shared probabilities after
EVERY generation (what pins `adapt_lag`)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "pydream")), reason="the reference is not on this machine")


@pytest.mark.parametrize("arm,n,seed", [
    ("", 26, 60601),               # both schedules, multitry 1..32, pairs, gamma levels, every prior / boundary combination, redraw rounds, history_lag, parallel tempering, half of the adaptations with an adapt_lag
    ("--adapt-lag", 18, 60602),    # every lockstep adaptation case with adapt_lag in {1, 2, 3, 9, 19}, burn-ins ending inside and beyond the run
    ("--tri", 12, 60603),          # MVN up to 100-D, each case also with the triangular factor bench.py's headline times
])
def test_oracle_equals_reference_on_random_configurations(arm, n, seed, tmp_path):
    env = dict(os.environ, PYTHONPATH=REFERENCE, OMP_NUM_THREADS="1")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "golden", "fuzz_reference.py"), "--n", str(n), "--seed", str(seed)] + ([arm] if arm else [])
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    tail = [ln for ln in r.stdout.splitlines() if ln.startswith("fuzz vs reference") or ln.startswith("MISMATCH")]
    assert r.returncode == 0 and tail, r.stdout[-3000:]
    m = re.search(r"(\d+) cases, (\d+) mismatches", tail[-1])
    assert m and int(m.group(1)) == n and int(m.group(2)) == 0, "\n".join(tail)
