"""A short turn of the randomised differential runs of tools/ inside the GPU suite (the long runs: profiles/r03_fuzz_parity.log):
engine vs oracle over random batch sequences and engine settings, and the whole CLI vs the real reference binary."""
import os
import subprocess
import sys

import pytest

import helpers as H

pytestmark = pytest.mark.gpu
TOOLS = os.path.join(H.ROOT, "tools")


def test_engine_against_oracle_random_cases(hip):
    sys.path.insert(0, TOOLS)
    try:
        import fuzz_parity
    finally:
        sys.path.remove(TOOLS)
    for seed in range(9000, 9008):
        r = fuzz_parity.one(seed)
        assert r is None, r


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("script,seed", [("fuzz_cli.py", 77), ("fuzz_cli_incremental.py", 78)])
def test_cli_against_reference_random_cases(hip, script, seed):
    p = subprocess.run([sys.executable, os.path.join(TOOLS, script), "12", str(seed)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and "0 mismatches" in out, out[-1500:]
