"""-m gpu: the C ABI driven by a host written in C (tests/c_host/scn_c_host.c, built by __graft_entry__.build with
gcc -std=c11): index build in one call, the hot kernel forward and backward-data, checked inside the program against a
brute-force restatement.  No Python, torch or oracle in that process."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "c_host", "scn_c_host")


@pytest.mark.gpu
def test_c_host_program(gpu):
    if not os.path.exists(BIN):
        import __graft_entry__ as g
        g.build_c_host()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "c host: ok" in r.stdout, r.stdout
