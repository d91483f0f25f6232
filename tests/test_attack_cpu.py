"""csrc/xq_attack.hpp - the king-centric legality / in-check test the HIP kernels use - is plain integer code: compiled
here with g++ and checked against the CPU oracle (the literal restatement of chess_env.py:431-548) on every candidate
move of random positions, consistent (random play from the start) and inconsistent (piece soup, stale / missing king
caches: SURVEY.md Appendix A5 / A6).  tests/cpu_harness/attack_check.cpp is the driver; the kernels themselves are
checked on the GPU by tests/test_gpu_parity.py (rules_random / rules_edge / rules_extra fixtures, 20,000 arbitrary
boards vs the oracle)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_king_centric_attack_test_equals_the_oracle(tmp_path):
    from oracle import xq_oracle
    xq_oracle.build()
    exe = os.path.join(tmp_path, "attack_check")
    subprocess.check_call(["g++", "-O2", "-I" + ROOT, os.path.join(ROOT, "tests", "cpu_harness", "attack_check.cpp"),
                           "-L" + os.path.join(ROOT, "oracle"), "-lxq_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
                           "-o", exe])
    r = subprocess.run([exe, "120", "8000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " 0 mismatches" in r.stdout.splitlines()[-1], r.stdout[-500:]
