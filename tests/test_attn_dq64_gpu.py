"""`attn_bwd_dq64_kernel` (csrc/attn_dq64.hip, the opt-in 64-query-rows-per-wave dQ kernel; MANTIS_ATTN_DQ64=1) against the oracle.

The switch is read once per process, so the forced run lives in a subprocess: a handful of the head-dim-128 backward checks of
tests/gpu_checks.py without a key-padding mask (causal and not, GQA groups 1 / 4 / 7, packed segments, cross attention, ragged lengths) --
the kernel takes exactly these launches when forced; batches with a mask keep attn_bwd_dq_kernel either way."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["attn_bwd_2_130_4_4_128_False_None", "attn_bwd_2_200_4_1_128_True_None", "attn_bwd_1_40_4_1_128_True_None",
         "attn_bwd_2_97_8_2_128_False_None", "attn_bwd_2_200_7_1_128_True_None", "attn_bwd_1_190_6_1_128_False_None",
         "attn_segments_0_L300_H8_2_hd128", "attn_segments_3_L450_H4_1_hd128", "attn_segments_7_L257_H14_2_hd128",
         "attn_cross_1_130_70_8_2_128_False"]


@pytest.mark.gpu
def test_forced_dq64_kernel_matches_the_oracle():
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from tests import gpu_checks as G\n"
            "c = G.all_checks()\n"
            "for n in %r:\n"
            "    c[n]()\n"
            "print('DQ64-OK')\n") % (ROOT, NAMES)
    env = dict(os.environ, MANTIS_ATTN_DQ64="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "DQ64-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
