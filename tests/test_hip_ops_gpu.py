"""-m gpu: every gfx950 kernel through the C-ABI vs the oracle (see tests/gpu_checks.py for inputs and tolerances)."""
import pytest

from tests import gpu_checks

CHECKS = gpu_checks.all_checks()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CHECKS))
def test_kernel_parity(name):
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from tests import helpers
    helpers.CURRENT_CASE = name          # whole-step checks file their per-tensor gradient report under it (helpers.GRAD_REPORTS)
    CHECKS[name]()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["1", "0"])
def test_attention_checks_with_forward_kernel_forced(mode):
    """The hd-128 attention forward has two kernels, chosen by query length (attn.hip: attn_fwd64 from 1024 rows on).  The parity cases
    are mostly short, so every attention check runs once more with attn_fwd64 forced for EVERY length (MANTIS_ATTN_FWD64=1: masks, left /
    right padding, packed segments, GQA groups, cross attention, ragged lengths) and once with it switched off (=0: the full-size cases
    on attn_fwd_kernel<128>).  The switch is read once per process, hence the subprocess."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MANTIS_ATTN_FWD64=mode)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_selftest.py"), "attn"], capture_output=True, text=True, env=env,
                       cwd=root, timeout=1500)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "FAIL" not in r.stdout, tail
