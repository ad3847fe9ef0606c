"""-m gpu: every gfx950 kernel through the C-ABI vs the oracle (see tests/gpu_checks.py for inputs and tolerances)."""
import pytest

from tests import gpu_checks

CHECKS = gpu_checks.all_checks()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CHECKS))
def test_kernel_parity(name):
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    CHECKS[name]()
