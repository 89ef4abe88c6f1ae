"""First test of the GPU suite: writes the human-scale reference / read files and starts the stock binary on the four human-scale
cases in the background (tests/humanscale.py).  Their ~4 minutes of CPU work then run beside the rest of the suite;
tests/test_gpu_zz_humanscale.py, the last module, compares."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_human_scale_inputs_written_and_stock_runs_started(human):
    for k in ("ref", "rl", "ns", "c4", "asm", "asm_full", "rr_ref", "rr"):
        assert os.path.getsize(human[k]) > 0
    assert set(human["stock"]) == {"northstar", "configs2", "configs2_full", "configs4", "repeat_rich"}
