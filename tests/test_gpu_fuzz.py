"""-m gpu: a short run of tests/fuzz_ragged.py (a checker that uses the oracle, hence under tests/) in the suite - random model kinds, batch sizes (1 .. 700) and pad patterns:
the ragged seq_len-50 pair against the full-row kernels (fused step, evaluation forward, ranking forward) and against the CPU
oracle (forward, every gradient, predict).  The long campaigns are under profiles/ (r03_fuzz_ragged.txt)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_random_batches_ragged_pair_equals_full_row_kernels():
    import fuzz_ragged
    assert fuzz_ragged.run_vs_full(24, seed=101, verbose=False) == 0


def test_random_small_batches_match_the_oracle():
    import fuzz_ragged
    assert fuzz_ragged.run_vs_oracle(16, seed=102, verbose=False) == 0
