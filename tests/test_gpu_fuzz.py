"""A fixed sample of the differential fuzzer (tests/fuzz_parity.py): random configurations, HIP engine == oracle bit for bit."""
import numpy as np
import pytest

from tests import fuzz_parity as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_random_configurations_equal_the_oracle(seed):
    from pydream_amd import _capi as G
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    for _ in range(12):
        c = F.draw_config(rng)
        assert F.run_one(G, O, c) is None, c
