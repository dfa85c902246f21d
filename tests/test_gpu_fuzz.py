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


@pytest.mark.parametrize("seed", [21, 22])
def test_random_configurations_with_an_adapt_lag_equal_the_oracle(seed):
    """the --adapt-lag arm: every configuration adapts with adapt_lag >= 1 -- launches that hold several burn-in generations (the kernels' own unit
    sums, or the ring of published positions), the multi-kernel path and sharded engines one generation per launch"""
    from pydream_amd import _capi as G
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    F.KINDS.clear()
    for _ in range(12):
        c = F.draw_config(rng, adapt_lag_arm=True)
        assert F.run_one(G, O, c) is None, c
    assert F.KINDS.get("ring", 0) + F.KINDS.get("multi", 0) > 0, F.KINDS
