"""The reference's algorithm-component tests (pydream/tests/test_dream.py) run on the HIP engine through the
C ABI; tests/test_oracle_reference_kat.py runs the same checks on the CPU oracle."""
import numpy as np
import pytest

from tests import reference_suite as RS

pytestmark = pytest.mark.gpu


def make(**cfg):
    from pydream_amd import _capi
    return _capi.Engine(**cfg)


def test_gamma_array_on_device():
    """the device-side table (dz_set_gamma_table(NULL) computes it) through proposals: gamma_arr[0][0][d'-1]."""
    from pydream_amd.Dream import Dream
    from pydream_amd.model import Model
    from pydream_amd.parameters import FlatParam
    dream = Dream(model=Model(likelihood=lambda x: 0.0, sampled_parameters=[FlatParam(test_value=np.zeros(1))]), DEpairs=5, p_gamma_unity=0)
    np.testing.assert_allclose(dream.gamma_arr[0, :, 0], [1.683, 1.19, .972, .841, .753], atol=5e-4)


def test_snooker_and_cr_fractions():
    RS.check_snooker_and_cr_fractions(make)


def test_gamma_choices():
    RS.check_gamma_choices(make)


def test_depair_selection():
    RS.check_depair_selection(make)


def test_crossover_fraction_of_dims():
    RS.check_crossover_fraction_of_dims(make)


def test_history_sampling():
    RS.check_history_sampling(make)


def test_multitry_selection():
    RS.check_multitry_selection(make)


def test_history_recording():
    RS.check_history_recording(make)


def test_boundaries():
    RS.check_boundaries(make)
