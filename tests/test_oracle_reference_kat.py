"""The reference's known-answer and algorithm-component tests (pydream/tests/test_dream.py) run on the CPU oracle.
The same checks run on the HIP engine in tests/test_reference_suite_gpu.py."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import reference_suite as RS


def make(**cfg):
    return O.Engine(**cfg)


def test_gamma_array():
    RS.check_gamma_array(O.gamma_table)


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
