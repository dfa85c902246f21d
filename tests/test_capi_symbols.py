"""CPU-only checks of the C-ABI boundary: the library builds, loads, exports every symbol that
include/dreamzs.h declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from pydream_amd import _capi, build as dzbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    dzbuild.build()
    return _capi.load_library()


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "dreamzs.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dz_[a-z0-9_]+)\s*\(", txt)) - {"dz_logp_cb", "dz_exchange_cb"})


def test_header_and_binding_agree():
    assert header_symbols() == sorted(_capi.SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    for s in header_symbols():
        assert hasattr(lib, s), s
    assert lib.dz_version() == 1


def test_config_struct_layout():
    # dz_config: 18 int32 (history_lag, adapt_lag and one reserved word last), 2 int64, 1 uint64, 5 double = 136 bytes, no padding surprises
    assert ctypes.sizeof(_capi.Config) == 18 * 4 + 3 * 8 + 5 * 8
    assert _capi.Config.history_capacity.offset == 18 * 4 and _capi.Config.adapt_lag.offset == 16 * 4


def test_no_silent_cpu_fallback(lib):
    n = _capi.device_count()
    if n > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_capi.DreamZSError, match="no HIP device|no CPU fallback"):
        _capi.Engine(nchains=4, ndim=3, history_capacity=16)
