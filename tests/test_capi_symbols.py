"""The C-ABI library exports every symbol include/yask_hip_c_api.h declares (no GPU needed)."""
import ctypes

import pytest

from yask_amd import _capi


def test_header_and_prototype_table_agree():
    assert set(_capi.header_symbols()) == set(_capi.PROTOTYPES), (
        set(_capi.header_symbols()) ^ set(_capi.PROTOTYPES))


@pytest.mark.parametrize("stencil", ["iso3dfd", "3axis", "ssg", "test_3d", "test_boundary_3d", "test_scratch_3d", "test_misc_2d"])
def test_library_exports_all_symbols(stencil):
    p = _capi.lib_path(stencil)
    assert p.exists(), f"{p} missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(str(p))
    for name in _capi.header_symbols():
        assert hasattr(lib, name), name
    lib.yk_get_version_string.restype = ctypes.c_char_p
    assert b"cdna4_hip" in lib.yk_get_version_string()


def test_missing_library_fails_loudly():
    with pytest.raises(ImportError):
        _capi.load("no_such_stencil")
