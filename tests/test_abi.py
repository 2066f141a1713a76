"""CPU: the C-ABI shared library loads, exports every symbol include/eeadv.h declares, and rejects bad
arguments before touching a device (no compute calls are made here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "eeadv.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ee_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def native():
    import eeadv._native as n
    return n


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ["ee_pgd_step_f32", "ee_pgd_init_f32", "ee_fgsm_step_f32", "ee_freeat_update_f32", "ee_add_clamp_f32",
                 "ee_edge125_fwd_f32", "ee_edge125_bwd_f32", "ee_frontend_fwd_f32", "ee_frontend_bwd_f32", "ee_ce_f32",
                 "ee_kl_f32", "ee_softce_f64", "ee_mse_f32", "ee_topk_i64", "ee_avmix_f32", "ee_add_square_fwd_f32"]:
        assert must in names


def test_library_exports_every_declared_symbol(native):
    lib = ctypes.CDLL(native.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    out = subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (ee_[a-z0-9_]+)", out))
    assert set(declared_functions()) <= exported
    assert set(native.SIGNATURES) == set(declared_functions()), "python binding and header drifted apart"


def test_no_torch_types_in_the_abi():
    src = open(HEADER).read()
    assert "at::" not in src and "c10::" not in src and "#include <torch" not in src


def test_argument_checks_happen_before_any_launch(native):
    L = native.lib
    assert native.abi_version() == 1
    assert L.ee_pgd_step_f32(None, None, None, 16, 0.1, 0.1, 0.0, 1.0, 1, None) == -1  # EE_ERR_NULL
    assert L.ee_pgd_step_f32(None, None, None, -1, 0.1, 0.1, 0.0, 1.0, 1, None) == -2  # EE_ERR_SHAPE
    assert L.ee_pgd_step_f32(None, None, None, 0, 0.1, 0.1, 0.0, 1.0, 1, None) == 0  # empty batch is fine
    assert L.ee_pgd_step_f32(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), 4, 0.1, 0.1, 0.0, 1.0, 7, None) == -2
    assert L.ee_pgd_step_f32(ctypes.c_void_p(18), ctypes.c_void_p(16), ctypes.c_void_p(16), 4, 0.1, 0.1, 0.0, 1.0, 1, None) == -4
    assert L.ee_edge125_fwd_f32(None, 1, 3, 8, 8, None, 0.0, 0.3, None, None, None) == -1
    w = (ctypes.c_float * 27)()
    p = ctypes.c_void_p(64)
    assert L.ee_edge125_fwd_f32(p, 1, 5, 8, 8, w, 0.0, 0.3, p, None, None) == -3  # C > 4: EE_ERR_UNSUPPORTED
    assert L.ee_edge125_fwd_f32(p, 1, 3, 0, 8, w, 0.0, 0.3, p, None, None) == -2
    assert L.ee_ce_f32(p, p, 4, 0, 0.0, 1.0, None, None, None) == -2
    assert L.ee_topk_i64(p, None, 4, 10, 17, p, None, None) == -2
    assert L.ee_mse_num_partials(0) == 0 and L.ee_mse_num_partials(4097) == 2
    assert b"NULL" in L.ee_strerror(-1) and b"not supported" in L.ee_strerror(-3)
    assert L.ee_prof_read(99, None, None) == -2


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import importlib
    import eeadv._native as n
    monkeypatch.setattr(n, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        n._load()
    importlib.reload(n)
