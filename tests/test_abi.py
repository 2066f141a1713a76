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
    # the weight-gradient entry points (round 3): shapes they do not take, missing pointers and workspaces, the workspace queries
    q = ctypes.c_void_p(4096)
    assert L.ee_wrw3x3_f32(q, q, q, q, 4, 48, 64, 8, None) == -3 and L.ee_wrw3x3_f32(q, q, q, q, 4, 64, 64, 6, None) == -3
    assert L.ee_wrw3x3_f32(q, q, None, q, 4, 64, 64, 8, None) == -1 and L.ee_wrw3x3_f32(None, q, q, q, 4, 64, 64, 8, None) == -1
    assert L.ee_wrw3x3_f32(q, q, q, None, 100, 64, 64, 16, None) == -1  # 58 splits need the workspace
    assert L.ee_wrw3x3_workspace_floats(100, 64, 64, 16) == 58 * 9 * 64 * 64 and L.ee_wrw3x3_workspace_floats(100, 512, 512, 2) == 0
    assert L.ee_wrw3x3_workspace_floats(100, 48, 64, 16) == 0
    assert L.ee_wrw3x3s2_f32(q, q, None, q, q, 4, 64, 128, 2, None) == -3 and L.ee_wrw3x3s2_f32(q, None, None, q, q, 4, 64, 128, 8, None) == -1
    assert L.ee_wrw3x3s2_workspace_floats(100, 64, 128, 16, 1) == 10 * L.ee_wrw3x3s2_workspace_floats(100, 64, 128, 16, 0) // 9 > 0
    assert L.ee_wrw_stem7x7s2_f32(q, q, q, q, 4, 64, 48, None) == -3 and L.ee_wrw_stem7x7s2_f32(q, q, None, q, 4, 64, 64, None) == -1
    assert L.ee_wrw_stem7x7s2_workspace_floats(100, 64, 64) == 256 * 64 * 147
    assert L.ee_wrw1x1_f32(q, q, q, q, 4, 64, 64, 49, None) == -3 and L.ee_wrw1x1_f32(q, q, q, q, 4, 96, 64, 16, None) == -3
    assert L.ee_wrw1x1_f32(ctypes.c_void_p(4100), q, q, q, 4, 64, 64, 16, None) == -4  # EE_ERR_ALIGN
    assert L.ee_net2_conv_wrw_workspace_floats(50) == 5 * 52096 and L.ee_net2_conv_wrw_workspace_floats(10) == 0
    assert L.ee_net2_conv_wrw_f32(q, q, q, q, q, q, q, None, 1.0, None, None, 4, None) == -1
    assert L.ee_conv_weight_prep_f32(7, q, None, q, 48, 64, None) == -3  # EE_WPREP_WINO_FB wants channels % 32 == 0


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import importlib
    import eeadv._native as n
    monkeypatch.setattr(n, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        n._load()
    importlib.reload(n)
