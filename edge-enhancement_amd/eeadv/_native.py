"""ctypes binding of libeeadv.so (include/eeadv.h) - the only doorway from Python to the HIP kernels.

The library is built in-tree by `make -C edge-enhancement_amd/csrc` (or `__graft_entry__.build()`).
There is NO fallback: if the shared object is missing, fails to load, or lacks a symbol declared in
include/eeadv.h, importing this module raises - a GPU run can never silently use a CPU path.
"""
import ctypes
import os

# PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7).  libeeadv.so must share THAT
# runtime instance - device pointers and streams come from torch - so torch is imported first: the dynamic
# loader then resolves libeeadv's NEEDED libamdhip64.so.7 to the copy already in the process.  Loading
# libeeadv first would pull /opt/rocm's copy in as a second, separate runtime (launches fail with
# hipErrorNoDevice).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libeeadv.so")

c_f, c_d, c_i, c_l, c_p = ctypes.c_float, ctypes.c_double, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p
c_u64 = ctypes.c_uint64

# symbol -> argtypes, exactly the prototypes of include/eeadv.h (return type int unless listed in _RESTYPE)
SIGNATURES = {
    "ee_abi_version": [],
    "ee_strerror": [c_i],
    "ee_device_name": [],
    "ee_pgd_init_f32": [c_p, c_p, c_p, c_l, c_f, c_f, c_p],
    "ee_pgd_init_rng_f32": [c_p, c_p, c_l, c_f, c_i, c_u64, c_u64, c_f, c_f, c_p],
    "ee_pgd_step_f32": [c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_i, c_p],
    "ee_l2_step_f32": [c_p, c_p, c_p, c_l, c_l, c_f, c_f, c_f, c_f, c_p],
    "ee_fgsm_step_f32": [c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_i, c_p],
    "ee_add_clamp_f32": [c_p, c_p, c_p, c_l, c_f, c_f, c_p],
    "ee_freeat_update_f32": [c_p, c_p, c_l, c_f, c_f, c_p],
    "ee_freeat_update_masked_f32": [c_p, c_p, c_p, c_l, c_f, c_f, c_p],
    "ee_avmix_f32": [c_p, c_p, c_p, c_p, c_l, c_l, c_f, c_p],
    "ee_avmix_labels_f64": [c_p, c_p, c_p, c_l, c_l, c_f, c_f, c_p],
    "ee_edge125_fwd_f32": [c_p, c_i, c_i, c_i, c_i, c_p, c_f, c_f, c_p, c_p, c_p],
    "ee_edge125_bwd_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_f, c_f, c_p, c_p],
    "ee_frontend_fwd_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_f, c_f, c_f, c_p, c_p, c_p, c_p],
    "ee_frontend_bwd_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_f, c_f, c_f, c_p, c_p, c_p],
    "ee_frontend_fwd_save_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_f, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p],
    "ee_frontend_bwd_saved_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_f, c_f, c_f, c_p, c_p, c_p],
    "ee_canny_fwd_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_f, c_f, c_f, c_f, c_p, c_p, c_p, c_p],
    "ee_canny_bwd_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_f, c_f, c_f, c_f, c_p, c_p, c_p],
    "ee_canny_bpda_fwd_f32": [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p],
    "ee_canny_bpda_bwd_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_f, c_f, c_p, c_p, c_p],
    "ee_pgd_step_bcast_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_l, c_f, c_f, c_f, c_f, c_i, c_p],
    "ee_ce_f32": [c_p, c_p, c_i, c_i, c_f, c_f, c_p, c_p, c_p],
    "ee_fc_ce_grad_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p],
    "ee_kl_f32": [c_p, c_p, c_i, c_i, c_f, c_p, c_p, c_p, c_p],
    "ee_softce_f64": [c_p, c_p, c_i, c_i, c_d, c_p, c_p, c_p],
    "ee_mse_f32": [c_p, c_p, c_l, c_f, c_p, c_p, c_p],
    "ee_mse_num_partials": [c_l],
    "ee_reduce_rows_f64": [c_p, c_l, c_d, c_p, c_p],
    "ee_topk_i64": [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p],
    "ee_add_square_fwd_f32": [c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_p, c_i, c_p, c_p],
    "ee_add_square_bwd_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_p, c_i, c_p, c_p],
    "ee_square_draw_f32": [c_p, c_l, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p],
    "ee_hfs_table_floats": [c_i, c_i, c_i, c_i],
    "ee_hfs_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_f, c_i, c_p, c_f, c_p, c_p, c_p, c_p, c_i, c_p],
    "ee_hfs_mfma_table_floats": [c_i, c_i, c_i],
    "ee_hfs_mfma_f32": [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_f, c_p, c_p, c_p, c_p, c_i, c_p],
    "ee_chain_supported": [c_i, c_i, c_i],
    "ee_chain_table_floats": [c_i, c_i],
    "ee_chain_fwd_f32": [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_f, c_f, c_f, c_i, c_f, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "ee_chain_bwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_p],
    "ee_bn_workspace_floats": [c_i, c_i, c_i],
    "ee_syncbn_workspace_floats": [c_i, c_i, c_i],
    "ee_syncbn_stats_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "ee_syncbn_apply_f32": [c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_f, c_f, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "ee_syncbn_bwd_sums_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_p],
    "ee_syncbn_bwd_apply_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_d, c_i, c_p, c_p, c_i, c_i, c_i, c_p],
    "ee_bn_act_fwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "ee_bn_act_bwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "ee_bn_act_bwd2_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "ee_bn_dual_supported": [c_i, c_i, c_i],
    "ee_bn_dual_fwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_p],
    "ee_bn_dual_bwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_f, c_i, c_p, c_p, c_p, c_p, c_p, c_p,
                           c_i, c_i, c_i, c_p],
    "ee_bn_relu_pool_workspace_floats": [c_i, c_i, c_i, c_i],
    # x, gamma, beta, rm, rv, momentum, eps, training, y_pool, code, save_mean, save_invstd, workspace, conv_stats, slices, B, C, H, W, stream
    "ee_bn_relu_pool_fwd_f32": [c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    # dy_pool, dy_pool2, code, x, gamma, beta, save_mean, save_invstd, rm, rv, eps, training, dx, dgamma, dbeta, workspace, B, C, H, W, stream
    "ee_bn_relu_pool_bwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "ee_bn_relu_pool_fwd_xa_f32": [c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "ee_bn_relu_pool_bwd_xa_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "ee_wino3x3_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    # x, u, mean, var, gamma, beta, eps, res, relu, y, B, Cin, Cout, H, stream
    "ee_wino3x3_bn_eval_fwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p],
    # dy, dy2, y, u_b, var, gamma, eps, dres, dx_add, dx, B, Cin, Cout, H, stream
    "ee_wino3x3_bn_eval_bwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "ee_wino3x3_stats_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    # x, stats, S, cnt, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, u, y, B, KC, RC, H, stream
    "ee_wino3x3_bn_train_pre_f32": [c_p, c_p, c_i, c_i, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "ee_dense2x2_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    # x, w2, mean, var, gamma, beta, eps, res, relu, y, B, Cin, Cout, stream
    "ee_dense2x2_bn_eval_fwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_i, c_p, c_i, c_i, c_i, c_p],
    # dy, dy2, y, w2t, var, gamma, eps, dres, dx_add, dx, B, Cin, Cout, stream
    "ee_dense2x2_bn_eval_bwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    # dc, u_b, x, save_mean, save_invstd, gamma, beta, dy, sums, B, Cin, Cout, H, stream
    "ee_wino3x3_bwd_sums_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    # dy, x, sums, S, cnt, save_mean, save_invstd, gamma, beta, u_b, dx, B, Cin, Cout, H, stream
    "ee_wino3x3_bn_train_bwd_pre_f32": [c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "ee_wrw3x3_workspace_floats": [c_i, c_i, c_i, c_i],
    "ee_wrw3x3_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "ee_wrw3x3s2_workspace_floats": [c_i, c_i, c_i, c_i, c_i],
    "ee_wrw_stem7x7s2_workspace_floats": [c_i, c_i, c_i],
    "ee_wrw1x1_workspace_floats": [c_i, c_i, c_i, c_i],
    "ee_wrw1x1_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "ee_wrw_stem7x7s2_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "ee_wrw3x3s2_f32": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "ee_conv3x3s2_small_fwd_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "ee_conv3x3s2_small_bwd_data_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "ee_conv_weight_prep_f32": [c_i, c_p, c_p, c_p, c_i, c_i, c_p],
    "ee_conv_weight_prep_batch_f32": [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "ee_conv3x3s2_pair_fwd_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "ee_conv3x3s2_pair_bwd_data_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    # x, w10, mean3, var3, gamma3, beta3, eps3, mean1, var1, gamma1, beta1, eps1, y3, y1, B, Cin, Cout, H, stream
    "ee_conv3x3s2_pair_bn_eval_fwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "ee_conv3x3s2_pair_stats_fwd_f32": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    # dy3, y3, dy1, w10, var3, gamma3, eps3, var1, gamma1, eps1, dx, B, Cin, Cout, H, stream
    "ee_conv3x3s2_pair_bn_eval_bwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_f, c_p, c_i, c_i, c_i, c_i, c_p],
    "ee_net2_conv_wrw_workspace_floats": [c_i],
    "ee_net2_conv_wrw_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_i, c_p],
    "ee_net2_conv_fwd_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p],
    "ee_net2_conv_bwd_f32": [c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p],
    "ee_maxpool3s2_fwd_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "ee_maxpool3s2_bwd_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "ee_conv1x1s2_fwd_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "ee_conv1x1s2_bwd_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "ee_stem7x7s2_bwd_data_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "ee_stem7x7s2_fwd_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "ee_stem7x7s2_fwd_stats_floats": [c_i, c_i, c_i, c_i],
    "ee_stem7x7s2_fwd_stats_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "ee_pool_linear_fwd_f32": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "ee_pool_linear_bwd_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "ee_ce_pool_linear_bwd_f32": [c_p, c_p, c_f, c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "ee_prof_enable": [c_i],
    "ee_prof_mark_empty": [c_p],
    "ee_prof_read": [c_i, c_p, c_p],
    "ee_prof_read_work": [c_i, c_p],
    "ee_prof_reset": [],
}
_RESTYPE = {"ee_strerror": ctypes.c_char_p, "ee_device_name": ctypes.c_char_p, "ee_mse_num_partials": c_l, "ee_wrw3x3_workspace_floats": c_l, "ee_wrw3x3s2_workspace_floats": c_l, "ee_wrw_stem7x7s2_workspace_floats": c_l, "ee_wrw1x1_workspace_floats": c_l, "ee_net2_conv_wrw_workspace_floats": c_l}

# kernel-family ids of include/eeadv.h (ee_prof_*)
(K_PGD_STEP, K_FRONTEND_FWD, K_FRONTEND_BWD, K_EDGE_FWD, K_EDGE_BWD, K_CE, K_PGD_STEP_BCAST, K_EMPTY, K_HFS, K_CHAIN_FWD, K_CHAIN_BWD,
 K_HFS_SQ_FWD, K_HFS_SQ_BWD, K_SQUARE_DRAW) = range(14)
K_WINO, K_CONV3S2_FWD, K_CONV3S2_BWD, K_WINO_FUSED = 18, 19, 20, 21  # 14 - 17: the direct 3x3 kernels removed in round 3


class EEError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libeeadv.so not found at %s - build it with `make -C edge-enhancement_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:  # header / library drift must be loud
            raise ImportError("libeeadv.so does not export %s (stale build?)" % name) from exc
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, c_i)
    return lib


lib = _load()


def check(rc, what):
    if rc != 0:
        msg = lib.ee_strerror(rc)
        raise EEError("%s failed: %s (code %d)" % (what, msg.decode() if msg else "?", rc))


def abi_version():
    return lib.ee_abi_version()
