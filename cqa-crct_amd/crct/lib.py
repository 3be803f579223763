"""ctypes binding of libcrct_hip.so (C ABI declared in include/crct_hip.h).

The library is built in-tree by ``cqa-crct_amd/csrc/Makefile`` (``__graft_entry__.build()``).
There is NO fallback: if the shared object is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcrct_hip.so")

c_i32, c_i64, c_u32, c_u64, c_f32 = C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float
ATTN_MAX_LEN = 512           # CRCT_ATTN_MAX_LEN of include/crct_hip.h: longest query / key sequence of the attention kernels
FP8_AMAX_LANES = 64          # CRCT_FP8_AMAX_LANES of include/crct_hip.h: fp32 words per amax value
vp = C.c_void_p


class GemmArgs(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("bias", vp), ("preact_out", vp), ("dact_src", vp), ("addend", vp),
                ("lda", c_i64), ("ldb", c_i64), ("ldc", c_i64), ("ld_aux", c_i64), ("ld_add", c_i64),
                ("M", c_i32), ("N", c_i32), ("K", c_i32), ("ta", c_i32), ("tb", c_i32), ("act", c_i32), ("dact", c_i32),
                ("c_is_f32", c_i32), ("accumulate", c_i32), ("tile", c_i32), ("alpha", c_f32), ("drop_thr", c_u32),
                ("drop_scale", c_f32), ("drop_site", c_u32), ("seed", c_u64), ("rowsum_out", vp),
                ("fp8", c_i32), ("scale_a", vp), ("scale_b", vp), ("q_out", vp), ("q_scale", vp), ("q_amax", vp), ("ld_q", c_i64),
                ("site", c_i32), ("split_k", c_i32), ("splitk_ws", vp), ("splitk_cnt", vp), ("addend_f32", c_i32), ("c_cached", c_i32)]


class LaunchRec(C.Structure):
    _fields_ = [("site", c_i32), ("kind", c_i32), ("M", c_i32), ("N", c_i32), ("K", c_i32), ("cfg", c_i32), ("split_k", c_i32),
                ("grid", c_i32), ("n_problems", c_i32), ("flops", C.c_double)]


SITE_NAMES = ["none", "t.qkv", "t.out", "t.ffn_up", "t.ffn_down", "v.qkv", "v.out", "v.ffn_up", "v.ffn_down",
              "c.qkv_t", "c.qkv_v", "c.out_t", "c.out_v", "img_emb", "head"]      # CRCT_SITE_* of include/crct_hip.h
KIND_NAMES = ["fwd", "dgrad", "wgrad"]
CLASS_NAMES = ["S.w", "S.n", "S.nl", "M.w", "M.n", "M.nl", "L.w", "L.n", "L.nl"]      # crct_gemm_class_config: rows bucket x column class


class HeadArgs(C.Structure):
    _fields_ = [("pooled_t", vp), ("pooled_v", vp), ("fus_h", vp), ("w_cls", vp), ("b_cls", vp), ("w_f6", vp), ("b_f6", vp),
                ("R", vp), ("labels", vp), ("logits", vp), ("reg", vp), ("stats", vp), ("scratch", vp),
                ("d_pooled_t", vp), ("d_pooled_v", vp), ("d_fus_h", vp),
                ("d_w_cls", vp), ("d_b_cls", vp), ("d_w_f6", vp), ("d_b_f6", vp), ("g_nsp_dev", vp), ("g_reg_dev", vp),
                ("g_loss_dev", vp), ("B", c_i32), ("Hb", c_i32), ("fusion_sum", c_i32), ("use_l1", c_i32), ("kind_l1", c_i32),
                ("tol_margin", c_f32), ("nsp_coeff", c_f32), ("reg_coeff", c_f32), ("grad_scale", c_f32),
                ("drop_thr", c_u32), ("drop_scale", c_f32), ("drop_site", c_u32), ("seed", c_u64)]


class ModelDims(C.Structure):
    _fields_ = [("vocab", c_i32), ("n_pos", c_i32), ("n_types", c_i32), ("H", c_i32), ("L", c_i32), ("heads", c_i32), ("I", c_i32),
                ("Fv", c_i32), ("Hv", c_i32), ("Lv", c_i32), ("v_heads", c_i32), ("Iv", c_i32), ("Hb", c_i32), ("b_heads", c_i32),
                ("n_color", c_i32), ("n_conn", c_i32), ("v_biatt", c_i32 * 32), ("t_biatt", c_i32 * 32),
                ("fusion_sum", c_i32), ("with_coattention", c_i32),
                ("p_hidden", c_f32), ("p_attn", c_f32), ("p_v_hidden", c_f32), ("p_v_attn", c_f32), ("p_cls", c_f32)]


class Batch(C.Structure):
    _fields_ = [("tokens", vp), ("segments", vp), ("loc", vp), ("text_keymask", vp), ("image_feat", vp), ("image_loc", vp),
                ("image_target", vp), ("image_keymask", vp), ("R", vp), ("labels", vp),
                ("B", c_i32), ("T", c_i32), ("V", c_i32),
                ("sep_indices", vp), ("hist_len", vp), ("image_mask", vp), ("sep_stride", c_i32), ("image_feat_bf16", c_i32)]


class LnFwdArgs(C.Structure):
    _fields_ = [("x", vp), ("gamma", vp), ("beta", vp), ("y", vp), ("mean", vp), ("rstd", vp), ("M", c_i32), ("H", c_i32),
                ("eps", c_f32), ("drop_thr", c_u32), ("drop_scale", c_f32), ("drop_site", c_u32), ("seed", c_u64),
                ("q_out", vp), ("q_scale", vp), ("q_amax", vp), ("x_f32", c_i32), ("y_f32", vp)]


class LnBwdArgs(C.Structure):
    _fields_ = [("dy", vp), ("x", vp), ("mean", vp), ("rstd", vp), ("gamma", vp), ("dx", vp), ("dx_lin", vp), ("partials", vp),
                ("M", c_i32), ("H", c_i32), ("post_thr", c_u32), ("post_scale", c_f32), ("post_site", c_u32),
                ("lin_thr", c_u32), ("lin_scale", c_f32), ("lin_site", c_u32), ("seed", c_u64),
                ("q_out", vp), ("q_scale", vp), ("q_amax", vp), ("x_f32", c_i32)]


class AmpState(C.Structure):
    _fields_ = [("grad_scale", vp), ("found_inf", vp), ("step", vp)]


class AttnQuant(C.Structure):
    _fields_ = [("ctx_q", vp), ("ctx_scale", vp), ("ctx_amax", vp), ("dq_q", vp), ("dk_q", vp), ("dv_q", vp),
                ("dq_scale", vp), ("dq_amax", vp), ("dkv_scale", vp), ("dkv_amax", vp),
                ("row_lse", vp), ("ctx", vp), ("ld_ctx", c_i64)]


class Fp8Shadow(C.Structure):
    _fields_ = [("q", vp), ("seg_slot", vp), ("scale", vp), ("amax", vp), ("qt", vp), ("seg_in", vp), ("seg_t_base", vp), ("seg_t_ld", vp)]


class StepCfg(C.Structure):
    _fields_ = [("training", c_i32), ("use_l1", c_i32), ("kind_l1", c_i32), ("tol_margin", c_f32), ("nsp_coeff", c_f32),
                ("reg_coeff", c_f32), ("grad_scale", c_f32), ("seed", c_u64), ("g_nsp_dev", vp), ("g_reg_dev", vp), ("g_loss_dev", vp),
                ("seg_ready_events", vp), ("seg_done_events", vp),
                ("fp8", c_i32), ("params_fp8", vp), ("fp8_w_scale", vp), ("fp8_act_scale", vp), ("fp8_act_amax", vp),
                ("fp8_bwd", c_i32), ("fp8_wgrad", c_i32), ("params_fp8_t", vp), ("fp8_grad_scale", vp), ("fp8_grad_amax", vp),
                ("seg_enqueued", vp), ("seg_enqueued_user", vp), ("seg_done_mask", vp), ("wgrad_overwrite", c_i32), ("grads_bf16", vp),
                ("residual_fp32", c_i32)]


SEG_ENQUEUED_FN = C.CFUNCTYPE(None, C.c_int, vp)      # void (*seg_enqueued)(int seg, void* user)


# name -> (restype, argtypes); every symbol include/crct_hip.h declares
_u8 = [c_u32, c_f32, c_u32, c_u64]   # drop_thr, drop_scale, drop_site, seed
PROTOTYPES = {
    "crct_last_error": (C.c_char_p, []),
    "crct_abi_version": (C.c_int, []),
    "crct_gemm_bf16": (C.c_int, [C.POINTER(GemmArgs), vp]),
    "crct_gemm_bf16_grouped": (C.c_int, [C.POINTER(GemmArgs), C.c_int, vp]),
    "crct_gemm_pick_tile": (C.c_int, [C.c_int, C.c_int]),
    "crct_gemm_group_max_workgroups": (C.c_int, [C.c_int]),
    "crct_gemm_group_target_workgroups": (C.c_int, [C.c_int]),
    "crct_gemm_group_concat": (C.c_int, [C.c_int]),
    "crct_embed_scatter_split": (None, [C.c_int]),
    "crct_gemm_force_generic": (C.c_int, [C.c_int]),
    "crct_prof_enable": (C.c_int, [C.c_int]),
    "crct_prof_reset": (C.c_int, []),
    "crct_prof_read": (C.c_int, [C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "crct_prof_read_site": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "crct_prof_stamp_count": (C.c_int, []),
    "crct_prof_stamp_read": (C.c_int, [C.c_int, C.POINTER(vp), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "crct_engine_streams": (C.c_int, [vp, C.POINTER(vp)]),
    "crct_launch_log_enable": (C.c_int, [C.c_int]),
    "crct_launch_log_count": (C.c_int, []),
    "crct_launch_log_read": (C.c_int, [C.c_int, C.POINTER(LaunchRec)]),
    "crct_gemm_splitk_ws_elems": (c_i64, [C.c_int, C.c_int, C.c_int]),
    "crct_gemm_splitk_tickets": (C.c_int, [C.c_int, C.c_int]),
    "crct_engine_set_site_policy": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "crct_layernorm_fwd": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, c_f32] + _u8 + [vp]),
    "crct_layernorm_bwd_blocks": (C.c_int, [C.c_int]),
    "crct_layernorm_bwd": (C.c_int, [vp] * 11 + [C.c_int, C.c_int, C.c_int, c_u32, c_f32, c_u32, c_u32, c_f32, c_u32, c_u64, vp]),
    "crct_layernorm_bwd_rows": (C.c_int, [vp] * 8 + [C.c_int, C.c_int, c_u32, c_f32, c_u32, c_u32, c_f32, c_u32, c_u64, vp]),
    "crct_layernorm_bwd_finalize": (C.c_int, [vp] * 4 + [C.c_int, C.c_int, C.c_int, vp]),
    "crct_colsum_blocks": (C.c_int, [C.c_int]),
    "crct_colsum_bf16": (C.c_int, [vp, c_i64, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "crct_build_keymasks": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "crct_softmax_rows_f32_bf16": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "crct_softmax_rows_bf16_bf16": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "crct_layernorm_fwd_q": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, c_f32] + _u8 + [vp, vp, vp, vp]),
    "crct_layernorm_bwd_rows_args": (C.c_int, [C.POINTER(LnBwdArgs), vp]),
    "crct_layernorm_fwd_args": (C.c_int, [C.POINTER(LnFwdArgs), vp]),
    "crct_fp8_transpose_weights": (C.c_int, [vp] * 6 + [C.c_int, c_i64, C.c_int, vp]),
    "crct_fp8_quantize_bf16": (C.c_int, [vp, vp, vp, vp, c_i64, vp]),
    "crct_fp8_update_scales": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, c_f32, vp]),
    "crct_fp8_quantize_weights": (C.c_int, [vp] * 7 + [c_i64, vp, vp, C.c_int, vp]),
    "crct_gemm_group_wgrad_config": (C.c_int, [C.c_int]),
    "crct_engine_set_wgrad_flush": (C.c_int, [vp, C.c_int]),
    "crct_engine_set_wgrad_workgroups": (C.c_int, [vp, C.c_int, C.c_int]),
    "crct_gemm_class_config": (C.c_int, [C.c_int, C.c_int]),
    "crct_gemm_fp8_scaled_mfma": (C.c_int, [C.c_int]),
    "crct_ghost_collective": (C.c_int, [vp, c_i64, C.c_int, C.c_int, C.c_double, vp]),
    "crct_cast_f32_bf16": (C.c_int, [vp, vp, c_i64, vp]),
    "crct_cast_bf16_f32": (C.c_int, [vp, vp, c_i64, vp]),
    "crct_cast_runs_f32_bf16": (C.c_int, [vp, vp, vp, vp, vp, vp, c_i64, vp]),
    "crct_cast_runs_bf16_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, c_i64, vp]),
    "crct_attention_fwd": (C.c_int, [vp] * 5 + [C.c_int] * 5 + [c_i64] * 4 + _u8 + [vp]),
    "crct_attention_bwd": (C.c_int, [vp] * 8 + [C.c_int] * 5 + [c_i64] * 7 + _u8 + [vp]),
    "crct_attention_fwd_q": (C.c_int, [vp] * 5 + [C.c_int] * 5 + [c_i64] * 4 + _u8 + [vp, vp]),
    "crct_attention_bwd_q": (C.c_int, [vp] * 8 + [C.c_int] * 5 + [c_i64] * 7 + _u8 + [vp, vp]),
    "crct_attention_quant_ok": (C.c_int, [C.c_int] * 3),
    "crct_attention_force_valu": (None, [C.c_int]),
    "crct_attention_force_long": (None, [C.c_int]),
    "crct_attention_force_split": (None, [C.c_int]),
    "crct_embed_text_fwd": (C.c_int, [vp] * 14 + [C.c_int] * 4 + [c_f32] + _u8 + [vp]),
    "crct_embed_text_bwd": (C.c_int, [vp] * 16 + [C.c_int] * 4 + _u8 + [vp, vp, C.c_int, vp]),
    "crct_embed_word_index": (None, [C.c_int]),
    "crct_embed_text_bwd_indexed": (C.c_int, [vp] * 16 + [C.c_int] * 4 + _u8 + [vp, vp, C.c_int, vp, C.c_int, vp]),
    "crct_embed_image_fwd": (C.c_int, [vp] * 12 + [C.c_int] * 2 + [c_f32] + _u8 + [vp]),
    "crct_embed_image_bwd": (C.c_int, [vp] * 15 + [C.c_int] * 2 + _u8 + [vp, vp, C.c_int, vp]),
    "crct_head_loss": (C.c_int, [C.POINTER(HeadArgs), vp]),
    "crct_eval_select": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, c_i64, vp, vp, vp, vp, vp, vp]),
    "crct_adamw_plan": (c_i64, [vp, C.c_int, vp, vp, c_i64]),
    "crct_adamw_step": (C.c_int, [vp] * 11 + [c_i64, c_f32, c_f32, c_f32, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, vp]),
    "crct_adamw_advance": (C.c_int, [vp, vp, vp]),
    "crct_engine_create": (vp, [C.POINTER(ModelDims), C.c_char_p, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "crct_engine_destroy": (None, [vp]),
    "crct_engine_workspace_bytes": (C.c_size_t, [vp]),
    "crct_engine_num_segments": (C.c_int, [vp]),
    "crct_engine_segment_range": (C.c_int, [vp, C.c_int, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "crct_engine_forward": (C.c_int, [vp, vp, vp, C.POINTER(Batch), C.POINTER(StepCfg), vp, vp, vp, vp, vp]),
    "crct_engine_backward": (C.c_int, [vp, vp, vp, C.POINTER(Batch), C.POINTER(StepCfg), vp, vp, vp, vp, vp, C.c_int, vp]),
    "crct_engine_set_streams": (C.c_int, [vp, C.c_int, C.c_int]),
    "crct_engine_aux_stream": (vp, [vp, vp, C.POINTER(C.c_int)]),
    "crct_event_create": (vp, []),
    "crct_event_destroy": (None, [vp]),
    "crct_event_record": (C.c_int, [vp, vp]),
    "crct_stream_wait_event": (C.c_int, [vp, vp]),
    "crct_event_synchronize": (C.c_int, [vp]),
    "crct_event_query": (C.c_int, [vp]),
    "crct_engine_wgrad_owned": (C.c_int, [vp, vp, vp, C.c_int]),
    "crct_engine_fp8_sites": (C.c_int, [vp]),
    "crct_engine_fp8_grad_sites": (C.c_int, [vp]),
    "crct_engine_fp8_weights": (C.c_int, [vp, vp, vp, C.c_int]),
    "crct_zero_runs": (C.c_int, [vp, vp, vp, vp, vp, c_i64, vp]),
    "crct_engine_tap": (c_i64, [vp, vp, C.c_char_p, C.c_int, C.c_int, C.c_int, vp, c_i64, vp]),
}

_lib = None


def load():
    """Load libcrct_hip.so; raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libcrct_hip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "-- there is no CPU fallback for the CRCT step" % LIB_PATH)
    import torch  # noqa: F401  -- torch must load ITS libamdhip64 first; a second HIP runtime in the process sees no device
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError("libcrct_hip %s failed (%d): %s" % (what, rc, load().crct_last_error().decode()))


def drop_threshold(p):
    t = float(p) * 4294967296.0
    if t <= 0.0:
        return 0
    return min(int(t), 4294967295)


def ptr(t):
    """Device (or host) address of a torch tensor / None."""
    return None if t is None else t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
