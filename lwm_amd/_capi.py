"""ctypes mirror of include/lwm_hip.h (struct layouts + prototypes).

Pure declarations: no library is loaded here.  `lwm_amd._lib` binds these to
liblwm_hip.so (the product); tests bind the same declarations to the
host-emulated build of the kernels.
"""
import ctypes as C

LWM_OK = 0
LWM_EINVAL = -1
LWM_EUNSUPPORTED = -2
LWM_ELAUNCH = -3


MAX_PIECES = 8     # LWM_MAX_PIECES


def set_pieces(args, side, pieces):
    """pieces = [(first row, position of that row), ...] of LwmAttnArgs' piecewise position map of `side` ("q" / "k");
    None or one piece leaves the single-piece fields alone."""
    if not pieces or len(pieces) < 2:
        return
    if len(pieces) > MAX_PIECES:
        raise ValueError(f"{side}: {len(pieces)} pieces, at most {MAX_PIECES}")
    setattr(args, side + "_pieces", len(pieces))
    rows, poss = getattr(args, side + "_piece_row"), getattr(args, side + "_piece_pos")
    for i, (r, p) in enumerate(pieces):
        rows[i], poss[i] = int(r), int(p)


class LwmTensor4(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("stride_b", C.c_int64), ("stride_s", C.c_int64),
                ("stride_h", C.c_int64)]


class LwmAttnArgs(C.Structure):
    _fields_ = [
        ("q", LwmTensor4), ("k", LwmTensor4), ("v", LwmTensor4),
        ("out", LwmTensor4),
        ("lse", C.c_void_p), ("out_acc", C.c_void_p), ("lse_acc", C.c_void_p),
        ("dout", LwmTensor4), ("dq", LwmTensor4), ("dk", LwmTensor4), ("dv", LwmTensor4),
        ("delta", C.c_void_p), ("dq_acc", C.c_void_p), ("dk_acc", C.c_void_p),
        ("dv_acc", C.c_void_p),
        ("segment_ids_q", C.c_void_p), ("segment_ids_k", C.c_void_p), ("key_valid", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("Sq", C.c_int32), ("Sk", C.c_int32), ("D", C.c_int32),
        ("q_start", C.c_int64), ("k_start", C.c_int64),
        ("scale", C.c_float),
        ("causal", C.c_int32), ("carry_in", C.c_int32), ("final_out", C.c_int32),
        ("dense_mask", C.c_void_p), ("mask_stride_b", C.c_int64), ("mask_stride_q", C.c_int64),
        ("k_splits", C.c_int32),
        ("seg_blocks_q", C.c_void_p), ("seg_blocks_k", C.c_void_p),
        ("dq_acc_head_major", C.c_int32),
        ("q_pieces", C.c_int32), ("k_pieces", C.c_int32),
        ("q_piece_row", C.c_int32 * MAX_PIECES), ("k_piece_row", C.c_int32 * MAX_PIECES),
        ("q_piece_pos", C.c_int64 * MAX_PIECES), ("k_piece_pos", C.c_int64 * MAX_PIECES),
        ("delta_bytes", C.c_int64),
    ]


class LwmRingArgs(C.Structure):
    _fields_ = [
        ("q", LwmTensor4), ("k", LwmTensor4), ("v", LwmTensor4), ("out", LwmTensor4),
        ("lse", C.c_void_p), ("dout", LwmTensor4), ("dq", LwmTensor4), ("dk", LwmTensor4), ("dv", LwmTensor4),
        ("segment_ids", C.c_void_p), ("key_valid", C.c_void_p),
        ("B", C.c_int32), ("c", C.c_int32), ("H", C.c_int32), ("D", C.c_int32),
        ("scale", C.c_float), ("causal", C.c_int32), ("workspace", C.c_void_p),
        ("layout", C.c_int32), ("schedule", C.c_int32),
        ("chunk_owner", C.c_void_p), ("n_chunks", C.c_int32),
        ("kv_keep", C.c_void_p), ("kv_kept", C.c_int32),
    ]


RING_LAYOUT = {"contiguous": 0, "zigzag": 1, "table": 2}
RING_SCHEDULE = {"ring": 0, "direct": 1, "mesh": 1}


class LwmGemvArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int64), ("nmat", C.c_int32), ("rows", C.c_int32), ("K", C.c_int32),
        ("w", C.c_void_p * 3), ("y", C.c_void_p * 3), ("ldy", C.c_int64 * 3), ("y_f32", C.c_void_p * 3), ("N", C.c_int32 * 3),
        ("workspace", C.c_void_p),
        ("norm_weight", C.c_void_p), ("ss_in", C.c_void_p), ("ss_n", C.c_int32), ("eps", C.c_float),
        ("residual", C.c_void_p * 3), ("ldres", C.c_int64 * 3),
        ("ss_out", C.c_void_p),
    ]


RING_GROUP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
RING_SEND_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p)


class LwmRingTransport(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("group_start", RING_GROUP_FN), ("send", RING_SEND_FN), ("recv", RING_SEND_FN),
                ("group_end", RING_GROUP_FN)]


class LwmConvArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p),
                ("y", C.c_void_p)] + [(n, C.c_int32) for n in (
                    "B", "Hin", "Win", "Cin", "Cout", "KH", "KW", "stride", "pad", "up_shift", "Ho",
                    "Wo", "clip")]


# name -> (restype, argtypes); every symbol include/lwm_hip.h declares
PROTOTYPES = {
    "lwm_attn_fwd": (C.c_int, [C.POINTER(LwmAttnArgs), C.c_void_p]),
    "lwm_attn_bwd_delta": (C.c_int, [C.POINTER(LwmAttnArgs), C.c_void_p]),
    "lwm_attn_bwd_dq": (C.c_int, [C.POINTER(LwmAttnArgs), C.c_void_p]),
    "lwm_attn_bwd_dkdv": (C.c_int, [C.POINTER(LwmAttnArgs), C.c_void_p]),
    "lwm_attn_bwd_delta_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "lwm_attn_fwd_f32": (C.c_int, [C.POINTER(LwmAttnArgs), C.c_void_p]),
    "lwm_attn_bwd_delta_f32": (C.c_int, [C.POINTER(LwmAttnArgs), C.c_void_p]),
    "lwm_attn_bwd_dq_f32": (C.c_int, [C.POINTER(LwmAttnArgs), C.c_void_p]),
    "lwm_attn_bwd_dkdv_f32": (C.c_int, [C.POINTER(LwmAttnArgs), C.c_void_p]),
    "lwm_rope_f32": (C.c_int, [LwmTensor4, LwmTensor4, C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p]),
    "lwm_rmsnorm_fwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p]),
    "lwm_rmsnorm_bwd_f32": (C.c_int, [C.c_void_p] * 7 + [C.c_int64, C.c_int32, C.c_void_p]),
    "lwm_swiglu_fwd_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_void_p]),
    "lwm_swiglu_bwd_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int64, C.c_void_p]),
    "lwm_softmax_ce_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int32, C.c_void_p]),
    "lwm_sum_f32": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "lwm_ring_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "lwm_ring_unique_id": (C.c_int, [C.c_void_p]),
    "lwm_ring_create_from_id": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "lwm_ring_create_transport": (C.c_int, [C.POINTER(LwmRingTransport), C.c_int32, C.c_int32, C.c_void_p,
                                            C.POINTER(C.c_void_p)]),
    "lwm_ring_destroy": (C.c_int, [C.c_void_p]),
    "lwm_ring_workspace_bytes": (C.c_int64, [C.c_int32] * 7),
    "lwm_ring_attn_fwd": (C.c_int, [C.c_void_p, C.POINTER(LwmRingArgs), C.c_void_p]),
    "lwm_ring_attn_bwd": (C.c_int, [C.c_void_p, C.POINTER(LwmRingArgs), C.c_void_p]),
    "lwm_ring_bytes_sent": (C.c_int64, [C.c_void_p]),
    "lwm_ring_planned_bytes": (C.c_int64, [C.c_int32] * 10),
    "lwm_ring_planned_bytes_table": (C.c_int64, [C.c_void_p] + [C.c_int32] * 7),
    "lwm_ring_selftest": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "lwm_ring_set_fetch_groups": (C.c_int, [C.c_void_p, C.c_int32]),
    "lwm_ring_last_form": (C.c_int, [C.c_void_p]),
    "lwm_ring_kv_keep_bytes": (C.c_int64, [C.c_int32] * 5),
    "lwm_ring_fetch_timeline": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32]),
    "lwm_ring_ipc_info_bytes": (C.c_int64, []),
    "lwm_ring_ipc_export": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "lwm_ring_ipc_connect": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lwm_ring_create_ipc": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "lwm_ring_ipc_destroy": (C.c_int, [C.c_void_p]),
    "lwm_attn_segment_blocks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "lwm_attn_combine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, LwmTensor4, C.c_void_p, C.c_void_p,
                                  C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "lwm_kv_cache_write": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int64,
                                    C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    "lwm_kv_cache_write_at": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                        C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    "lwm_rope_bf16": (C.c_int, [LwmTensor4, LwmTensor4, C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p]),
    "lwm_rmsnorm_fwd_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                      C.c_float, C.c_void_p]),
    "lwm_rmsnorm_bwd_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int32]),
    "lwm_rmsnorm_bwd_bf16": (C.c_int, [C.c_void_p] * 7 + [C.c_int64, C.c_int32, C.c_void_p]),
    "lwm_rmsnorm_bwd_res_bf16": (C.c_int, [C.c_void_p] * 8 + [C.c_int64, C.c_int32, C.c_void_p]),
    "lwm_swiglu_fwd_ld_bf16": (C.c_int, [C.c_void_p, C.c_int64] * 3 + [C.c_int64, C.c_int64, C.c_void_p]),
    "lwm_swiglu_bwd_ld_bf16": (C.c_int, [C.c_void_p, C.c_int64] * 5 + [C.c_int64, C.c_int64, C.c_void_p]),
    "lwm_wgrad_workspace_bytes": (C.c_int64, [C.c_int64] * 3),
    "lwm_wgrad_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p] + [C.c_int64] * 4 +
                       [C.c_void_p, C.c_int64, C.c_void_p]),
    "lwm_transpose_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "lwm_swiglu_fwd_bf16": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_void_p]),
    "lwm_swiglu_bwd_bf16": (C.c_int, [C.c_void_p] * 5 + [C.c_int64, C.c_void_p]),
    "lwm_softmax_ce_bf16": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int32, C.c_void_p]),
    "lwm_gemv_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "lwm_gemv_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "lwm_gemv_multi_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_void_p,
                                      C.c_int32, C.c_int32, C.c_void_p]),
    "lwm_gemv_fused_bf16": (C.c_int, [C.POINTER(LwmGemvArgs), C.c_void_p]),
    "lwm_cast_f32_to_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "lwm_sum_f32_to_bf16": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "lwm_conv2d_nhwc_f32": (C.c_int, [C.POINTER(LwmConvArgs), C.c_void_p]),
    "lwm_groupnorm_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int64, C.c_int32, C.c_int32]),
    "lwm_groupnorm_silu_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                                            C.c_float, C.c_int32, C.c_void_p]),
    "lwm_vq_sqnorm_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "lwm_vq_argmin_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "lwm_vq_gather_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "lwm_last_error": (C.c_char_p, []),
    "lwm_version": (C.c_int, []),
    "lwm_sizeof": (C.c_int, [C.c_int]),
}


def bind(lib):
    """Attach restype/argtypes for every declared symbol; raises if one is missing or if
    the struct mirrors above do not have the library's layout."""
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    for which, cls in ((0, LwmAttnArgs), (1, LwmConvArgs), (2, LwmRingArgs)):
        if lib.lwm_sizeof(which) != C.sizeof(cls):
            raise ImportError(f"{cls.__name__}: ctypes mirror is {C.sizeof(cls)} bytes, library has "
                              f"{lib.lwm_sizeof(which)} (include/lwm_hip.h and lwm_amd/_capi.py out of step)")
    return lib


class LwmError(RuntimeError):
    pass


def check(lib, rc, what):
    if rc != LWM_OK:
        msg = lib.lwm_last_error()
        raise LwmError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")
