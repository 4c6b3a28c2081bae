"""ctypes binding of libvlaser_hip.so (include/vlaser_hip.h).  The product path FAILS LOUDLY when the HIP
library is missing -- there is no CPU / PyTorch fallback (the CPU oracle lives in oracle/ and is test-only)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('VLASER_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libvlaser_hip.so')     # env: an alternative build of the same ABI
_lib = None

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_longlong, C.c_float


class GemmArgs(C.Structure):
    _fields_ = [('A', vp), ('W', vp), ('out', vp), ('M', i32), ('N', i32), ('K', i32), ('lda', i32), ('ldw', i32),
                ('ldo', i32), ('bias', vp), ('res', vp), ('ls', vp), ('q_out', vp), ('k_cache', vp), ('vt_cache', vp),
                ('rope_cos', vp), ('rope_sin', vp), ('pos_ids', vp), ('n_q_heads', i32), ('n_kv_heads', i32),
                ('s_max', i32), ('tok_per_batch', i32), ('slot_base', i32), ('vq', vp), ('vk', vp), ('vvt', vp),
                ('vit_heads', i32), ('vit_seq', i32), ('vit_seq_pad', i32), ('q_scale', f32), ('out_f32', vp), ('k_splits', i32),
                ('force_bm', i32), ('batch', i32), ('a_bs', i64), ('w_bs', i64), ('o_bs', i64), ('w_group', i32), ('aux_out', vp), ('ld_aux', i32), ('sumsq_part', vp), ('sumsq_cap', i32)]


class AttnArgs(C.Structure):
    _fields_ = [('q', vp), ('k', vp), ('vt', vp), ('out', vp), ('batch', i32), ('sq', i32), ('kv_len', i32),
                ('n_q_heads', i32), ('n_kv_heads', i32), ('head_dim', i32), ('q_bs', i64), ('q_hs', i64), ('q_ss', i64),
                ('k_bs', i64), ('k_hs', i64), ('vt_bs', i64), ('vt_hs', i64), ('o_bs', i64), ('o_ss', i64),
                ('ld_vt', i32), ('scale', f32), ('mode', i32), ('causal_off', i32), ('valid_len', vp),
                ('blk_start', i32), ('q_row_off', i32), ('part_m', vp), ('part_l', vp), ('part_o', vp), ('n_splits', i32),
                ('first_tok_kv_len', i32), ('lse_out', vp), ('dbg', vp), ('mask', vp), ('mask_bs', i64), ('mask_rs', i64)]


class SkinnyArgs(C.Structure):
    _fields_ = [('x', vp), ('partials', vp), ('n_partials', i32), ('norm_w', vp), ('eps', f32), ('h_out', vp),
                ('W', vp), ('M', i32), ('N', i32), ('K', i32), ('ldw', i32), ('n_valid', i32), ('tiles_per_unit', i32), ('k_splits', i32), ('out_f32', vp),
                ('out', vp), ('ldo', i32), ('bias', vp), ('q_out', vp), ('k_cache', vp), ('vt_cache', vp),
                ('rope_cos', vp), ('rope_sin', vp), ('pos_ids', vp), ('n_q_heads', i32), ('n_kv_heads', i32),
                ('s_max', i32), ('tok_per_batch', i32), ('slot_base', i32), ('attn_m', vp), ('attn_l', vp), ('attn_o', vp),
                ('attn_splits', i32), ('attn_group', i32), ('attn_nq', i32), ('dbg', vp)]


class VlaStageArgs(C.Structure):
    _fields_ = [('ids', vp), ('ids_out', vp), ('B', i32), ('T', i32), ('pad_id', i64), ('valid_in', vp), ('valid_is_i64', i32), ('valid_out', vp),
                ('proprio', vp), ('proprio_out', vp), ('n_proprio', i32), ('noise', vp), ('noise_out', vp), ('n_noise', i32),
                ('pix', vp), ('pix_out', vp), ('n_pix', i64), ('pix_dtype', i32), ('hw', i32), ('mean', f32 * 3), ('std', f32 * 3), ('call_ctr', vp),
                ('call_no', i32), ('itp_mask', vp), ('action_mask', vp), ('mask_dtype', i32), ('n_act', i32), ('pos_vlm', vp), ('pos_pro', vp), ('pos_act', vp),
                ('pos_vlm_out', vp), ('pos_pro_out', vp), ('pos_act_out', vp), ('pos_ride_out', vp), ('itp_bs', i64), ('itp_rs', i64), ('act_bs', i64), ('act_rs', i64),
                ('mask_slot', vp), ('mask_ld', i32)]


# enums (include/vlaser_hip.h)
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_LS_RES, EPI_RES, EPI_SWIGLU, EPI_QKV_ROPE, EPI_VIT_QKV, EPI_F32, EPI_PARTIAL, EPI_SWIGLU_BWD = range(11)
ATTN_FULL, ATTN_CAUSAL, ATTN_PREFIX, ATTN_DENSE = range(4)
PRO_PLAIN, PRO_NORM, PRO_ATTN = range(3)
SK_PARTIAL, SK_QKV_ROPE, SK_SWIGLU, SK_F32, SK_BIAS, SK_BIAS_SILU = range(6)

_SIGS = {
    'vlaser_gemm': [i32, C.POINTER(GemmArgs), vp],
    'vlaser_gemm_nn': [i32, C.POINTER(GemmArgs), vp],
    'vlaser_set_cu_budget': [i32],
    'vlaser_get_cu_budget': [],
    'vlaser_stream_create_cumask': [i32, i32, C.POINTER(vp)],
    'vlaser_stream_destroy': [vp],
    'vlaser_attn_prefill': [C.POINTER(AttnArgs), vp],
    'vlaser_attn_skinny': [C.POINTER(AttnArgs), vp],
    'vlaser_attn_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, i32, i32, vp],
    'vlaser_skinny': [i32, i32, C.POINTER(SkinnyArgs), vp],
    'vlaser_chain_qkv_supported': [i32, i32, i32],
    'vlaser_chain_gu_supported': [i32, i32, i32, i32, i32],
    'vlaser_chain_down_supported': [i32, i32, i32],
    'vlaser_chain_down_geometry': [i32, C.POINTER(i32), C.POINTER(i32)],
    'vlaser_chain_qkv': [C.POINTER(SkinnyArgs), vp],
    'vlaser_chain_attn_splits': [i32],
    'vlaser_chain_attn': [C.POINTER(AttnArgs), vp],
    'vlaser_chain_oproj_supported': [i32, i32, i32, i32, i32, i32],
    'vlaser_chain_oproj': [C.POINTER(SkinnyArgs), vp],
    'vlaser_chain_gu': [C.POINTER(SkinnyArgs), vp],
    'vlaser_chain_down': [vp, i32, vp, vp, vp, i32, i32, i32, vp, vp],
    'vlaser_chain_down2_supported': [i32, i32, i32],
    'vlaser_chain_qkv2_supported': [i32, i32, i32],
    'vlaser_chain_qkv_set_waves': [i32],
    'vlaser_chain_down2': [vp, i32, vp, vp, i32, i32, i32, vp, vp],
    'vlaser_layernorm': [vp, vp, vp, vp, i32, i32, f32, vp],
    'vlaser_rmsnorm': [vp, vp, vp, i32, i32, f32, vp],
    'vlaser_im2col': [vp, vp, i32, i32, i32, vp],
    'vlaser_vit_assemble': [vp, vp, vp, vp, i32, i32, i32, vp],
    'vlaser_pixel_shuffle_ln': [vp, vp, vp, vp, i32, i32, i32, f32, i32, vp],
    'vlaser_pixel_shuffle': [vp, vp, i32, i32, i32, i32, vp],
    'vlaser_embed_merge': [vp, i32, vp, vp, i32, vp, i32, i64, i64, i32, vp, vp, vp],
    'vlaser_argmax': [vp, i32, i32, vp, vp, vp, i32, vp, i32, vp],
    'vlaser_argmax_ws_bytes': [i32],
    'vlaser_vla_prep': [vp, vp, vp, vp, i32, i32, i32, f32, f32, vp],
    'vlaser_small_linear': [vp, vp, vp, vp, i32, i32, i32, vp],
    'vlaser_vla_step': [vp, vp, i32, i32, i32, vp, f32, vp, vp, vp, vp, vp, f32, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    'vlaser_vla_euler': [vp, vp, i32, i32, vp, f32, vp, vp, vp, i32, i32, f32, f32, i32, vp, vp, vp, i32, i32, i32, vp],
    'vlaser_vla_stage': [C.POINTER(VlaStageArgs), vp],
    'vlaser_cast_f32_bf16': [vp, vp, i64, vp],
    'vlaser_normalize_u8': [vp, vp, i32, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), vp],
    'vlaser_resample_ksize': [i32, i32],
    'vlaser_resample_coeffs': [i32, i32, vp, vp],
    'vlaser_resize_u8': [vp, i32, i32, i64, vp, i64, vp, i32, i32, i64, vp, vp, i32, vp, vp, i32, vp],
    'vlaser_tiles_normalize_u8': [vp, i64, i32, i32, i32, vp, i32, C.POINTER(f32), C.POINTER(f32), vp],
    'vlaser_avg_update': [vp, vp, i64, f32, i32, vp],
    'vlaser_ce_rows': [vp, vp, i32, i32, i64, vp, vp, i64, vp],
    'vlaser_reduce_norm': [vp, vp, i32, vp, vp, i32, vp, vp, f32, vp, vp, i32, i32, vp],
    'vlaser_gemm_tn': [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp],
    'vlaser_gemm_tn_lds': [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp],
    'vlaser_gemm_tn_grouped': [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i64, i64, i32, i64, i64, i64, vp],
    'vlaser_transpose': [vp, vp, i32, i32, i32, i32, i32, i32, i64, i64, i32, i64, i64, vp],
    'vlaser_rope_bwd_pack': [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    'vlaser_rmsnorm_bwd': [vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp, i32, vp],
    'vlaser_colsum_partials_multi': [vp, i64, i32, i32, i32, vp, vp, vp],
    'vlaser_colsum_bf16': [vp, vp, i32, i32, i32, vp],
    'vlaser_colsum_mul': [vp, vp, vp, i32, i32, i32, f32, vp, vp],
    'vlaser_swiglu': [vp, vp, i32, i32, vp],
    'vlaser_swiglu_bwd': [vp, vp, vp, i32, i32, vp],
    'vlaser_ce_dlogits': [vp, vp, vp, vp, i32, i32, i64, i32, f32, i64, vp],
    'vlaser_embed_scatter_add': [vp, vp, vp, vp, vp, i32, i32, i64, vp],
    'vlaser_gelu_bwd': [vp, vp, vp, i64, vp],
    'vlaser_adamw': [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, f32, i32, vp],
    'vlaser_sumsq': [vp, i64, vp, vp, vp],
    'vlaser_sumsq_chunks': [vp, vp, i32, vp, vp],
    'vlaser_sumsq_rows': [vp, vp, vp, i32, i32, i64, vp, i32, vp],
    'vlaser_sum_partials': [vp, i64, vp, i32, vp],
    'vlaser_adamw_clipped': [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, f32, vp, f32, i32, vp],
    'vlaser_grad_accumulate': [vp, vp, i64, f32, i32, i32, vp],
    'vlaser_silu': [vp, vp, i64, vp],
    'vlaser_silu_bwd': [vp, vp, vp, i64, vp],
    'vlaser_attn_rows_bwd': [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp],
    'vlaser_attn_rows_bwd_ex': [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp],
    'vlaser_attn_rows_bwd_ws_floats': [i32],
    'vlaser_attn_bwd_pds_masked': [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, i32, i32, vp],
    'vlaser_rope_bwd_pack_ex': [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp],
    'vlaser_layernorm_bwd': [vp, vp, vp, vp, vp, i32, i32, f32, vp],
    'vlaser_scale_cols': [vp, vp, vp, i32, i32, i32, i32, f32, vp],
    'vlaser_pixel_unshuffle': [vp, vp, i32, i32, i32, i32, vp],
}


class VlaserHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VlaserHipError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                                 f'or `make -C vlaser_amd/csrc` -- there is no CPU fallback')
        # torch first: it ships its own libamdhip64 / libhsa-runtime64, and whichever copy of that soname is mapped first serves the whole
        # process.  Loading this library before torch maps /opt/rocm's runtime instead, and torch's later device initialisation then ends in
        # "no ROCm-capable device is detected" (seen with build() followed by smoke() in one process on the GPU box).
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        l.vlaser_last_error.restype = C.c_char_p
        l.vlaser_abi_version.restype = i32
        for name, sig in _SIGS.items():
            fn = getattr(l, name)
            fn.argtypes = sig
            fn.restype = i32
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        raise VlaserHipError(f'{what} failed ({rc}): {lib().vlaser_last_error().decode()}')
