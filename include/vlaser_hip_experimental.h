/*
 * libvlaser_hip.so -- EXPERIMENTAL entry points: measured, kept for their A/B tests and timelines, NOT used by any default path and not part of the
 * drop-in boundary (INTEGRATION.md does not list them).  They may change without an ABI bump.  Build: the same `make` (csrc/euler.hip, csrc/attn_o.hip).
 *   vlaser_fused_ogu   (r03)  PiZeroInference(euler_opts='...,gu16,fuse_ogu'):  +2.0 us per layer-step in-chain  (profiles/r03*_fused_ogu*)
 *   vlaser_attn_oproj  (r04)  PiZeroInference(euler_opts='...,fuse_ao'):        10.2-10.7 us vs 10.5 us for the pair it replaces (profiles/r04i_attn_oproj_timeline.md)
 */
#ifndef VLASER_HIP_EXPERIMENTAL_H
#define VLASER_HIP_EXPERIMENTAL_H
#include "vlaser_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- fused layer-step launches (ABI 4, r03): one launch instead of two of the five of a <= 5-row decoder layer-step -------------------------
 * vlaser_fused_ogu = vlaser_skinny(VL_PRO_ATTN, VL_SK_PARTIAL) [o_proj: flash-decoding merge of the vlaser_attn_skinny partials + split-K GEMV]
 *                  + vlaser_skinny(VL_PRO_NORM, VL_SK_SWIGLU)  [residual + split-K reduce + Qwen2RMSNorm + gate/up GEMV + SwiGLU],
 * bit-identical to the pair (16-row units), replacing the same reference call sites (joint_model.py:140-232,410-696 for the action mixture of
 * every Euler step, pizero_internvl.py:884-924).  The first H/16 * ks_o workgroups compute the o_proj partial tiles and hand them to all 256
 * workgroups INSIDE the launch (write-through stores + arrival counters), while every workgroup's share of the gate/up weights is already in
 * flight.  sync: device uint32[VL_FUSED_SYNC_WORDS], ZEROED BY THE CALLER on the stream before every launch (one memset over all launch
 * slots of a HIP graph); sync[VL_FUSED_SYNC_ERR] != 0 afterwards = a bounded wait expired (result invalid, nothing hangs). */
#define VL_FUSED_SYNC_WORDS 160
#define VL_FUSED_SYNC_ERR 128
typedef struct {
  /* o_proj: as VlaserSkinnyArgs' VL_PRO_ATTN prologue */
  const float* attn_m; const float* attn_l; const float* attn_o; int attn_splits, attn_group, attn_nq;
  const void* Wo;      /* ops.pack_skinny(o_proj.weight, ks_o, tiles_per_unit = 1) */
  int K_o, ks_o;       /* n_q_heads * 128; cross-workgroup K splits */
  float* part_o;       /* fp32 [ks_o][M][H] split-K slabs (written write-through, read by every workgroup) */
  /* gate/up: as VlaserSkinnyArgs' VL_PRO_NORM prologue + VL_SK_SWIGLU on 16-row lane-local units */
  const void* h_in;    /* bf16 [M,H] residual stream */
  const void* norm_w; float eps;
  void* h_out;         /* bf16 [M,H] = bf16(h_in + o_proj), written by workgroup 0 (may be null) */
  const void* Wgu;     /* ops.pack_skinny(ops.pack_gate_up8(gate, up), 1, 1) */
  int M, H, N_gu, n_valid_gu;   /* rows; hidden; packed gate/up rows (2 I) and their un-padded count */
  void* act; int ld_act;        /* bf16 [M, ld_act]: silu(gate) * up */
  unsigned int* sync;
  int cons_delay;               /* tuning: consumer-only workgroups start their weight stream this many 10-ns ticks after their start (0 = at once) */
  unsigned long long* dbg;      /* optional: per-workgroup timestamps [256][8] (wall_clock64, 100 MHz) for kernel tuning */
} VlaserFusedOguArgs;
int vlaser_fused_ogu(const VlaserFusedOguArgs* args, vl_stream_t stream);

/* ---- attention + o_proj of a <= 16-row decoder layer-step in one launch (ABI 5, r04; batch 1): replaces vlaser_attn_skinny followed by
 * vlaser_skinny(VL_PRO_ATTN, VL_SK_PARTIAL) at joint_model.py:636-671 (attention of the proprio / action rows over the cached prefix + their own block,
 * then o_proj).  Every workgroup recomputes the attention of its kv group from K / V^T tiles staged in LDS by coalesced LDS-DMA (only the keys a row can
 * see: the valid prefix and [blk_start, kv_len)) and owns 16 output columns of W_o; out_f32 = n_kv_heads partial slabs [sq, N] fp32 (one per kv head)
 * for the consumer's split-K reduction.  wo_packed = ops.pack_skinny(o_proj.weight [N, n_q_heads*128], k_splits = n_kv_heads, tiles_per_unit = 1).
 * `a` as for vlaser_attn_skinny (part_* / n_splits unused); GQA group 2 / 4 / 6 / 8, group * sq <= 32, blk_start a multiple of 16. */
int vlaser_attn_oproj(const VlaserAttnArgs* a, const void* wo_packed, float* out_f32, int N, vl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
