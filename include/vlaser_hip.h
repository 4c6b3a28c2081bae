/*
 * libvlaser_hip.so -- C ABI of the MI355X-native (gfx950) Vlaser forward hot path.
 *
 * The reference (OpenGVLab/Vlaser) is pure Python and has no FFI of its own; its operator seams
 * are Python duck-typing.  This header is the C-ABI the seams bind to through ctypes
 * (vlaser_amd/_lib.py; INTEGRATION.md shows the reference-side stubs).  Each entry point cites the reference
 * function it replaces.
 *
 * Conventions: every pointer is a DEVICE pointer owned by the caller (PyTorch allocates); bf16 tensors are raw
 * uint16 bits; kernels are asynchronous on `stream`, never allocate, never synchronise; return 0 on success,
 * negative on error (vlaser_last_error() gives the message).  Thread-safe w.r.t. distinct streams.
 */
#ifndef VLASER_HIP_H
#define VLASER_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vl_stream_t; /* hipStream_t */

const char* vlaser_last_error(void);
int vlaser_abi_version(void);

/* ---- GEMM: out = epilogue(A[M,K] @ W[N,K]^T), bf16 in, fp32 accumulate on MFMA ------------------------------
 * replaces nn.Linear / F.linear call sites: modeling_intern_vit.py:196,208,256-257 (qkv, proj, fc1, fc2),
 * modeling_internvl_chat.py:91-93 (mlp1), HF Qwen2Attention q/k/v/o_proj and Qwen2MLP gate/up/down
 * (joint_model.py:449-450,573-578,694; modeling_internvl_chat.py:194-203). */
enum {
  VL_EPI_NONE = 0,       /* out = acc                                              */
  VL_EPI_BIAS = 1,       /* out = acc + bias[n]                                    */
  VL_EPI_BIAS_GELU = 2,  /* out = gelu_erf(acc + bias[n])                          */
  VL_EPI_BIAS_LS_RES = 3,/* out = res[m,n] + ls[n]*(acc + bias[n])   (ViT layer-scale residual, :291-293) */
  VL_EPI_RES = 4,        /* out = res[m,n] + acc                     (Qwen2 residual) */
  VL_EPI_SWIGLU = 5,     /* W = gate/up interleaved in 16-row groups; out[m, n/2] = silu(g)*u */
  VL_EPI_QKV_ROPE = 6,   /* fused q/k/v projection + bias + RoPE + KV-cache write (Qwen2, head_dim 128) */
  VL_EPI_VIT_QKV = 7,    /* fused ViT qkv + bias, q*scale; writes Q,K [T,H,S,64] and V^T [T,H,64,Spad] */
  VL_EPI_F32 = 8,        /* out (float32) = acc   (full-vocab logits) */
  VL_EPI_PARTIAL = 9,    /* split-K: out_f32[ks, m, n] = partial over K slice ks (reduced by vlaser_reduce_norm) */
  VL_EPI_SWIGLU_BWD = 10 /* vlaser_gemm_nn only: acc = d(act) [M, I]; res = the forward's bf16 pre-activations [M, 2I] (packed [gate16|up16], row
                            stride ldo); out [M, 2I] = d(gate), d(up) in the same packing -- the dgrad of down_proj with swiglu's backward as its epilogue */
};

typedef struct {
  const void* A; const void* W; void* out;
  int M, N, K;
  int lda, ldw, ldo;
  const void* bias;   /* bf16 [N] */
  const void* res;    /* bf16 [M, ldo] */
  const void* ls;     /* bf16 [N] */
  /* VL_EPI_QKV_ROPE */
  void* q_out;              /* bf16 [M, n_q_heads*128] */
  void* k_cache;            /* bf16 [B, n_kv, S_max, 128] */
  void* vt_cache;           /* bf16 [B, n_kv, 128, S_max]  (V transposed) */
  const float* rope_cos;    /* fp32 [n_pos, 64] */
  const float* rope_sin;
  const int32_t* pos_ids;   /* int32 [M] */
  int n_q_heads, n_kv_heads, s_max, tok_per_batch, slot_base;
  /* VL_EPI_VIT_QKV */
  void* vq; void* vk; void* vvt; int vit_heads, vit_seq, vit_seq_pad; float q_scale;
  /* VL_EPI_PARTIAL */
  float* out_f32; int k_splits;
  int force_bm;             /* 0 = heuristic; 32 / 64 / 128 = register-staged tile height; 1100 / 1200 / 1300 / 1440 / 1500 = LDS-DMA pipeline
                               128x128 / 128x256 / 256x256 / 144x128 / 64x128 (tests, tools/micro/gemm_lab) */
  /* batched GEMM (attention backward through materialised per-head matrices): blockIdx.z = batch index z;
   * A += z*a_bs, out += z*o_bs, W += (z / w_group)*w_bs  (element strides; batch 0/1 = plain GEMM) */
  int batch; long long a_bs, w_bs, o_bs; int w_group;
  /* VL_EPI_SWIGLU, optional: also keep the bf16 pre-activations [M, N] (gate / up in the packed 16-row interleave, exactly what
   * VL_EPI_NONE would write) -- the SFT forward saves them for swiglu's backward instead of running the GEMM unfused + a swiglu pass.
   * VL_EPI_BIAS_GELU, optional (ABI 4): aux_out [M, N] = bf16(acc + bias), the pre-activation GELU's backward needs (the projector's first
   * Linear, modeling_internvl_chat.py:89-94, was evaluated twice in r02: once with BIAS, once with BIAS_GELU); ld_aux % 4 == 0, 8-byte aligned */
  void* aux_out; int ld_aux;
  /* ABI 5, set by vlaser_gemm_tn_lds only (vlaser_gemm / vlaser_gemm_nn refuse it): see there */
  float* sumsq_part; int sumsq_cap;
} VlaserGemmArgs;

int vlaser_gemm(int epi, const VlaserGemmArgs* args, vl_stream_t stream);
/* (ABI 6) CUs the GEMM tile heuristics may count on (64..256, default 256; set returns the previous value, out-of-range values are refused): for a host that runs these
 * launches on a CU-masked stream beside another kernel's resident workgroups (RCCL's channels during zero_stage1_config.json's overlap_comm) -- the grids are then sized for
 * the CUs the mask leaves.  Without a mask a lower budget measured no gain (profiles/r05_rccl_contention.md). */
int vlaser_set_cu_budget(int cus);
int vlaser_get_cu_budget(void);
/* (ABI 7) A HIP stream whose kernels may only run on CUs [first_cu, first_cu + n_cus) (hipExtStreamCreateWithCUMask).  The data-parallel SFT step's switchable
 * exchange (`VLASER_DP_EXCHANGE=capi`, vlaser_amd/rccl_capi.py) puts RCCL's reduce-scatter / all-gather on such a stream and the forward + backward on streams masked to
 * the complement, so that RCCL's channel workgroups never share a CU with a GEMM workgroup -- what DeepSpeed's `overlap_comm: true` comm stream
 * (Vlaser_VLM/internvl_chat/zero_stage1_config.json, the engine behind internvl_chat_finetune.py:1041-1057) leaves to the hardware scheduler: measured x1.13 instead of
 * x1.23-1.34 on the forward + backward beside 8-32 resident streaming workgroups (profiles/r05j_rccl_shadow_masks.md).  The caller owns the stream (wrap it with
 * torch.cuda.ExternalStream) and destroys it with vlaser_stream_destroy. */
int vlaser_stream_create_cumask(int first_cu, int n_cus, vl_stream_t* out);
int vlaser_stream_destroy(vl_stream_t stream);

/* NN form: out[M,N] = A[M,K] @ B[K,N], B = args->W row-major with row stride args->ldw ("k-major").  The dgrad of an nn.Linear
 * (dX = dY @ W, autograd of modeling_internvl_chat.py:194-203 / joint_model.py:410-696) reads the forward weight [N_out, K_in] as it
 * is stored: no transposed copy.  Epilogues VL_EPI_NONE (bf16 out), VL_EPI_F32, VL_EPI_PARTIAL (split-K fp32 slabs) and VL_EPI_SWIGLU_BWD; batched like
 * vlaser_gemm for NONE / F32 (the attention backward's dP = dO V^T and dQ = dS K read V^T / K as the cache holds them); K % 64 == 0, N % 8 == 0;
 * force_bm: 0 or an LDS-DMA configuration code. */
int vlaser_gemm_nn(int epi, const VlaserGemmArgs* args, vl_stream_t stream);

/* ---- attention ------------------------------------------------------------------------------------------------
 * vlaser_attn_prefill replaces FlashAttention.forward / InternAttention._naive_attn (modeling_intern_vit.py:51-96,
 * 210-227; non-causal, hd 64) and HF eager_attention_forward / flash_attention_2 for Qwen2 prefill (causal GQA,
 * hd 128; joint_model.py:631-656 with the vlm rows of the block mask).
 * vlaser_attn_skinny (writes split partials, merged by vlaser_skinny's VL_PRO_ATTN prologue) replaces the same call for
 * <=16 query tokens over the KV cache: the proprio row of the joint
 * prefill, the 4 action tokens of every Euler step (pizero_internvl.py:896-908) and single-token greedy decode.
 * Masks are passed as descriptors instead of dense [B,1,Sq,Skv] additive tensors (pizero_internvl.py:517-603):
 * key j is visible to query row i iff  j < lim1(i)  ||  blk_start <= j < kv_len (rows >= blk_start only). */
enum { VL_ATTN_FULL = 0, VL_ATTN_CAUSAL = 1, VL_ATTN_PREFIX = 2,
       VL_ATTN_DENSE = 3 /* (ABI 8) a dense ADDITIVE mask as the reference hands it to eager_attention_forward (joint_model.py:636-656): see `mask` below */ };

typedef struct {
  const void* q;   /* bf16, element (b,h,s,d) at q + b*q_bs + h*q_hs + s*q_ss + d */
  const void* k;   /* bf16 [B, n_kv, S_max, hd]:  b*k_bs + kvh*k_hs + s*hd + d (RoPE already applied) */
  const void* vt;  /* bf16 [B, n_kv, hd, ld_vt]:  b*vt_bs + kvh*vt_hs + d*ld_vt + s  (V transposed, zero padded) */
  void* out;       /* bf16: b*o_bs + s*o_ss + h*hd + d */
  int batch, sq, kv_len, n_q_heads, n_kv_heads, head_dim;
  long long q_bs, q_hs, q_ss, k_bs, k_hs, vt_bs, vt_hs, o_bs, o_ss;
  int ld_vt;
  float scale;
  int mode;                 /* VL_ATTN_* */
  int causal_off;           /* CAUSAL: row i sees keys <= i + causal_off */
  const int32_t* valid_len; /* PREFIX: int32 [B] valid image/text prefix length (device) */
  int blk_start;            /* PREFIX: first key of the proprio/action block */
  int q_row_off;            /* PREFIX: global row index of query row 0 */
  /* vlaser_attn_skinny only: flash-decoding partials per (b, kv head, split), rows r = hg*sq + tok (<= 32) */
  float* part_m; float* part_l; /* fp32 [B, n_kv, n_splits, 32] */
  float* part_o;                /* fp32 [B, n_kv, n_splits, 32, 128], unnormalised */
  int n_splits;                 /* 1..8 key splits (grid.y) */
  int first_tok_kv_len;         /* > 0: query token 0 of each batch element sees the block keys [blk_start, first_tok_kv_len) only -- the
                                   proprio row riding in front of the action rows (pizero_internvl.py:517-587: proprio sees prefix + self) */
  float* lse_out;               /* vlaser_attn_prefill, optional (ABI 4): fp32 [B, n_q_heads, sq] base-2 log-sum-exp of the scaled scores, kept for vlaser_attn_bwd */
  unsigned long long* dbg;      /* vlaser_attn_skinny, optional (ABI 6): per-workgroup timestamps [grid][8] (wall_clock64, 100 MHz) for kernel tuning */
  /* (ABI 8) VL_ATTN_DENSE (vlaser_attn_prefill at head_dim 128, vlaser_chain_attn): fp32 additive mask, element (b, query token i, key j) at
   * mask[b * mask_bs + i * mask_rs + j] -- the same value for every head, as the reference's [B,1,Sq,Skv] tensors.  score = q.k * scale + mask; a value below -1e30
   * (the reference writes finfo(dtype).min) hides the key outright; a query row with no visible key returns 0 (the reference: the mean of V -- nobody reads such
   * rows).  mask_rs must be a multiple of 4 floats >= kv_len rounded up to 64 (prefill) / 32 (chain) keys, the base 16-byte aligned: values past kv_len are read and ignored. */
  const float* mask;
  long long mask_bs, mask_rs;
} VlaserAttnArgs;

int vlaser_attn_prefill(const VlaserAttnArgs* args, vl_stream_t stream);
int vlaser_attn_skinny(const VlaserAttnArgs* args, vl_stream_t stream);

/* ---- skinny (M <= 16 rows) weight-streaming GEMV on MFMA, HBM-bound --------------------------------------------
 * out = epilogue( prologue(x)[M,K] @ W[N,K]^T ).  Replaces the same nn.Linear call sites as vlaser_gemm when the
 * activation has <= 16 rows: every layer of the 10 Euler steps (joint_model.py:140-232 with only the action
 * mixture active) and of greedy decode, plus lm_head on the last position (modeling_internvl_chat.py:204).
 * Prologue VL_PRO_NORM fuses: residual + split-K partial reduction of the producer, bf16 rounding of the residual
 * stream, Qwen2RMSNorm.  Split-K partials are reduced by the CONSUMER's prologue (deterministic, no atomics). */
enum { VL_PRO_PLAIN = 0, VL_PRO_NORM = 1, VL_PRO_ATTN = 2 /* x = merge of vlaser_attn_skinny partials (o_proj) */ };
enum {
  VL_SK_PARTIAL = 0,   /* out_f32[ks, m, n] = partial over this block's K slice */
  VL_SK_QKV_ROPE = 1,  /* as VL_EPI_QKV_ROPE */
  VL_SK_SWIGLU = 2,    /* as VL_EPI_SWIGLU */
  VL_SK_F32 = 3,       /* out_f32[m, n] = acc (+bias)   (logits) */
  VL_SK_BIAS = 4,      /* out bf16 = acc + bias */
  VL_SK_BIAS_SILU = 5  /* out bf16 = silu(acc + bias) */
};

typedef struct {
  const void* x;          /* PLAIN: bf16 [M,K]; NORM: residual stream h_in bf16 [M,K] */
  const float* partials;  /* NORM: fp32 [n_partials, M, K] added to h_in (may be null) */
  int n_partials;
  const void* norm_w;     /* NORM: bf16 [K] */
  float eps;
  void* h_out;            /* NORM: bf16 [M,K] = bf16(h_in + sum partials), written by block 0 (may be null) */
  const void* W;          /* bf16, FRAGMENT-MAJOR packed [k_splits][N/(16 T)][8 waves][K/(k_splits*256)][T tiles][64][8], T = tiles_per_unit (ops.pack_skinny) */
  int M, N, K, ldw;       /* N = padded row count (multiple of 32); ldw unused */
  int n_valid;            /* un-padded N (0 = N): logits / partial row length */
  int tiles_per_unit;     /* 0/2: units of 32 rows; 6: units of 96 rows; 1: units of 16 rows (split-K partial outputs only); packing must match */
  int k_splits;           /* grid.y; K % (k_splits*256) == 0 */
  float* out_f32;
  void* out; int ldo;
  const void* bias;       /* bf16 [N] */
  /* VL_SK_QKV_ROPE (same meaning as in VlaserGemmArgs) */
  void* q_out; void* k_cache; void* vt_cache;
  const float* rope_cos; const float* rope_sin; const int32_t* pos_ids;
  int n_q_heads, n_kv_heads, s_max, tok_per_batch, slot_base;   /* slot_base < 0: cache slot of row m = pos_ids[m] (read on the device: graph-replayable decode) */
  /* VL_PRO_ATTN: x[m = b*nq+tok][h*128+d] = sum_s o_s e^(m_s-M) / sum_s l_s e^(m_s-M) */
  const float* attn_m; const float* attn_l; const float* attn_o; int attn_splits, attn_group, attn_nq;
  unsigned long long* dbg; /* optional: per-block timestamps [grid][8] (wall_clock64, 100 MHz) for kernel tuning */
} VlaserSkinnyArgs;

int vlaser_skinny(int prologue, int epi, const VlaserSkinnyArgs* args, vl_stream_t stream);

/* ---- (ABI 6, r05) the <= 16-row layer-step chain rebuilt for latency (csrc/chain.hip): same call sites as vlaser_skinny -- joint_model.py:140-232 with only the
 * action mixture active (10 Euler steps x 28 layers), HF Qwen2DecoderLayer under greedy decode -- with every shape a compile-time parameter, every load of a launch
 * issued in one straight-line block, and the down projection publishing the next layer's residual stream ONCE (bf16) instead of fp32 split-K slabs that every
 * consumer workgroup re-reduces.  All three take device pointers, are asynchronous on `stream`, return 0 / negative like everything else here.
 *   vlaser_chain_qkv : args as vlaser_skinny(VL_PRO_NORM, VL_SK_QKV_ROPE) with tiles_per_unit = 1, n_partials = 0 (x = the published residual stream) or 2 (see vlaser_chain_down2): one wave per 16-row
 *                      unit, RMSNorm + q/k/v GEMV + bias + RoPE + KV-cache scatter;  hidden 768 / 1536
 *   vlaser_chain_gu  : args as vlaser_skinny(VL_PRO_NORM, VL_SK_SWIGLU) with tiles_per_unit = 2 (or 1: 16-row lane-local units) and 2 or 3 producer slabs: bit-identical outputs, all units of a workgroup
 *                      requested up front; h_out (nullable) = bf16(x + sum partials), the residual vlaser_chain_down adds back
 *   vlaser_chain_down: h_out [M, N] = bf16(res + x [M, K] @ W^T), W = down_proj.weight packed by ops.pack_down4 ([workgroup][7 waves][10 loads][groups][16 blocks][cols][8]):
 *                      a workgroup owns `groups` x `cols` output columns (vlaser_chain_down_geometry) over the whole K = 8960 (no cross-workgroup split-K); res != h_out
 * *_supported(...) != 0 tells the host surface whether a geometry has a variant (it keeps vlaser_skinny otherwise). */
/*   vlaser_chain_attn : vlaser_attn_skinny's arguments; one WAVE per (kv head, key split, batch element) with 2 chunks of 32 keys each (no LDS merge, no barrier);
 *                       part_m = fp32 [B, n_kv, n_splits, 32, 2] (m, l) pairs, part_o = BF16 [B, n_kv, n_splits, 32, 128] NORMALISED rows (part_l unused);
 *                       n_splits = vlaser_chain_attn_splits(kv_len) = ceil(ceil(kv_len / 32) / 2) <= 16
 *   vlaser_chain_oproj: args as vlaser_skinny(VL_PRO_ATTN, VL_SK_PARTIAL) with tiles_per_unit = 1, attn_m / attn_o = what vlaser_chain_attn wrote: merge of the splits
 *                       (weights l 2^(m - max m)) -> o_proj GEMV -> fp32 split-K slabs */
int vlaser_chain_attn_splits(int kv_len);
int vlaser_chain_attn(const VlaserAttnArgs* args, vl_stream_t stream);
int vlaser_chain_oproj_supported(int M, int N, int K, int k_splits, int attn_splits, int group);
int vlaser_chain_oproj(const VlaserSkinnyArgs* args, vl_stream_t stream);
int vlaser_chain_qkv_supported(int M, int N, int K);
int vlaser_chain_gu_supported(int M, int N, int K, int n_partials, int tiles_per_unit);
int vlaser_chain_down_supported(int M, int N, int K);
int vlaser_chain_down_geometry(int N, int* cols, int* groups);      /* returns the workgroup count; cols (3 / 4) per column group, groups (1 / 2) per workgroup: the layout ops.pack_down4 writes */
int vlaser_chain_qkv(const VlaserSkinnyArgs* args, vl_stream_t stream);
int vlaser_chain_gu(const VlaserSkinnyArgs* args, vl_stream_t stream);
int vlaser_chain_down(const void* x, int ldx, const void* W, const void* res, void* h_out, int M, int N, int K, unsigned long long* dbg, vl_stream_t stream);
/*   vlaser_chain_down2: the same contraction as two K halves on 256 workgroups of six columns (grid.y = half): out_f32 [2][M][N] fp32 slabs, no residual, no rounding;
 *                      W = ops.pack_down4(W, k_splits=2).  The consumer is vlaser_chain_qkv with n_partials = 2 (h = bf16(x + slab 0 + slab 1), stored to h_out by unit 0's
 *                      wave for the o_proj -> gate/up seam): half the activation bytes per CU of vlaser_chain_down, 24 more L2 loads per lane in the q/k/v launch */
int vlaser_chain_down2_supported(int M, int N, int K);
int vlaser_chain_qkv2_supported(int M, int N, int K);
/* tools / tests: 1 = the one-wave-per-unit kernel of vlaser_chain_qkv at hidden 1536 too (default 0: two waves there, one K half each); returns the previous value */
int vlaser_chain_qkv_set_waves(int n);
int vlaser_chain_down2(const void* x, int ldx, const void* W, float* out_f32, int M, int N, int K, unsigned long long* dbg, vl_stream_t stream);

/* (The two fused layer-step launches measured in r03 / r04 -- o_proj -> gate/up with an in-launch hand-off, +2.0 us in-chain; attention + o_proj in one launch,
 * break-even -- lost to the chain kernels above and left the library in r05 together with include/vlaser_hip_experimental.h: profiles/r03c_euler_fusion.md,
 * profiles/r04i_attn_oproj_timeline.md keep their measurements.) */

/* ---- fused attention backward (ABI 4, r03): the backward of vlaser_attn_prefill's causal / valid-prefix attention (HF eager_attention_forward /
 * flash_attention_2 autograd, modeling_internvl_chat.py:194-203) without materialised score matrices: two deterministic kernels (no atomics) --
 *   dQ:      one workgroup per (64 query rows, q head) walks the visible key tiles:  S^T, P^T = exp2(S^T - lse), dP^T = V dO^T, dS^T, dQ^T += K^T dS^T
 *   dK, dV:  one workgroup per (64 keys, q head) walks the query tiles that see them: S, P, dP, dS, dV^T += dO^T P, dK^T += Q^T dS
 * q / o / d_o / dq: bf16 [S, n_q*128]; k: [n_kv, s_max, 128] (post-RoPE), vt: [n_kv, 128, s_max]; lse: fp32 [n_q, S] from vlaser_attn_prefill;
 * dk / dv: bf16 [S, n_q*128] -- ONE partial per Q head (vlaser_rope_bwd_pack sums the heads of a kv group); delta_ws: fp32 [n_q, S] scratch.
 * Key k is visible to query q iff k < kv_valid and (k <= q when causal != 0).  head_dim 128. */
int vlaser_attn_bwd(const void* q, const void* k, const void* vt, const void* o, const void* d_o, const float* lse, float* delta_ws, void* dq, void* dk,
                    void* dv, int S, int n_q, int n_kv, int s_max, float scale, int causal, int kv_valid, int head_dim /* 128, checked */, vl_stream_t stream);

/* Weight-gradient GEMM: out[M,N] (bf16) = At^T @ Wt, At [K,M] and Wt [K,N] row-major bf16 (contraction along rows).
 * dW = dY^T X of every nn.Linear on the SFT path (autograd of modeling_internvl_chat.py:194-203): At = dY [S,N_out],
 * Wt = X [S,K_in].  Rows 16-byte aligned and readable up to M / N rounded up to 8 columns (ld >= that).
 * sumsq_part (ABI 5, optional, NULL = off): float[sumsq_cap]; every (workgroup, wave) of the launch writes the sum of the squares of the bf16 values it
 * stored into its own slot (slots the launch does not reach are left untouched; which slots it reaches depends on M, N only).  The caller adds the
 * slots in index order (vlaser_sum_partials): the weight gradient's share of clip_grad_norm_'s global norm (internvl_chat_finetune.py:1041-1057,
 * max_grad_norm 1.0) without reading the gradient back.  sumsq_cap >= ceil(M/64) * ceil(N/128) * 4 always suffices. */
int vlaser_gemm_tn(const void* At, const void* Wt, void* out, int M, int N, int K, int ldat, int ldwt, int ldo, float* sumsq_part, int sumsq_cap,
                   vl_stream_t stream);
/* The same TN product on the LDS-DMA GEMM pipeline (8 waves, 128x128 .. 256x256 tiles, both operands global -> LDS without staging registers,
 * fragments through the transposing LDS read).  Contract: K is a whole number of 64-row tiles; rows K_true..K of At are ZERO and those of Wt finite
 * (the SFT step pads its sequence axis: sft.py `_wgrad`); ldat, ldwt multiples of 8 and >= M / N rounded up to 8 (rows are read in 16-byte pieces).  force_cfg: 0 = heuristic, or 1100 / 1105 / 1200 / 1300.
 * sumsq_part / sumsq_cap: as for vlaser_gemm_tn; sumsq_cap >= ceil(M/128) * ceil(N/128) * 8 always suffices. */
int vlaser_gemm_tn_lds(const void* At, const void* Wt, void* out, int M, int N, int K, int ldat, int ldwt, int ldo, int force_cfg, float* sumsq_part,
                       int sumsq_cap, vl_stream_t stream);
/* Grouped + batched form: out[b] = sum_{g < groups} At[b, g]^T @ Wt[b, g], run (b, g) starting at At + b*a_bs + g*a_gs (K rows each);
 * grouped-query attention backward: dK[kvh] = sum_g dS[kvh*G+g]^T Q_g, dV[kvh] = sum_g P[kvh*G+g]^T dO_g (HF sdpa/eager autograd). */
int vlaser_gemm_tn_grouped(const void* At, const void* Wt, void* out, int M, int N, int K, int ldat, int ldwt, int ldo, int groups,
                           long long a_gs, long long w_gs, int batch, long long a_bs, long long w_bs, long long o_bs, vl_stream_t stream);

/* ---- memory-bound helpers -------------------------------------------------------------------------------------- */
/* nn.LayerNorm (NORM2FN['layer_norm'], modeling_intern_vit.py:127-130,275-276) and Qwen2RMSNorm. bf16 [rows, C]. */
int vlaser_layernorm(const void* x, const void* w, const void* b, void* out, int rows, int C, float eps, vl_stream_t stream);
int vlaser_rmsnorm(const void* x, const void* w, void* out, int rows, int C, float eps, vl_stream_t stream);
/* patch embedding as im2col (+ vlaser_gemm) and token assembly: InternVisionEmbeddings.forward
 * (modeling_intern_vit.py:162-174).  pix bf16 [T,3,img,img] -> A bf16 [T*(img/14)^2, Kpad]. */
int vlaser_im2col(const void* pix, void* A, int T, int img, int Kpad, vl_stream_t stream);
int vlaser_vit_assemble(const void* patch, const void* cls, const void* pos, void* h, int T, int P, int C, vl_stream_t stream);
/* pixel_shuffle (modeling_internvl_chat.py:257-271, drop CLS :284) fused with mlp1[0] LayerNorm (:89-94).
 * x bf16 [T, G*G+1, C] -> out bf16 [T*(G/2)^2, 4C]. vlaser_pixel_shuffle is the bare permutation. */
int vlaser_pixel_shuffle_ln(const void* x, const void* w, const void* b, void* out, int T, int G, int C, float eps, int ps_v1, vl_stream_t stream);
int vlaser_pixel_shuffle(const void* x, void* out, int T, int G, int C, int ps_v1, vl_stream_t stream);
/* embed_tokens + visual-token scatter (modeling_internvl_chat.py:418-427; VLA variant with zeroed pad rows
 * pizero_internvl.py:757-791). ids int64 [n]; rank_ws int32 [n] receives the <IMG_CONTEXT> rank (or -1);
 * count_out (optional, device int32) receives the number of <IMG_CONTEXT> tokens. */
int vlaser_embed_merge(const int64_t* ids, int n, const void* embed, const void* vit, int n_vit_rows, void* out, int H,
                       long long img_id, long long pad_id, int zero_pad, int32_t* rank_ws, int32_t* count_out, vl_stream_t stream);
/* greedy argmax over fp32 logits [M,N] (GenerationMixin greedy step; lowest index wins ties, as torch.argmax) + optional embedding gather of the winner.
 * (ABI 6) ws: optional device workspace of vlaser_argmax_ws_bytes(M) bytes, ZEROED ONCE by the caller and then owned by these launches (one stream at a time): the row is
 * spread over 64 workgroups whose last arriver folds their (value, index) pairs -- 18 -> ~4 us at N = 151 674; ws = NULL (or N < 4096): one workgroup per row. */
int vlaser_argmax_ws_bytes(int M);
int vlaser_argmax(const float* logits, int M, int N, int64_t* out_id, const void* embed, void* next_h, int H, void* ws, int ws_bytes, vl_stream_t stream);
/* pi0 head glue: SinusoidalPosEmb + ActionEncoder.linear_1 (modules.py:9-22,45-50); proprio_encoder
 * (pizero_internvl.py:823); final norm + action_decoder + Euler update (+clamp) (pizero_internvl.py:911-932). */
int vlaser_vla_prep(const float* action, const void* w1, const void* b1, void* xcat, int M, int W, int adim, float t, float max_period, vl_stream_t stream);
int vlaser_small_linear(const float* x, const void* w, const void* b, void* out, int M, int N, int K, vl_stream_t stream);
/* (ABI 5) ring (nullable): the updated actions are ALSO written to slot (ring_ctr[0] mod ring_n) of `ring` (ring_stride floats per slot), so that the host
 * surface can hand out a view of the result instead of launching a copy (the reference returns a fresh tensor: pizero_internvl.py:934-936).
 * (ABI 6) ring_ctr = int32[3] = VlaserVlaStageArgs.call_ctr: {call number k, error word of even calls, error word of odd calls}; when this call's error word
 * (ring_ctr[1 + (k & 1)]) is non-zero the ring copy is NaN: a dense mask the kernels cannot honour is never served silently.
 * (ABI 7) method: the reference's `integration_method` (pizero_internvl.py:164,910-922, `integration_step` :1309-1331): 0 euler a + dt v, 1 heun a + (0.5 dt)(v + v),
 * 2 rk4 a + (dt / 6)(((v + 2 v) + 2 v) + v) -- the reference's model_step returns this step's decoder output whatever it is handed, so the higher-order methods re-combine
 * ONE velocity (golden G7c).  `dt` is then the method's coefficient (dt | 0.5 dt | dt / 6), rounded from double by the caller. */
int vlaser_vla_euler(const void* h_in, const float* partials, int n_partials, int M, const void* norm_w, float eps, const void* wd,
                     const void* bd, float* action, int W, int adim, float dt, float clip, int do_clip, float* vel_out, float* ring, const int* ring_ctr,
                     int ring_n, int ring_stride, int method, vl_stream_t stream);
/* (ABI 5) Every per-call input of PiZero.infer_action (pizero_internvl.py:798-808: input_ids, pixel_values, proprios; :879-881 the noise the reference
 * draws inside) into the static input slots of the captured chunk graph in ONE launch: ids int64 [B, T] copied; valid_out[b] = valid_in[b] (int32, or int64
 * when valid_is_i64) or, with valid_in null, the number of ids != pad_id in row b; proprio / noise fp32 copied; pixels -> bf16 (pix_dtype 0: bf16 copy,
 * 1: fp32 cast, 2: uint8 planar [n,3,H,W] normalised as vlaser_normalize_u8 mode 0 with mean / std, hw = H*W).
 * (ABI 6) The reference's other five call tensors, exactly as Vlaser_VLA/Simpler/src/agent/eval.py:110-128 builds them per control step and moves them to the
 * device, are consumed ON the device -- no host copy of a mask, no synchronisation inside the call:
 *   itp_mask [B,1,T+1,T+1] / action_mask [B,1,n_act,T+1+n_act] (additive, 0 or dtype-min; pizero_internvl.py:517-603): with valid_in null, valid_out[b] = zero
 *   count of the proprio row over the T image / text columns; every row that matters (valid prefix rows, proprio row, action rows) is compared with the
 *   prefix + trailing-block pattern that (valid_len, blk_start) descriptors express; a mismatch ORs bits into this call's error word: 1 = the valid prefix is
 *   not contiguous, 2 = another image_text_proprio_mask row, 4 = an action_mask row (the reference feeds action_mask to every Euler step, :894-896);
 *   pos_vlm / pos_pro / pos_act int64 [B,T] / [B,1] / [B,n_act] -> the int32 position slots of the graph (+ pos_ride_out = [proprio | action], batch 1);
 *   call_ctr int32[3] = {call number, error word of even calls, error word of odd calls}: call_ctr[0] = call_no (given by the HOST, which advances it only
 *   after the launch was accepted), this call's word is call_ctr[1 + (call_no & 1)], the next call's word is cleared. */
typedef struct {
  const int64_t* ids; int64_t* ids_out; int B, T; long long pad_id;
  const void* valid_in; int valid_is_i64; int32_t* valid_out;
  const float* proprio; float* proprio_out; int n_proprio;
  const float* noise; float* noise_out; int n_noise;
  const void* pix; void* pix_out; long long n_pix; int pix_dtype; int hw; float mean[3]; float std[3];
  int* call_ctr;
  /* ABI 6 */
  int call_no;
  const void* itp_mask; const void* action_mask; int mask_dtype /* 0 bf16, 1 f32, 2 f16 */; int n_act;
  const int64_t* pos_vlm; const int64_t* pos_pro; const int64_t* pos_act;
  int32_t* pos_vlm_out; int32_t* pos_pro_out; int32_t* pos_act_out; int32_t* pos_ride_out;
  long long itp_bs, itp_rs, act_bs, act_rs;   /* element strides of the masks' batch / row axes (0 = dense): split_full_mask_into_submasks (:589-603) returns slices of the full mask */
  /* (ABI 8) mask_slot != NULL: GENERAL masks -- instead of being checked against the prefix + trailing-block pattern, both masks are COPIED as fp32 into
   * mask_slot [B, T + 1 + n_act, mask_ld] (rows 0..T = image_text_proprio_mask, rows T+1.. = action_mask; columns past a row's width filled with -FLT_MAX) for the
   * VL_ATTN_DENSE attention launches of the graph.  The one thing still refused (error bit 8): an image / text row that sees the proprio key -- the cached-prefix
   * schedule computes the prefix rows' attention before the proprio token's K / V exist.  Both masks must be given; mask_ld % 32 == 0, >= T + 1 + n_act. */
  float* mask_slot; int mask_ld;
} VlaserVlaStageArgs;
int vlaser_vla_stage(const VlaserVlaStageArgs* args, vl_stream_t stream);
/* (ABI 4) Everything between two passes through the expert's layers in ONE launch: [finish != 0: the tail of the previous Euler step exactly as
 * vlaser_vla_euler computes it, without the clamp, on rows row_off .. row_off + M of h_in / partials (slabs of rows_in rows), a_out = a_in + dt * vel,
 * vel_out optional] + the action encoder of the next step (modules.py:25-56 ActionEncoder: linear_1, time embedding, cat, linear_2, swish, linear_3)
 * with linear_1 / the time embedding folded into linear_2 by the host: w21 = W2[:, W:] @ W1 (fp32 [W, adim]), cs = W2[:, :W] @ temb(t) + W2[:, W:] @ b1 + b2
 * (fp32 [W], this step's row).  h_out [M, W] bf16 = linear_3's output.  a_in / a_out must differ when finish != 0.  W multiple of 256, <= 1024. */
int vlaser_vla_step(const void* h_in, const float* partials, int n_partials, int rows_in, int row_off, const void* norm_w, float eps, const void* wd,
                    const void* bd, const float* a_in, float* a_out, float* vel_out, float dt, int finish, const float* w21, const float* cs, const void* w3,
                    const void* b3, void* h_out, int M, int W, int adim, int method /* (ABI 7) as vlaser_vla_euler's; dt = the method's coefficient */, vl_stream_t stream);
int vlaser_cast_f32_bf16(const float* x, void* y, long long n, vl_stream_t stream);
/* uint8 image -> normalised bf16 pixel_values [n_img, 3, H, W] (ABI 4).  Replaces the host-side fp32 normalisation of
 * InternVLAProcessor.__call__ (Vlaser_VLA/Simpler/src/model/vla/processing.py:51-63,303-311; mode 0: (u8 * (1/255) - mean) / std) and of
 * build_transform's ToTensor + Normalize (Vlaser_VLM/internvl_chat/internvl/train/dataset.py:293-300; mode 1: (u8 / 255 - mean) / std), same fp32
 * operation order, one bf16 rounding.  layout 0: planar [n,3,H,W]; 1: interleaved [n,H,W,3].  hw = H*W (multiple of 4); mean3 / std3: HOST float[3]. */
int vlaser_normalize_u8(const void* in_u8, void* out_bf16, int n_img, int hw, int layout, int mode, const float* mean3, const float* std3,
                        vl_stream_t stream);
/* ---- (ABI 8, r06) image preparation on the device: the bicubic resize + tile cut of `load_image` (eval_example.py:38-82: `dynamic_preprocess` =
 * dataset.py:830-866 `image.resize((target_width, target_height))` :849, the crop loop :851-862, the thumbnail :864; `build_transform` dataset.py:276-310).  The
 * reference resizes on the host through Pillow (pillow==11.2.1, Vlaser_VLA/Simpler/requirements.txt:165; `Image.resize`, default filter BICUBIC ->
 * libImaging/Resample.c): 8-bit fixed-point separable resampling, horizontal pass then vertical pass, the intermediate image rounded to 8 bits.  Bit-exact with
 * Pillow (tests/test_resize.py, tests/test_image_gpu.py).
 *
 * vlaser_resample_ksize / vlaser_resample_coeffs: HOST functions, no GPU work -- Pillow's precompute_coeffs + normalize_coeffs_8bpc for one axis (whole-image box):
 * bounds[2 * out_size] = (first input index, tap count) per output index, kk_t[ksize * out_size] = the 22-bit fixed-point weights TRANSPOSED (kk_t[k * out_size + i]:
 * tap k of output i; taps beyond the count are 0).  Returns ksize = 2 ceil(2 max(in / out, 1)) + 1, or -1.  The caller copies both tables to the device.
 *
 * vlaser_resize_u8: src [H, W, 3] uint8 (row stride ld_src bytes) -> dst [h, w, 3] (ld_dst); tmp [H, w, 3] (ld_tmp) is the intermediate image, needed only when
 * both axes change.  bounds_* / kk_* are DEVICE copies of the tables for (W -> w) and (H -> h); an axis that keeps its size takes NULL tables and is skipped, equal
 * sizes on both axes are a copy (as Pillow).  Row strides that are multiples of 4 on dword-aligned images take the dword paths. */
int vlaser_resample_ksize(int in_size, int out_size);
int vlaser_resample_coeffs(int in_size, int out_size, int* bounds, int* kk_t);
int vlaser_resize_u8(const void* src, int H, int W, long long ld_src, void* tmp, long long ld_tmp, void* dst, int h, int w, long long ld_dst, const int* bounds_x,
                     const int* kk_x, int ksize_x, const int* bounds_y, const int* kk_y, int ksize_y, vl_stream_t stream);
/* [rows * tile, cols * tile, 3] uint8 (row stride ld, multiple of 4) -> bf16 pixel_values [cols * rows, 3, tile, tile]: tile i = box ((i % cols) tile, (i / cols) tile, ...)
 * of dynamic_preprocess's crop loop (dataset.py:851-862), normalised as vlaser_normalize_u8 (mode 1 = ToTensor + Normalize, dataset.py:297-299).  tile % 4 == 0. */
int vlaser_tiles_normalize_u8(const void* src_u8, long long ld, int cols, int rows, int tile, void* out_bf16, int mode, const float* mean3, const float* std3,
                              vl_stream_t stream);
/* CrossEntropyLoss rows (modeling_internvl_chat.py:231-243): loss_row[r] = lse(logits[r]) - logits[r,label], 0 for
 * ignore_index; lse_row optional. */
int vlaser_ce_rows(const float* logits, const int64_t* labels, int R, int N, long long ld, float* loss_row, float* lse_row,
                   long long ignore_index, vl_stream_t stream);
/* Fused split-K reduction + residual (+bias, +layer-scale) + norm: the seam between a VL_EPI_PARTIAL GEMM and the
 * next GEMM.   h = h_in + [ls *] (sum_s partials[s] [+ bias]);  x_out = norm(h) (kind 0 none / 1 RMS (Qwen2) / 2 LayerNorm).
 * Replaces `residual + o_proj(...)` + post_attention_layernorm / `+ mlp(...)` + next input_layernorm (Qwen2DecoderLayer)
 * and `h + ls*attn(...)`, norm2 / `h + ls*mlp(...)`, next norm1 (modeling_intern_vit.py:291-293). Row-wise, in place OK. */
int vlaser_reduce_norm(const void* h_in /* may be null = 0 */, const float* partials, int n_partials, const void* bias, const void* ls, int norm_kind,
                       const void* norm_w, const void* norm_b, float eps, void* h_out, void* x_out, int M, int C, vl_stream_t stream);

/* ---- SFT step: backward + optimizer kernels ---------------------------------------------------
 * The trainable step reproduces HF Trainer + DeepSpeed bf16 semantics for InternVLChatModel.forward with labels
 * (modeling_internvl_chat.py:143-255; internvl_chat_finetune.py:1041-1057; zero_stage1_config.json): bf16 params
 * and grads, fp32 master weights and AdamW moments, per-layer activation recompute. GEMM-shaped work (dgrad, wgrad,
 * attention backward through materialised per-head score matrices) reuses vlaser_gemm on transposed operands. */
/* out[c*ld_out + r] = in[r*ld_in + c] for r < rows, 0 for rows <= r < pad_rows (bf16); two-level batch with element
 * strides: matrix (b, i), b < batch, i < inner (inner >= 1), starts at in + b*in_bs + i*in_is / out + b*out_bs + i*out_is */
int vlaser_transpose(const void* in, void* out, int rows, int cols, int ld_in, int ld_out, int pad_rows, int batch, long long in_bs, long long out_bs,
                     int inner, long long in_is, long long out_is, vl_stream_t stream);
/* inverse RoPE on dq/dk + pack [dq | dk | dv] (natural [S, heads*128]) into the packed q/k/v column order.
 * kv_per_q_head != 0: dk / dv are [S, n_q*128] partials, one per Q head, summed here over the heads of each kv group */
int vlaser_rope_bwd_pack(const void* dq, const void* dk, const void* dv, const float* rope_cos, const float* rope_sin, const int32_t* pos_ids,
                         void* out_packed, int S, int n_q, int n_kv, int kv_per_q_head, vl_stream_t stream);
/* Qwen2RMSNorm backward: dx_out = dres + rmsnorm_bwd(dy, x, w).  dw_out (bf16 [C], nullable) = sum_s dy x rs, the weight
 * gradient, accumulated in the same pass through dw_ws (fp32 [ceil(S/4)][C] scratch).
 * dw_ws WITHOUT dw_out (ABI 5): only the per-block partials [ceil(S/4)][C] are left in dw_ws; vlaser_colsum_partials_multi finishes several tensors in one launch.
 * dy_partials (ABI 5, nullable): dy handed over as the n_partials fp32 split-K slabs [n_partials][S][C] of the dgrad GEMM that produced it (then `dy` may be NULL): summed
 * in slab order and rounded to bf16 once, exactly as vlaser_reduce_norm would -- the reduction launch between the two is not needed. */
int vlaser_rmsnorm_bwd(const void* dy, const void* x, const void* w, const void* dres, void* dx_out, void* dw_out, float* dw_ws, int S, int C,
                       float eps, const float* dy_partials, int n_partials, vl_stream_t stream);
/* n_tensors norm-weight gradients whose partials vlaser_rmsnorm_bwd left at ws + t * slot_stride (n_part = ceil(S/4) rows of C): out_base[out_off[t] + c] (bf16) =
 * sum over the rows, the sums and their order of the per-call reduction (out_off: device int64[n_tensors], element offsets) */
int vlaser_colsum_partials_multi(const float* ws, long long slot_stride, int n_tensors, int n_part, int C, void* out_base, const long long* out_off,
                                 vl_stream_t stream);
/* out[c] = sum_s a[s,c] * f(b)[s,c]: mode 0: 1; 1: b; 2: rmsnorm-normalised b (b = norm input); 3: layernorm-normalised b */
/* bias gradient of an nn.Linear: out[c] (bf16) = sum_s a[s, c], one launch */
int vlaser_colsum_bf16(const void* a, void* out, int S, int C, int lda, vl_stream_t stream);
int vlaser_colsum_mul(const void* a, const void* b, float* out, int S, int C, int mode, float eps, float* ws /* float[2*S + 16*C] scratch */, vl_stream_t stream);
/* SwiGLU on the packed [gate16|up16] layout: act[s, I] from gu[s, 2I]; backward: dgu from (gu, dact) */
int vlaser_swiglu(const void* gu, void* act, int S, int I, vl_stream_t stream);
int vlaser_swiglu_bwd(const void* gu, const void* dact, void* dgu, int S, int I, vl_stream_t stream);
/* dlogits[r, v] = (exp(logit - lse[r]) - [v == label]) * scale for label != ignore, else 0; bf16 [R, ld_out] zero padded */
int vlaser_ce_dlogits(const float* logits, const float* lse, const int64_t* labels, void* out, int R, int V, long long ld_in, int ld_out,
                      float scale, long long ignore_index, vl_stream_t stream);
/* dEmbed[id, :] += sum of dh[s, :] over the text positions s (rank < 0) with ids[s] == id: fp32 accumulation per id, rounded once
 * (torch embedding backward).  order = int32 [n], the positions sorted by id (stable); ids outside [0, vocab) are skipped. */
int vlaser_embed_scatter_add(const int64_t* ids, const int32_t* rank, const int32_t* order, const void* dh, void* dembed, int n, int H,
                             long long vocab, vl_stream_t stream);
/* GELU(erf) backward: dx = dy * gelu'(x)  (x = pre-activation, bf16) */
int vlaser_gelu_bwd(const void* x, const void* dy, void* dx, long long n, vl_stream_t stream);
/* fused AdamW on a flat shard: g bf16 -> fp32 (times grad_scale), m/v/master fp32 updated, bf16 param written.
 * DeepSpeed FusedAdam semantics (adam_w_mode): p = p*(1 - lr*wd) - lr * mhat / (sqrt(vhat) + eps). */
int vlaser_adamw(void* param_bf16, float* master, float* m, float* v, const void* grad_bf16, long long n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, float grad_scale, int step, vl_stream_t stream);
/* same update with global-norm clipping resolved on the device: grad_scale *= max_norm / (sqrt(gnorm2[0]) + 1e-6) when the norm exceeds
 * max_norm > 0 (torch clip_grad_norm_; HF TrainingArguments.max_grad_norm) -- no host round trip between the norm and the update. */
int vlaser_adamw_clipped(void* param_bf16, float* master, float* m, float* v, const void* grad_bf16, long long n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, float grad_scale, const float* gnorm2, float max_norm, int step, vl_stream_t stream);
/* acc (fp32) = (first ? 0 : acc) + w * g (bf16); finalize: g = bf16(acc).  Gradient accumulation over micro-batches / the samples of a
 * per-device batch (…2nd_finetune_full.sh:5-6,49-50: PER_DEVICE_BATCH_SIZE, GRADIENT_ACC). */
int vlaser_grad_accumulate(void* g_bf16, float* acc, long long n, float w, int first, int finalize, vl_stream_t stream);
/* out[0] += sum of squares of a bf16 buffer (gradient norm); out must be zeroed by the caller */
int vlaser_sumsq(const void* x, long long n, float* out, float* partial_ws /* float[1024] */, vl_stream_t stream);
/* ABI 5 -- the global gradient norm of clip_grad_norm_ (internvl_chat_finetune.py:1041-1057, max_grad_norm 1.0) WITHOUT a second pass over the 3.6 GB of
 * gradients: the weight-gradient GEMMs leave per-wave partial sums (vlaser_gemm_tn[_lds], `sumsq_part`), and
 *   vlaser_sumsq_chunks: part[c] = sum of squares of x[tab[c][0] .. + tab[c][1])  (tab = int64 [n_chunks][2] on the device; the small tensors: norm
 *                        weights, biases, the projector), one workgroup per chunk;
 *   vlaser_sumsq_rows:   the embedding gradient after vlaser_embed_scatter_add: part[i] = sum of squares of row ids[order[i]] of x [vocab, H] when
 *                        position i of the id-sorted order starts a run of equal ids, else 0; part[n .. cap) = 0;
 *   vlaser_sum_partials: out[0] = (accumulate ? out[0] : 0) + part[0] + ... + part[n-1], fixed association (deterministic). */
int vlaser_sumsq_chunks(const void* x, const long long* tab, int n_chunks, float* part, vl_stream_t stream);
int vlaser_sumsq_rows(const int64_t* ids, const int32_t* order, const void* x, int n, int H, long long vocab, float* part, int cap, vl_stream_t stream);
int vlaser_sum_partials(const float* part, long long n, float* out, int accumulate, vl_stream_t stream);

/* ---- VLA flow-matching training step: PiZero.forward, pizero_internvl.py:1064-1197; train.py:470-513 -----------
 * SiLU of ActionEncoder.linear_2 (modules.py:45-52) and its backward (x = pre-activation). */
int vlaser_silu(const void* x, void* y, long long n, vl_stream_t stream);
int vlaser_silu_bwd(const void* x, const void* dy, void* dx, long long n, vl_stream_t stream);
/* Backward of the joint attention (joint_model.py:410-696) for the R <= 16 expert rows (proprio + action tokens, cache slots
 * [blk_start, blk_start + R)) over the frozen VLM prefix [0, valid_len) + their own block; first_tok_self: row 0 sees only itself in
 * the block (proprio row of the block mask, pizero_internvl.py:517-587).  q / dO / O / dq bf16 [R, n_q*128]; K [n_kv, s_max, 128]
 * (post-RoPE), VT [n_kv, 128, s_max]; dk / dv bf16 [R, n_kv*128] (gradients of the block keys; dk still needs the inverse RoPE).
 * (ABI 8) ws: device workspace of vlaser_attn_rows_bwd_ws_floats(n_q) floats owned by the call (one workgroup per (query head, row) leaves the block keys' P / dS
 * there; a second launch sums dK / dV from them in a fixed order: r03-r05 walked the (head, row) pairs of a kv head one after the other in ONE workgroup, 1.04 ms per layer). */
int vlaser_attn_rows_bwd_ws_floats(int n_q);
int vlaser_attn_rows_bwd(const void* q, const void* K, const void* VT, const void* dO, const void* O, void* dq, void* dk, void* dv, int R, int n_q,
                         int n_kv, int s_max, int valid_len, int blk_start, int first_tok_self, float scale, float* ws, vl_stream_t stream);

/* ---- f1 with `train_vlm: True` (ABI 4, r03): the reference's second parameter group `trainable_vlm_parameters` (pizero_internvl.py:405-411 =
 * vision tower + projector + the VLM mixture's decoder layers; optimiser + schedule train.py:270-295, stepped :509-520).  The gradient reaches the
 * VLM through the keys / values its rows hand to the proprio / action rows, then runs back through the Qwen2 layers (bidirectional valid-prefix
 * mask), mlp1, pixel_shuffle and the 24 InternViT blocks (modeling_intern_vit.py:177-295: LayerNorm, full attention, GELU MLP, layer scale). */
/* softmax + dS of an attention backward through materialised score matrices, in one pass: P = softmax(scale * scores) over the visible keys (bf16),
 * dS = P o (dP - D) * scale with D[q] = sum_d dO[q,d] O[q,d]; scores, dP fp32 [H, S, ld]; dO, O bf16 [S, H*hd]; P, dS bf16 [H, S, ld] (0 on masked keys).
 * Key k is visible to query row q iff k < kv_valid and (k <= q + q_off when causal != 0); q_off = global index of row 0 when the S rows are a block
 * of a longer sequence (`VLASER_SFT_ATTN_BWD=materialised` walks the queries in blocks: the score matrices stay [heads, block, keys]).  On the default
 * path this serves the bidirectional valid-prefix / full masks of the VLM group (vla_vlm_group.py); the causal SFT attention uses vlaser_attn_bwd. */
int vlaser_attn_bwd_pds_masked(const float* scores, const float* dP, const void* dO, const void* O, void* P, void* dS, int n_heads, int S, int ld, int hd,
                               float scale, int causal, int kv_valid, int q_off, vl_stream_t stream);
/* vlaser_attn_rows_bwd that also stores P and dS of EVERY key, bf16 [n_q][16][s_max] (row r of head h at (h*16 + r)*s_max): the prefix keys' dK / dV
 * then come from vlaser_gemm_tn_grouped (contraction over the R rows, summed over the q heads of a kv group) */
int vlaser_attn_rows_bwd_ex(const void* q, const void* K, const void* VT, const void* dO, const void* O, void* dq, void* dk, void* dv, int R, int n_q,
                            int n_kv, int s_max, int valid_len, int blk_start, int first_tok_self, float scale, void* p_out, void* ds_out, float* ws, vl_stream_t stream);
/* vlaser_rope_bwd_pack with a second source of key / value gradients for the same rows (bf16 [S, n_kv*128], nullable), added in fp32 before the rotation */
int vlaser_rope_bwd_pack_ex(const void* dq, const void* dk, const void* dv, const float* rope_cos, const float* rope_sin, const int32_t* pos_ids,
                            void* out_packed, int S, int n_q, int n_kv, int kv_per_q_head, const void* dk_extra, const void* dv_extra, vl_stream_t stream);
/* nn.LayerNorm backward (NORM2FN['layer_norm'], modeling_intern_vit.py:127-130): dx_out = dres + LN'(dy); the affine gradients are column sums
 * (vlaser_colsum_mul mode 3 / mode 0) */
int vlaser_layernorm_bwd(const void* dy, const void* x, const void* w, const void* dres /* nullable */, void* dx_out, int S, int C, float eps, vl_stream_t stream);
/* out[s,c] = x[s,c] * alpha * (vec ? vec[c] : 1) (bf16; row strides ldx / ldo): layer-scale factor in the backward of a ViT residual branch */
int vlaser_scale_cols(const void* x, const void* vec /* nullable */, void* out, int S, int C, int ldx, int ldo, float alpha, vl_stream_t stream);
/* backward of vlaser_pixel_shuffle: dx bf16 [T, G*G+1, C] (CLS rows zero) from dout [T*(G/2)^2, 4C] */
int vlaser_pixel_unshuffle(const void* dout, void* dx, int T, int G, int C, int ps_v1, vl_stream_t stream);

/* EMA / SWA of the trained parameters (ABI 4): `ModelAveraging.maybe_update` (Vlaser_VLA/Simpler/src/agent/model_averaging.py:8-72, called at
 * train.py:524-528) = torch.optim.swa_utils.AveragedModel.update_parameters on the rank's fp32 shard: first != 0: avg = p; else
 * avg += (p - avg) * c, c = 1 - ema_decay (EMA) or 1 / (n_averaged + 1) (SWA). */
int vlaser_avg_update(float* avg, const float* p, long long n, float c, int first, vl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
