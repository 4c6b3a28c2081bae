// Attention + o_proj of a <= 16-row decoder layer-step in ONE launch (r03; batch 1).
//
// Today the pair is two launches at the ~5 us floor of a dependent load -> store kernel each (attn_skinny 5.65 us: 14 workgroups = 2 kv heads x 7 key
// splits; o_proj 4.88 us: 144 workgroups that each merge the split partials and stream a slice of the 2.4 MB weight).  A hand-off inside one launch costs
// more than the kernel boundary (profiles/r03c_euler_fusion.md), so this kernel needs NO hand-off: every workgroup recomputes the WHOLE attention of its kv
// group -- 24-30 (head, token) rows x 389 keys, 199 KB of K / V^T that hit L2 after the first workgroup of an XCD touched them -- with its 4 waves each
// holding up to four 32-key chunks in registers (one wave per SIMD: 512 registers; every K / V^T request of the launch is issued up front) (the key split of attn_skinny moved INSIDE the workgroup, merged through LDS), and then multiplies the merged rows with
// its own 16 columns of W_o restricted to the group's heads: out[kv head][token][n] = sum_{head in group, d} attn[head, token, d] W_o[n, head*128 + d].
// The consumer (gate/up's NORM prologue) sums n_kv_heads slabs instead of the 3 split-K slabs of the stand-alone o_proj.
// Measured beforehand with a traffic-only stand-in (tools/micro/attn_o_probe.hip): 96 workgroups pulling 199 KB of shared K / V with COALESCED 1 KiB wave
// requests + 24 KB of their own weights run 4.8 us in the chain (2.5 us without the K / V pull).
//
// RESULT (profiles/r03u_attn_oproj.md): correct (tests/test_ops_gpu.py::test_attn_oproj_one_launch, chunk within 3.8e-3 of the two-launch path) and SLOWER --
// 15.1-17.3 us against 10.5 us for the pair, +1.2...1.6 ms per chunk -- so it is OFF by default (`VLASER_EULER=...,fuse_ao`).  The MFMA fragment loads of
// attn_skinny fetch 64-byte pieces (16 key rows x 64 bytes per wave instruction): fine for 14 workgroups, but 96 workgroups x 199 KB in 64-byte pieces is 300 K
// requests per launch and the request path, not the bytes, sets the time: with every request issued up front (this version: 4 waves x 4 chunks in registers)
// the wave needs 7.6 us just to get its 80 load instructions accepted; two passes of plain loads (8 waves) 12.7 us in-kernel with 4 us for the second round
// trip; the second chunk through LDS-DMA costs ~230 issue cycles per scattered 1 KiB piece.  A version that stages K / V^T row-major through LDS with
// coalesced requests (as the probe does) is the remaining candidate; its best case is ~2 us per layer-step.
#include "common.h"
#include "../../include/vlaser_hip.h"

#define AO_WAVES 4
#define AO_CH 4          // chunks of 32 keys a wave holds in registers at once (4 waves x 4 chunks = 512 keys in ONE round trip)
#define AO_NEG_BIG (-1.0e30f)

struct AttnOP {
  VlaserAttnArgs a;
  const bf16_t* wo;      // [N][ldw] bf16 row-major (nn.Linear weight as stored), ldw >= n_q_heads * 128
  float* out;            // [n_kv_heads][sq][N] fp32 partial slabs
  int N, ldw;
};

__device__ __forceinline__ float ao_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// lab build only (-DAO_TIMELINE, tools/micro/attn_o_timeline.py): cycle stamps of wave 0 of workgroup (0, 0)
#ifdef AO_TIMELINE
__device__ long long ao_dbg[32];
#define AO_STAMP(i) { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) ao_dbg[i] = clock64(); __builtin_amdgcn_sched_barrier(0); }
extern "C" int vlaser_attn_oproj_debug_read(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ao_dbg), sizeof(long long) * 32); }
#else
#define AO_STAMP(i)
#endif

__global__ __launch_bounds__(64 * AO_WAVES) void attn_oproj_kernel(AttnOP p) {
  constexpr int HD = 128, DC = 4, DT = 8, RS = 132, WS = 64 + 32 * RS, XP = 136;   // RS: floats per partial row (128 + 4: the 16 rows of a store hit different banks; 128 was a 16-way
                                                                                   // conflict = 3 us of this launch); XP: bf16 elements per merged row
  extern __shared__ __attribute__((aligned(16))) char smem[];                // [8 waves][ m[32] l[32] o[32][128] ] fp32 | merged rows bf16 [32][XP]
  const VlaserAttnArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16, kvh = blockIdx.y;
  const int G = a.n_q_heads / a.n_kv_heads, nq = a.sq, nrows = G * nq;
  const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + (size_t)kvh * a.k_hs;
  const bf16_t* VT = reinterpret_cast<const bf16_t*>(a.vt) + (size_t)kvh * a.vt_hs;
  const int n_chunks = (a.kv_len + 31) >> 5;
  AO_STAMP(0)
  bf16x8 qf[2][DC];
  int row_hi2[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int r = min(qt * 16 + fr, nrows - 1);            // rows past nrows shadow the last real row (unconditional loads, results never used)
    const int hg = (int)(((float)r + 0.5f) * __builtin_amdgcn_rcpf((float)nq)), tok = r - hg * nq;
    row_hi2[qt] = (tok == 0 && a.first_tok_kv_len > 0) ? a.first_tok_kv_len : 0x7fffffff;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + (size_t)(kvh * G + hg) * a.q_hs + (size_t)tok * a.q_ss;
#pragma unroll
    for (int dc = 0; dc < DC; ++dc) qf[qt][dc] = as_bf16x8(ld_global_16(Q + dc * 32 + g * 8));
  }
  int lim1 = a.kv_len, lo2 = 0x7fffffff, hi2 = 0;
  if (a.mode == VL_ATTN_PREFIX) {
    lim1 = min(a.valid_len ? a.valid_len[0] : a.kv_len, a.kv_len);
    lo2 = a.blk_start; hi2 = a.kv_len;
  }
  f32x4 o[2][DT];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int i = 0; i < DT; ++i) o[qt][i] = f32x4{0, 0, 0, 0};
  float m_run[2] = {AO_NEG_BIG, AO_NEG_BIG}, l_run[2] = {0.f, 0.f};
  const float sc = a.scale * 1.4426950408889634f;

  auto load_chunk = [&](int key0, u32x4 (&kf)[2][DC], u32x4 (&vf)[DT]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int key = key0 + (fr >> 2) * 8 + t * 4 + (fr & 3);
#pragma unroll
      for (int dc = 0; dc < DC; ++dc) kf[t][dc] = ld_global_16(K + (size_t)min(key, a.kv_len - 1) * HD + dc * 32 + g * 8);
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vf[dt] = ld_global_16(VT + (size_t)(dt * 16 + fr) * a.ld_vt + key0 + g * 8);
  };
  auto process_chunk = [&](int key0, u32x4 (&kf)[2][DC], u32x4 (&vf)[DT]) __attribute__((always_inline)) {
    bool visk[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + g * 8 + t * 4 + r;
        visk[t][r] = (key < lim1) || (key >= lo2 && key < hi2);
      }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      if (qt * 16 >= nrows) continue;
      bool vis[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = key0 + g * 8 + t * 4 + r;
          vis[t][r] = visk[t][r] && (key < lim1 || key < row_hi2[qt]);
        }
      f32x4 s[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int dc = 0; dc < DC; ++dc) acc = mfma16(as_bf16x8(kf[t][dc]), qf[qt][dc], acc);
        s[t] = acc;
      }
      float mx = AO_NEG_BIG;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[t][r] *= sc;
          if (vis[t][r]) mx = fmaxf(mx, s[t][r]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[qt], mx);
      const float alpha = ao_exp2(m_run[qt] - m_new);
      m_run[qt] = m_new;
      float pv[8], psum = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pe = vis[t][r] ? ao_exp2(s[t][r] - m_new) : 0.f;
          psum += pe;
          pv[t * 4 + r] = pe;
        }
      l_run[qt] = l_run[qt] * alpha + psum;
      const u32x4 pk = {pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3]), pack_bf16x2(pv[4], pv[5]), pack_bf16x2(pv[6], pv[7])};
      const bf16x8 pf = as_bf16x8(pk);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        f32x4 acc = o[qt][dt];
        acc[0] *= alpha; acc[1] *= alpha; acc[2] *= alpha; acc[3] *= alpha;
        o[qt][dt] = mfma16(as_bf16x8(vf[dt]), pf, acc);
      }
    }
  };
  // Every request of the launch goes out up front, in straight-line code: the wave's (up to) four chunks -- a chunk requested after another chunk's
  // arithmetic is a second full round trip, 4 us in the first version of this kernel -- and, BEHIND them (vmcnt retires in order: the L2-resident K / V^T must
  // not queue behind the one HBM-cold stream), its share of W_o: the contraction (G heads x 4 blocks of 32) dealt to the waves, <= 8 blocks each.
  const int KC = G * DC, cpw = (KC + AO_WAVES - 1) / AO_WAVES;
  u32x4 wof[8];
  {
    u32x4 kf[AO_CH][2][DC], vf[AO_CH][DT];
#pragma unroll
    for (int c = 0; c < AO_CH; ++c) load_chunk(min(wave + AO_WAVES * c, n_chunks - 1) << 5, kf[c], vf[c]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kc = min(wave * cpw + min(j, cpw - 1), KC - 1), hg = kc >> 2, dc = kc & 3;
      wof[j] = ld_global_16(p.wo + (size_t)(n0 + fr) * p.ldw + (size_t)(kvh * G + hg) * HD + dc * 32 + g * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
    AO_STAMP(1)
#pragma unroll
    for (int c = 0; c < AO_CH; ++c) {
      if (wave + AO_WAVES * c < n_chunks) process_chunk((wave + AO_WAVES * c) << 5, kf[c], vf[c]);
      if (c == 0) { AO_STAMP(2) }
    }
    for (int ci = wave + AO_WAVES * AO_CH; ci < n_chunks; ci += AO_WAVES) {     // more than 512 keys: one more round trip per 128 keys
      load_chunk(ci << 5, kf[0], vf[0]);
      process_chunk(ci << 5, kf[0], vf[0]);
    }
  }
  AO_STAMP(3)
  // ---- merge of the waves' partial softmax states through LDS -> normalised bf16 rows (the A tile of o_proj, as VL_PRO_ATTN builds it)
  float* wm = reinterpret_cast<float*>(smem) + wave * WS;
  const int nwa = min(AO_WAVES, n_chunks);
  if (wave < nwa) {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float l_tot = l_run[qt] + __shfl_xor(l_run[qt], 16, 64);
      l_tot += __shfl_xor(l_tot, 32, 64);
      if (g == 0) { wm[qt * 16 + fr] = m_run[qt]; wm[32 + qt * 16 + fr] = l_tot; }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4*>(wm + 64 + (qt * 16 + fr) * RS + dt * 16 + g * 4) = o[qt][dt];
    }
  }
  __syncthreads();
  AO_STAMP(4)
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem + (size_t)AO_WAVES * WS * 4);
  {
    const float* base = reinterpret_cast<const float*>(smem);
    const int row = tid >> 3, d0 = (tid & 7) * 16;          // 256 threads: 32 rows x 8 pieces of 16 d
    if (row < nrows) {
      float M = AO_NEG_BIG;
#pragma unroll
      for (int w = 0; w < AO_WAVES; ++w)
        if (w < nwa) M = fmaxf(M, base[w * WS + row]);
      float Ls = 0.f, v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = 0.f;
#pragma unroll
      for (int w = 0; w < AO_WAVES; ++w) {
        if (w >= nwa) break;
        const float* bw = base + w * WS;
        const float f = ao_exp2(bw[row] - M);
        Ls += bw[32 + row] * f;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const f32x4 ov = *reinterpret_cast<const f32x4*>(bw + 64 + row * RS + d0 + q4 * 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q4 * 4 + q] += ov[q] * f;
        }
      }
      const float inv = Ls > 0.f ? 1.0f / Ls : 0.f;
#pragma unroll
      for (int h8 = 0; h8 < 2; ++h8)
        *reinterpret_cast<u32x4*>(xs + row * XP + d0 + h8 * 8) = u32x4{pack_bf16x2(v[h8 * 8] * inv, v[h8 * 8 + 1] * inv), pack_bf16x2(v[h8 * 8 + 2] * inv, v[h8 * 8 + 3] * inv),
                                                                    pack_bf16x2(v[h8 * 8 + 4] * inv, v[h8 * 8 + 5] * inv), pack_bf16x2(v[h8 * 8 + 6] * inv, v[h8 * 8 + 7] * inv)};
    }
  }
  __syncthreads();
  AO_STAMP(5)
  // ---- o_proj on this workgroup's 16 output columns: wave w contracts its <= 4 blocks of 32, the 8 partial tiles are summed in a fixed order
  f32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int kc = wave * cpw + j;
    if (j < cpw && kc < KC) {                                // wave-uniform
      const int hg = kc >> 2, dc = kc & 3;
      const bf16x8 bf = as_bf16x8(*reinterpret_cast<const u32x4*>(xs + (hg * nq + min(fr, nq - 1)) * XP + dc * 32 + g * 8));
      acc = mfma16(as_bf16x8(wof[j]), bf, acc);              // rows = output columns n0 + g*4 + r, columns = tokens
    }
  }
  float* red = reinterpret_cast<float*>(smem);               // the waves' partial states are dead: [waves][16 n][16 tok]
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave * 256 + (g * 4 + r) * 16 + fr] = acc[r];
  __syncthreads();
  AO_STAMP(6)
  if (tid < 256) {
    const int n = tid >> 4, tok = tid & 15;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < AO_WAVES; ++w) s += red[w * 256 + tid];
    if (tok < nq) p.out[((size_t)kvh * nq + tok) * p.N + n0 + n] = s;
  }
  AO_STAMP(7)
}

extern "C" int vlaser_attn_oproj(const VlaserAttnArgs* a, const void* wo, int ldw, float* out_f32, int N, vl_stream_t s) {
  VL_CHECK(a && a->q && a->k && a->vt && wo && out_f32, "vlaser_attn_oproj: null pointer");
  VL_CHECK(a->batch == 1 && a->head_dim == 128 && a->n_q_heads % a->n_kv_heads == 0, "vlaser_attn_oproj: batch 1, head_dim 128, whole GQA groups");
  const int G = a->n_q_heads / a->n_kv_heads;
  VL_CHECK(a->sq >= 1 && a->sq <= 16 && G * a->sq <= 32 && G <= 8, "vlaser_attn_oproj: group * tokens must be <= 32, group <= 8 (tokens=%d)", a->sq);
  VL_CHECK(a->mode == VL_ATTN_FULL || a->mode == VL_ATTN_PREFIX, "vlaser_attn_oproj: mode must be FULL or PREFIX");
  VL_CHECK(a->kv_len >= 1 && a->kv_len <= a->ld_vt && a->ld_vt % 32 == 0, "vlaser_attn_oproj: kv_len / padded V^T row");
  VL_CHECK(N % 16 == 0 && ldw >= a->n_q_heads * 128 && ldw % 8 == 0 && ((uintptr_t)wo & 15) == 0, "vlaser_attn_oproj: N must be a multiple of 16, W_o rows 16-byte aligned");
  AttnOP p;
  p.a = *a; p.wo = (const bf16_t*)wo; p.out = out_f32; p.N = N; p.ldw = ldw;
  const int lds = AO_WAVES * (64 + 32 * 132) * 4 + 32 * 136 * 2;
  if (int rc = set_max_lds_once(attn_oproj_kernel, lds)) return rc;
  hipLaunchKernelGGL(attn_oproj_kernel, dim3(N / 16, a->n_kv_heads), dim3(64 * AO_WAVES), lds, (hipStream_t)s, p);
  VL_LAUNCH_CHECK();
  return 0;
}
