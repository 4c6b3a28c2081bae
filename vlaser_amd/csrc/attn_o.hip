// Attention + o_proj of a <= 16-row decoder layer-step in ONE launch (r04; the r03 attempt is in git history and in profiles/r03u_attn_oproj.md).
//
// Why.  The pair attn_skinny (5.7 us, 14 workgroups) + o_proj (4.8 us, 144 workgroups that each re-merge the 7 split partials) sits twice on the ~4.9 us
// floor of a dependent load -> store launch, 280 times per action chunk.  An in-launch hand-off costs more than the kernel boundary on this chip
// (profiles/r03c_euler_fusion.md, MI355X_MICROARCH.md "fanin" / "allgather"), so this kernel has none: every workgroup recomputes the WHOLE attention of
// its kv group (24-30 (head, token) rows x <= 304 visible keys) and then contracts its own 16 output columns of W_o over the group's heads; the consumer
// (gate/up's NORM prologue) sums n_kv_heads slabs.  r03 built exactly that on attn_skinny's MFMA-fragment loads (16 key rows x 64 bytes per wave
// instruction: 300 K requests per launch) and measured 15-17 us.  What is different here:
//   * K and V^T are pulled with COALESCED 1 KiB LDS-DMA pieces (whole 256-byte key rows; 640-byte runs of a V^T row) -- the traffic-only probe of
//     profiles/r03u_attn_oproj.md priced that at 4.8 us for the whole launch -- and re-shaped into MFMA fragments by conflict-free `ds_read_b128`
//     (XOR swizzles applied on the SOURCE side of the DMA, guide rule 21);
//   * only the keys a row can see are staged: the valid prefix [0, valid_len) and the block [blk_start, kv_len), as a compact list of 16-key tiles
//     (19 tiles = 152 KB of LDS per pass; a prompt with more visible keys takes more passes with an online-softmax rescale between them);
//   * the softmax is two-step (all waves exchange their row maxima through LDS, then exponentiate against the GLOBAL maximum), and P goes through LDS so
//     that the P V product is split over the 8 waves by head-dim tile instead of by key: no flash-decoding merge of 8 partial [32 x 128] fp32 states
//     (128 KB of LDS writes + reads and two barriers in r03);
//   * W_o comes fragment-major (ops.pack_skinny(w_o, n_kv_heads, 1)): one contiguous 1 KiB request per wave and K-step.
// Row order, masks (VL_ATTN_FULL / VL_ATTN_PREFIX with valid_len, blk_start, first_tok_kv_len) and rounding points are attn_skinny's + VL_PRO_ATTN's:
// scores in fp32, P rounded to bf16 for the MFMA, attention rows rounded to bf16 before o_proj, fp32 partial slabs out.
#include "common.h"
#include "../../include/vlaser_hip_experimental.h"

#define AO_WAVES 8
#define AO_THREADS (AO_WAVES * 64)
#define AO_TILES 19                       // 16-key tiles staged per pass
#define AO_VT_PITCH 640                   // bytes per V^T row in LDS = 40 slots of 16 B (20 tiles)
#define AO_VT_BYTES (128 * AO_VT_PITCH)   // 81 920
#define AO_K_BYTES (AO_TILES * 16 * 256)  // 77 824
#define AO_STAT_OFF (AO_VT_BYTES + AO_K_BYTES)
#define AO_LDS_BYTES (AO_STAT_OFF + 2 * AO_WAVES * 32 * 4)
// P of list tile t lives in the first KiB of the tile's own K rows (written by the wave that owns the tile, after its own K reads); x and the o_proj partial
// tiles live in the waves' private V^T rows once their P V phase is over
#define AO_NEG_BIG (-1.0e30f)

struct AttnOP {
  VlaserAttnArgs a;
  const u32x4* wo;       // fragment-major: [kv head][unit = N/16][8 waves][NS steps][64 lanes] x 16 bytes
  float* out;            // [n_kv_heads][sq][N] fp32 partial slabs
  int N;
};

__device__ __forceinline__ float ao_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ void ao_glds16(const void* gsrc, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory");
}

// lab build only (-DAO_TIMELINE, tools/micro/attn_o_timeline.py): cycle stamps of wave 0 of workgroup (0, 0)
#ifdef AO_TIMELINE
__device__ long long ao_dbg[256 * 16];
#define AO_STAMP(i) { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) { ao_dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = wall_clock64(); if ((i) == 0 || (i) == 7) ao_dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 14 + ((i) == 7)] = clock64(); } __builtin_amdgcn_sched_barrier(0); }
extern "C" int vlaser_attn_oproj_debug_read(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ao_dbg), sizeof(long long) * 256 * 16); }
#else
#define AO_STAMP(i)
#endif

typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ f32x4 ao_mfma_k16(u32x2 a, u32x2 b, f32x4 c) {      // v_mfma_f32_16x16x16_bf16: lane (i = l & 15, k = 4 (l >> 4) .. + 3) for both operands
  union { u32x2 u; s16x4 h; } x, y;
  x.u = a; y.u = b;
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x.h, y.h, c, 0, 0, 0);
}

template <int NS>      // K-steps of 32 per wave in the o_proj contraction: group heads * 128 = 8 waves * NS * 32
__global__ __launch_bounds__(AO_THREADS) void attn_oproj_kernel(AttnOP p) {
  constexpr int HD = 128, DC = 4;
  constexpr int VPP = AO_VT_BYTES / 1024 / AO_WAVES;                    // V^T pieces per wave (10) = its 16 rows of the image
  constexpr int TPW = (AO_TILES + AO_WAVES - 1) / AO_WAVES;            // 16-key tiles a wave owns in the Q K^T phase (3)
  constexpr int VAREA = VPP * 1024;                                     // bytes of a wave's private V^T rows; after its P V phase: x at +0, o_proj partial tile at +2048
  static_assert(AO_VT_BYTES % (1024 * AO_WAVES) == 0, "V^T image = whole pieces per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef VL_KERNARG_UP_FRONT
  vl_kernargs_up_front(p);
#endif
  const VlaserAttnArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr_k = lane & 15, g_k = lane >> 4;
  const int kvh = blockIdx.y;
  const int G = a.n_q_heads / a.n_kv_heads, nq = a.sq, nrows = G * nq;
  const bf16_t* K = reinterpret_cast<const bf16_t*>(a.k) + (size_t)kvh * a.k_hs;
  const bf16_t* VT = reinterpret_cast<const bf16_t*>(a.vt) + (size_t)kvh * a.vt_hs;
  // buffer descriptors (wave-uniform bases from kernel arguments and blockIdx): every K / V^T request carries ONE 32-bit offset register instead of a 64-bit address pair
  const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(K), 0, a.ld_vt * HD * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(VT), 0, HD * a.ld_vt * 2, 0x00020000);
  AO_STAMP(0)
  // ---- the list of 16-key tiles.  The FIRST pass is a fixed list -- prefix tiles 0 .. 17 (keys 0 .. 287) and, as tile 18, the first tile of the block
  // [blk_start, kv_len) -- so that none of its requests depends on valid_len (a dependent load in front of them cost 1-1.5 us); keys past valid_len are
  // staged and masked.  Only a prompt with more visible keys has further tiles (the rest of the prefix, then the rest of the block), in passes of <= 19
  // with an online-softmax rescale between them.
  const bool pfx = a.mode == VL_ATTN_PREFIX;
  const int lo2 = pfx ? a.blk_start : 0x3fffffff, hi2 = pfx ? a.kv_len : 0;
  const int nb16 = pfx ? (max(hi2 - lo2, 0) + 15) >> 4 : 0;
  const int t18_key0 = nb16 > 0 ? lo2 : 288, pref_next = nb16 > 0 ? 18 : 19;
  const int kmax = a.ld_vt - 16;                              // clamp of tile starts: every staged address stays inside the padded cache rows
  int lim1 = pfx ? min(a.kv_len, lo2) : a.kv_len;             // PREFIX: min'ed with valid_len once that (vector) load is back -- see pass 0
  const float sc = a.scale * 1.4426950408889634f;
  float m_run[2] = {AO_NEG_BIG, AO_NEG_BIG}, l_run[2] = {0.f, 0.f};
  f32x4 o[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};      // this wave's head-dim tile (d = 16 wave + 4 g + j) of the two row tiles, unnormalised
  float* smax = reinterpret_cast<float*>(smem + AO_STAT_OFF);
  float* ssum = smax + AO_WAVES * 32;
  const u32x4* wsrc = p.wo + ((((size_t)kvh * (p.N >> 4) + blockIdx.x) * AO_WAVES + wave) * NS) * 64 + lane;
  uint32_t wwarm = 0;
  bf16x8 qf[2][DC];
  int row_hi2[2] = {0x7fffffff, 0x7fffffff};
  u32x4 wof[NS];
  int nt = AO_TILES, xp = 0;                                  // total tiles / prefix tiles beyond the first pass: known once valid_len is (not needed by pass 0)

  for (int t0 = 0; t0 < nt;) {
    const int cnt = min(nt - t0, AO_TILES);                   // tiles of this pass
    // the pass's tiles are at most two linear runs of source keys (pass 0: prefix tiles 0 .. 17 | first block tile; later: rest of the prefix | rest of the block):
    // tile tl starts at key (tl < n1 ? b1 : b2) + 16 tl, three wave-uniform scalars
    int n1, b1, b2;
    if (t0 == 0) { n1 = pref_next; b1 = 0; b2 = t18_key0; }
    else {
      const int u0 = t0 - AO_TILES;
      n1 = max(0, min(cnt, xp - u0)); b1 = (pref_next + u0) << 4; b2 = lo2 + ((1 + max(0, u0 - xp)) << 4);
    }
    n1 = __builtin_amdgcn_readfirstlane(n1); b1 = __builtin_amdgcn_readfirstlane(b1); b2 = __builtin_amdgcn_readfirstlane(b2 - (n1 << 4));
    auto tile_key0 = [&](int tl) { return (tl < n1 ? b1 : b2) + (tl << 4); };
    // ---------------------------------------------------------------- requests: coalesced 1 KiB loads to registers, ALL up front (unconditional, clamped), in the
    // order they are needed.  K: the wave stages exactly the tiles it owns in the Q K^T phase (tiles wave, wave + 8, wave + 16 of the pass: 4 pieces of 4 key rows
    // each) -- no barrier between its ds_write and its own fragment reads; lane -> key row 4 pc + (lane >> 4) of the tile, PHYSICAL 16-byte slot lane & 15
    // holding logical slot (lane & 15) ^ row (conflict-free ds_read_b128 of the MFMA fragments).
    u32x4 kreg[TPW][4], vreg[VPP];
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
      const int tl = min(wave + ti * AO_WAVES, cnt - 1);
      const int k0 = min(tile_key0(tl), kmax);
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) {
        const int row = pc * 4 + (lane >> 4);
        kreg[ti][pc] = __builtin_amdgcn_raw_buffer_load_b128(krs, (k0 + row) * (HD * 2) + (((lane & 15) ^ row) << 4), 0, 0);
      }
    }
    AO_STAMP(12)
    int vl_raw = 0x7fffffff;
    if (t0 == 0) {                                            // (uniform) first pass only: Q fragments, valid_len, the W_o pull
      const int fr = fr_k, g = g_k;
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const int r = min(qt * 16 + fr, nrows - 1);          // rows past nrows shadow the last real row (unconditional loads, results never used)
        const int hg = (int)(((float)r + 0.5f) * __builtin_amdgcn_rcpf((float)nq)), tok = r - hg * nq;
        row_hi2[qt] = (tok == 0 && a.first_tok_kv_len > 0) ? a.first_tok_kv_len : 0x7fffffff;
        const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + (size_t)(kvh * G + hg) * a.q_hs + (size_t)tok * a.q_ss;
#pragma unroll
        for (int dc = 0; dc < DC; ++dc) qf[qt][dc] = as_bf16x8(ld_global_16(Q + dc * 32 + g * 8));
      }
      // valid_len through the VECTOR path, behind the K requests: as a scalar load it was waited for (1.5 us: the scalar cache is cold and lgkmcnt cannot be
      // counted past it) before the first request went out; nothing needs it before the masks of the Q K^T phase
      if (pfx && a.valid_len) {
        int zero = 0;
        asm volatile("" : "+v"(zero));
        vl_raw = a.valid_len[zero];
      }
      // W_o (the only HBM-cold bytes of the launch, needed last): one dword per lane pulls the wave's NS KiB into L2 now, the fragments themselves are requested
      // at the start of the P V phase (held from here on, their NS x 4 registers were spilled -- and a scratch reload is a vmcnt(0))
      wwarm = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(wsrc - lane) + lane * (NS * 4));
    }
    // V^T: the wave stages the 16 rows d = 16 wave .. + 15 of the [128 d][40 slots of 16 B] image -- exactly the rows its own P V phase reads (no barrier for V^T
    // either): 10 pieces of 64 consecutive slots, slot S = 64 i + lane -> row S / 40, physical slot S % 40 by a 3-instruction recurrence (64 = 40 + 24),
    // logical slot = physical ^ ((d >> 1) & 7) = 8 keys of tile logical >> 1; slots of tiles this pass does not have re-fetch its last tile (never weighted)
    {
      int v_dl = lane >= 40 ? 1 : 0, v_ps = lane >= 40 ? lane - 40 : lane;
#pragma unroll
      for (int i = 0; i < VPP; ++i) {
        const int d = wave * 16 + v_dl;
        const int ls = v_ps ^ ((d >> 1) & 7);
        const int src = min(tile_key0(min(ls >> 1, cnt - 1)), kmax) + ((ls & 1) << 3);
        vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(vrs, (d * a.ld_vt + src) * 2, 0, 0);
        v_ps += 24; v_dl += 1;
        if (v_ps >= 40) { v_ps -= 40; v_dl += 1; }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    AO_STAMP(8)
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
      const int tl = wave + ti * AO_WAVES;
      if (tl < cnt) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) *reinterpret_cast<u32x4*>(smem + AO_VT_BYTES + tl * 4096 + pc * 1024 + lane * 16) = kreg[ti][pc];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    AO_STAMP(9)
    // ---------------------------------------------------------------- S = Q K^T for the tiles this wave owns, softmax against the wave-LOCAL row maximum
    // (flash-decoding: the P V phase rescales by exp2(m_wave - M)), P (bf16) into the first KiB of the tile's own K rows
    if (t0 == 0) lim1 = min(lim1, vl_raw);
    f32x4 s[TPW][2];
    float mloc[2] = {AO_NEG_BIG, AO_NEG_BIG};
    int fr = fr_k, g = g_k;                                   // opaque per-phase copies: this phase's LDS offsets are computed HERE (hoisted out of the
    asm volatile("" : "+v"(fr), "+v"(g));                    // pass loop, ~40 swizzled offsets stayed live across all phases and were spilled)
    {
      bf16x8 kf[TPW][DC];                                     // every fragment requested before the first MFMA: one LDS round trip for the phase
#pragma unroll
      for (int ti = 0; ti < TPW; ++ti) {
        const char* krow = smem + AO_VT_BYTES + (min(wave + ti * AO_WAVES, AO_TILES - 1) * 16 + fr) * 256;
#pragma unroll
        for (int dc = 0; dc < DC; ++dc) kf[ti][dc] = as_bf16x8(*reinterpret_cast<const u32x4*>(krow + (((dc * 4 + g) ^ fr) << 4)));
      }
#pragma unroll
      for (int ti = 0; ti < TPW; ++ti) {
        const int tl = wave + ti * AO_WAVES;
        const int k0 = tile_key0(tl) + g * 4;
        const bool blk = tl >= n1 && pfx;                     // (uniform) a tile of the block [blk_start, kv_len): its limit depends on the row (first_tok_kv_len)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          // visibility as the accumulator's initial value: 0 for a visible key, -1e30 for a masked one (x scale stays hugely negative, exp2 gives 0) -- one compare
          // + select per key instead of ~10 mask instructions per score
          const int limit = tl < cnt ? (blk ? min(hi2, row_hi2[qt]) : lim1) : (int)0x80000000;      // (a tile the pass does not have: INT_MIN masks all its keys)
          f32x4 acc;
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = k0 + j < limit ? 0.f : AO_NEG_BIG;
#pragma unroll
          for (int dc = 0; dc < DC; ++dc) acc = mfma16(kf[ti][dc], qf[qt][dc], acc);
          // a tile index past the image (wave + 16 >= 19) read ANOTHER wave's tile 18, possibly before its owner wrote it: stale LDS bits may be NaN, and NaN
          // survives the additive mask (found as rare non-finite outputs after other kernels had used the LDS) -- select, do not add
          if (ti * AO_WAVES + AO_WAVES > AO_TILES) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = tl < AO_TILES ? acc[j] : AO_NEG_BIG;
          }
          s[ti][qt] = acc;
        }
      }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
      for (int ti = 0; ti < TPW; ++ti)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[ti][qt][j] *= sc;
          mloc[qt] = fmaxf(mloc[qt], s[ti][qt][j]);
        }
      mloc[qt] = fmaxf(mloc[qt], __shfl_xor(mloc[qt], 16, 64));
      mloc[qt] = fmaxf(mloc[qt], __shfl_xor(mloc[qt], 32, 64));
      const float meff = fmaxf(mloc[qt], -1.0e20f);          // a row this wave sees nothing of: exp2(-1e29 + 1e20) = 0 without a select
      float psum = 0.f;
      const int row = qt * 16 + fr;
#pragma unroll
      for (int ti = 0; ti < TPW; ++ti) {
        const int tl = wave + ti * AO_WAVES;
        float pe[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pe[j] = ao_exp2(s[ti][qt][j] - meff);
          psum += pe[j];
        }
        if (tl < AO_TILES)                                    // (compile time) a tile the pass does not have is all-masked: its P is written as zeros
          *reinterpret_cast<u32x2*>(smem + AO_VT_BYTES + tl * 4096 + row * 32 + ((g ^ (((row >> 3) & 1) << 1)) << 3)) = u32x2{pack_bf16x2(pe[0], pe[1]), pack_bf16x2(pe[2], pe[3])};
      }
      psum += __shfl_xor(psum, 16, 64);
      psum += __shfl_xor(psum, 32, 64);
      if (g == 0) { smax[wave * 32 + row] = mloc[qt]; ssum[wave * 32 + row] = psum; }
    }
    AO_STAMP(2)
#pragma unroll
    for (int i = 0; i < VPP; ++i) *reinterpret_cast<u32x4*>(smem + wave * VAREA + i * 1024 + lane * 16) = vreg[i];
    __syncthreads();                     // P tiles and row statistics of every wave are visible (V^T rows are wave-private: written and read by the same wave)
    AO_STAMP(3)
    if (t0 == 0) {                       // W_o now (L2-warm by the dword pull): it lands under the P V phase
#pragma unroll
      for (int st = 0; st < NS; ++st) wof[st] = __builtin_nontemporal_load(wsrc + st * 64);
    }
    // ---------------------------------------------------------------- O^T[d tile of this wave][row] += sum over owner waves w of exp2(m_w - M) * V^T_w P_w^T
    fr = fr_k; g = g_k;
    asm volatile("" : "+v"(fr), "+v"(g));
    {
      // every operand of the phase is requested before the first MFMA (a read behind each MFMA pair was a serial LDS round trip per tile)
      const int drow = wave * 16 + fr;
      const char* vrow = smem + drow * AO_VT_PITCH + ((g & 1) << 3);
      const int vkey = (drow >> 1) & 7;
      const int prow0 = fr * 32 + ((g ^ (((fr >> 3) & 1) << 1)) << 3), prow1 = (16 + fr) * 32 + ((g ^ ((((16 + fr) >> 3) & 1) << 1)) << 3);
      u32x2 vf[AO_TILES], pf0[AO_TILES], pf1[AO_TILES];
#pragma unroll
      for (int tl = 0; tl < AO_TILES; ++tl) {
        vf[tl] = *reinterpret_cast<const u32x2*>(vrow + (((tl * 2 + (g >> 1)) ^ vkey) << 4));
        pf0[tl] = *reinterpret_cast<const u32x2*>(smem + AO_VT_BYTES + tl * 4096 + prow0);
        pf1[tl] = *reinterpret_cast<const u32x2*>(smem + AO_VT_BYTES + tl * 4096 + prow1);
      }
      float fw[2][AO_WAVES];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const int row = qt * 16 + fr;
        float mw[AO_WAVES], sw[AO_WAVES], M = m_run[qt], L = 0.f;
#pragma unroll
        for (int w = 0; w < AO_WAVES; ++w) { mw[w] = smax[w * 32 + row]; sw[w] = ssum[w * 32 + row]; }
#pragma unroll
        for (int w = 0; w < AO_WAVES; ++w) M = fmaxf(M, mw[w]);
        const float alpha = ao_exp2(m_run[qt] - M);
        m_run[qt] = M;
#pragma unroll
        for (int w = 0; w < AO_WAVES; ++w) { fw[qt][w] = ao_exp2(mw[w] - M); L += fw[qt][w] * sw[w]; }
        l_run[qt] = l_run[qt] * alpha + L;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[qt][j] *= alpha;
      }
#pragma unroll
      for (int w = 0; w < AO_WAVES; ++w) {
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) {
          const int tl = w + ti * AO_WAVES;
          if (tl < AO_TILES) {                               // (compile time) a tile the pass does not have holds P = 0 and finite V^T bytes
            acc0 = ao_mfma_k16(vf[tl], pf0[tl], acc0);
            acc1 = ao_mfma_k16(vf[tl], pf1[tl], acc1);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[0][j] = __builtin_fmaf(fw[0][w], acc0[j], o[0][j]);
          o[1][j] = __builtin_fmaf(fw[1][w], acc1[j], o[1][j]);
        }
      }
    }
    if (t0 == 0) {                                            // (uniform) the rest of the list, now that valid_len is known
      const int np16 = (lim1 + 15) >> 4;
      xp = __builtin_amdgcn_readfirstlane(max(0, np16 - pref_next));
      nt = __builtin_amdgcn_readfirstlane(pfx ? AO_TILES + xp + max(0, nb16 - 1) : max(AO_TILES, np16));
    }
    t0 += cnt;
    if (t0 < nt) __syncthreads();        // another pass re-stages both regions
  }
  AO_STAMP(4)
  const int fr = fr_k, g = g_k;
  // ---------------------------------------------------------------- normalised attention rows (bf16): this wave's 16 head-dim columns of x [row = head-in-group * nq +
  // token][128 d] go to the start of its OWN V^T rows (dead: only this wave read them) -- no barrier between the P V phase and this write
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const float inv = l_run[qt] > 0.f ? 1.0f / l_run[qt] : 0.f;
    const int row = qt * 16 + fr;
    *reinterpret_cast<u32x2*>(smem + wave * VAREA + row * 32 + g * 8) = u32x2{pack_bf16x2(o[qt][0] * inv, o[qt][1] * inv), pack_bf16x2(o[qt][2] * inv, o[qt][3] * inv)};
  }
  __syncthreads();
  AO_STAMP(5)
  // ---------------------------------------------------------------- o_proj on this workgroup's 16 output columns: wave w contracts k = 32 (NS w + s) .. + 32 of the group
  f32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    const int k = (wave * NS + st) * 32, hg = k >> 7, dcol = k & 127;
    const int d0 = dcol + g * 8;                              // 8 consecutive head-dim columns: half of the 16 the wave d0 >> 4 wrote
    const bf16x8 xf = as_bf16x8(*reinterpret_cast<const u32x4*>(smem + (d0 >> 4) * VAREA + (hg * nq + min(fr, nq - 1)) * 32 + (d0 & 15) * 2));
    acc = mfma16(as_bf16x8(wof[st]), xf, acc);               // rows = output columns 4 g + j of the unit, columns = tokens
  }
  *reinterpret_cast<f32x4*>(smem + wave * VAREA + 2048 + (fr * 16 + g * 4) * 4) = acc;      // [token][column] partial tile, behind this wave's x columns
  __syncthreads();
  AO_STAMP(6)
  if (tid < 256) {
    const int tok = tid >> 4, n = tid & 15;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < AO_WAVES; ++w) v += *reinterpret_cast<const float*>(smem + w * VAREA + 2048 + tid * 4);
    if (tok < nq) p.out[((size_t)kvh * nq + tok) * p.N + blockIdx.x * 16 + n] = v;
  }
  asm volatile("" ::"v"(wwarm));
  AO_STAMP(7)
}

extern "C" int vlaser_attn_oproj(const VlaserAttnArgs* a, const void* wo_packed, float* out_f32, int N, vl_stream_t s) {
  VL_CHECK(a && a->q && a->k && a->vt && wo_packed && out_f32, "vlaser_attn_oproj: null pointer");
  VL_CHECK(a->batch == 1 && a->head_dim == 128 && a->n_q_heads % a->n_kv_heads == 0, "vlaser_attn_oproj: batch 1, head_dim 128, whole GQA groups");
  const int G = a->n_q_heads / a->n_kv_heads;
  VL_CHECK(a->sq >= 1 && a->sq <= 16 && G * a->sq <= 32, "vlaser_attn_oproj: group * tokens must be <= 32 (tokens=%d)", a->sq);
  VL_CHECK(G == 2 || G == 4 || G == 6 || G == 8, "vlaser_attn_oproj: GQA group %d unsupported (the contraction group * 128 is dealt to 8 waves in K-steps of 32: group 2 / 4 / 6 / 8)", G);
  VL_CHECK(a->mode == VL_ATTN_FULL || a->mode == VL_ATTN_PREFIX, "vlaser_attn_oproj: mode must be FULL or PREFIX");
  VL_CHECK(a->kv_len >= 1 && a->kv_len <= a->ld_vt && a->ld_vt % 32 == 0, "vlaser_attn_oproj: kv_len / padded V^T row");
  VL_CHECK(a->mode != VL_ATTN_PREFIX || (a->blk_start % 16 == 0 && a->blk_start >= 0), "vlaser_attn_oproj: blk_start must be a multiple of 16");
  VL_CHECK(N % 16 == 0 && ((uintptr_t)wo_packed & 15) == 0, "vlaser_attn_oproj: N must be a multiple of 16, packed W_o 16-byte aligned");
  AttnOP p;
  p.a = *a; p.wo = (const u32x4*)wo_packed; p.out = out_f32; p.N = N;
  const dim3 grid(N / 16, a->n_kv_heads);
#define AO_LAUNCH(NS_)                                                                                        \
  do {                                                                                                        \
    if (int rc = set_max_lds_once(attn_oproj_kernel<NS_>, AO_LDS_BYTES)) return rc;                           \
    hipLaunchKernelGGL(attn_oproj_kernel<NS_>, grid, dim3(AO_THREADS), AO_LDS_BYTES, (hipStream_t)s, p);     \
  } while (0)
  switch (G) {
    case 2: AO_LAUNCH(1); break;
    case 4: AO_LAUNCH(2); break;
    case 6: AO_LAUNCH(3); break;
    default: AO_LAUNCH(4); break;
  }
#undef AO_LAUNCH
  VL_LAUNCH_CHECK();
  return 0;
}
