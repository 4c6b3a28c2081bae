"""Thin torch-tensor wrappers over the C ABI (include/vlaser_hip.h).  PyTorch is plumbing here: it owns device
memory and the stream; every op below is a launch of a hand-written gfx950 kernel in libvlaser_hip.so."""
import ctypes as C
import math
import os


import threading

import torch

from . import _lib as L

BF16 = torch.bfloat16


_TLS = threading.local()      # .stream: raw hipStream_t pinned by a caller that queues a run of launches without looking the current stream up per launch
                              # (sft.py: `torch.cuda.current_stream()` was 30 % of a forward + backward's host time); per thread, like torch's current stream


def pinned_stream():
    return getattr(_TLS, 'stream', None)


def pin_stream(handle):
    """Pin (handle) or lift (None) the stream of this thread's C-ABI launches; returns the previous pin so that callers can restore it."""
    prev = getattr(_TLS, 'stream', None)
    _TLS.stream = handle
    return prev


def _stream():
    h = getattr(_TLS, 'stream', None)
    return torch.cuda.current_stream().cuda_stream if h is None else h


def _p(t):
    return None if t is None else t.data_ptr()


def _chk(t, dtype=BF16):
    assert t.is_cuda and t.dtype == dtype and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())
    return t


# ------------------------------------------------------------------------------------------------ weight packing
def head_perm(head_dim=128):
    """packed row p of a 128-row head <-> natural d = 16*(p//32) + p%16 + 64*((p%32)//16): puts the RoPE pair
    (d, d+64) into the same MFMA lane (gemm.hip / skinny.hip QKV_ROPE epilogue)."""
    p = torch.arange(head_dim)
    return 16 * (p // 32) + (p % 16) + 64 * ((p % 32) // 16)


def pack_qkv(qw, kw, vw, qb, kb, vb, head_dim=128):
    w = torch.cat([qw, kw, vw], dim=0)
    b = torch.cat([qb, kb, vb], dim=0)
    nh = w.shape[0] // head_dim
    idx = (torch.arange(nh)[:, None] * head_dim + head_perm(head_dim)[None, :]).reshape(-1).to(w.device)
    return w[idx].contiguous(), b[idx].contiguous()


def pack_gate_up(gw, uw):
    """rows [32j, 32j+16) = gate rows 16j..16j+15, rows [32j+16, 32j+32) = up rows 16j..16j+15."""
    I, K = gw.shape
    assert I % 16 == 0
    return torch.stack([gw.view(I // 16, 16, K), uw.view(I // 16, 16, K)], dim=1).reshape(2 * I, K).contiguous()


def head_perm16(head_dim=128):
    """16-row lane-local units (skinny.hip, TPU = 1): packed row p of a head <-> natural d = 8*(p//16) + 2*((p%16)//4) + (p%4)%2 + 64*((p%4)//2):
    lane group g of tile t holds [d, d+1, d+64, d+65], d = 8t + 2g -- the RoPE pair (d, d+64) in one lane."""
    p = torch.arange(head_dim)
    q = p % 16
    return 8 * (p // 16) + 2 * (q // 4) + (q % 4) % 2 + 64 * ((q % 4) // 2)


def pack_qkv16(qw, kw, vw, qb, kb, vb, head_dim=128):
    """q/k/v fused for the 16-row-unit weight-streaming kernel: rows permuted inside each head by head_perm16."""
    w = torch.cat([qw, kw, vw], dim=0)
    b = torch.cat([qb, kb, vb], dim=0)
    nh = w.shape[0] // head_dim
    idx = (torch.arange(nh)[:, None] * head_dim + head_perm16(head_dim)[None, :]).reshape(-1).to(w.device)
    return w[idx].contiguous(), b[idx].contiguous()


def pack_gate_up8(gw, uw):
    """16-row lane-local units: rows [16j + 4g, +4) = [gate 8j+2g, gate 8j+2g+1, up 8j+2g, up 8j+2g+1] -> unit j yields activation columns 8j .. 8j+7."""
    I, K = gw.shape
    assert I % 8 == 0
    g4 = gw.view(I // 8, 4, 2, K)          # [unit, lane group, pair, K]
    u4 = uw.view(I // 8, 4, 2, K)
    return torch.cat([g4, u4], dim=2).reshape(2 * I, K).contiguous()


def pack_patch_embed(w, kpad=640):
    C_, k = w.shape[0], w[0].numel()
    out = torch.zeros(C_, kpad, dtype=w.dtype, device=w.device)
    out[:, :k] = w.reshape(C_, k)
    return out


def rope_table(n_pos, head_dim=128, theta=1e6, device='cuda'):
    """fp32 cos/sin [n_pos, head_dim/2], same fp32 arithmetic as Qwen2RotaryEmbedding."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    f = torch.arange(n_pos).float()[:, None] * inv_freq[None, :]
    return f.cos().contiguous().to(device), f.sin().contiguous().to(device)


# ------------------------------------------------------------------------------------------------ GEMM
def gemm(epi, A, W, out=None, bias=None, res=None, ls=None, N=None, **kw):
    _chk(A); _chk(W)
    a = L.GemmArgs()
    M, K = A.shape
    a.A, a.W = A.data_ptr(), W.data_ptr()
    a.M, a.N, a.K = M, (W.shape[0] if N is None else N), K
    a.lda, a.ldw = A.stride(0), W.stride(0)
    if out is not None:
        a.out, a.ldo = out.data_ptr(), out.stride(0)
    a.bias, a.res, a.ls = _p(bias), _p(res), _p(ls)
    part = kw.get('out_f32')
    if epi == L.EPI_PARTIAL and isinstance(part, torch.Tensor) and part.numel() < kw.get('k_splits', 1) * M * a.N:
        raise ValueError(f'vlaser_gemm: {kw.get("k_splits", 1)} fp32 slabs of [{M},{a.N}] do not fit the {part.numel()}-element partial buffer')
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    L.check(L.lib().vlaser_gemm(epi, C.byref(a), _stream()), 'vlaser_gemm')
    return out


def gemm_nn(epi, A, B, out=None, **kw):
    """out[M,N] = A[M,K] @ B[K,N] with B row-major as stored (a forward weight seen from its dgrad): NONE (bf16 `out`) or PARTIAL
    (`out_f32` slabs, `k_splits`)."""
    _chk(A); _chk(B)
    a = L.GemmArgs()
    M, K = A.shape
    assert B.shape[0] == K, (A.shape, B.shape)
    a.A, a.W = A.data_ptr(), B.data_ptr()
    a.M, a.N, a.K = M, B.shape[1], K
    a.lda, a.ldw = A.stride(0), B.stride(0)
    if out is not None:
        a.out, a.ldo = out.data_ptr(), out.stride(0)
    part = kw.get('out_f32')
    if epi == L.EPI_PARTIAL and isinstance(part, torch.Tensor) and part.numel() < kw.get('k_splits', 1) * M * a.N:
        raise ValueError(f'vlaser_gemm_nn: {kw.get("k_splits", 1)} fp32 slabs of [{M},{a.N}] do not fit the {part.numel()}-element partial buffer')
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    L.check(L.lib().vlaser_gemm_nn(epi, C.byref(a), _stream()), 'vlaser_gemm_nn')
    return out


def gemm_raw_nn(epi, A, B, out, M, N, K, lda, ldb, ldo, **kw):
    """Fully explicit NN GEMM call (views / batched operands): out[M,N] = A[M,K] @ B[K,N], B row-major with row stride ldb."""
    a = L.GemmArgs()
    a.A, a.W, a.out = A.data_ptr(), B.data_ptr(), out.data_ptr()
    a.M, a.N, a.K, a.lda, a.ldw, a.ldo = M, N, K, lda, ldb, ldo
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    L.check(L.lib().vlaser_gemm_nn(epi, C.byref(a), _stream()), 'vlaser_gemm_nn')


def gemm_raw(epi, A, W, out, M, N, K, lda, ldw, ldo, **kw):
    """Fully explicit GEMM call (views / batched operands): pointers from the tensors, geometry from the arguments."""
    a = L.GemmArgs()
    a.A, a.W = A.data_ptr(), W.data_ptr()
    a.M, a.N, a.K, a.lda, a.ldw, a.ldo = M, N, K, lda, ldw, ldo
    if epi in (L.EPI_PARTIAL,):
        a.out_f32 = out.data_ptr()
    else:
        a.out = out.data_ptr()
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    L.check(L.lib().vlaser_gemm(epi, C.byref(a), _stream()), 'vlaser_gemm')


def linear(x, W, bias=None, epi=None, res=None, ls=None, out=None, out_dtype=BF16):
    """out[M,N] = epi(x @ W^T)."""
    M, N = x.shape[0], W.shape[0]
    if epi is None:
        epi = L.EPI_BIAS if bias is not None else L.EPI_NONE
    if out is None:
        n_out = N // 2 if epi == L.EPI_SWIGLU else N
        out = torch.empty(M, n_out, device=x.device, dtype=torch.float32 if epi == L.EPI_F32 else out_dtype)
    return gemm(epi, x, W, out=out, bias=bias, res=res, ls=ls)


# ------------------------------------------------------------------------------------------------ attention
def _attn_args(q, k, vt, out, batch, sq, kv_len, n_q, n_kv, hd, q_str, k_str, vt_str, o_str, ld_vt, scale, mode,
               causal_off=0, valid_len=None, blk_start=0, q_row_off=0, parts=None, n_splits=1, first_tok_kv_len=0, lse_out=None, dense_mask=None):
    a = L.AttnArgs()
    a.q, a.k, a.vt, a.out = q.data_ptr(), k.data_ptr(), vt.data_ptr(), _p(out)
    a.batch, a.sq, a.kv_len, a.n_q_heads, a.n_kv_heads, a.head_dim = batch, sq, kv_len, n_q, n_kv, hd
    a.q_bs, a.q_hs, a.q_ss = q_str
    a.k_bs, a.k_hs = k_str
    a.vt_bs, a.vt_hs = vt_str
    a.o_bs, a.o_ss = o_str
    a.ld_vt, a.scale, a.mode, a.causal_off = ld_vt, scale, mode, causal_off
    a.valid_len = _p(valid_len)
    a.blk_start, a.q_row_off = blk_start, q_row_off
    if parts is not None:
        a.part_m, a.part_l, a.part_o = parts[0].data_ptr(), parts[1].data_ptr(), parts[2].data_ptr()
    a.n_splits = n_splits
    a.first_tok_kv_len = first_tok_kv_len
    a.lse_out = _p(lse_out)
    if dense_mask is not None:
        # (ABI 8) VL_ATTN_DENSE: fp32 additive mask VIEW [B, sq, >= kv_len] (row 0 = query token 0, column 0 = key 0) of a buffer with padded rows
        assert dense_mask.dtype == torch.float32 and dense_mask.dim() == 3 and dense_mask.stride(2) == 1 and dense_mask.shape[1] >= sq
        a.mask, a.mask_bs, a.mask_rs = dense_mask.data_ptr(), dense_mask.stride(0), dense_mask.stride(1)
    return a


def attn_prefill(*args, **kw):
    a = _attn_args(*args, **kw)
    L.check(L.lib().vlaser_attn_prefill(C.byref(a), _stream()), 'vlaser_attn_prefill')


def attn_bwd(q, k, vt, o, d_o, lse, delta_ws, dq, dk, dv, S, n_q, n_kv, s_max, scale, causal=True, kv_valid=None, head_dim=128):
    """Fused backward of the prefill attention (csrc/attn_bwd.hip): dq, and one dk / dv partial per Q head, from q / k / v^T / o / d_o and the forward's lse."""
    L.check(L.lib().vlaser_attn_bwd(q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), delta_ws.data_ptr(), dq.data_ptr(),
                                    dk.data_ptr(), dv.data_ptr(), S, n_q, n_kv, s_max, scale, int(causal), S if kv_valid is None else kv_valid, head_dim, _stream()), 'vlaser_attn_bwd')


def attn_splits(kv_len):
    """Key splits of the skinny attention: about 2 chunks (of 32 keys) per block, at most 8 splits (measured: 389 keys,
    1/2/4/7 splits -> 13.5/9.5/7.7/6.3 us per launch)."""
    return max(1, min(8, (((kv_len + 31) // 32) + 1) // 2))


def attn_skinny_args(q, k, vt, parts, batch, sq, kv_len, n_q, n_kv, hd, q_str, k_str, vt_str, ld_vt, scale, mode, n_splits, **kw):
    return _attn_args(q, k, vt, None, batch, sq, kv_len, n_q, n_kv, hd, q_str, k_str, vt_str, (0, 0), ld_vt, scale, mode, parts=parts,
                      n_splits=n_splits, **kw)


def launch_attn_skinny(a, stream=None):
    L.check(L.lib().vlaser_attn_skinny(C.byref(a), _stream() if stream is None else stream), 'vlaser_attn_skinny')


def attn_skinny(q, k, vt, parts, batch, sq, kv_len, n_q, n_kv, hd, q_str, k_str, vt_str, ld_vt, scale, mode, n_splits, **kw):
    """Writes flash-decoding partials (m, l, o) per (b, kv head, split) into `parts`; merged by skinny(PRO_ATTN)."""
    launch_attn_skinny(attn_skinny_args(q, k, vt, parts, batch, sq, kv_len, n_q, n_kv, hd, q_str, k_str, vt_str, ld_vt, scale, mode, n_splits,
                                        **kw))


def attn_partial_buffers(batch, n_kv, device, max_splits=8):
    z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)
    return z(batch, n_kv, max_splits, 32), z(batch, n_kv, max_splits, 32), z(batch, n_kv, max_splits, 32, 128)


# ------------------------------------------------------------------------------------------------ skinny GEMV
class PackedW:
    """Weight [N,K] packed for the skinny kernel: fragment-major [k_splits][N/32][8 waves][steps][2 tiles][64 lanes][8]."""
    __slots__ = ('t', 'N', 'n_valid', 'K', 'k_splits', 'tpu')

    def __init__(self, t, N, n_valid, K, k_splits, tpu=2):
        self.t, self.N, self.n_valid, self.K, self.k_splits, self.tpu = t, N, n_valid, K, k_splits, tpu


SK_WAVES = 8
SK_STEPS = (1, 2, 3, 4, 5, 6, 7, 8, 14, 16)      # K-steps (of 32) per wave the kernel is built for; 14 / 16 = two chunks of 7 / 8


def pack_skinny(W, k_splits=1, tpu=2, k_pad=None):
    """Row-major [N,K] bf16 -> PackedW.  Lane (r = l&15, g = l>>4) of wave w, K-step s, tile t of unit u, split ks holds
    W[u*32 + t*16 + r, ks*kb + w*kw + s*32 + g*8 : +8]; each wave-level load is a contiguous 1 KiB.  k_pad > K appends zero
    columns (widths that do not factor into k_splits x 8 waves x an available step count; the activation is zero-padded alike)."""
    N, K = W.shape
    if k_pad is not None and k_pad > K:
        W = torch.cat([W, torch.zeros(N, k_pad - K, dtype=W.dtype, device=W.device)], 1)
        K = k_pad
    assert K % (k_splits * 32 * SK_WAVES) == 0, (K, k_splits)
    rpu = 16 * tpu
    Np = (N + rpu - 1) // rpu * rpu
    if Np != N:
        W = torch.cat([W, torch.zeros(Np - N, K, dtype=W.dtype, device=W.device)], 0)
    ns = K // (k_splits * SK_WAVES * 32)
    assert ns in SK_STEPS, f'{ns} K-steps per wave: not a kernel variant {SK_STEPS}'
    v = W.view(Np // rpu, tpu, 16, k_splits, SK_WAVES, ns, 4, 8)       # [u, t, r, ks, w, s, g, e]
    v = v.permute(3, 0, 4, 5, 1, 6, 2, 7).contiguous()                 # [ks, u, w, s, t, g, r, e]
    return PackedW(v.reshape(-1), Np, N, K, k_splits, tpu)


def skinny_args(x, W: PackedW, M, **kw):
    """Filled VlaserSkinnyArgs (+ tensors it must keep alive).  Building the ctypes struct costs ~8 us of host time; callers
    on a host-bound path (greedy decode: 141 launches per token) build it once and re-launch it (`launch_skinny`)."""
    a = L.SkinnyArgs()
    a.x, a.W = _p(x), W.t.data_ptr()
    a.M, a.N, a.K, a.ldw, a.n_valid, a.tiles_per_unit = M, W.N, W.K, W.K, W.n_valid, W.tpu
    a.k_splits = W.k_splits
    a.eps = kw.pop('eps', 1e-6)
    keep = None
    b = kw.get('bias')
    if b is not None and b.numel() < W.N:       # the epilogue reads bias with unconditional vector loads over the padded N
        keep = kw['bias'] = torch.cat([b, torch.zeros(W.N - b.numel(), dtype=b.dtype, device=b.device)])
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    return a, keep


def launch_skinny(pro, epi, a, stream=None):
    L.check(L.lib().vlaser_skinny(pro, epi, C.byref(a), _stream() if stream is None else stream), 'vlaser_skinny')


def skinny(pro, epi, x, W: PackedW, M, **kw):
    a, _keep = skinny_args(x, W, M, **kw)
    launch_skinny(pro, epi, a)


# ---- r05 latency chain (csrc/chain.hip)
DOWN4_WAVES, DOWN4_LOADS = 7, 10          # K = 7 waves x 10 loads x 128 = 8960 (Qwen2.5-1.5B / action-expert MLP width)


def chain_down_geometry(N):
    """(workgroups, columns per group, groups per workgroup) of vlaser_chain_down for an output width N."""
    c, g = C.c_int(), C.c_int()
    wgs = L.lib().vlaser_chain_down_geometry(N, C.byref(c), C.byref(g))
    return wgs, c.value, g.value


def pack_down4(W, nw=DOWN4_WAVES, nl=DOWN4_LOADS, k_splits=1):
    """down_proj.weight [N, K] -> [K split][workgroup][nw waves][loads][groups][16 blocks][cols][8] bf16 for vlaser_chain_down (k_splits = 1) / vlaser_chain_down2
    (k_splits = 2: two K halves, always 2 column groups per workgroup): block b, column i of group c of wave w, load l holds
    W[(wg groups + c) cols + i, kh K/k_splits + (w loads + l) 128 + 8 b : + 8] -- one contiguous 16 x cols x 16 bytes per wave-level load, one contiguous stream per workgroup."""
    N, K = W.shape
    if k_splits == 1:
        wgs, cols, groups = chain_down_geometry(N)
    else:
        assert k_splits == 2 and nl % 2 == 0 and (N % 6 == 0 or N % 8 == 0), (N, K, k_splits)
        cols, groups = (3 if N % 6 == 0 else 4), 2
        wgs = N // (cols * groups)
    nlk = nl // k_splits
    assert N == wgs * cols * groups and K == nw * nl * 128, (N, K)
    v = W.view(wgs, groups, cols, k_splits, nw, nlk, 16, 8)                # [wg, c, i, kh, w, l, b, e]
    return v.permute(3, 0, 4, 5, 1, 6, 2, 7).contiguous().reshape(-1)      # [kh, wg, w, l, c, b, i, e]


def chain_down2(x, W42, out_f32, M, N, K, dbg=None, stream=None):
    """out_f32[2, M, N] = the two K halves of x[M, :K] @ W^T (fp32, no residual), W42 = pack_down4(W, k_splits=2)."""
    assert x.dtype == BF16 and out_f32.dtype == torch.float32 and x.stride(-1) == 1 and out_f32.numel() >= 2 * M * N
    L.check(L.lib().vlaser_chain_down2(x.data_ptr(), x.stride(0), W42.data_ptr(), out_f32.data_ptr(), M, N, K, dbg if isinstance(dbg, int) else _p(dbg),
                                       _stream() if stream is None else stream), 'vlaser_chain_down2')


def chain_down2_supported(M, N, K):
    return bool(L.lib().vlaser_chain_down2_supported(M, N, K)) and bool(L.lib().vlaser_chain_qkv2_supported(M, 128, 768 if N == 768 else N))


def chain_qkv_supported(M, N, K):
    return bool(L.lib().vlaser_chain_qkv_supported(M, N, K))


def chain_gu_supported(M, N, K, n_partials, tpu=2):
    return bool(L.lib().vlaser_chain_gu_supported(M, N, K, n_partials, tpu))


def chain_down_supported(M, N, K):
    return bool(L.lib().vlaser_chain_down_supported(M, N, K))


def chain_attn_splits(kv_len):
    return ((kv_len + 31) // 32 + 1) // 2


def chain_attn_buffers(batch, n_kv, device, max_splits=16):
    """(m, l) pairs fp32 [B, n_kv, splits, 32, 2] and normalised rows bf16 [B, n_kv, splits, 32, 128] of vlaser_chain_attn."""
    return (torch.zeros(batch, n_kv, max_splits, 32, 2, dtype=torch.float32, device=device), torch.zeros(batch, n_kv, max_splits, 32, 128, dtype=BF16, device=device))


def chain_oproj_supported(M, N, K, k_splits, attn_splits, group):
    return bool(L.lib().vlaser_chain_oproj_supported(M, N, K, k_splits, attn_splits, group))


def launch_chain_attn(a, stream=None):
    L.check(L.lib().vlaser_chain_attn(C.byref(a), _stream() if stream is None else stream), 'vlaser_chain_attn')


def launch_chain_oproj(a, stream=None):
    L.check(L.lib().vlaser_chain_oproj(C.byref(a), _stream() if stream is None else stream), 'vlaser_chain_oproj')


def launch_chain_qkv(a, stream=None):
    L.check(L.lib().vlaser_chain_qkv(C.byref(a), _stream() if stream is None else stream), 'vlaser_chain_qkv')


def launch_chain_gu(a, stream=None):
    L.check(L.lib().vlaser_chain_gu(C.byref(a), _stream() if stream is None else stream), 'vlaser_chain_gu')


def chain_down(x, W4, res, h_out, M, N, K, dbg=None, stream=None):
    """h_out[M, N] = bf16(res + x[M, :K] @ W^T), W4 = pack_down4(W)."""
    assert x.dtype == BF16 and res.dtype == BF16 and h_out.dtype == BF16 and x.stride(-1) == 1 and res.data_ptr() != h_out.data_ptr()
    L.check(L.lib().vlaser_chain_down(x.data_ptr(), x.stride(0), W4.data_ptr(), res.data_ptr(), h_out.data_ptr(), M, N, K, dbg if isinstance(dbg, int) else _p(dbg),
                                      _stream() if stream is None else stream), 'vlaser_chain_down')


# measured on the action-expert chunk: 72 / 100 / 130 / 160 target blocks -> 16.63 / 16.52 / 16.48 / 16.50 ms
SK_TARGET_BLOCKS = 130


def pick_k_splits(K, N, target_blocks=None, rows_per_unit=32):
    """Smallest cross-block split-K factor that (a) keeps K/k_splits a multiple of 256 (8 waves x 32) with a step count the kernel
    has, and (b) gives about one block per CU (units = N/rows_per_unit); None when K does not factor that way (see skinny_geometry)."""
    if target_blocks is None:
        target_blocks = SK_TARGET_BLOCKS
    units = (N + rows_per_unit - 1) // rows_per_unit
    best = None
    for s in range(1, 9):            # <= 8 slabs: the consumer's prologue sums them in ONE batch of loads
        if K % (s * 256) or K // (s * 256) not in SK_STEPS:
            continue
        best = s
        if units * s >= target_blocks:
            break
    return best


def skinny_geometry(K, N, target_blocks=None, rows_per_unit=32):
    """(K_pad, k_splits) of a [N, K] weight for the weight-streaming kernel.  K_pad == K whenever K factors into
    k_splits x 8 waves x an available step count; otherwise the least zero-padding that does (Vlaser-8B's MLP width
    18944 = 2^9 x 37 -> 20480 = 5 splits x 16 steps, +8 % on the down projection only)."""
    ks = pick_k_splits(K, N, target_blocks, rows_per_unit)
    if ks is not None:
        return K, ks
    best = None
    for s in range(1, 9):
        for ns in SK_STEPS:
            kp = s * ns * 256
            if kp >= K:
                if best is None or kp < best[0]:
                    best = (kp, s)
                break
    return best


def skinny_supported(llm):
    """True when every matrix of a Qwen2 layer (+ lm_head) has a weight-streaming geometry: Vlaser-2B, the 768-wide action expert
    and (with chunked K / a zero-padded MLP width) Vlaser-8B."""
    H, I, nqd = llm.hidden_size, llm.intermediate_size, llm.num_attention_heads * llm.head_dim
    if H % 256 or H // 256 not in SK_STEPS:          # NORM-prologue kernels (qkv, gate/up, lm_head) cannot split or pad K
        return False
    return skinny_geometry(nqd, H)[0] == nqd and skinny_geometry(I, H) is not None


# ------------------------------------------------------------------------------------------------ helpers
def layernorm(x, w, b, eps, out=None):
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().vlaser_layernorm(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), x.numel() // x.shape[-1],
                                     x.shape[-1], eps, _stream()), 'vlaser_layernorm')
    return out


def rmsnorm(x, w, eps, out=None):
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().vlaser_rmsnorm(x.data_ptr(), w.data_ptr(), out.data_ptr(), x.numel() // x.shape[-1], x.shape[-1], eps,
                                   _stream()), 'vlaser_rmsnorm')
    return out


def im2col(pix, A, T, img, kpad):
    L.check(L.lib().vlaser_im2col(pix.data_ptr(), A.data_ptr(), T, img, kpad, _stream()), 'vlaser_im2col')


def vit_assemble(patch, cls, pos, h, T, P, Cc):
    L.check(L.lib().vlaser_vit_assemble(patch.data_ptr(), cls.data_ptr(), pos.data_ptr(), h.data_ptr(), T, P, Cc, _stream()),
            'vlaser_vit_assemble')


def pixel_shuffle_ln(x, w, b, out, T, G, Cc, eps, ps_v1=0):
    assert out.numel() >= T * (G // 2) ** 2 * 4 * Cc and x.numel() >= T * (G * G + 1) * Cc, 'pixel_shuffle_ln: buffer smaller than T tiles'
    L.check(L.lib().vlaser_pixel_shuffle_ln(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), T, G, Cc, eps, ps_v1,
                                            _stream()), 'vlaser_pixel_shuffle_ln')


def pixel_shuffle(x, out, T, G, Cc, ps_v1=0):
    assert out.numel() >= T * (G // 2) ** 2 * 4 * Cc and x.numel() >= T * (G * G + 1) * Cc, 'pixel_shuffle: buffer smaller than T tiles'
    L.check(L.lib().vlaser_pixel_shuffle(x.data_ptr(), out.data_ptr(), T, G, Cc, ps_v1, _stream()), 'vlaser_pixel_shuffle')


def embed_merge(ids, embed, vit, out, img_id, pad_id, zero_pad, rank_ws, count_out=None):
    n = ids.numel()
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    L.check(L.lib().vlaser_embed_merge(ids.data_ptr(), n, embed.data_ptr(), _p(vit), 0 if vit is None else vit.shape[0] if vit.dim() == 2 else vit.numel() // vit.shape[-1],
                                       out.data_ptr(), embed.shape[1], img_id, pad_id, int(zero_pad), rank_ws.data_ptr(),
                                       _p(count_out), _stream()), 'vlaser_embed_merge')


def argmax_workspace(max_rows, device):
    """Zeroed workspace of vlaser_argmax's many-workgroup form (pairs + arrival counters); owned by the launches that receive it, one stream at a time."""
    return torch.zeros(L.lib().vlaser_argmax_ws_bytes(max_rows), dtype=torch.uint8, device=device)


def argmax(logits, out_id, embed=None, next_h=None, ws=None):
    M, N = logits.shape
    L.check(L.lib().vlaser_argmax(logits.data_ptr(), M, N, out_id.data_ptr(), _p(embed), _p(next_h),
                                  0 if embed is None else embed.shape[1], _p(ws), 0 if ws is None else ws.numel(), _stream()), 'vlaser_argmax')


def vla_prep(action, w1, b1, xcat, M, W, adim, t, max_period):
    L.check(L.lib().vlaser_vla_prep(action.data_ptr(), w1.data_ptr(), b1.data_ptr(), xcat.data_ptr(), M, W, adim, t, max_period,
                                    _stream()), 'vlaser_vla_prep')


def small_linear(x, w, b, out, M, N, K):
    L.check(L.lib().vlaser_small_linear(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, _stream()),
            'vlaser_small_linear')


INTEGRATION_METHODS = {'euler': 0, 'heun': 1, 'rk4': 2}


def integration_coef(dt, method='euler'):
    """The scalar the reference's `integration_step` multiplies the (re-combined) velocity with (pizero_internvl.py:1309-1331): dt | 0.5 dt | dt / 6, evaluated in
    Python double exactly as the reference evaluates it; ctypes rounds it to fp32 the way torch rounds a Python scalar."""
    return {'euler': dt, 'heun': 0.5 * dt, 'rk4': dt / 6.0}[method]


def vla_euler(h_in, partials, n_partials, M, norm_w, eps, wd, bd, action, W, adim, dt, clip, do_clip, vel_out=None, ring=None, ring_ctr=None, method='euler'):
    """ring (fp32 [slots, stride]) + ring_ctr (device int32): the result also goes to slot (*ring_ctr mod slots) -- infer_action returns that view.
    method: the reference's `integration_method` (euler | heun | rk4); dt is the STEP (1 / num_inference_steps), the method's coefficient is derived here."""
    L.check(L.lib().vlaser_vla_euler(h_in.data_ptr(), _p(partials), n_partials, M, norm_w.data_ptr(), eps, wd.data_ptr(),
                                     bd.data_ptr(), action.data_ptr(), W, adim, integration_coef(dt, method), clip, int(do_clip), _p(vel_out), _p(ring), _p(ring_ctr),
                                     0 if ring is None else ring.shape[0], 0 if ring is None else ring.stride(0), INTEGRATION_METHODS[method], _stream()),
            'vlaser_vla_euler')


_PIX_DTYPES = {torch.bfloat16: 0, torch.float32: 1, torch.uint8: 2}


_MASK_DTYPES = {torch.bfloat16: 0, torch.float32: 1, torch.float16: 2}


def vla_stage(ids, ids_out, valid_in, valid_out, proprio, proprio_out, noise, noise_out, pix, pix_out, pad_id, mean, std, call_ctr=None, call_no=0,
              masks=None, n_act=0, positions=None, pos_out=None, mask_slot=None):
    """All per-call inputs of infer_action into the chunk graph's static slots in ONE launch (device tensors, contiguous).  pix: bf16 / fp32
    (already normalised) or uint8 [n,3,H,W] (normalised here, InternVLAProcessor arithmetic); valid_in: int32 / int64 [B] or None (= zero count of the dense
    mask's proprio row when `masks` is given, else the count of ids != pad_id).  masks = (image_text_proprio_mask | None, action_mask | None): the
    reference's dense additive masks, checked on the device against the prefix + trailing-block pattern (error word in call_ctr, see the header).
    positions = (vlm | None, proprio | None, action | None) int64 device tensors -> pos_out = (vlm, proprio, action, ride | None) int32 slots.
    call_ctr: int32[3] {call number, error word of even calls, of odd calls}; call_no: this call's number (host-owned).
    mask_slot (ABI 8): fp32 [B, T + 1 + n_act, ld] -- GENERAL masks: both masks are copied there for the VL_ATTN_DENSE launches instead of being checked."""
    a = L.VlaStageArgs()
    B, T = ids.shape
    assert ids.dtype == torch.int64 and ids.is_cuda and ids.is_contiguous()
    a.ids, a.ids_out, a.B, a.T, a.pad_id = ids.data_ptr(), ids_out.data_ptr(), B, T, pad_id
    if valid_in is not None:
        assert valid_in.dtype in (torch.int32, torch.int64) and valid_in.is_cuda and valid_in.is_contiguous() and valid_in.numel() == B
        a.valid_in, a.valid_is_i64 = valid_in.data_ptr(), int(valid_in.dtype == torch.int64)
    a.valid_out = valid_out.data_ptr()
    for t in (proprio, noise):
        assert t.dtype == torch.float32 and t.is_cuda and t.is_contiguous()
    # the C side never sees the capacities of the slots it copies into: a wrong proprio_dim / action_dim / horizon must not become an out-of-bounds device write
    if proprio_out.numel() < proprio.numel() or noise_out.numel() < noise.numel() or valid_out.numel() < B or ids_out.numel() < ids.numel():
        raise ValueError(f'vla_stage: slot smaller than its input (proprio {proprio.numel()} -> {proprio_out.numel()}, noise {noise.numel()} -> {noise_out.numel()}, '
                         f'ids {ids.numel()} -> {ids_out.numel()}, valid_len {B} -> {valid_out.numel()})')
    a.proprio, a.proprio_out, a.n_proprio = proprio.data_ptr(), proprio_out.data_ptr(), proprio.numel()
    a.noise, a.noise_out, a.n_noise = noise.data_ptr(), noise_out.data_ptr(), noise.numel()
    assert pix.is_cuda and pix.is_contiguous() and pix.dtype in _PIX_DTYPES and pix_out.dtype == BF16 and pix_out.numel() >= pix.numel()
    a.pix, a.pix_out, a.n_pix, a.pix_dtype = pix.data_ptr(), pix_out.data_ptr(), pix.numel(), _PIX_DTYPES[pix.dtype]
    a.hw = pix.shape[-1] * pix.shape[-2]
    for i in range(3):
        a.mean[i], a.std[i] = mean[i], std[i]
    a.call_ctr, a.call_no = _p(call_ctr), call_no
    if call_ctr is not None:
        assert call_ctr.dtype == torch.int32 and call_ctr.numel() >= 3
    a.n_act = n_act
    if masks is not None and (masks[0] is not None or masks[1] is not None):
        m1, m2 = masks
        dts = {m.dtype for m in (m1, m2) if m is not None}
        if len(dts) != 1 or next(iter(dts)) not in _MASK_DTYPES:
            raise ValueError(f'vla_stage: the dense masks must share one dtype out of bf16 / fp32 / fp16, got {dts}')
        a.mask_dtype = _MASK_DTYPES[next(iter(dts))]
        if m1 is not None:
            if tuple(m1.shape) != (B, 1, T + 1, T + 1):
                raise ValueError(f'image_text_proprio_mask must be [{B},1,{T + 1},{T + 1}], got {tuple(m1.shape)}')
            assert m1.is_cuda and m1.stride(-1) == 1, 'image_text_proprio_mask: rows must be contiguous (any batch / row stride)'
            a.itp_mask, a.itp_bs, a.itp_rs = m1.data_ptr(), m1.stride(0), m1.stride(2)
        if m2 is not None:
            if tuple(m2.shape) != (B, 1, n_act, T + 1 + n_act):
                raise ValueError(f'action_mask must be [{B},1,{n_act},{T + 1 + n_act}], got {tuple(m2.shape)}')
            assert m2.is_cuda and m2.stride(-1) == 1, 'action_mask: rows must be contiguous (any batch / row stride)'
            a.action_mask, a.act_bs, a.act_rs = m2.data_ptr(), m2.stride(0), m2.stride(2)
    if mask_slot is not None:
        if masks is None or masks[0] is None or masks[1] is None:
            raise ValueError('general masks: image_text_proprio_mask AND action_mask are required')
        assert mask_slot.dtype == torch.float32 and mask_slot.is_contiguous() and mask_slot.dim() == 3 and mask_slot.shape[0] >= B and mask_slot.shape[1] == T + 1 + n_act
        a.mask_slot, a.mask_ld = mask_slot.data_ptr(), mask_slot.shape[2]
    if positions is not None:
        want = ((B, T), (B, 1), (B, n_act))
        for name, t, o, shp in zip(('pos_vlm', 'pos_pro', 'pos_act'), positions, pos_out[:3], want):
            if t is None:
                continue
            if tuple(t.shape) != shp:
                raise ValueError(f'{name}: position ids must be {list(shp)}, got {tuple(t.shape)}')
            assert t.dtype == torch.int64 and t.is_cuda and t.is_contiguous() and o.dtype == torch.int32 and o.numel() >= t.numel()
            setattr(a, name, t.data_ptr())
            setattr(a, name + '_out', o.data_ptr())
        if len(pos_out) > 3 and pos_out[3] is not None and positions[1] is not None and positions[2] is not None:
            assert B == 1 and pos_out[3].numel() >= 1 + n_act
            a.pos_ride_out = pos_out[3].data_ptr()
    L.check(L.lib().vlaser_vla_stage(C.byref(a), _stream()), 'vlaser_vla_stage')


def vla_step(a_in, a_out, w21, cs, w3, b3, h_out, M, W, adim, finish=None, vel_out=None, dt=0.0, method='euler'):
    """One launch between two passes through the expert: `finish` = (h_in, partials, n_partials, rows_in, row_off, norm_w, eps, wd, bd) completes the
    previous Euler step (a_out = a_in + dt * vel), then the action encoder (folded linear_1 / time embedding: w21, cs) writes h_out."""
    if finish is None:
        f = (None, None, 0, M, 0, None, 0.0, None, None)
    else:
        f = finish
    L.check(L.lib().vlaser_vla_step(_p(f[0]), _p(f[1]), f[2], f[3], f[4], _p(f[5]), f[6], _p(f[7]), _p(f[8]), a_in.data_ptr(), a_out.data_ptr(), _p(vel_out),
                                    integration_coef(dt, method), 0 if finish is None else 1, w21.data_ptr(), cs.data_ptr(), w3.data_ptr(), b3.data_ptr(), h_out.data_ptr(), M, W,
                                    adim, INTEGRATION_METHODS[method], _stream()),
            'vlaser_vla_step')


def fold_action_encoder(w1, b1, w2, b2, W, adim, n_steps, max_period):
    """Host-side constants of vlaser_vla_step (fp32): W21 = W2[:, W:] @ W1 and, per Euler step s (t = s / n_steps), C[s] = W2[:, :W] @ temb(t) + W2[:, W:] @ b1 + b2,
    temb = SinusoidalPosEmb (modules.py:9-22) rounded to bf16 as the reference's bf16 module sees it."""
    w1f, b1f, w2f, b2f = w1.float(), b1.float(), w2.float(), b2.float()
    w21 = (w2f[:, W:] @ w1f).contiguous()
    half = W // 2
    e = math.log(max_period) / (half - 1)
    freq = torch.exp(-e * torch.arange(half, dtype=torch.float32, device=w1.device))
    t = torch.arange(n_steps, dtype=torch.float32, device=w1.device)[:, None] / n_steps
    ang = t * freq[None]
    temb = torch.cat([ang.sin(), ang.cos()], -1).to(torch.bfloat16).float()
    cs = (temb @ w2f[:, :W].t() + (w2f[:, W:] @ b1f + b2f)[None]).contiguous()
    return w21, cs


def reduce_norm(h_in, partials, n_partials, M, C, h_out, x_out=None, bias=None, ls=None, norm=0, norm_w=None, norm_b=None, eps=1e-6):
    """h_out = h_in + [ls*](sum partials [+bias]); x_out = norm(h_out) (norm: 0 none, 1 RMS, 2 LayerNorm)."""
    L.check(L.lib().vlaser_reduce_norm(_p(h_in), _p(partials), n_partials, _p(bias), _p(ls), norm, _p(norm_w), _p(norm_b), eps,
                                       h_out.data_ptr(), _p(x_out), M, C, _stream()), 'vlaser_reduce_norm')


_GEMM_CFGS = ((1564, 64, 64, 420.0), (1500, 64, 128, 701.0), (1100, 128, 128, 850.0), (1440, 144, 128, 900.0), (1200, 128, 256, 1040.0), (1300, 256, 256, 1208.0))


_CU_BUDGET = 256


def set_cu_budget(cus):
    """CUs the GEMM tile / split heuristics (here and in csrc/gemm.hip) may count on; returns the previous value.  For launches on a CU-masked stream (the grids are then
    sized for the CUs the mask leaves: tools/micro/rccl_shadow_lab.py); everything else leaves the default of 256."""
    global _CU_BUDGET
    prev = L.lib().vlaser_set_cu_budget(int(cus))
    _CU_BUDGET = L.lib().vlaser_get_cu_budget()            # (an out-of-range request is refused: mirror what the library holds)
    return prev


def get_cu_budget():
    return L.lib().vlaser_get_cu_budget()


def gemm_tile_config(M, N, splits=1, batch=1, nn=False):
    """Mirror of the tile choice in csrc/gemm.hip `launch<EPI, WKM>` (single-round rule, then least modelled time): (code, BM, BN, rate);
    nn: the NN form (vlaser_gemm_nn), which has no 64x64 and no 32-row configuration."""
    blocks = lambda bm, bn: -(-M // bm) * -(-N // bn) * splits * batch
    if M <= 32 and not nn:
        return (32, 32, 128, 500.0)
    for c in _GEMM_CFGS:
        if c[0] == 1300 and blocks(192, 256) <= _CU_BUDGET and -(-M // 192) * 192 < -(-M // 256) * 256:
            return (1900, 192, 256, 1208.0)
        if blocks(c[1], c[2]) <= _CU_BUDGET and not (nn and c[0] == 1564):
            return c
    return min((c for c in _GEMM_CFGS if c[0] not in (1440, 1564)), key=lambda c: -(-blocks(c[1], c[2]) // _CU_BUDGET) * c[1] * c[2] / c[3])


def split_slab_elems(max_rows, N):
    """fp32 elements of a split-K slab workspace sized ONCE for up to max_rows output rows of width N: 8 slabs for outputs of up to
    1024 rows, at least 2 for any size.  Workspaces must not be re-allocated after a HIP graph has captured their address, so the
    callers allocate this at construction and pass it to `gemm_splits` as the budget."""
    return max(8 * min(max_rows, 1024), 2 * max_rows) * N


# per-launch time model of the LDS-DMA pipelines, fitted on MI355X (profiles/r03e_gemm_lab.md: 24 vs 64 K-steps per configuration):
# a launch costs a fixed ~5-7 us (launch, cold first tiles, epilogue) + a per-K-step time that grows with the tile area
_GEMM_STEP_US = {32: (5.0, 0.30), 1564: (4.9, 0.177), 1500: (5.1, 0.263), 1100: (5.3, 0.377), 1440: (5.6, 0.509), 1200: (7.0, 0.62), 1300: (9.0, 1.55), 1900: (8.5, 1.2)}


_SPLIT_MIN_K = int(os.environ.get('VLASER_GEMM_SPLIT_MINK', '512'))      # >= 8 K-steps per slice: the fitted model does not extrapolate to shorter loops


def gemm_splits(M, N, K, max_elems=None, max_splits=8, nn=False):
    """Split-K factor for a [M,N] output whose tiles alone cannot fill 256 CUs: the factor (K/splits a multiple of 64 and >= 256, at
    most max_splits, splits*M*N fp32 slab elements within max_elems) with the least modelled time = rounds x (fixed + K-steps x step time
    of the tile configuration the kernel will pick) + the consumer's cost of summing the extra fp32 slabs."""
    best = (None, 1)
    for s in range(1, max_splits + 1):
        if K % (s * 64) or (s > 1 and K // s < _SPLIT_MIN_K) or (s > 1 and max_elems is not None and s * M * N > max_elems):
            continue
        code, bm, bn, _ = gemm_tile_config(M, N, s, nn=nn)
        blocks = -(-M // bm) * -(-N // bn) * s
        fixed, step = _GEMM_STEP_US[code]
        # + what the extra fp32 slabs cost their consumer: the seam kernel takes 5.3 / 5.3 / 5.4 us with 1 / 3 / 7 slabs of 2.4 MB, 5.3 / 5.9 with 1 / 4 slabs
        # of 4.2 MB (profiles/r03l_chunk_kernel_stats.md) -- about 0.04 us per MB of slab
        t = -(-blocks // _CU_BUDGET) * (fixed + (K // s // 64) * step) + (s > 1) * 0.04 * s * M * N * 4.0 / 1e6
        if best[0] is None or t < best[0] - 1e-9:
            best = (t, s)
    return best[1]


def cast_f32_bf16(x, y):
    L.check(L.lib().vlaser_cast_f32_bf16(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), 'vlaser_cast_f32_bf16')


_NORM_CONST = {}


def normalize_u8(img_u8, out, mean, std, layout='chw', mode='vla'):
    """uint8 images -> normalised bf16 pixel_values on the device.  img_u8: [N,3,H,W] (layout 'chw') or [N,H,W,3] ('hwc'), contiguous,
    on the GPU; out: bf16 [N,3,H,W].  mode 'vla' = InternVLAProcessor's (u8 * (1/255) - mean) / std, 'totensor' = torchvision's
    (u8 / 255 - mean) / std (see include/vlaser_hip.h)."""
    assert img_u8.dtype == torch.uint8 and img_u8.is_cuda and img_u8.is_contiguous() and img_u8.dim() == 4
    N = img_u8.shape[0]
    H, W = (img_u8.shape[2], img_u8.shape[3]) if layout == 'chw' else (img_u8.shape[1], img_u8.shape[2])
    assert (img_u8.shape[1] if layout == 'chw' else img_u8.shape[3]) == 3
    _chk(out)
    assert out.numel() == N * 3 * H * W
    key = (tuple(mean), tuple(std))
    if key not in _NORM_CONST:
        _NORM_CONST[key] = ((C.c_float * 3)(*mean), (C.c_float * 3)(*std))
    m3, s3 = _NORM_CONST[key]
    L.check(L.lib().vlaser_normalize_u8(img_u8.data_ptr(), out.data_ptr(), N, H * W, 0 if layout == 'chw' else 1, 0 if mode == 'vla' else 1, m3, s3,
                                        _stream()), 'vlaser_normalize_u8')
    return out


def avg_update(avg, p, c, first):
    """avg (fp32) = p on the first update, else avg += (p - avg) * c (EMA / SWA of the fp32 master shard)."""
    assert avg.dtype == torch.float32 and p.dtype == torch.float32 and avg.numel() == p.numel()
    L.check(L.lib().vlaser_avg_update(avg.data_ptr(), p.data_ptr(), avg.numel(), c, int(first), _stream()), 'vlaser_avg_update')


# ------------------------------------------------------------------------------------------------ SFT (backward / optimizer)
_TN_LDS = os.environ.get('VLASER_TN_LDS', '1') == '1'


def _ssq(part):
    """(pointer, capacity) of an optional fp32 slot array for the per-wave sums of squares a weight-gradient GEMM leaves behind."""
    if part is None:
        return None, 0
    assert part.dtype == torch.float32 and part.is_contiguous()
    return part.data_ptr(), part.numel()


def gemm_tn(At, Wt, out, K=None, sumsq_part=None):
    """out[M,N] = At[:K]^T @ Wt[:K] (bf16): At [K,M], Wt [K,N] row-major views (row strides honoured).  `sumsq_part`: see gemm_tn_lds."""
    K = At.shape[0] if K is None else K
    M, N = At.shape[1], Wt.shape[1]
    c8 = lambda n: (n + 7) // 8 * 8
    if K % 64 == 0 and At.stride(0) >= c8(M) and Wt.stride(0) >= c8(N) and At.stride(0) % 8 == 0 and Wt.stride(0) % 8 == 0 \
            and (At.data_ptr() | Wt.data_ptr()) % 16 == 0 and M * N >= 128 * 128 and _TN_LDS:       # ragged M / N: rows must be readable up to the next multiple of 8
        return gemm_tn_lds(At, Wt, out, K, sumsq_part=sumsq_part)          # whole 64-row tiles: nothing to pad, the LDS-DMA pipeline applies as is
    sp, cap = _ssq(sumsq_part)
    L.check(L.lib().vlaser_gemm_tn(At.data_ptr(), Wt.data_ptr(), out.data_ptr(), M, N, K, At.stride(0), Wt.stride(0), out.stride(0), sp, cap, _stream()),
            'vlaser_gemm_tn')
    return out


def gemm_tn_lds(At, Wt, out, K_pad, force_cfg=0, sumsq_part=None):
    """out[M,N] = At[:K_pad]^T @ Wt[:K_pad] on the LDS-DMA pipeline: K_pad a multiple of 64, rows past the true K of At ZERO and of Wt finite.
    `sumsq_part` (fp32 slots, see `tn_sumsq_slots`): every wave of the launch leaves the sum of the squares of the bf16 values it stored in its own
    slot -- the gradient norm's share of this tensor without reading it back (`sum_partials` adds the slots in a fixed order)."""
    M, N = At.shape[1], Wt.shape[1]
    sp, cap = _ssq(sumsq_part)
    L.check(L.lib().vlaser_gemm_tn_lds(At.data_ptr(), Wt.data_ptr(), out.data_ptr(), M, N, K_pad, At.stride(0), Wt.stride(0), out.stride(0), force_cfg, sp, cap,
                                        _stream()), 'vlaser_gemm_tn_lds')
    return out


def tn_sumsq_slots(M, N):
    """Slots that always suffice for `sumsq_part` of a TN GEMM with an [M, N] output, whichever kernel / tile takes it (include/vlaser_hip.h)."""
    c = lambda a, b: (a + b - 1) // b
    return max(c(M, 64) * c(N, 128) * 4, c(M, 128) * c(N, 128) * 8)


def sumsq_chunks(x, tab, part):
    """part[c] = sum of squares of the bf16 buffer x[tab[c, 0] : tab[c, 0] + tab[c, 1]] (tab int64 [n, 2] on the device)."""
    assert tab.dtype == torch.int64 and tab.is_contiguous() and part.numel() >= tab.shape[0]
    L.check(L.lib().vlaser_sumsq_chunks(x.data_ptr(), tab.data_ptr(), tab.shape[0], part.data_ptr(), _stream()), 'vlaser_sumsq_chunks')


def sum_partials(part, out, accumulate=True):
    """out[0] (+)= sum(part) in a fixed association."""
    L.check(L.lib().vlaser_sum_partials(part.data_ptr(), part.numel(), out.data_ptr(), 1 if accumulate else 0, _stream()), 'vlaser_sum_partials')


def gemm_tn_grouped(At, Wt, out, M, N, K, ldat, ldwt, ldo, groups, a_gs, w_gs, batch, a_bs, w_bs, o_bs):
    """out[b] = sum_g At[b,g]^T @ Wt[b,g] (raw pointers + element strides; see include/vlaser_hip.h)."""
    L.check(L.lib().vlaser_gemm_tn_grouped(At.data_ptr(), Wt.data_ptr(), out.data_ptr(), M, N, K, ldat, ldwt, ldo, groups, a_gs, w_gs, batch,
                                           a_bs, w_bs, o_bs, _stream()), 'vlaser_gemm_tn_grouped')


def transpose(x, out, rows, cols, ld_in, ld_out, pad_rows=None, batch=1, in_bs=0, out_bs=0, inner=1, in_is=0, out_is=0):
    L.check(L.lib().vlaser_transpose(x.data_ptr(), out.data_ptr(), rows, cols, ld_in, ld_out, ld_out if pad_rows is None else pad_rows, batch,
                                     in_bs, out_bs, inner, in_is, out_is, _stream()), 'vlaser_transpose')


def rope_bwd_pack(dq, dk, dv, cos, sin, pos, out, S, n_q, n_kv, kv_per_q_head=False):
    L.check(L.lib().vlaser_rope_bwd_pack(dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), cos.data_ptr(), sin.data_ptr(), pos.data_ptr(),
                                         out.data_ptr(), S, n_q, n_kv, 1 if kv_per_q_head else 0, _stream()), 'vlaser_rope_bwd_pack')


def rmsnorm_bwd(dy, x, w, dres, dx, S, Cc, eps, dw_out=None, dw_ws=None, dy_partials=None, n_partials=0):
    """dx = dres + RMSNorm backward; with dw_out (bf16 [C]) also the weight gradient (dw_ws: fp32 [ceil(S/4) * C] scratch).  `dy_partials` (fp32, >= n_partials * S * C):
    dy comes as the split-K slabs of the dgrad GEMM before (summed + rounded in the kernel exactly as reduce_norm would; `dy` may be None)."""
    L.check(L.lib().vlaser_rmsnorm_bwd(_p(dy), x.data_ptr(), w.data_ptr(), _p(dres), dx.data_ptr(), _p(dw_out), _p(dw_ws), S, Cc, eps,
                                       _p(dy_partials), n_partials if dy_partials is not None else 0, _stream()), 'vlaser_rmsnorm_bwd')


def colsum_partials_multi(ws, slot_stride, n_tensors, n_part, Cc, out_base, out_off):
    """Finish n_tensors norm-weight gradients (partials left by rmsnorm_bwd(dw_out=None, dw_ws=slot)) in one launch; out_off: device int64 element offsets into out_base."""
    assert out_off.dtype == torch.int64 and out_off.numel() >= n_tensors
    L.check(L.lib().vlaser_colsum_partials_multi(ws.data_ptr(), slot_stride, n_tensors, n_part, Cc, out_base.data_ptr(), out_off.data_ptr(), _stream()),
            'vlaser_colsum_partials_multi')


def colsum_bf16(a, out, S, Cc):
    L.check(L.lib().vlaser_colsum_bf16(a.data_ptr(), out.data_ptr(), S, Cc, a.stride(0), _stream()), 'vlaser_colsum_bf16')


def colsum_mul(a, b, out, S, Cc, mode=0, eps=1e-6, ws=None):
    L.check(L.lib().vlaser_colsum_mul(a.data_ptr(), _p(b), out.data_ptr(), S, Cc, mode, eps, _p(ws), _stream()), 'vlaser_colsum_mul')


def swiglu(gu, act, S, I):
    L.check(L.lib().vlaser_swiglu(gu.data_ptr(), act.data_ptr(), S, I, _stream()), 'vlaser_swiglu')


def swiglu_bwd(gu, dact, dgu, S, I):
    L.check(L.lib().vlaser_swiglu_bwd(gu.data_ptr(), dact.data_ptr(), dgu.data_ptr(), S, I, _stream()), 'vlaser_swiglu_bwd')


def ce_rows(logits, labels, loss_rows, lse_rows, ignore_index=-100):
    R, V = logits.shape
    L.check(L.lib().vlaser_ce_rows(logits.data_ptr(), labels.data_ptr(), R, V, logits.stride(0), loss_rows.data_ptr(), _p(lse_rows),
                                   ignore_index, _stream()), 'vlaser_ce_rows')


def ce_dlogits(logits, lse, labels, out, scale, ignore_index=-100):
    R, V = logits.shape
    L.check(L.lib().vlaser_ce_dlogits(logits.data_ptr(), lse.data_ptr(), labels.data_ptr(), out.data_ptr(), R, V, logits.stride(0), out.stride(0),
                                      scale, ignore_index, _stream()), 'vlaser_ce_dlogits')


def embed_scatter_add(ids, rank, dh, dembed, n, H, sumsq_part=None):
    """dEmbed[id] += fp32 sum of dh over the text positions holding id (duplicates summed before the single bf16 rounding).  `sumsq_part` (fp32, >= n
    slots): afterwards slot i = the sum of squares of the table row of the i-th distinct id (0 elsewhere) -- the rows this step touched; with the
    table gradient cleared at the start of the step that is its whole share of the gradient norm."""
    order = torch.sort(ids.reshape(-1)[:n], stable=True).indices.to(torch.int32)          # index bookkeeping only: equal ids -> contiguous runs
    L.check(L.lib().vlaser_embed_scatter_add(ids.data_ptr(), rank.data_ptr(), order.data_ptr(), dh.data_ptr(), dembed.data_ptr(), n, H,
                                             dembed.shape[0], _stream()), 'vlaser_embed_scatter_add')
    if sumsq_part is not None:
        sp, cap = _ssq(sumsq_part)
        L.check(L.lib().vlaser_sumsq_rows(ids.data_ptr(), order.data_ptr(), dembed.data_ptr(), n, H, dembed.shape[0], sp, cap, _stream()), 'vlaser_sumsq_rows')


def gelu_bwd(x, dy, dx):
    L.check(L.lib().vlaser_gelu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), _stream()), 'vlaser_gelu_bwd')


def adamw(param, master, m, v, grad, lr, beta1, beta2, eps, wd, gscale, step):
    L.check(L.lib().vlaser_adamw(param.data_ptr(), master.data_ptr(), m.data_ptr(), v.data_ptr(), grad.data_ptr(), param.numel(), lr, beta1, beta2,
                                 eps, wd, gscale, step, _stream()), 'vlaser_adamw')


def adamw_clipped(param, master, m, v, grad, lr, beta1, beta2, eps, wd, gscale, gnorm2, max_norm, step):
    L.check(L.lib().vlaser_adamw_clipped(param.data_ptr(), master.data_ptr(), m.data_ptr(), v.data_ptr(), grad.data_ptr(), param.numel(), lr, beta1,
                                         beta2, eps, wd, gscale, gnorm2.data_ptr(), max_norm or 0.0, step, _stream()), 'vlaser_adamw_clipped')


def grad_accumulate(g, acc, w, first, finalize):
    L.check(L.lib().vlaser_grad_accumulate(g.data_ptr(), acc.data_ptr(), g.numel(), w, int(first), int(finalize), _stream()), 'vlaser_grad_accumulate')


def sumsq(x, out, ws):
    L.check(L.lib().vlaser_sumsq(x.data_ptr(), x.numel(), out.data_ptr(), ws.data_ptr(), _stream()), 'vlaser_sumsq')


# ------------------------------------------------------------------------------------------------ VLA flow-matching training step
def silu(x, y):
    L.check(L.lib().vlaser_silu(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), 'vlaser_silu')


def silu_bwd(x, dy, dx):
    L.check(L.lib().vlaser_silu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), _stream()), 'vlaser_silu_bwd')


def attn_rows_bwd(q, K, VT, dO, O, dq, dk, dv, R, n_q, n_kv, s_max, valid_len, blk_start, first_tok_self, scale, p_out=None, ds_out=None, ws=None):
    if ws is None:         # (a trainer passes its own; stream-ordered scratch otherwise)
        ws = torch.empty(L.lib().vlaser_attn_rows_bwd_ws_floats(n_q), dtype=torch.float32, device=q.device)
    assert ws.dtype == torch.float32 and ws.numel() >= L.lib().vlaser_attn_rows_bwd_ws_floats(n_q)
    L.check(L.lib().vlaser_attn_rows_bwd_ex(q.data_ptr(), K.data_ptr(), VT.data_ptr(), dO.data_ptr(), O.data_ptr(), dq.data_ptr(), dk.data_ptr(),
                                            dv.data_ptr(), R, n_q, n_kv, s_max, valid_len, blk_start, int(first_tok_self), scale, _p(p_out), _p(ds_out),
                                            ws.data_ptr(), _stream()), 'vlaser_attn_rows_bwd')


# ------------------------------------------------------------------------------------------------ f1 with train_vlm: VLM-side backward
def attn_bwd_pds_masked(scores, dP, dO, O, P, dS, H, S, ld, hd, scale, causal, kv_valid, q_off=0):
    L.check(L.lib().vlaser_attn_bwd_pds_masked(scores.data_ptr(), dP.data_ptr(), dO.data_ptr(), O.data_ptr(), P.data_ptr(), dS.data_ptr(), H, S, ld, hd,
                                               scale, int(causal), kv_valid, q_off, _stream()), 'vlaser_attn_bwd_pds_masked')


def rope_bwd_pack_ex(dq, dk, dv, cos, sin, pos, out, S, n_q, n_kv, kv_per_q_head=False, dk_extra=None, dv_extra=None):
    L.check(L.lib().vlaser_rope_bwd_pack_ex(dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), cos.data_ptr(), sin.data_ptr(), pos.data_ptr(), out.data_ptr(), S,
                                            n_q, n_kv, 1 if kv_per_q_head else 0, _p(dk_extra), _p(dv_extra), _stream()), 'vlaser_rope_bwd_pack_ex')


def layernorm_bwd(dy, x, w, dres, dx, S, Cc, eps):
    L.check(L.lib().vlaser_layernorm_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), _p(dres), dx.data_ptr(), S, Cc, eps, _stream()), 'vlaser_layernorm_bwd')


def scale_cols(x, vec, out, S, Cc, alpha=1.0, ldx=None, ldo=None):
    L.check(L.lib().vlaser_scale_cols(x.data_ptr(), _p(vec), out.data_ptr(), S, Cc, x.stride(0) if ldx is None else ldx, out.stride(0) if ldo is None else ldo,
                                      alpha, _stream()), 'vlaser_scale_cols')


def pixel_unshuffle(dout, dx, T, G, Cc, ps_v1=0):
    assert dout.numel() >= T * (G // 2) ** 2 * 4 * Cc and dx.numel() >= T * (G * G + 1) * Cc
    L.check(L.lib().vlaser_pixel_unshuffle(dout.data_ptr(), dx.data_ptr(), T, G, Cc, ps_v1, _stream()), 'vlaser_pixel_unshuffle')
