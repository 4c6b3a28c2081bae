"""Kernel sequencing for the three hot loops: InternViT tile encoder, Qwen2.5 prefill / greedy
decode, and the pi0 joint prefill + 10-step Euler sampler.  Every arithmetic step is a launch of a gfx950 kernel
from libvlaser_hip.so through vlaser_amd.ops; PyTorch only owns buffers and the stream.  All launches are
asynchronous and allocation-free after construction, so whole phases can be captured in one HIP graph.

Weights are stored in HBM in the layouts the kernels want (packed once at load):
  * q/k/v fused into one [N,K] matrix with in-head row permutation (RoPE pair in one MFMA lane),
  * gate/up interleaved in 16-row groups (SwiGLU is lane-local),
  * patch-embedding conv flattened to [1024, 640] (K zero-padded from 588 to a multiple of 64),
  * K cache [L][B,n_kv,S_max,128], V cache TRANSPOSED [L][B,n_kv,128,S_max] (MFMA A-fragment = one 16-byte load).
"""

import os

import torch
from types import SimpleNamespace

from . import _lib as L
from . import ops
from .config import LLMConfig, VlaserConfig

BF = torch.bfloat16


def _dev(t, device):
    return t.to(device=device, dtype=BF).contiguous()


class QwenLayerWeights:
    """One Qwen2DecoderLayer in kernel layout: row-major packed matrices for the MFMA GEMM (prefill) and/or
    fragment-major copies for the weight-streaming skinny kernel (decode / action tokens)."""

    def __init__(self, sd, p, llm: LLMConfig, device, ks_o, ks_down, gemm=True, skinny=True, tpu_down=2, tpu_o=2, i_pad=None, opts=()):
        g = lambda k: _dev(sd[p + k], device)
        wqkv, self.bqkv = ops.pack_qkv(g('self_attn.q_proj.weight'), g('self_attn.k_proj.weight'),
                                       g('self_attn.v_proj.weight'), g('self_attn.q_proj.bias'),
                                       g('self_attn.k_proj.bias'), g('self_attn.v_proj.bias'), llm.head_dim)
        wo = g('self_attn.o_proj.weight')
        wgu = ops.pack_gate_up(g('mlp.gate_proj.weight'), g('mlp.up_proj.weight'))
        wdown = g('mlp.down_proj.weight')
        self.ln_in = g('input_layernorm.weight')
        self.ln_post = g('post_attention_layernorm.weight')
        if gemm:
            self.wqkv, self.wo, self.wgu, self.wdown = wqkv, wo, wgu, wdown
        if skinny:
            self.bqkv_sk = self.bqkv
            if 'qkv16' in opts:           # 16-row lane-local units (r03): 2x the workgroups on the q/k/v GEMV
                w16, self.bqkv_sk = ops.pack_qkv16(g('self_attn.q_proj.weight'), g('self_attn.k_proj.weight'), g('self_attn.v_proj.weight'),
                                                   g('self_attn.q_proj.bias'), g('self_attn.k_proj.bias'), g('self_attn.v_proj.bias'), llm.head_dim)
                self.sk_qkv = ops.pack_skinny(w16, 1, 1)
            else:
                self.sk_qkv = ops.pack_skinny(wqkv, 1)
            self.sk_o = ops.pack_skinny(wo, ks_o, tpu_o)
            # wide output + short K (action expert: 17920 x 768): 96-row units (tpu = 6) measured slower on MI355X (12.3 vs 11.2 us): kept in
            # the kernel, not used; 16-row lane-local units ('gu16', r03) balance 1120 units over 256 workgroups (80 vs 96 rows on the longest)
            if 'gu16' in opts:
                self.sk_gu = ops.pack_skinny(ops.pack_gate_up8(g('mlp.gate_proj.weight'), g('mlp.up_proj.weight')), 1, 1)
            else:
                self.sk_gu = ops.pack_skinny(wgu, 1, 2)
            self.sk_down = ops.pack_skinny(wdown, ks_down, tpu_down, k_pad=i_pad)
            # 'chain' (r05, csrc/chain.hip): the down projection without cross-workgroup split-K (4 output columns per workgroup over the whole K) publishes the next
            # layer's residual stream as bf16; needs the 16-row q/k/v packing and the MLP width the kernel is built for
            ok4 = 'chain' in opts and 'qkv16' in opts and wdown.shape[1] == ops.DOWN4_WAVES * ops.DOWN4_LOADS * 128
            self.sk_down4 = ops.pack_down4(wdown) if (ok4 and (wdown.shape[0] % 3 == 0 or wdown.shape[0] % 4 == 0)) else None
            # 'down2' (with 'chain'): the down projection as two K halves on 256 six-column workgroups leaving two fp32 slabs (vlaser_chain_down2) that the q/k/v launch reduces
            self.sk_down42 = ops.pack_down4(wdown, k_splits=2) if (ok4 and 'down2' in opts and (wdown.shape[0] % 6 == 0 or wdown.shape[0] % 8 == 0)) else None


class QwenStack:
    """Weights + geometry of one Qwen2 decoder stack (the VLM LLM or the action expert)."""

    def __init__(self, sd, prefix, llm: LLMConfig, device, with_embed=True, with_head=True, gemm=True, skinny=True, opts=()):
        self.llm = llm
        self.opts = tuple(opts)
        H, I = llm.hidden_size, llm.intermediate_size
        nqd = llm.num_attention_heads * llm.head_dim
        # o_proj on 16-row units where the kernel has the variant: the same ~144 workgroups with half the split-K slabs for the
        # gate/up prologue to sum (chunk time unchanged, the dominant gate/up GEMV 8.9 -> 8.6 us; same-box A/B)
        ks16o = ops.pick_k_splits(nqd, H, rows_per_unit=16)
        self.tpu_o = 1 if (ks16o is not None and nqd // (ks16o * 256) in (2, 3)) else 2
        self.ks_o = ks16o if self.tpu_o == 1 else ops.skinny_geometry(nqd, H)[1]
        # down_proj (narrow output, long K): 16-row units double the workgroups that stream it (48 units x 5 splits = 240 for the expert)
        # when the kernel has that variant (5 or 7 K-steps per wave); measured +0.9 % chunks/s over 32-row units x 7 splits
        ks16 = ops.pick_k_splits(I, H, rows_per_unit=16)
        self.tpu_down = 1 if (ks16 is not None and I // (ks16 * 256) in (5, 7)) else 2
        self.I_pad, ks32 = ops.skinny_geometry(I, H)            # widths that do not factor (Vlaser-8B: 18944) are zero-padded: act buffer + down weight
        self.ks_down = ks16 if self.tpu_down == 1 else ks32
        self.nqd = nqd
        self.layers = [QwenLayerWeights(sd, f'{prefix}model.layers.{i}.', llm, device, self.ks_o, self.ks_down, gemm, skinny, self.tpu_down, self.tpu_o, self.I_pad, self.opts)
                       for i in range(llm.num_hidden_layers)]
        self.norm = _dev(sd[prefix + 'model.norm.weight'], device)
        self.embed = _dev(sd[prefix + 'model.embed_tokens.weight'], device) if with_embed else None
        self.head = _dev(sd[prefix + 'lm_head.weight'], device) if with_head and (prefix + 'lm_head.weight') in sd else None
        self.sk_head = ops.pack_skinny(self.head, 1) if (self.head is not None and skinny) else None

    @property
    def nq(self):
        return self.llm.num_attention_heads

    @property
    def nkv(self):
        return self.llm.num_key_value_heads


class KVCache:
    """K [L][B,n_kv,S_max,128] and V^T [L][B,n_kv,128,S_max], zero-initialised (padding must stay finite)."""

    def __init__(self, n_layers, batch, n_kv, s_max, device, head_dim=128):
        assert s_max % 64 == 0
        self.s_max, self.batch, self.n_kv, self.hd = s_max, batch, n_kv, head_dim
        self.k = torch.zeros(n_layers, batch, n_kv, s_max, head_dim, dtype=BF, device=device)
        self.vt = torch.zeros(n_layers, batch, n_kv, head_dim, s_max, dtype=BF, device=device)

    def strides(self):
        return (self.n_kv * self.s_max * self.hd, self.s_max * self.hd), (self.n_kv * self.hd * self.s_max, self.hd * self.s_max)


# ------------------------------------------------------------------------------------------------------ ViT
class VitEngine:
    """InternViT-300M + pixel_shuffle + mlp1 (modeling_intern_vit.py:133-431, modeling_internvl_chat.py:257-291)."""

    KPAD = 640

    def __init__(self, sd, cfg: VlaserConfig, device, max_tiles=1):
        v = cfg.vision
        self.cfg, self.v, self.device = cfg, v, device
        g = lambda k: _dev(sd[k], device)
        e = 'vision_model.embeddings.'
        self.w_pe = ops.pack_patch_embed(g(e + 'patch_embedding.weight'), self.KPAD)
        self.b_pe = g(e + 'patch_embedding.bias')
        self.cls = g(e + 'class_embedding').reshape(-1)
        self.pos = g(e + 'position_embedding').reshape(v.num_positions, v.hidden_size)
        self.layers = []
        for i in range(v.num_hidden_layers):
            p = f'vision_model.encoder.layers.{i}.'
            self.layers.append({k: g(p + n) for k, n in [
                ('wqkv', 'attn.qkv.weight'), ('bqkv', 'attn.qkv.bias'), ('wproj', 'attn.proj.weight'), ('bproj', 'attn.proj.bias'),
                ('wfc1', 'mlp.fc1.weight'), ('bfc1', 'mlp.fc1.bias'), ('wfc2', 'mlp.fc2.weight'), ('bfc2', 'mlp.fc2.bias'),
                ('n1w', 'norm1.weight'), ('n1b', 'norm1.bias'), ('n2w', 'norm2.weight'), ('n2b', 'norm2.bias'),
                ('ls1', 'ls1'), ('ls2', 'ls2')]})
        self.m0w, self.m0b = g('mlp1.0.weight'), g('mlp1.0.bias')
        self.m1w, self.m1b = g('mlp1.1.weight'), g('mlp1.1.bias')
        self.m3w, self.m3b = g('mlp1.3.weight'), g('mlp1.3.bias')
        self.max_tiles = 0
        self._alloc(max_tiles)

    def _alloc(self, T):
        if T <= self.max_tiles:
            return
        v, dev = self.v, self.device
        S, C, Hn = v.num_positions, v.hidden_size, v.num_attention_heads
        self.s_pad = (S + 63) // 64 * 64
        z = lambda *s: torch.zeros(*s, dtype=BF, device=dev)
        self.col = z(T * v.num_patches, self.KPAD)
        self.patch = z(T * v.num_patches, C)
        self.h = z(T * S, C)
        self.x = z(T * S, C)
        self.q = z(T, Hn, self.s_pad, v.head_dim)
        self.k = z(T, Hn, self.s_pad, v.head_dim)
        self.vt = z(T, Hn, v.head_dim, self.s_pad)
        self.ao = z(T * S, C)
        self.f = z(T * S, v.intermediate_size)
        n_tok = T * self.cfg.num_image_token
        self.psln = z(n_tok, 4 * C)
        self.g = z(n_tok, self.cfg.llm.hidden_size)
        self.feat = z(n_tok, self.cfg.llm.hidden_size)
        self.part = torch.zeros(ops.split_slab_elems(T * S, C), dtype=torch.float32, device=dev)      # split-K slabs of proj / fc2
        self.max_tiles = T

    def forward(self, pixel_values, return_layers=False, project=True):
        """pixel_values bf16 [T,3,448,448] -> projected visual tokens bf16 [T*256, H_llm] (view into a workspace).  `project=False` stops after the
        encoder (last hidden state in `self.h`): the SFT step runs the trainable projector itself, keeping the intermediates its backward needs."""
        v, cfg = self.v, self.cfg
        T = pixel_values.shape[0]
        assert pixel_values.dtype == BF and pixel_values.is_contiguous() and pixel_values.shape[1:] == (3, v.image_size, v.image_size)
        self._alloc(T)
        S, C, Hn, hd, sp = v.num_positions, v.hidden_size, v.num_attention_heads, v.head_dim, self.s_pad
        M = T * S
        h, x, ao, f = self.h[:M], self.x[:M], self.ao[:M], self.f[:M]
        ops.im2col(pixel_values, self.col, T, v.image_size, self.KPAD)
        ops.linear(self.col[:T * v.num_patches], self.w_pe, self.b_pe, out=self.patch[:T * v.num_patches])
        ops.vit_assemble(self.patch, self.cls, self.pos, h, T, v.num_patches, C)
        layers_out = []
        if return_layers:
            layers_out.append(h.clone())
        # split-K factors for the two N = C outputs (proj, fc2): M*C/128^2 tiles alone cannot fill 256 CUs
        sp_proj, sp_fc2 = ops.gemm_splits(M, C, C, self.part.numel()), ops.gemm_splits(M, C, v.intermediate_size, self.part.numel())
        nl = len(self.layers)
        ops.layernorm(h, self.layers[0]['n1w'], self.layers[0]['n1b'], v.layer_norm_eps, out=x)
        for li, lw in enumerate(self.layers):
            ops.gemm(L.EPI_VIT_QKV, x, lw['wqkv'], bias=lw['bqkv'], vq=self.q, vk=self.k, vvt=self.vt, vit_heads=Hn, vit_seq=S,
                     vit_seq_pad=sp, q_scale=hd ** -0.5)
            ops.attn_prefill(self.q, self.k, self.vt, ao, T, S, S, Hn, Hn, hd, (Hn * sp * hd, sp * hd, hd), (Hn * sp * hd, sp * hd),
                             (Hn * hd * sp, hd * sp), (S * C, C), sp, 1.0, L.ATTN_FULL)
            # h += ls1 * (proj(ao) + b); x = LN2(h)      (modeling_intern_vit.py:291)
            residual_seam(ao, lw['wproj'], h, self.part, M, C, C, x, bias=lw['bproj'], ls=lw['ls1'], norm=2, norm_w=lw['n2w'], norm_b=lw['n2b'],
                          eps=v.layer_norm_eps, splits=sp_proj)
            ops.gemm(L.EPI_BIAS_GELU, x, lw['wfc1'], out=f, bias=lw['bfc1'])
            # h += ls2 * (fc2(f) + b); x = LN1 of the NEXT layer (:293)
            nxt = self.layers[li + 1] if li + 1 < nl else None
            residual_seam(f, lw['wfc2'], h, self.part, M, C, v.intermediate_size, x if nxt else None, bias=lw['bfc2'], ls=lw['ls2'], norm=2 if nxt else 0,
                          norm_w=nxt['n1w'] if nxt else None, norm_b=nxt['n1b'] if nxt else None, eps=v.layer_norm_eps, splits=sp_fc2)
            if return_layers:
                layers_out.append(h.clone())
        if not project:
            return layers_out if return_layers else None
        n_tok = T * cfg.num_image_token
        G = v.image_size // v.patch_size
        ops.pixel_shuffle_ln(h, self.m0w, self.m0b, self.psln, T, G, C, 1e-5, 1 if cfg.ps_version == 'v1' else 0)
        ops.gemm(L.EPI_BIAS_GELU, self.psln[:n_tok], self.m1w, out=self.g[:n_tok], bias=self.m1b)
        ops.gemm(L.EPI_BIAS, self.g[:n_tok], self.m3w, out=self.feat[:n_tok], bias=self.m3b)
        if return_layers:
            return self.feat[:n_tok], layers_out
        return self.feat[:n_tok]


def residual_seam(a, w, h, part, M, N, K, x_out=None, bias=None, ls=None, norm=0, norm_w=None, norm_b=None, eps=1e-6, splits=None):
    """The seam behind an N = hidden GEMM: h += [ls *] (a @ w^T [+ bias]); x_out = norm(h) (norm: 0 none / 1 RMS / 2 LayerNorm).  Two compositions of the same arithmetic
    and the same rounding points (h rounded to bf16 once, the norm taken of the rounded h):
      * few row tiles, long K (the output tiles alone cannot fill 256 CUs): split-K GEMM -> fp32 slabs -> ONE fused reduce + residual + norm launch;
      * ONE slab, or two slabs of >= 2048 rows: whole-K GEMM with the residual in its epilogue + a stand-alone norm launch.  Measured (r06, tools/micro/seam_ab.py,
        profiles/r06x_seam_ab.md): 9-13 us less for the 13-tile ViT proj / fc2 (1 slab), 11 us for the 8B o_proj (1 slab), 96 us for the 8B down_proj at S = 3408 (2 slabs):
        with one or two large slabs the fp32 slab round trip (2 x 49 MB at 3408 x 3584) costs more than the split buys.  One slab is bit-identical to the slab path; two differ in
        the fp32 summation order.  (The one-tile ViT proj, 2 slabs of 1025 rows, is 3 us faster as a pair in isolation and EQUAL inside the chunk -- same-box A/B 11.775 vs
        11.778 ms -- so the small shapes keep the slab path and their bit-exact history.)"""
    sp = ops.gemm_splits(M, N, K, part.numel()) if splits is None else splits
    if sp >= 3 or (sp == 2 and M < 2048) or os.environ.get('VLASER_SEAM_SPLITK') == '1':
        ops.gemm(L.EPI_PARTIAL, a, w, out_f32=part, k_splits=sp)
        ops.reduce_norm(h, part, sp, M, N, h, x_out, bias=bias, ls=ls, norm=norm if x_out is not None else 0, norm_w=norm_w, norm_b=norm_b, eps=eps)
        return
    if ls is not None:
        ops.gemm(L.EPI_BIAS_LS_RES, a, w, out=h, bias=bias, res=h, ls=ls)
    else:
        assert bias is None
        ops.gemm(L.EPI_RES, a, w, out=h, res=h)
    if x_out is not None and norm == 2:
        ops.layernorm(h, norm_w, norm_b, eps, out=x_out)
    elif x_out is not None and norm == 1:
        ops.rmsnorm(h, norm_w, eps, out=x_out)


# ------------------------------------------------------------------------------------------------------ LLM
class PrefillBuffers:
    def __init__(self, stack: QwenStack, max_rows, device):
        llm = stack.llm
        z = lambda *s: torch.zeros(*s, dtype=BF, device=device)
        self.x = z(max_rows, llm.hidden_size)
        self.q = z(max_rows, stack.nq * llm.head_dim)
        self.ao = z(max_rows, stack.nq * llm.head_dim)
        self.act = z(max_rows, llm.intermediate_size)
        self.max_rows = max_rows
        # split-K slabs of o_proj / down_proj: sized once (a HIP graph may have captured the address)
        self.part = torch.zeros(ops.split_slab_elems(max_rows, llm.hidden_size), dtype=torch.float32, device=device)
        self.device = device


def prefill_begin(stack: QwenStack, buf: PrefillBuffers, h, M):
    """x = input_layernorm_0(h): the only stand-alone norm launch of a prefill (later norms are fused into the split-K seam)."""
    ops.rmsnorm(h, stack.layers[0].ln_in, stack.llm.rms_norm_eps, out=buf.x[:M])


def prefill_layer(stack: QwenStack, lw: QwenLayerWeights, buf: PrefillBuffers, h, cache: KVCache, layer, rope, pos_ids, batch,
                  tok_per_batch, attn_mode, valid_len=None, blk_start=0, causal_off=0, kv_len=None, skip_post_attn=False,
                  next_norm_w=None, slot_base=0, dense_mask=None):
    """One Qwen2DecoderLayer over M = batch*tok_per_batch rows with the big-GEMM kernels; K/V written to slots
    [slot_base, slot_base + tok_per_batch) of the cache (slot_base > 0: decode steps of models too wide for the
    weight-streaming kernels).  Expects buf.x = input_layernorm(h); leaves buf.x = next_norm(h_out) when
    next_norm_w is given (next layer's input_layernorm or the final norm).  h is updated in place.
    o_proj / down_proj run split-K (their [M, H] outputs have too few tiles to fill 256 CUs); the fp32 slabs are
    reduced by ONE fused kernel that also adds the residual and applies the following RMSNorm."""
    llm = stack.llm
    M = batch * tok_per_batch
    nq, nkv, hd = stack.nq, stack.nkv, llm.head_dim
    H, I = llm.hidden_size, llm.intermediate_size
    x, q, ao, act = buf.x[:M], buf.q[:M], buf.ao[:M], buf.act[:M]
    ops.gemm(L.EPI_QKV_ROPE, x, lw.wqkv, bias=lw.bqkv, q_out=q, k_cache=cache.k[layer], vt_cache=cache.vt[layer], rope_cos=rope[0],
             rope_sin=rope[1], pos_ids=pos_ids, n_q_heads=nq, n_kv_heads=nkv, s_max=cache.s_max, tok_per_batch=tok_per_batch, slot_base=slot_base)
    if skip_post_attn:          # the last layer of a mixture whose hidden states nobody reads (the VLM / proprio rows of the VLA): only its K / V are needed
        return
    ks, vs = cache.strides()
    ops.attn_prefill(q, cache.k[layer], cache.vt[layer], ao, batch, tok_per_batch, tok_per_batch if kv_len is None else kv_len, nq, nkv, hd,
                     (tok_per_batch * nq * hd, hd, nq * hd), ks, vs, (tok_per_batch * nq * hd, nq * hd), cache.s_max, hd ** -0.5,
                     L.ATTN_DENSE if dense_mask is not None else attn_mode, causal_off=causal_off, valid_len=valid_len, blk_start=blk_start, q_row_off=slot_base,
                     dense_mask=dense_mask)        # dense_mask: fp32 view [B, tok_per_batch, >= kv_len] of the reference's additive mask (general masks, ABI 8)
    part = buf.part
    sp_o, sp_d = ops.gemm_splits(M, H, nq * hd, part.numel()), ops.gemm_splits(M, H, I, part.numel())
    residual_seam(ao, lw.wo, h, part, M, H, nq * hd, x, norm=1, norm_w=lw.ln_post, eps=llm.rms_norm_eps, splits=sp_o)
    ops.gemm(L.EPI_SWIGLU, x, lw.wgu, out=act)
    residual_seam(act, lw.wdown, h, part, M, H, I, x if next_norm_w is not None else None, norm=1 if next_norm_w is not None else 0, norm_w=next_norm_w,
                  eps=llm.rms_norm_eps, splits=sp_d)


class SkinnyBuffers:
    """Workspace of the M <= 16 weight-streaming path (decode / proprio row / action tokens)."""

    def __init__(self, stack: QwenStack, max_rows, device):
        llm = stack.llm
        H, I = llm.hidden_size, llm.intermediate_size
        z = lambda *s: torch.zeros(*s, dtype=BF, device=device)
        self.hA, self.hB, self.hC = z(max_rows, H), z(max_rows, H), z(max_rows, H)      # hC: the residual stream as `vlaser_chain_down` publishes it
        self.q = z(max_rows, stack.nq * llm.head_dim)
        self.ao = z(max_rows, stack.nq * llm.head_dim)
        self.act = z(max_rows, getattr(stack, 'I_pad', I))      # zero padding columns (never written) feed the zero-padded down weight
        self.attn_parts = ops.attn_partial_buffers(max_rows, stack.nkv, device)
        self.chain_parts = ops.chain_attn_buffers(max_rows, stack.nkv, device)          # r05: (m, l) pairs + normalised bf16 rows of vlaser_chain_attn
        self.part_o = torch.zeros(stack.ks_o, max_rows, H, dtype=torch.float32, device=device)
        self.part_d = torch.zeros(stack.ks_down, max_rows, H, dtype=torch.float32, device=device)
        self.plans = {}      # cached launch argument structs of skinny_layer


def skinny_layer(stack: QwenStack, lw: QwenLayerWeights, sb: SkinnyBuffers, h_in, partials, n_partials, cache: KVCache, layer, rope,
                 pos_ids, batch, tok_per_batch, slot_base, kv_len, attn_mode, valid_len=None, blk_start=0, skip_post_attn=False,
                 first_tok_kv_len=0, skip=(), dense_mask=None):
    """One decoder layer over M = batch*tok_per_batch <= 16 rows with the weight-streaming kernels (5 launches).
    Input residual = h_in + sum(partials) (partials = down_proj slabs of the previous layer).  Returns
    (h, partials, n_partials) describing this layer's output residual the same way."""
    llm = stack.llm
    M = batch * tok_per_batch
    nq, nkv, hd = stack.nq, stack.nkv, llm.head_dim
    # the five argument structs of this (layer, input buffers, geometry) are built once and re-launched: only the cache slot,
    # the key count and its split factor change from step to step (greedy decode is otherwise host-bound on struct building)
    # (the two fused launches of r03 / r04 -- o_proj -> gate/up with an in-launch hand-off, attention + o_proj in one launch -- lost to the r05 chain kernels below and were
    # removed from the library: DESIGN.md section 3)
    nsp = ops.attn_splits(kv_len)
    # 'chain' (r05): qkv / gate-up / down on the latency-built kernels of csrc/chain.hip -- the layer takes a residual stream that is already reduced (n_partials == 0:
    # the action encoder's output, or what the previous layer's vlaser_chain_down published) and hands on (hC, None, 0)
    H_, I_ = llm.hidden_size, llm.intermediate_size
    down2 = getattr(lw, 'sk_down42', None) is not None and ops.chain_down2_supported(M, H_, I_)
    from_down2 = n_partials == 2 and partials is not None and partials.data_ptr() == sb.part_d.data_ptr() and down2
    chain = ('chain' in stack.opts and getattr(lw, 'sk_down4', None) is not None and lw.sk_qkv.tpu == 1
             and ((n_partials == 0 and h_in.data_ptr() != sb.hB.data_ptr()) or from_down2)
             and ops.chain_qkv_supported(M, lw.sk_qkv.N, H_) and ops.chain_gu_supported(M, lw.sk_gu.N, H_, stack.ks_o, lw.sk_gu.tpu)
             and ops.chain_down_supported(M, H_, I_) and sb.act.shape[1] == I_)
    # ... and attention + o_proj on the one-wave-per-split attention / bf16-partial merge pair when the key count gives the split count they are built for
    nsp2 = ops.chain_attn_splits(kv_len)
    chain_ao = (chain and not skip_post_attn and lw.sk_o.tpu == 1 and tok_per_batch * (nq // nkv) <= 32
                and ops.chain_oproj_supported(M, lw.sk_o.N, nq * hd, stack.ks_o, nsp2, nq // nkv) and 'chain_noao' not in stack.opts)
    if dense_mask is not None:
        # general additive masks (ABI 8) are served by the chain attention only (every <= 16-row launch of the VLA at its shipped widths)
        if not (chain_ao or (chain and skip_post_attn)):
            raise NotImplementedError('dense additive masks need the chain attention (hidden 768 / 1536, <= 16 rows, group * tokens <= 32)')
        attn_mode = L.ATTN_DENSE
    key = (layer, M, tok_per_batch, attn_mode, h_in.data_ptr(), 0 if partials is None else partials.data_ptr(), n_partials,
           0 if valid_len is None else valid_len.data_ptr(), pos_ids.data_ptr(), cache.k.data_ptr(), skip_post_attn, first_tok_kv_len,
           chain, chain_ao, down2, 0 if dense_mask is None else dense_mask.data_ptr())
    plan = sb.plans.get(key)
    if plan is None:
        ks, vs = cache.strides()
        plan = SimpleNamespace()
        plan.qkv = ops.skinny_args(h_in, lw.sk_qkv, M, partials=partials, n_partials=n_partials, norm_w=lw.ln_in, eps=llm.rms_norm_eps,
                                   h_out=sb.hA, bias=lw.bqkv_sk, q_out=sb.q, k_cache=cache.k[layer], vt_cache=cache.vt[layer], rope_cos=rope[0],
                                   rope_sin=rope[1], pos_ids=pos_ids, n_q_heads=nq, n_kv_heads=nkv, s_max=cache.s_max,
                                   tok_per_batch=tok_per_batch, slot_base=slot_base)
        plan.attn = ops.attn_skinny_args(sb.q, cache.k[layer], cache.vt[layer], sb.attn_parts, batch, tok_per_batch, kv_len, nq, nkv, hd,
                                         (tok_per_batch * nq * hd, hd, nq * hd), ks, vs, cache.s_max, hd ** -0.5, attn_mode, 1,
                                         valid_len=valid_len, blk_start=blk_start, first_tok_kv_len=first_tok_kv_len)
        if not skip_post_attn:
            plan.o = ops.skinny_args(None, lw.sk_o, M, out_f32=sb.part_o, attn_m=sb.attn_parts[0], attn_l=sb.attn_parts[1],
                                     attn_o=sb.attn_parts[2], attn_splits=1, attn_group=nq // nkv, attn_nq=tok_per_batch)
            if chain_ao:
                plan.attn = ops.attn_skinny_args(sb.q, cache.k[layer], cache.vt[layer], (sb.chain_parts[0], sb.chain_parts[0], sb.chain_parts[1]), batch, tok_per_batch,
                                                 kv_len, nq, nkv, hd, (tok_per_batch * nq * hd, hd, nq * hd), ks, vs, cache.s_max, hd ** -0.5, attn_mode, nsp2,
                                                 valid_len=valid_len, blk_start=blk_start, first_tok_kv_len=first_tok_kv_len, dense_mask=dense_mask)
                plan.o = ops.skinny_args(None, lw.sk_o, M, out_f32=sb.part_o, attn_m=sb.chain_parts[0], attn_o=sb.chain_parts[1], attn_splits=nsp2,
                                         attn_group=nq // nkv, attn_nq=tok_per_batch)
            # (chain: vlaser_chain_qkv leaves no rounded copy of the residual stream behind -- its input already IS the rounded stream)
            plan.gu = ops.skinny_args(h_in if (chain and n_partials == 0) else sb.hA, lw.sk_gu, M, partials=sb.part_o, n_partials=stack.ks_o, norm_w=lw.ln_post,
                                      eps=llm.rms_norm_eps, h_out=sb.hB, out=sb.act, ldo=sb.act.shape[1])
            plan.down = SimpleNamespace(dbg=None) if chain else ops.skinny_args(sb.act, lw.sk_down, M, out_f32=sb.part_d)
        if len(sb.plans) > 4096:         # keys hold buffer addresses of per-call tensors (ragged lengths): bound the cache
            sb.plans.clear()
        sb.plans[key] = plan
    stream = torch.cuda.current_stream().cuda_stream
    plan.qkv[0].slot_base = slot_base
    if 'qkv' not in skip:                 # `skip` (bench.py only): in-chain timing of one launch = chain with it - chain without it
        if chain:
            ops.launch_chain_qkv(plan.qkv[0], stream)
        else:
            ops.launch_skinny(L.PRO_NORM, L.SK_QKV_ROPE, plan.qkv[0], stream)
    a = plan.attn
    a.kv_len, a.n_splits, a.blk_start = kv_len, (nsp2 if chain_ao else nsp), blk_start
    if chain:
        if skip_post_attn:
            return h_in, None, 0
        if chain_ao:
            if 'attn' not in skip:
                ops.launch_chain_attn(a, stream)
            plan.o[0].attn_splits = nsp2          # (the key count, hence the split count, moves from step to step in an eager decode)
            if 'o' not in skip:
                ops.launch_chain_oproj(plan.o[0], stream)
        else:
            if 'attn' not in skip:
                ops.launch_attn_skinny(a, stream)
            plan.o[0].attn_splits = nsp
            if 'o' not in skip:
                ops.launch_skinny(L.PRO_ATTN, L.SK_PARTIAL, plan.o[0], stream)
        if 'gu' not in skip:
            ops.launch_chain_gu(plan.gu[0], stream)
        if down2:
            if 'down' not in skip:
                ops.chain_down2(sb.act, lw.sk_down42, sb.part_d, M, H_, I_, dbg=plan.down.dbg, stream=stream)
            return sb.hB, sb.part_d, 2
        if 'down' not in skip:
            ops.chain_down(sb.act, lw.sk_down4, sb.hB, sb.hC, M, H_, I_, dbg=plan.down.dbg, stream=stream)
        return sb.hC, None, 0
    if skip_post_attn:          # only this layer's K / V are needed (written by the qkv launch above)
        return sb.hA, None, 0
    if 'attn' not in skip:
        ops.launch_attn_skinny(a, stream)
    # o_proj: the prologue merges the attention split partials (flash-decoding) straight into its activation tile
    plan.o[0].attn_splits = nsp
    if 'o' not in skip:
        ops.launch_skinny(L.PRO_ATTN, L.SK_PARTIAL, plan.o[0], stream)
    if 'gu' not in skip:
        ops.launch_skinny(L.PRO_NORM, L.SK_SWIGLU, plan.gu[0], stream)
    if 'down' not in skip:
        ops.launch_skinny(L.PRO_PLAIN, L.SK_PARTIAL, plan.down[0], stream)
    return sb.hB, sb.part_d, stack.ks_down
