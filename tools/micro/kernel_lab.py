"""GPU-box kernel lab: per-launch time of individual C-ABI kernels inside a HIP graph (no host gaps), cycling over
distinct weight buffers so every launch is HBM-cold like in the real layer sequence.  Prints us/launch and GB/s."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vlaser_amd import ops, _lib as L
BF = torch.bfloat16
dev = 'cuda'


def timeit(fn_list, reps=20):
    """fn_list: callables launched back-to-back inside one graph; returns us per launch."""
    for f in fn_list: f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fn_list: f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(fn_list))


def rnd(*s, std=0.03):
    return (torch.randn(*s, device=dev) * std).to(BF)


def skinny_cases():
    NL = 28
    M, H, I = 4, 768, 8960
    h = rnd(M, H, std=1.0); nw = torch.ones(H, dtype=BF, device=dev)
    parts = torch.randn(8, M, H, device=dev) * 0.1
    TPU = int(os.environ.get('TPU', '6'))
    wgu = [ops.pack_skinny(rnd(2 * I, H), 1, TPU) for _ in range(NL)]
    out = torch.zeros(M, I, dtype=BF, device=dev); hout = torch.zeros(M, H, dtype=BF, device=dev)
    byts = 2 * I * H * 2
    for npart in (0, 3, 6, 7):
        us = timeit([lambda w=w: ops.skinny(L.PRO_NORM, L.SK_SWIGLU, h, w, M, partials=parts, n_partials=npart, norm_w=nw, h_out=hout, out=out, ldo=I) for w in wgu])
        print(f'gate/up NORM+SWIGLU n_partials={npart}: {us:.2f} us  {byts / us / 1e3:.0f} GB/s')
    pf = torch.zeros(1, M, 2 * I, dtype=torch.float32, device=dev)
    if TPU == 2:
        us = timeit([lambda w=w: ops.skinny(L.PRO_PLAIN, L.SK_PARTIAL, h, w, M, out_f32=pf) for w in wgu])
        print(f'gate/up PLAIN+PARTIAL (no norm, f32 out): {us:.2f} us  {byts / us / 1e3:.0f} GB/s')
    # down
    wd_raw = [rnd(H, I) for _ in range(NL)]
    act = rnd(M, I, std=1.0)
    for ks in (5, 7):
        wd = [ops.pack_skinny(w, ks) for w in wd_raw]
        pd = torch.zeros(ks, M, H, dtype=torch.float32, device=dev)
        us = timeit([lambda w=w: ops.skinny(L.PRO_PLAIN, L.SK_PARTIAL, act, w, M, out_f32=pd) for w in wd])
        print(f'down PLAIN+PARTIAL ks={ks}: {us:.2f} us  {H * I * 2 / us / 1e3:.0f} GB/s')
    # qkv
    nq, nkv = 12, 2
    wq = [ops.pack_skinny(rnd(2048, H)) for _ in range(NL)]; bq = rnd(2048)
    cos, sin = ops.rope_table(64)
    pos = torch.arange(2, 6, dtype=torch.int32, device=dev)
    q_out = torch.zeros(M, 1536, dtype=BF, device=dev)
    kc = torch.zeros(1, nkv, 448, 128, dtype=BF, device=dev); vtc = torch.zeros(1, nkv, 128, 448, dtype=BF, device=dev)
    for npart in (0, 7):
        us = timeit([lambda w=w: ops.skinny(L.PRO_NORM, L.SK_QKV_ROPE, h, w, M, partials=parts, n_partials=npart, norm_w=nw, h_out=hout, bias=bq, q_out=q_out,
                                            k_cache=kc, vt_cache=vtc, rope_cos=cos, rope_sin=sin, pos_ids=pos, n_q_heads=nq, n_kv_heads=nkv, s_max=448,
                                            tok_per_batch=4, slot_base=385) for w in wq])
        print(f'qkv NORM+QKV_ROPE n_partials={npart}: {us:.2f} us  {2048 * H * 2 / us / 1e3:.0f} GB/s')
    # attention + o_proj
    k = [rnd(1, nkv, 448, 128, std=1.0) for _ in range(NL)]; vt = [rnd(1, nkv, 128, 448, std=1.0) for _ in range(NL)]
    valid = torch.tensor([277], dtype=torch.int32, device=dev)
    aparts = ops.attn_partial_buffers(1, nkv, dev)
    for nsp in (1, 2, 4, 7):
        us = timeit([lambda kk=kk, vv=vv: ops.attn_skinny(q_out, kk, vv, aparts, 1, 4, 389, nq, nkv, 128, (4 * 1536, 128, 1536), (nkv * 448 * 128, 448 * 128),
                                                          (nkv * 128 * 448, 128 * 448), 448, 128 ** -0.5, L.ATTN_PREFIX, nsp, valid_len=valid, blk_start=384)
                     for kk, vv in zip(k, vt)])
        print(f'attn_skinny splits={nsp}: {us:.2f} us')
    wo_raw = [rnd(H, 1536) for _ in range(NL)]
    for ks in (1, 3, 6):
        wo = [ops.pack_skinny(w, ks) for w in wo_raw]
        po = torch.zeros(ks, M, H, dtype=torch.float32, device=dev)
        us = timeit([lambda w=w: ops.skinny(L.PRO_ATTN, L.SK_PARTIAL, None, w, M, out_f32=po, attn_m=aparts[0], attn_l=aparts[1], attn_o=aparts[2],
                                            attn_splits=4, attn_group=6, attn_nq=4) for w in wo])
        print(f'o_proj ATTN+PARTIAL ks={ks}: {us:.2f} us  {H * 1536 * 2 / us / 1e3:.0f} GB/s')


def gemm_cases():
    NL = 8
    for (M, N, K, name) in [(384, 2048, 1536, 'llm qkv'), (384, 1536, 1536, 'llm o'), (384, 17920, 1536, 'llm gate/up'), (384, 1536, 8960, 'llm down'),
                            (1025, 3072, 1024, 'vit qkv'), (1025, 1024, 1024, 'vit proj'), (1025, 4096, 1024, 'vit fc1'), (1025, 1024, 4096, 'vit fc2'),
                            (560, 17920, 1536, 'sft gate/up'), (560, 1536, 8960, 'sft down'), (17920, 1536, 576, 'sft wgrad gu'), (1536, 8960, 576, 'sft wgrad down'),
                            (560, 1536, 17920, 'sft dgrad gu')]:
        x = rnd(M, K, std=1.0); ws = [rnd(N, K) for _ in range(NL)]
        out = torch.zeros(M, N, dtype=BF, device=dev)
        fl = 2.0 * M * N * K
        for bm in (32, 64, 128):
            us = timeit([lambda w=w: ops.gemm(L.EPI_NONE, x, w, out=out, force_bm=bm) for w in ws])
            print(f'{name:12s} M={M} N={N} K={K} NONE bm={bm}: {us:7.2f} us {fl / us / 1e6:7.1f} TF')
        for S in (2, 3, 4, 7):
            if K % (S * 64): continue
            part = torch.zeros(S, M, N, dtype=torch.float32, device=dev)
            for bm in (64, 128):
                us = timeit([lambda w=w: ops.gemm(L.EPI_PARTIAL, x, w, out_f32=part, k_splits=S, force_bm=bm) for w in ws])
                print(f'{name:12s} M={M} N={N} K={K} PARTIAL S={S} bm={bm}: {us:7.2f} us {fl / us / 1e6:7.1f} TF')


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if which in ('all', 'skinny'): skinny_cases()
    if which in ('all', 'gemm'): gemm_cases()
