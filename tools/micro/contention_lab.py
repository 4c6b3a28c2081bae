"""How much each forward kernel of the SFT step stretches while AdamW streams on another stream (the step overlaps the optimizer with the next forward):
us per launch alone vs under a continuously running AdamW (198 M-parameter bucket, one workgroup per CU), per tile configuration.
    python tools/micro/contention_lab.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from vlaser_amd import ops, _lib as L
from kernel_lab import rnd
BF, F32 = torch.bfloat16, torch.float32
dev = 'cuda'
S, H, I, NQ, nq, nkv, hd = 560, 1536, 8960, 2048, 12, 2, 128

# background: AdamW over a 198 M-parameter bucket, queued back to back on its own stream
n = 198_000_000
p_, g_ = torch.zeros(n, dtype=BF, device=dev), torch.zeros(n, dtype=BF, device=dev)
ms_, m_, v_ = torch.zeros(n, dtype=F32, device=dev), torch.zeros(n, dtype=F32, device=dev), torch.zeros(n, dtype=F32, device=dev)
bg = torch.cuda.Stream()


def timed(fns, busy):
    for f in fns: f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns: f()
    g.replay(); torch.cuda.synchronize()
    reps = 10
    end_bg = torch.cuda.Event(enable_timing=True)
    if busy:
        with torch.cuda.stream(bg):
            for i in range(40):            # ~40 ms of optimizer traffic
                ops.adamw(p_, ms_, m_, v_, g_, 1e-5, 0.9, 0.999, 1e-8, 0.05, 1.0, i + 1)
            end_bg.record()
        import time; time.sleep(0.003)     # let the background get going
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record()
    torch.cuda.synchronize()
    if busy:
        assert e1.elapsed_time(end_bg) > 0, 'background finished before the foreground: lengthen it'
    return e0.elapsed_time(e1) * 1e3 / (reps * len(fns))


def row(name, make, cfgs=(0,)):
    out = []
    for c in cfgs:
        fns = make(c)
        a = timed(fns, False); b = timed(fns, True)
        out.append(f'cfg {c}: {a:6.1f} -> {b:6.1f} us (x{b / a:.2f})')
    print(f'{name:38s} ' + '   '.join(out), flush=True)


x = rnd(S, H, std=1.0)
# LLM layer forward pieces
wqkv = [rnd(NQ, H) for _ in range(4)]
o_ = torch.zeros(S, NQ, dtype=BF, device=dev)
row('qkv NT [560x2048x1536]', lambda c: [lambda w=w: ops.gemm(L.EPI_NONE, x, w, out=o_, force_bm=c) for w in wqkv], (0, 1500, 1100, 1506, 1105))
sm = 576
q = rnd(S, nq * hd, std=1.0); Kc = rnd(1, nkv, sm, hd, std=1.0); vt = rnd(1, nkv, hd, sm, std=1.0)
ao = torch.zeros(S, nq * hd, dtype=BF, device=dev)
lse = torch.zeros(nq * S, dtype=F32, device=dev)
row(f'attention causal S=560 (KS={os.environ.get("VLASER_ATTN_KS", "auto")})', lambda c: [lambda: ops.attn_prefill(q, Kc, vt, ao, 1, S, S, nq, nkv, hd, (S * nq * hd, hd, nq * hd), (nkv * sm * hd, sm * hd),
            (nkv * hd * sm, hd * sm), (S * nq * hd, nq * hd), sm, hd ** -0.5, L.ATTN_CAUSAL, lse_out=lse)] * 4)
wo = [rnd(H, nq * hd) for _ in range(4)]
part = torch.zeros(8 * S * H, dtype=F32, device=dev)
a_in = rnd(S, nq * hd, std=1.0)
for sp in (1, 2, 3):
    row(f'o_proj NT PARTIAL x{sp} [560x1536x1536]', lambda c: [lambda w=w: ops.gemm(L.EPI_PARTIAL, a_in, w, out_f32=part, k_splits=sp, force_bm=c) for w in wo], (0, 1500, 1100))
wgu = [rnd(2 * I, H) for _ in range(4)]
act, gu = torch.zeros(S, I, dtype=BF, device=dev), torch.zeros(S, 2 * I, dtype=BF, device=dev)
row('gate/up NT SWIGLU+aux', lambda c: [lambda w=w: ops.gemm(L.EPI_SWIGLU, x, w, out=act, aux_out=gu, ld_aux=gu.stride(0), force_bm=c) for w in wgu], (0, 1100, 1105, 1200, 1300))
wd = [rnd(H, I) for _ in range(4)]
a2 = rnd(S, I, std=1.0)
for sp in (2, 4):
    row(f'down NT PARTIAL x{sp} [560x1536x8960]', lambda c: [lambda w=w: ops.gemm(L.EPI_PARTIAL, a2, w, out_f32=part, k_splits=sp, force_bm=c) for w in wd], (0, 1500, 1100, 1105))
h2, x2 = torch.zeros(S, H, dtype=BF, device=dev), torch.zeros(S, H, dtype=BF, device=dev)
nw = torch.ones(H, dtype=BF, device=dev)
row('reduce_norm x4', lambda c: [lambda: ops.reduce_norm(x, part, 4, S, H, h2, x2, norm=1, norm_w=nw, eps=1e-6)] * 4)
# ViT pieces (1025 rows)
T, C = 1025, 1024
xv = rnd(T, C, std=1.0)
wv = [rnd(4 * C, C) for _ in range(4)]; bv = rnd(4 * C)
ov = torch.zeros(T, 4 * C, dtype=BF, device=dev)
row('ViT fc1 BIAS_GELU [1025x4096x1024]', lambda c: [lambda w=w: ops.gemm(L.EPI_BIAS_GELU, xv, w, out=ov, bias=bv, force_bm=c) for w in wv], (0, 1100, 1105, 1440))
Sp = 1088
qv = rnd(1, 16, Sp, 64, std=1.0); kv = rnd(1, 16, Sp, 64, std=1.0); vv = rnd(1, 16, 64, Sp, std=1.0)
outv = torch.zeros(1, T, C, dtype=BF, device=dev)
row('ViT attention 1025x16x64', lambda c: [lambda: ops.attn_prefill(qv, kv, vv, outv, 1, T, T, 16, 16, 64, (16 * Sp * 64, Sp * 64, 64), (16 * Sp * 64, Sp * 64), (16 * 64 * Sp, 64 * Sp),
                                                                    (T * C, C), Sp, 1.0, L.ATTN_FULL)] * 4)
