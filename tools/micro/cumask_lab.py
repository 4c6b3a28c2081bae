"""Does CU partitioning fix the AdamW / forward contention of the SFT step?  AdamW on a stream masked to a fraction of the CUs, forward kernels on the
complementary mask: us per forward kernel alone / beside an unmasked AdamW / beside the masked AdamW, and AdamW's own rate in each case.
    python tools/micro/cumask_lab.py"""
import ctypes as C
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from vlaser_amd import ops, _lib as L
from kernel_lab import rnd
BF, F32 = torch.bfloat16, torch.float32
dev = 'cuda'
torch.zeros(1, device=dev)
lib = C.CDLL(os.path.join(ROOT, 'tools', 'micro', 'lab_build', 'libcumask.so'))


def masked_stream(pred):
    words = (C.c_uint32 * 8)()
    for i in range(256):
        if pred(i):
            words[i // 32] |= 1 << (i % 32)
    out = C.c_void_p()
    rc = lib.cumask_stream_create(words, 8, C.byref(out))
    assert rc == 0, rc
    return torch.cuda.ExternalStream(out.value)


S, H, I, nq, nkv, hd = 560, 1536, 8960, 12, 2, 128
n = 198_000_000
p_, g_ = torch.zeros(n, dtype=BF, device=dev), torch.zeros(n, dtype=BF, device=dev)
ms_, m_, v_ = torch.zeros(n, dtype=F32, device=dev), torch.zeros(n, dtype=F32, device=dev), torch.zeros(n, dtype=F32, device=dev)
x = rnd(S, H, std=1.0)
wgu = [rnd(2 * I, H) for _ in range(4)]
act, gu = torch.zeros(S, I, dtype=BF, device=dev), torch.zeros(S, 2 * I, dtype=BF, device=dev)
wd = [rnd(H, I) for _ in range(4)]
a2 = rnd(S, I, std=1.0)
part = torch.zeros(8 * S * H, dtype=F32, device=dev)
sm = 576
q = rnd(S, nq * hd, std=1.0); Kc = rnd(1, nkv, sm, hd, std=1.0); vt = rnd(1, nkv, hd, sm, std=1.0)
ao = torch.zeros(S, nq * hd, dtype=BF, device=dev)
fg_kernels = {
    'gate/up fwd': [lambda w=w: ops.gemm(L.EPI_SWIGLU, x, w, out=act, aux_out=gu, ld_aux=gu.stride(0)) for w in wgu],
    'down fwd x4': [lambda w=w: ops.gemm(L.EPI_PARTIAL, a2, w, out_f32=part, k_splits=4) for w in wd],
    'attention S=560': [lambda: ops.attn_prefill(q, Kc, vt, ao, 1, S, S, nq, nkv, hd, (S * nq * hd, hd, nq * hd), (nkv * sm * hd, sm * hd), (nkv * hd * sm, hd * sm),
                                                  (S * nq * hd, nq * hd), sm, hd ** -0.5, L.ATTN_CAUSAL)] * 4,
}


def run(fg_stream, bg_stream, fns, busy):
    with torch.cuda.stream(fg_stream):
        for f in fns: f()
    torch.cuda.synchronize()
    reps = 40
    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if busy:
        with torch.cuda.stream(bg_stream):
            b0.record()
            for i in range(30):
                ops.adamw(p_, ms_, m_, v_, g_, 1e-5, 0.9, 0.999, 1e-8, 0.05, 1.0, i + 1)
            b1.record()
        time.sleep(0.003)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(fg_stream):
        e0.record()
        for _ in range(reps):
            for f in fns: f()
        e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * len(fns))
    tb = 28.0 * n * 30 / (b0.elapsed_time(b1) * 1e-3) / 1e12 if busy else 0.0
    return us, tb


default = torch.cuda.current_stream()
plain_bg = torch.cuda.Stream()
for frac_name, bgp in (('1/4 of the CUs (i % 4 == 0)', lambda i: i % 4 == 0), ('1/2 of the CUs (i % 2 == 0)', lambda i: i % 2 == 0), ('CUs 0..63', lambda i: i < 64)):
    bg = masked_stream(bgp)
    fg = masked_stream(lambda i: not bgp(i))
    print(f'--- AdamW masked to {frac_name}, forward on the complement')
    for name, fns in fg_kernels.items():
        alone, _ = run(default, plain_bg, fns, False)
        both, tb0 = run(default, plain_bg, fns, True)
        alone_m, _ = run(fg, bg, fns, False)
        masked, tb1 = run(fg, bg, fns, True)
        print(f'{name:16s} alone {alone:6.1f} us | beside AdamW {both:6.1f} us (AdamW {tb0:.2f} TB/s) | masked: alone {alone_m:6.1f}, beside {masked:6.1f} us (AdamW {tb1:.2f} TB/s)', flush=True)
