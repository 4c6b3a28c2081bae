"""GPU-box lab: the five GEMM-shaped pieces of the SFT attention backward at S = 560 (12 q heads, 2 kv heads, hd 128), HIP-graph timed."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from vlaser_amd import ops, _lib as L
from kernel_lab import timeit, rnd
BF = torch.bfloat16
S, nq, nkv, hd = 560, 12, 2, 128
G, Sp, sm = nq // nkv, 576, 576
q = rnd(S, nq * hd); dao = rnd(S, nq * hd); Kc = rnd(nkv, sm, hd); Vn = rnd(nkv, Sp, hd); KT = rnd(nkv, hd, Sp)
sc = torch.zeros(nq, S, Sp, dtype=torch.float32, device='cuda'); dP = torch.zeros_like(sc)
P = rnd(nq, S, Sp); dS = rnd(nq, S, Sp)
dq = torch.zeros(S, nq * hd, dtype=BF, device='cuda'); dk = torch.zeros(S, nq * hd, dtype=BF, device='cuda'); dv = torch.zeros_like(dk)
fs = {
    'Q K^T (F32, batch 12)': lambda: ops.gemm_raw(L.EPI_F32, q, Kc, sc, S, S, hd, nq * hd, hd, Sp, batch=nq, a_bs=hd, w_bs=sm * hd, o_bs=S * Sp, w_group=G),
    'dO V^T (F32, batch 12)': lambda: ops.gemm_raw(L.EPI_F32, dao, Vn, dP, S, S, hd, nq * hd, hd, Sp, batch=nq, a_bs=hd, w_bs=Sp * hd, o_bs=S * Sp, w_group=G),
    'dQ = dS K (batch 12)': lambda: ops.gemm_raw(L.EPI_NONE, dS, KT, dq, S, hd, Sp, Sp, Sp, nq * hd, batch=nq, a_bs=S * Sp, w_bs=hd * Sp, o_bs=hd, w_group=G),
    'dK = dS^T Q (TN, 12)': lambda: ops.gemm_tn_grouped(dS, q, dk, S, hd, S, Sp, nq * hd, nq * hd, 1, 0, 0, nq, S * Sp, hd, hd),
    'dV = P^T dO (TN, 12)': lambda: ops.gemm_tn_grouped(P, dao, dv, S, hd, S, Sp, nq * hd, nq * hd, 1, 0, 0, nq, S * Sp, hd, hd),
}
for name, f in fs.items():
    us = timeit([f] * 8)
    print(f'{name:26s}: {us:7.2f} us   ({2.0 * nq * S * S * hd / us / 1e6:6.1f} TF)')

# ---- the fused backward (csrc/attn_bwd.hip; VLASER_ATTN_BWD_KS=1|2 picks the number of wave groups), forward with lse for reference
vt = Vn.transpose(-1, -2).contiguous()
out = torch.zeros(S, nq * hd, dtype=BF, device='cuda')
lse = torch.zeros(nq * S, dtype=torch.float32, device='cuda'); delta = torch.zeros_like(lse)
fwd = lambda: ops.attn_prefill(q, Kc, vt, out, 1, S, S, nq, nkv, hd, (S * nq * hd, hd, nq * hd), (nkv * sm * hd, sm * hd), (nkv * hd * sm, hd * sm),
                               (S * nq * hd, nq * hd), sm, hd ** -0.5, L.ATTN_CAUSAL, lse_out=lse)
fwd()
bwd = lambda: ops.attn_bwd(q, Kc, vt, out, dao, lse, delta, dq, dk, dv, S, nq, nkv, sm, hd ** -0.5)
print(f'forward (causal, lse)     : {timeit([fwd] * 8):7.2f} us')
print(f'fused backward (2 kernels): {timeit([bwd] * 8):7.2f} us   KS={os.environ.get("VLASER_ATTN_BWD_KS", "auto")}')
