#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
{
echo "== tail path on"; python tools/micro/vit_attn_probe.py
echo "== VLASER_ATTN_NO_TAIL=1"; VLASER_ATTN_NO_TAIL=1 python tools/micro/vit_attn_probe.py
} > gpurun_out/r04l_vit_attn.log 2>&1
cat gpurun_out/r04l_vit_attn.log
python - <<'PY' > gpurun_out/r04l_nan_debug.log 2>&1
import torch, sys
sys.path.insert(0, '.')
from vlaser_amd import ops, _lib as L
BF = torch.bfloat16
def rnd(*s, std=1.0, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed + sum(s)); return (torch.randn(*s, generator=g) * std).to(BF).cuda()
def run(tag):
    nq_tok, kv_len, nq, nkv, smax, H, valid = 4, 389, 12, 2, 1536, 768, 277
    q = rnd(nq_tok, nq * 128, seed=1); k = rnd(1, nkv, smax, 128, seed=2); v = rnd(1, nkv, smax, 128, seed=3)
    vt = v.transpose(-1, -2).contiguous()
    vl = torch.tensor([valid], dtype=torch.int32, device='cuda')
    parts = ops.attn_partial_buffers(1, nkv, 'cuda')
    a = ops.attn_skinny_args(q, k, vt, parts, 1, nq_tok, kv_len, nq, nkv, 128, (nq_tok * nq * 128, 128, nq * 128), (nkv * smax * 128, smax * 128), (nkv * 128 * smax, 128 * smax), smax, 128 ** -0.5, L.ATTN_PREFIX, 1, valid_len=vl, blk_start=384)
    wp = ops.pack_skinny(rnd(H, nq * 128, std=0.03, seed=9), nkv, 1)
    out = torch.full((nkv, nq_tok, H), 5.0, dtype=torch.float32, device='cuda')
    ops.launch_attn_oproj(a, wp, out, H); torch.cuda.synchronize()
    bad = ~torch.isfinite(out)
    print(tag, 'non-finite:', int(bad.sum()), 'of', out.numel(), 'slabs', bad.sum((1, 2)).tolist(), 'toks', bad.sum((0, 2)).tolist(), 'first cols', bad.any(0).any(0).nonzero().flatten()[:20].tolist())
run('fresh')
# poison LDS with NaN patterns through another kernel that uses a lot of LDS, then run again
x = torch.full((1025, 1024), float('nan'), dtype=BF, device='cuda'); w = rnd(1024, 1024, std=0.03)
for i in range(3):
    ops.linear(x, w)
torch.cuda.synchronize()
run('after NaN GEMMs')
run('again')
PY
cat gpurun_out/r04l_nan_debug.log
