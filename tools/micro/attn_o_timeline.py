"""In-kernel timeline of attn_oproj_kernel (r04 rewrite): lab build of csrc/attn_o.hip with -DAO_TIMELINE (wall_clock64 stamps, 10 ns, thread 0 of EVERY
workgroup), expert geometry, inside a graph of 8 launches with distinct K / V / W_o (HBM-cold weights, L2-cold K / V per XCD).
    python tools/micro/attn_o_timeline.py            # builds tools/micro/lab_build/libvlaser_aotl.so itself"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LAB = os.path.join(ROOT, 'tools', 'micro', 'lab_build')
os.makedirs(LAB, exist_ok=True)
so = os.path.join(LAB, 'libvlaser_aotl.so')
src = os.path.join(ROOT, 'vlaser_amd', 'csrc')
flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result', '-mllvm', '-amdgpu-mfma-vgpr-form']
subprocess.check_call(['/opt/rocm/bin/hipcc'] + flags + ['-DAO_TIMELINE', '-c', os.path.join(src, 'attn_o.hip'), '-o', os.path.join(LAB, 'attn_o_tl.o')])
objs = [os.path.join(src, f) for f in ('gemm.o', 'attn.o', 'skinny.o', 'euler.o', 'misc.o', 'train.o', 'attn_bwd.o', 'api.o') if os.path.exists(os.path.join(src, f))]
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + [os.path.join(LAB, 'attn_o_tl.o'), '-o', so])
os.environ['VLASER_HIP_LIB'] = so

import torch  # noqa: E402
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from vlaser_amd import ops, _lib as L  # noqa: E402
from kernel_lab import rnd, timeit  # noqa: E402

nq_tok, kv_len, nq, nkv, smax, H = int(os.environ.get('NQ', 4)), 389, 12, 2, 448, 768
q = rnd(nq_tok, nq * 128, std=1.0)
ks = [rnd(1, nkv, smax, 128, std=1.0) for _ in range(8)]; vts = [rnd(1, nkv, 128, smax, std=1.0) for _ in range(8)]
wos = [ops.pack_skinny(rnd(H, nq * 128), nkv, 1) for _ in range(8)]
valid = torch.tensor([int(os.environ.get('VALID', 277))], dtype=torch.int32, device='cuda')
parts = ops.attn_partial_buffers(1, nkv, 'cuda')
out = torch.zeros(nkv, nq_tok, H, dtype=torch.float32, device='cuda')
args = [ops.attn_skinny_args(q, k, vt, parts, 1, nq_tok, kv_len, nq, nkv, 128, (nq_tok * nq * 128, 128, nq * 128), (nkv * smax * 128, smax * 128), (nkv * 128 * smax, 128 * smax),
                             smax, 128 ** -0.5, L.ATTN_PREFIX, 1, valid_len=valid, blk_start=384) for k, vt in zip(ks, vts)]
us = timeit([lambda a=a, w=w: ops.launch_attn_oproj(a, w, out, H) for a, w in zip(args, wos)])
print(f'attn_oproj: {us:.2f} us per launch (8 K / V / W_o sets cycled, graph of 8 launches)')
nb = (H // 16) * nkv
buf = (C.c_longlong * (256 * 16))()
L.lib().vlaser_attn_oproj_debug_read(buf)
t = torch.tensor(list(buf), dtype=torch.int64).view(256, 16)[:nb]
order = [(0, 'start'), (12, 'K requested'), (8, 'Q / valid_len / V^T requested'), (9, 'K landed + written to LDS'), (2, 'Q K^T + local softmax + P written'),
         (3, 'V^T written, P + stats visible (barrier)'), (4, 'P V done'), (5, 'x written (barrier)'), (6, 'o_proj partial tiles (barrier)'), (7, 'end')]
t0 = t[:, 0].min()
print(f'{nb} workgroups; ns after the earliest workgroup start: min / median / max    (+ median since the previous stamp)')
prev = None
for i, n in order:
    c = (t[:, i] - t0).float() * 10
    d = '' if prev is None else f'   (+{(c - prev).median():.0f})'
    print(f'  {n:36s} {c.min():7.0f} {c.median():7.0f} {c.max():7.0f}{d}')
    prev = c

cyc = (t[:, 15] - t[:, 14]).float(); wall = (t[:, 7] - t[:, 0]).float() * 10
print(f'shader clock while the kernel runs: {(cyc / wall).median() * 1000:.0f} MHz (s_memtime cycles / wall ns, median over workgroups)')
