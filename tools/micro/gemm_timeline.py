"""In-kernel timeline of gemm_glds_kernel (r04, VERDICT r03 #1c): lab build of csrc/gemm.hip with -DGEMM_TIMELINE (s_memtime stamps of thread 0 of every
workgroup: start, prologue issued, every K-step's barrier passed, loop end, epilogue start / end), one launch per shape inside a graph of 6 launches
with distinct operands.   python tools/micro/gemm_timeline.py"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LAB = os.path.join(ROOT, 'tools', 'micro', 'lab_build')
os.makedirs(LAB, exist_ok=True)
so = os.path.join(LAB, 'libvlaser_gemmtl.so')
src = os.path.join(ROOT, 'vlaser_amd', 'csrc')
flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result', '-mllvm', '-amdgpu-mfma-vgpr-form']
subprocess.check_call(['/opt/rocm/bin/hipcc'] + flags + ['-DGEMM_TIMELINE', '-c', os.path.join(src, 'gemm.hip'), '-o', os.path.join(LAB, 'gemm_tl.o')])
objs = [os.path.join(src, f) for f in ('attn.o', 'skinny.o', 'chain.o', 'misc.o', 'train.o', 'attn_bwd.o', 'api.o')]
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + [os.path.join(LAB, 'gemm_tl.o'), '-o', so])
os.environ['VLASER_HIP_LIB'] = so

import torch  # noqa: E402
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from vlaser_amd import ops, _lib as L  # noqa: E402
from kernel_lab import rnd, timeit  # noqa: E402
BF = torch.bfloat16
lib = C.CDLL(so)
lib.vlaser_gemm_debug_read.argtypes = [C.c_void_p]


def report(name, us, nk, nwg):
    buf = (C.c_longlong * (1024 * 40))()
    lib.vlaser_gemm_debug_read(buf)
    t = torch.tensor(list(buf), dtype=torch.int64).view(1024, 40)[:min(nwg, 1024)].double()
    t0 = t[:, 0].min()
    ticks_us = 1.0 / float(os.environ.get('SHADER_MHZ', '2100'))        # clock64() counts shader cycles on this chip (~2.1 GHz under MFMA load, guide: DVFS)
    rows = [('prologue tiles requested', t[:, 1] - t[:, 0]), ('first tile landed + barrier', t[:, 2] - t[:, 1])]
    if nk > 1:
        steps = (t[:, 2 + nk - 1] - t[:, 2]) / (nk - 1)
        rows.append((f'one K-step (mean of {nk - 1})', steps))
    rows += [('last K-step -> loop end', t[:, 36] - t[:, 2 + nk - 1]), ('drain + barrier before the epilogue', t[:, 37] - t[:, 36]), ('epilogue (math + stores issued)', t[:, 38] - t[:, 37]),
             ('whole workgroup', t[:, 38] - t[:, 0])]
    print(f'{name}: {us:.2f} us per launch, {nwg} workgroups, {nk} K-steps; us at 2.1 GHz, min / median / max over workgroups')
    for n, v in rows:
        v = v * ticks_us
        print(f'    {n:42s} {v.min():7.2f} {v.median():7.2f} {v.max():7.2f}')


S, Sp = 560, 576
for (N, K, name, cfg, bm, bn) in [(17920, 1536, 'wgrad gate/up, TN 256x256', 1300, 256, 256), (1536, 8960, 'wgrad down, TN 256x256', 1300, 256, 256), (2048, 1536, 'wgrad qkv, TN 128x128', 1100, 128, 128)]:
    dps = [torch.zeros(Sp, N, dtype=BF, device='cuda') for _ in range(6)]
    xp = torch.zeros(Sp, K, dtype=BF, device='cuda')
    for d in dps:
        d[:S] = rnd(S, N, std=1.0)
    xp[:S] = rnd(S, K, std=1.0)
    out = torch.zeros(N, K, dtype=BF, device='cuda')
    us = timeit([lambda d=d: ops.gemm_tn_lds(d, xp, out, Sp, force_cfg=cfg) for d in dps])
    report(name, us, Sp // 64, -(-N // bm) * -(-K // bn))
for (M, N, K, name, bm, bn) in [(560, 17920, 1536, 'forward gate/up-sized NT (NONE), 192x256', 192, 256), (1025, 4096, 1024, 'ViT fc1-sized NT (NONE), 144x128', 144, 128),
                                (384, 2048, 1536, 'prefill qkv-sized NT (NONE), 64x64', 64, 64)]:
    ws = [rnd(N, K) for _ in range(6)]
    x = rnd(M, K, std=1.0)
    out = torch.zeros(M, N, dtype=BF, device='cuda')
    us = timeit([lambda w=w: ops.gemm(L.EPI_NONE, x, w, out=out) for w in ws])
    report(name, us, K // 64, -(-M // bm) * -(-N // bn))

# the epilogues with their own store code: SwiGLU (+ the training forward's aux), SwiGLU backward (NN form), ViT qkv, Qwen qkv + RoPE
from vlaser_amd import config as Cfg  # noqa: E402
H, I = 1536, 8960
x560, x384 = rnd(560, H, std=1.0), rnd(384, H, std=1.0)
wgus = [ops.pack_gate_up(rnd(I, H), rnd(I, H)) for _ in range(6)]
act = torch.zeros(560, I, dtype=BF, device='cuda'); aux = torch.zeros(560, 2 * I, dtype=BF, device='cuda')
us = timeit([lambda w=w: ops.gemm(L.EPI_SWIGLU, x560, w, out=act, aux_out=aux, ld_aux=aux.stride(0)) for w in wgus])
report('SFT forward gate/up, SwiGLU + aux, M = 560 (192x256)', us, H // 64, 3 * 70)
us = timeit([lambda w=w: ops.gemm(L.EPI_SWIGLU, x384, w, out=act[:384]) for w in wgus])
report('prefill gate/up, SwiGLU, M = 384', us, H // 64, -(-384 // 128) * 70)
wds = [rnd(H, I) for _ in range(6)]
dh = rnd(560, H, std=1.0); dgu = torch.zeros(560, 2 * I, dtype=BF, device='cuda')
us = timeit([lambda w=w: ops.gemm_nn(L.EPI_SWIGLU_BWD, dh, w, out=dgu, res=aux) for w in wds])
report('SFT dgrad down + SwiGLU backward, NN, M = 560 (128x256)', us, H // 64, 5 * 35)
