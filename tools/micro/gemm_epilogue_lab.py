"""What the GEMM epilogue's global stores cost (r04): the same launches from the product library and from a lab build whose epilogue stores are predicated
off (-DGEMM_LAB_NOSTORE), on the SFT step's weight-gradient and forward shapes.   python tools/micro/gemm_epilogue_lab.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LAB = os.path.join(ROOT, 'tools', 'micro', 'lab_build')
src = os.path.join(ROOT, 'vlaser_amd', 'csrc')
if os.environ.get('GEMM_LAB_CHILD') != '1':
    os.makedirs(LAB, exist_ok=True)
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result', '-mllvm', '-amdgpu-mfma-vgpr-form']
    objs = [os.path.join(src, f) for f in ('attn.o', 'skinny.o', 'chain.o', 'misc.o', 'train.o', 'attn_bwd.o', 'api.o')]
    variants = [('product', None), ('fragments', ['-DGEMM_STORE_MODE=0']), ('via LDS', ['-DGEMM_STORE_MODE=1']), ('lane swap', ['-DGEMM_STORE_MODE=2']),
                ('no stores', ['-DGEMM_LAB_NOSTORE']), ('issue 1st', ['-DGLDS_ISSUE_FIRST=1'])]
    if os.environ.get('GEMM_LAB_ONLY'):
        variants = [v for v in variants if v[0] in os.environ['GEMM_LAB_ONLY'].split(',')]
    for tag, defs in variants:
        env = dict(os.environ, GEMM_LAB_CHILD='1', GEMM_LAB_TAG=tag)
        if defs is not None:
            name = tag.replace(' ', '_')
            so = os.path.join(LAB, f'libvlaser_epi_{name}.so')
            subprocess.check_call(['/opt/rocm/bin/hipcc'] + flags + defs + ['-c', os.path.join(src, 'gemm.hip'), '-o', os.path.join(LAB, f'gemm_{name}.o')], stderr=subprocess.DEVNULL)
            subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + [os.path.join(LAB, f'gemm_{name}.o'), '-o', so])
            env['VLASER_HIP_LIB'] = so
        subprocess.check_call([sys.executable, os.path.abspath(__file__)], env=env)
    sys.exit(0)

import torch  # noqa: E402
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from vlaser_amd import ops, _lib as L  # noqa: E402
from kernel_lab import rnd, timeit  # noqa: E402
BF = torch.bfloat16
tag = os.environ['GEMM_LAB_TAG']
S, Sp = 560, 576
res = []
for (N, K, name, cfgs) in [(17920, 1536, 'wgrad gate/up', (1300, 1340)), (1536, 8960, 'wgrad down', (1300, 1340)), (2048, 1536, 'wgrad qkv', (1100, 1140))]:
    dps = [torch.zeros(Sp, N, dtype=BF, device='cuda') for _ in range(6)]
    xp = torch.zeros(Sp, K, dtype=BF, device='cuda')
    for t in dps:
        t[:S] = rnd(S, N, std=1.0)
    xp[:S] = rnd(S, K, std=1.0)
    out = torch.zeros(N, K, dtype=BF, device='cuda')
    for cfg in cfgs:
        res.append(f'{name} {cfg}: {timeit([lambda t=t: ops.gemm_tn_lds(t, xp, out, Sp, force_cfg=cfg) for t in dps]):6.2f} us')
# forward shapes (NT): gate/up with SwiGLU at M = 560, ViT fc1 at M = 1025, LLM prefill qkv-sized at M = 384
for (M, N, K, name) in [(560, 17920, 1536, 'fwd gate/up (NONE)'), (1025, 4096, 1024, 'ViT fc1 (NONE)'), (384, 2048, 1536, 'prefill 384x2048'), (384, 17920, 1536, 'prefill gate/up-sized'),
                        (1025, 3072, 1024, 'ViT qkv-sized'), (560, 1536, 8960, 'SFT down-sized')]:
    ws = [rnd(N, K) for _ in range(6)]
    x = rnd(M, K, std=1.0)
    out = torch.zeros(M, N, dtype=BF, device='cuda')
    res.append(f'{name}: {timeit([lambda w=w: ops.gemm(L.EPI_NONE, x, w, out=out) for w in ws]):6.2f} us')
print(f'[{tag:9s}] ' + '   '.join(res))
# 8B prefill shapes (13 tiles, M = 3408): the two-stage 256x256 ring with the refill requested behind the first half's reads (1300) or first (1301)
if tag == 'product':
    res = []
    for (M, N, K, name) in [(3408, 4608, 3584, '8B qkv-sized'), (3408, 3584, 3584, '8B o-sized'), (3408, 18944, 3584, '8B half gate/up'), (3408, 3584, 18944, '8B down-sized')]:
        ws = [rnd(N, K) for _ in range(3)]
        x = rnd(M, K, std=1.0)
        out = torch.zeros(M, N, dtype=BF, device='cuda')
        for cfg in (0, 1300, 1301):
            res.append(f'{name} {cfg}: {timeit([lambda w=w: ops.gemm(L.EPI_NONE, x, w, out=out, force_bm=cfg) for w in ws]):7.2f} us')
    print('[8B shapes] ' + '   '.join(res))
