"""Lab for the fused o_proj -> gate/up launch (csrc/euler.hip) at the action expert's shapes: per-layer time of
   attn_skinny -> o_proj -> gate/up   (two launches, 32-row / 16-row gate/up units)   vs   attn_skinny -> fused_ogu
inside HIP graphs of 12 layers (every layer streams its own weights: HBM-cold as in the chunk), HIP events on the launch stream, interleaved
rounds; then the in-kernel timeline (wall_clock64 stamps per workgroup) of the fused launch.   python tools/micro/ogu_lab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vlaser_amd import ops, _lib as L  # noqa: E402

BF = torch.bfloat16
dev = 'cuda'
rnd = lambda *s, std=0.03: (torch.randn(*s, device=dev) * std).to(BF)
M, H, I, nq, nkv, hd = 4, 768, 8960, 12, 2, 128
G, NL, S, kv_len, s_max = nq // nkv, 12, 7, 389, 448


def main():
    torch.manual_seed(0)
    lay = []
    for _ in range(NL):
        gw, uw = rnd(I, H), rnd(I, H)
        lay.append(dict(wo=ops.pack_skinny(rnd(H, nq * hd), 3, 1), gu16=ops.pack_skinny(ops.pack_gate_up8(gw, uw), 1, 1),
                        gu32=ops.pack_skinny(ops.pack_gate_up(gw, uw), 1, 2), k=rnd(1, nkv, s_max, hd, std=1.0), vt=rnd(1, nkv, hd, s_max, std=1.0)))
    q = rnd(M, nq * hd, std=1.0)
    nw = torch.ones(H, dtype=BF, device=dev)
    h = rnd(M, H, std=1.0)
    parts = ops.attn_partial_buffers(1, nkv, dev)
    part_o = torch.zeros(3, M, H, dtype=torch.float32, device=dev)
    act = torch.zeros(M, I, dtype=BF, device=dev)
    hout = torch.zeros(M, H, dtype=BF, device=dev)
    sync = torch.zeros(NL, L.FUSED_SYNC_WORDS, dtype=torch.int32, device=dev)
    valid = torch.tensor([277], dtype=torch.int32, device=dev)
    ks, vs = (nkv * s_max * hd, s_max * hd), (nkv * hd * s_max, hd * s_max)

    def attn(l):
        ops.attn_skinny(q, l['k'], l['vt'], parts, 1, M, kv_len, nq, nkv, hd, (M * nq * hd, hd, nq * hd), ks, vs, s_max, hd ** -0.5, L.ATTN_PREFIX, S,
                        valid_len=valid, blk_start=384)

    def seq(kind, delay=0, dbg=None, dbg_layer=-1):
        if kind.startswith('fused'):
            sync.zero_()
        for i, l in enumerate(lay):
            attn(l)
            if kind == 'attn':
                continue
            if kind.startswith('fused'):
                a = ops.fused_ogu_args(parts, l['wo'], part_o, h, nw, 1e-6, hout, l['gu16'], M, act, sync[i], S, G, M, cons_delay=delay,
                                       dbg=dbg if i == dbg_layer else None)
                ops.launch_fused_ogu(a)
            else:
                ops.skinny(L.PRO_ATTN, L.SK_PARTIAL, None, l['wo'], M, out_f32=part_o, attn_m=parts[0], attn_l=parts[1], attn_o=parts[2], attn_splits=S,
                           attn_group=G, attn_nq=M)
                ops.skinny(L.PRO_NORM, L.SK_SWIGLU, h, l['gu16' if kind == 'gu16' else 'gu32'], M, partials=part_o, n_partials=3, norm_w=nw, h_out=hout, out=act, ldo=I)

    variants = [('attn', 0), ('gu32', 0), ('gu16', 0)] + [('fused', d) for d in (0, 50, 100, 150, 200, 300)]
    graphs = []
    for kind, d in variants:
        seq(kind, d)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            seq(kind, d)
        g.replay()
        torch.cuda.synchronize()
        graphs.append(g)
    assert int((sync[:, L.FUSED_SYNC_ERR] != 0).sum()) == 0
    res = [[] for _ in variants]
    for r in range(5):
        for i, g in enumerate(graphs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            res[i].append(e0.elapsed_time(e1) / 10 / NL * 1e3)
    base = sorted(res[0])[len(res[0]) // 2]
    print('| sequence per layer | median us | minus attn-only | rounds |')
    print('|---|---|---|---|')
    for (kind, d), ts in zip(variants, res):
        m = sorted(ts)[len(ts) // 2]
        print(f'| attn -> {kind}{f" delay {d * 10} ns" if kind == "fused" else ""} | {m:.2f} | {m - base:.2f} | {" ".join(f"{t:.2f}" for t in ts)} |')

    # ---- in-kernel timeline of the fused launch (layer NL-1 of an eager sequence), producers / consumers separately
    names = ['start', 'producer: merge in LDS', 'producer: published', 'poll ok (wave 0)', 'gather + residual done', 'norm done', 'end']
    for d in (0, 150):
        dbg = torch.zeros(256 * 8, dtype=torch.int64, device=dev)
        for _ in range(3):
            dbg.zero_()
            seq('fused', d, dbg=dbg, dbg_layer=NL - 1)
        torch.cuda.synchronize()
        t = dbg.view(256, 8).cpu()
        t0 = t[:, 0].min()
        rel = (t - t0).float() * 10      # ns
        print(f'\n--- fused_ogu timeline, consumer delay {d * 10} ns: ns after the earliest workgroup start (min / median / max)')
        for role, sl in (('producers (144)', slice(0, 144)), ('consumers (112)', slice(144, 256))):
            print(f'  {role}')
            for i, n in enumerate(names):
                c = rel[sl, i]
                c = c[t[sl, i] > 0]
                if c.numel():
                    print(f'    {n:28s} {c.min():8.0f} {c.median():8.0f} {c.max():8.0f}')


if __name__ == '__main__':
    main()
