"""In-kernel timeline of the <= 16-row layer-step chain AS IT RUNS INSIDE THE CHUNK'S HIP GRAPH (r05, VERDICT r04 #1c).

Every launch of the Euler phase (5 per layer: qkv -> attn_skinny -> o_proj -> gate/up -> down) or of a greedy decode step gets its own
stamp buffer (`dbg` of VlaserSkinnyArgs / VlaserAttnArgs: wall_clock64, a chip-wide 100 MHz counter, so stamps of different kernels
share one time axis); the phase is captured into a HIP graph with those pointers and replayed; the LAST pass through the layers is what
the buffers hold afterwards.  Per kernel, relative to the end of the last workgroup of the PREVIOUS kernel of the chain:

    first / median / last workgroup start   (boundary + dispatch spread)
    prologue stamps (median)                (skinny: phase-1 data back, prologue done, first MFMA batch, first unit reduced)
    first / median / last workgroup end

averaged over the layers of the pass (first two layers skipped).  Usage: python tools/micro/chain_timeline.py [euler|decode] [reps]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vlaser_amd import config as C, synth            # noqa: E402
from vlaser_amd.pizero import PiZeroInference        # noqa: E402

KERNELS = ('qkv', 'attn', 'o', 'gu', 'down')
SK_NAMES = ['start', 'phase1 data', 'prologue done', 'first MFMA', 'unit0 reduced', 'end']
AT_NAMES = ['start', 'chunk0 done', 'merged', 'end']


def euler_model(dev):
    vla = C.VLAConfig(base=C.vlaser_2b())
    sd = synth.vla_state_dict(vla, device=dev, dtype=torch.bfloat16)
    m = PiZeroInference(vla, device=dev, max_batch=1, use_graph=False)
    m.load_state_dict(sd)
    del sd
    g = torch.Generator().manual_seed(0)
    ids = torch.full((1, 384), vla.base.pad_token_id)
    ids[:, :10] = torch.randint(0, 151643, (1, 10), generator=g)
    ids[:, 10:266] = vla.base.img_context_token_id
    ids[:, 266:277] = torch.randint(0, 151643, (1, 11), generator=g)
    pv = torch.randn(1, 3, 448, 448, generator=g)
    m.infer_action(ids.to(dev), pv.to(dev).to(torch.bfloat16), proprios=torch.rand(1, 1, 7, generator=g).to(dev), noise=torch.randn(1, 4, 7, generator=g).to(dev),
                   valid_len=torch.tensor([277], device=dev))
    torch.cuda.synchronize()
    return m


def attach(plans, nL, dev):
    """One stamp buffer per launch of the plans of `nL` layers; returns {(layer, kernel): (tensor, n_blocks_max)}."""
    bufs = {}
    for key, plan in plans.items():
        layer = key[0]
        for kn in KERNELS:
            if not hasattr(plan, kn):
                continue
            t = torch.zeros(512 * 8, dtype=torch.int64, device=dev)
            st = getattr(plan, kn)
            st = st[0] if isinstance(st, tuple) else st
            st.dbg = t.data_ptr()
            bufs[(layer, kn)] = t
    return bufs


def report(bufs, nL, title):
    rows = {kn: [] for kn in KERNELS}
    prev_end = None
    for layer in range(nL):
        for kn in KERNELS:
            t = bufs.get((layer, kn))
            if t is None:
                continue
            d = t.view(-1, 8).cpu()
            d = d[d[:, 0] > 0]
            if d.shape[0] == 0:
                continue
            ncol = 6 if kn != 'attn' else 4
            end_col = ncol - 1
            if prev_end is not None and layer >= 2:
                rel = (d[:, :ncol] - prev_end).double() * 10.0        # ns
                st, en = rel[:, 0], rel[:, end_col]
                mid = [rel[:, c][d[:, c] > 0].median().item() if (d[:, c] > 0).any() else float('nan') for c in range(1, end_col)]
                rows[kn].append([d.shape[0], st.min().item(), st.median().item(), st.max().item()] + mid + [en.min().item(), en.median().item(), en.max().item()])
            prev_end = int(d[:, end_col].max())
    print(f'=== {title}: ns after the LAST workgroup of the previous kernel in the chain ended (mean over layers 2..{nL - 1})')
    tot = 0.0
    for kn in KERNELS:
        if not rows[kn]:
            continue
        r = torch.tensor(rows[kn], dtype=torch.float64).nanmean(0).tolist()
        names = (SK_NAMES if kn != 'attn' else AT_NAMES)[1:-1]
        mids = '  '.join(f'{n} {v:6.0f}' for n, v in zip(names, r[4:-3]))
        print(f'{kn:5s} wgs {r[0]:5.0f} | start first {r[1]:6.0f} med {r[2]:6.0f} last {r[3]:6.0f} | {mids} | end first {r[-3]:6.0f} med {r[-2]:6.0f} last {r[-1]:6.0f}')
        tot += r[-1]
    print(f'sum of (previous end -> this end) over the 5 kernels: {tot / 1e3:.2f} us per layer-step')


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else 'euler'
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = 'cuda:0'
    torch.set_grad_enabled(False)
    if what == 'euler':
        m = euler_model(dev)
        nL = m.cfg.base.llm.num_hidden_layers
        bufs = attach(m.sb_act.plans, nL, dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            m._run_euler(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        for t in bufs.values():
            t.zero_()
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        n = nL * m.num_inference_steps
        print(f'Euler phase WITH stamps: {e0.elapsed_time(e1):.3f} ms = {e0.elapsed_time(e1) * 1e3 / n:.2f} us per layer-step (bench without stamps: phases.euler_us_per_layer_step)')
        report(bufs, nL, 'action expert, M = 4, last Euler step')
    else:
        raise SystemExit('usage: chain_timeline.py euler [reps]')


if __name__ == '__main__':
    main()
