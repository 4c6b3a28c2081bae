"""Where inside one pipelined SFT step (BASELINE configs[4], 1 GPU) the phases sit: device-time stamps of forward end (the loss kernel), each gradient
bucket's completion in the backward, and each bucket's AdamW on the optimizer stream, relative to the step's first launch.
    python tools/micro/sft_timeline.py"""
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from vlaser_amd import config as C, synth, ops
    from vlaser_amd import sft as sft_mod
    from vlaser_amd.sft import SFTModel
    torch.set_grad_enabled(False)
    dev = 'cuda:0'
    cfg = C.vlaser_2b()
    sd = synth.vlm_state_dict(cfg, device=dev, dtype=torch.bfloat16)
    m = SFTModel(cfg, device=dev, max_seq_len=576)
    m.load_state_dict(sd)
    del sd
    g = torch.Generator().manual_seed(1000)
    S, R = 560, 128
    ids = torch.cat([torch.randint(1, 151643, (41,), generator=g), torch.full((256,), cfg.img_context_token_id), torch.randint(1, 151643, (S - 41 - 256,), generator=g)])[None]
    labels = torch.full_like(ids, -100)
    labels[0, -R:] = ids[0, -R:]
    pv = torch.randn(1, 3, 448, 448, generator=g).to(dev).to(torch.bfloat16)
    for _ in range(4):
        m.step(pv, ids, labels)
    torch.cuda.synchronize()

    marks = []

    def stamp(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()                       # on the CURRENT stream (main or optimizer)
        marks.append((name, e))

    real_ce, real_adamw, real_norm, real_wait = ops.ce_rows, ops.adamw_clipped, m._norm_bucket, m._wait_params
    nb = len(m.buckets)
    state = {'adamw': 0, 'bucket': 0}

    def ce(*a, **k):
        r = real_ce(*a, **k); stamp('forward end (loss kernel)'); return r

    def adamw(*a, **k):
        r = real_adamw(*a, **k); state['adamw'] += 1; stamp(f'AdamW {state["adamw"]}/{nb} done (optimizer stream)'); return r

    def norm_bucket(b):
        stamp(f'backward: bucket {b} complete'); return real_norm(b)

    def wait_params(b):
        r = real_wait(b); stamp(f'forward: waited for bucket {b}'); return r

    ops.ce_rows, ops.adamw_clipped, m._norm_bucket, m._wait_params = ce, adamw, norm_bucket, wait_params
    NSTEP = 5
    for it in range(NSTEP):                      # NO sync between the steps: the pipelined schedule bench.py measures
        state['adamw'] = 0
        stamp(f'[{it}] step start')
        cur = it
        n0 = len(marks)
        m.step(pv, ids, labels)
        for i in range(n0, len(marks)):
            marks[i] = (f'[{it}] ' + marks[i][0], marks[i][1])
        stamp(f'[{it}] step end (main stream)')
    torch.cuda.synchronize()
    t0 = [e for n, e in marks if n == f'[{NSTEP - 2}] step start'][0]
    t1 = [e for n, e in marks if n == f'[{NSTEP - 1}] step start'][0]
    print(f'pipelined steps, device ms relative to the start of step {NSTEP - 2}; step period = {t0.elapsed_time(t1):.2f} ms')
    rows = [(n, t0.elapsed_time(e)) for n, e in marks]
    last = None
    for n, t in sorted(rows, key=lambda r: r[1]):
        if t < -9 or t > t0.elapsed_time(t1) + 1:
            continue
        short = re.sub(r'waited for bucket (\d+)', r'waited for bucket \1', n)
        if 'waited for bucket' in n and last is not None and last[0] == n:
            continue
        last = (n, t)
        print(f'  {t:8.2f}  {n}')
    ops.ce_rows, ops.adamw_clipped = real_ce, real_adamw


if __name__ == '__main__':
    main()
