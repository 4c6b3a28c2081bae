"""Within-process interleaved A/B of Euler-phase kernel options on the whole chunk (guide 5.4 rule 24: N variants x M rounds in ONE process).

    python tools/micro/ab_chunk.py none gu16 gu16,qkv16 ...      # each argument = one VLASER_EULER option string

Every variant is its own PiZeroInference (own packed weights, own HIP graph) on the same synthetic checkpoint and inputs; rounds replay
the variants in turn (20 chunks each) and the table reports median / min ms per chunk per variant + the per-phase split (HIP events,
bench._phases) + max |action - action of variant 0| (the variants must agree to bf16 tolerance; 16-row units are bit-identical)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import bench
    from vlaser_amd import config as C, synth
    from vlaser_amd.pizero import PiZeroInference
    variants = sys.argv[1:] or ['none', 'gu16,qkv16']
    rounds = int(os.environ.get('AB_ROUNDS', 5))
    torch.set_grad_enabled(False)
    vla = C.VLAConfig(base=C.vlaser_2b())
    dev = 'cuda:0'
    sd = synth.vla_state_dict(vla, device=dev, dtype=torch.bfloat16)
    ids, pv, proprio, noise = bench.make_inputs(vla.base, 1, seed=0)
    ids_d, pv_d, pro_d, noise_d = ids.to(dev), pv.to(dev).to(torch.bfloat16), proprio.to(dev), noise.to(dev)
    valid = (ids != vla.base.pad_token_id).sum(-1).to(dev)
    models, outs = [], []
    for v in variants:
        m = PiZeroInference(vla, device=dev, max_batch=1, euler_opts=v)
        m.load_state_dict(sd)
        call = lambda m=m: m.infer_action(ids_d, pv_d, proprios=pro_d, noise=noise_d, valid_len=valid)
        for _ in range(3):
            o = call()
        torch.cuda.synchronize()
        models.append((m, call))
        outs.append(o.float().cpu())
        print(f'variant {v!r}: ready, max|action - variant0| = {(outs[-1] - outs[0]).abs().max().item():.3e}', flush=True)
    times = [[] for _ in variants]
    n = 20
    for r in range(rounds):
        for i, (m, call) in enumerate(models):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                call()
            torch.cuda.synchronize()
            times[i].append((time.perf_counter() - t0) / n * 1e3)
    print(f'\n| variant | median ms/chunk | min | all rounds |')
    print('|---|---|---|---|')
    for v, ts in zip(variants, times):
        s = sorted(ts)
        print(f'| `{v}` | {s[len(s) // 2]:.3f} | {s[0]:.3f} | {" ".join(f"{t:.3f}" for t in ts)} |')
    if os.environ.get('AB_PHASES', '1') == '1':
        print('\n| variant | ViT+proj ms | prefill ms | Euler ms | us / layer-step |')
        print('|---|---|---|---|---|')
        for v, (m, _) in zip(variants, models):
            vit = bench._graph_ms(lambda: m._run_vit(1))
            pre = bench._graph_ms(lambda: m._run_prefill(1))
            eul = bench._graph_ms(lambda: m._run_euler(1))
            nls = m.cfg.expert.num_hidden_layers * m.num_inference_steps
            print(f'| `{v}` | {vit:.3f} | {pre:.3f} | {eul:.3f} | {eul * 1e3 / nls:.2f} |')


if __name__ == '__main__':
    main()
