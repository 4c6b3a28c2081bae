#!/usr/bin/env python3
"""Headline benchmark of the Vlaser hot path on MI355X (contract: one JSON line on rank 0).

    python bench.py --gpus N --steps K --warmup W [--workload vla_chunk]

Workload `vla_chunk` = BASELINE.json configs[2]: Vlaser-2B-VLA image -> action-chunk forward (one 448x448 WidowX
observation, 384-token prompt with 277 valid tokens, 7-DoF x 4-step chunk, 10 flow-matching Euler steps), batch 1,
bf16 storage / fp32 accumulate, synthetic inputs and deterministic random-init weights of the true architecture
(SURVEY.md 8d).  A "step" is one full infer_action() call with inputs already resident in HBM.
Inference does not shard: N > 1 runs N independent replicas (one process per GPU, no data-path collective) and
`value` is the whole-job rate = N*K chunks / max-over-ranks time ("scaling": "weak").

Extra objects on the JSON line (tier contract 4):
  roofline     : dominant kernel, algorithmic bytes per launch / HIP-event time per launch vs 8 TB/s HBM peak
  cpu_baseline : this repo's CPU oracle (a port of the reference's fp32 CPU path) timed on the host cores on a
                 bounded, depth-truncated sample and scaled linearly in depth (stated in `sample`)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def make_inputs(cfg, B, seed=0):
    g = torch.Generator().manual_seed(seed)
    pv = torch.randn(B, 3, 448, 448, generator=g)
    ids = torch.full((B, 384), cfg.pad_token_id)
    ids[:, :10] = torch.randint(0, 151643, (B, 10), generator=g)
    ids[:, 10:266] = cfg.img_context_token_id
    ids[:, 266:277] = torch.randint(0, 151643, (B, 11), generator=g)
    proprio = torch.rand(B, 1, 7, generator=g) * 2 - 1
    noise = torch.randn(B, 4, 7, generator=g)
    return ids, pv, proprio, noise


def _cpu_name():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown CPU'


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def cpu_baseline(vla_full, seed=0):
    """this repo's CPU oracle (the fp32 torch-CPU restatement of the reference path, pinned against the
    reference's own outputs by the golden fixtures) on the host cores, same synthetic inputs as the GPU run: 1 warm-up + 3 timed
    runs, median, on min(host cores, 32) threads (r02 also timed all 256 threads of the GPU box's host: the oracle's small matmuls
    collapse there -- 0.001 chunks/s -- and that leg cost a minute of every bench run; dropped).  Bounded sample (~10 s of CPU work):
    full widths, a QUARTER of the depth (6 of 24 ViT layers, 7 of 28 LLM + expert layers: generating fp32 weights for more costs more wall
    time than timing them), all 10 Euler steps; phase times scaled x4 to full depth (r02 sampled 2 layers and 2 steps)."""
    from vlaser_amd import config as C, synth
    from oracle import vla as ovla
    import torch.nn.functional as F
    dv, dl, de = 6, 7, 10                     # ViT layers, LLM/expert layers, Euler steps in the sample
    cfg = C.truncated(vla_full.base, dv, dl)
    vla = C.VLAConfig(base=cfg)
    sd = synth.vla_state_dict(vla)
    ids, pv, proprio, noise = make_inputs(cfg, 1, seed)
    am = (ids != cfg.pad_token_id).long()
    mask, vp, pp, ap = ovla.build_causal_mask_and_position_ids(am, torch.float32, vla)
    m1, m2 = ovla.split_full_mask_into_submasks(mask, vla)
    nv, nl, ne = vla_full.base.vision.num_hidden_layers, vla_full.base.llm.num_hidden_layers, vla_full.num_inference_steps

    def run():
        t0 = time.perf_counter()
        emb = ovla.embed_image_text(sd, vla, ids, pv)
        t1 = time.perf_counter()
        caches = {'vlm': [], 'proprio': []}
        pro = F.linear(proprio, sd['proprio_encoder.weight'], sd['proprio_encoder.bias'])
        ovla.joint_forward(sd, vla, {'vlm': emb, 'proprio': pro}, {'vlm': vp, 'proprio': pp}, m1, caches)
        t2 = time.perf_counter()
        a = noise.clone()
        for s in range(de):
            temb = ovla.sinusoidal_pos_emb(torch.full((1,), s * 0.1), vla.action_hidden_size, vla.time_max_period)
            ae = ovla.action_encoder(sd, a, temb)
            out = ovla.joint_forward(sd, vla, {'action': ae}, {'action': ap}, m2, caches, final_skip=())['action']
            a = a + 0.1 * F.linear(out, sd['action_decoder.weight'], sd['action_decoder.bias'])
        t3 = time.perf_counter()
        return t1 - t0, t2 - t1, t3 - t2

    res = {}
    with torch.no_grad():
        for threads in (min(os.cpu_count(), 32),):
            torch.set_num_threads(threads)
            run()                       # warm-up
            ts = [run() for _ in range(3)]
            tv, tp, te = [_median([t[i] for t in ts]) for i in range(3)]
            res[threads] = (tv * nv / dv + tp * nl / dl + te * (nl / dl) * (ne / de), tv, tp, te)
    best = min(res, key=lambda k: res[k][0])
    full, tv, tp, te = res[best]
    others = '; '.join(f'{k} threads -> {1.0 / v[0]:.3f} chunks/s' for k, v in res.items())
    return {'value': round(1.0 / full, 4), 'unit': 'action-chunks/s', 'cores': best, 'kind': 'port',
            'sample': f'oracle fp32 torch-CPU on {_cpu_name()} ({os.cpu_count()} host cores); 1 warm-up + 3 runs, median; full widths, ViT {dv}/{nv} layers, '
                      f'LLM/expert {dl}/{nl} layers, {de}/{ne} Euler steps, batch 1; phase times scaled linearly '
                      f'(median {tv:.2f}+{tp:.2f}+{te:.2f} s -> est. {full:.1f} s/chunk at {best} threads); {others}'}


def sft_cpu_baseline(cfg_full, S=560, R=128):
    """CPU baseline of the SFT step: forward + torch-autograd backward through the fp32 oracle + torch AdamW on
    a depth-truncated, full-width model (2 of 24 ViT layers frozen, 2 of 28 LLM layers, full 151 674-row head and embedding),
    S = 560 with 128 supervised positions; the ViT time is scaled by 24/2, the LLM-layer time by 28/2 (measured as the difference of a
    2-layer and a 1-layer run), head + embedding + projector counted once, AdamW by parameter count.  1 warm-up + 3 runs, median."""
    from vlaser_amd import config as C, synth
    from oracle import vlm as ovlm, vit as ovit
    threads = min(os.cpu_count(), 32)
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(1000)
    ids = torch.cat([torch.randint(1, 151643, (41,), generator=g), torch.full((256,), cfg_full.img_context_token_id),
                     torch.randint(1, 151643, (S - 41 - 256,), generator=g)])[None]
    labels = torch.full_like(ids, -100)
    labels[0, -R:] = ids[0, -R:]
    pv = torch.randn(1, 3, 448, 448, generator=g)

    def timed(dl):
        cfg = C.truncated(cfg_full, 2, dl)
        sd = synth.vlm_state_dict(cfg)
        params = [k for k in sd if k.startswith(('language_model.', 'mlp1.'))]
        for k in params:
            sd[k] = sd[k].clone().requires_grad_(True)
        opt = torch.optim.AdamW([sd[k] for k in params], lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
        n_params = sum(sd[k].numel() for k in params)

        def step():
            t0 = time.perf_counter()
            with torch.no_grad():
                ovit.vision_forward(sd, cfg.vision, pv)
            t1 = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            with torch.enable_grad():
                loss = ovlm.sft_loss(ovlm.forward_logits(sd, cfg, pv, ids), labels)
                loss.backward()
            t2 = time.perf_counter()
            opt.step()
            t3 = time.perf_counter()
            return t1 - t0, t2 - t1, t3 - t2
        step()
        ts = [step() for _ in range(3)]
        return [_median([t[i] for t in ts]) for i in range(3)] + [n_params]

    tv2, tfb2, to2, np2 = timed(2)
    tv1, tfb1, to1, np1 = timed(1)
    nv, nl = cfg_full.vision.num_hidden_layers, cfg_full.llm.num_hidden_layers
    t_layer = max(tfb2 - tfb1, 1e-6)                       # one LLM layer, forward + backward
    t_fixed = tfb1 - tv1                                   # ONE layer + head + embedding + projector (the frozen 2-layer ViT forward inside forward_logits removed)
    per_param = to2 / np2
    n_full = np2 + (np2 - np1) * (nl - 2)
    full = tv2 * nv / 2 + t_fixed + t_layer * (nl - 1) + per_param * n_full
    return {'value': round(S / full, 2), 'unit': 'tokens/s', 'cores': threads, 'kind': 'port',
            'sample': f'oracle fp32 torch-CPU autograd + torch AdamW on {_cpu_name()} ({threads} of {os.cpu_count()} host cores); 1 warm-up + 3 runs, median; full widths, ViT 2/{nv} '
                      f'layers, LLM 2 and 1 of {nl} layers (layer time = difference {t_layer:.2f} s; 1 layer + head/embedding/projector {t_fixed:.2f} s), AdamW '
                      f'{per_param * 1e9:.2f} ns/param x {n_full / 1e9:.2f} B params -> est. {full:.1f} s per S={S} step'}


def sft_flops(cfg, S, R, n_tiles):
    """FLOPs one rank actually executes per optimizer step (1 MAC = 2 FLOP, full S x S attention counted): frozen ViT forward; projector
    forward (the GELU epilogue also keeps the pre-activation for its backward: ONE evaluation of the first Linear, r03) + dgrad + wgrad on the 256 visual rows per
    tile; every LLM matmul forward + dgrad + wgrad on S rows; attention forward + the five S x S products of its backward; lm_head
    forward + dgrad + wgrad on the R supervised rows ONLY (the step never forms logits of unlabelled positions)."""
    v, l = cfg.vision, cfg.llm
    P = v.num_positions
    vit = n_tiles * v.num_hidden_layers * (2 * P * v.hidden_size * (4 * v.hidden_size + 2 * v.intermediate_size) + 4 * P * P * v.hidden_size)
    vit += n_tiles * 2 * v.num_patches * v.hidden_size * 3 * v.patch_size ** 2
    nt, C4, H = n_tiles * cfg.num_image_token, 4 * v.hidden_size, l.hidden_size
    proj = 2 * nt * (C4 * H * (1 + 2) + H * H * (1 + 2))
    nqd, nkvd = l.num_attention_heads * l.head_dim, l.num_key_value_heads * l.head_dim
    per_layer = H * (nqd + 2 * nkvd) + nqd * H + 3 * H * l.intermediate_size
    llm = l.num_hidden_layers * 3 * 2 * S * per_layer
    attn = l.num_hidden_layers * (4 + 10) * S * S * nqd
    head = 3 * 2 * R * l.vocab_size * H
    return float(vit + proj + llm + attn + head)


def sft_bench(rank, world, local, dist, steps, warmup=2):
    """BASELINE configs[4]: Vlaser-2B SFT, data parallel, per-GPU micro-batch 1, T = 1 tile, S = 560 (48 + 256 image + 256 text),
    labels on the last 128 positions, ViT frozen, layer activations KEPT (1.1 GB; the reference recomputes them to fit 80 GB parts --
    `SFTModel(recompute=True)` restores that policy), bf16 params / fp32 AdamW, ZeRO-1 RCCL exchange.  ids / labels are handed over
    as CPU tensors, as the reference's collator produces them: the step then runs without a host<->device round trip."""
    from vlaser_amd import config as C, synth
    from vlaser_amd.sft import SFTModel
    dev = f'cuda:{local}'
    cfg = C.vlaser_2b()
    sd = synth.vlm_state_dict(cfg, device=dev, dtype=torch.bfloat16)
    model = SFTModel(cfg, device=dev, max_seq_len=576, process_group=(dist.group.WORLD if dist is not None else None))
    model.load_state_dict(sd)
    del sd
    torch.cuda.empty_cache()
    g = torch.Generator().manual_seed(1000 + rank)
    S, R = 560, 128
    ids = torch.cat([torch.randint(1, 151643, (41,), generator=g), torch.full((256,), cfg.img_context_token_id),
                     torch.randint(1, 151643, (S - 41 - 256,), generator=g)])[None]
    labels = torch.full_like(ids, -100)
    labels[0, -R:] = ids[0, -R:]
    pv = torch.randn(1, 3, 448, 448, generator=g).to(dev).to(torch.bfloat16)
    out = None
    exchange_info = {}
    if dist is not None:
        # the first collective of the process builds the communicator (ring / direct choice, xGMI links): its wall time and the RCCL version on the
        # line let a first 8-GPU run be read without a second one
        t = torch.ones(1024, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dist.all_reduce(t)
        torch.cuda.synchronize()
        first_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        dist.all_reduce(t)
        torch.cuda.synchronize()
        try:
            ver = '.'.join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:          # noqa: BLE001 -- a missing version query must not cost the run
            ver = f'unavailable ({type(e).__name__})'
        exchange_info = {'rccl_version': ver, 'first_collective_ms': round(first_ms, 2), 'second_collective_ms': round((time.perf_counter() - t0) * 1e3, 3)}
    exchange_info.update(model.exchange_info())            # mode (none | pg | capi), CU-mask widths of the capi mode
    for _ in range(warmup):
        out = model.step(pv, ids, labels)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = model.step(pv, ids, labels)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss = float(out.loss)
    assert loss == loss, 'SFT loss is NaN'
    # forward + backward alone (no gradient norm, no AdamW, no exchange): the part of the step an N-GPU ZeRO-1 rank runs unchanged (its AdamW is 1/N)
    model.wait_optimizer()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.forward_backward(pv, ids, labels)
    torch.cuda.synchronize()
    fwd_bwd_ms = (time.perf_counter() - t0) / steps * 1e3
    buckets_mb = [round((hi - lo) * 2 / 2 ** 20) for lo, hi in model.buckets]
    side = {}
    if world == 1 and dist is None and os.environ.get('VLASER_BENCH_NO_SFT_SIDE') != '1':
        # (a) per-device batch 4, the reference launcher's PER_DEVICE_BATCH_SIZE (internvl3_2b_dynamic_res_2nd_finetune_full.sh:47-67): one optimizer step over four samples.
        #     The kernels run one sample at a time (M = 560 rows each, fp32 gradient accumulation); AdamW and the norm are paid once per 2 240 tokens
        try:
            g4 = torch.Generator().manual_seed(2000)
            ids4 = torch.stack([torch.cat([torch.randint(1, 151643, (41,), generator=g4), torch.full((256,), cfg.img_context_token_id),
                                           torch.randint(1, 151643, (S - 41 - 256,), generator=g4)]) for _ in range(4)])
            lab4 = torch.full_like(ids4, -100)
            lab4[:, -R:] = ids4[:, -R:]
            pv4 = torch.randn(4, 3, 448, 448, generator=g4).to(dev).to(torch.bfloat16)
            model.wait_optimizer()
            for _ in range(2):
                o4 = model.train_step([(pv4, ids4, lab4)])
            torch.cuda.synchronize()
            n4 = max(steps // 2, 3)
            t0 = time.perf_counter()
            for _ in range(n4):
                o4 = model.train_step([(pv4, ids4, lab4)])
            torch.cuda.synchronize()
            d4 = (time.perf_counter() - t0) / n4
            assert float(o4.loss) == float(o4.loss)
            side['micro_batch4'] = {'ms_per_step': round(d4 * 1e3, 2), 'tokens_per_s': round(4 * S / d4, 1), 'tokens_per_step': 4 * S,
                                    'note': 'per-device batch 4 (…2nd_finetune_full.sh:47-67), one optimizer step; samples run one at a time (M = 560 rows per GEMM, not 2 240), '
                                            'weighted fp32 gradient accumulation, AdamW once per step'}
        except Exception as e:          # noqa: BLE001 -- a side number
            side['micro_batch4'] = {'error': f'{type(e).__name__}: {e}'[:200]}
    del model
    torch.cuda.empty_cache()
    if world == 1 and dist is None and os.environ.get('VLASER_BENCH_NO_SFT_SIDE') != '1':
        # (b) the reference's activation policy: grad_checkpoint on (BASELINE configs[4], SURVEY 8d, …2nd_finetune_full.sh:46) -- every layer's forward re-run inside the
        #     backward instead of 1.1 GB of kept activations; same values bit for bit (tests/test_sft_gpu.py)
        try:
            sd = synth.vlm_state_dict(cfg, device=dev, dtype=torch.bfloat16)
            mr = SFTModel(cfg, device=dev, max_seq_len=576, recompute=True)
            mr.load_state_dict(sd)
            del sd
            for _ in range(warmup):
                mr.step(pv, ids, labels)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                orc = mr.step(pv, ids, labels)
            torch.cuda.synchronize()
            dr = (time.perf_counter() - t0) / steps
            mr.wait_optimizer()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                mr.forward_backward(pv, ids, labels)
            torch.cuda.synchronize()
            side['recompute_ms_per_step'] = round(dr * 1e3, 2)
            side['recompute_fwd_bwd_ms'] = round((time.perf_counter() - t0) / steps * 1e3, 2)
            side['recompute_note'] = 'SFTModel(recompute=True): per-layer activation recompute as the reference\'s grad_checkpoint; the headline sft line keeps activations (1.1 GB of 288 GB)'
            assert float(orc.loss) == float(orc.loss)
            del mr
            torch.cuda.empty_cache()
        except Exception as e:          # noqa: BLE001
            side['recompute_ms_per_step'] = None
            side['recompute_error'] = f'{type(e).__name__}: {e}'[:200]
    fl = sft_flops(cfg, S, R - 1 + 1, 1)
    return {'metric': 'sft_tokens_per_sec', 'value': round(world * steps * S / dt, 1), 'unit': 'tokens/s', 'ms_per_step': round(dt / steps * 1e3, 2),
            'steps': steps, 'tokens_per_rank_step': S, 'last_loss': round(loss, 4), 'fwd_bwd_ms': round(fwd_bwd_ms, 2),
            'parallelism': f'dp{world} (ZeRO-1: bucketed RCCL reduce-scatter(mean, bf16) issued from the backward + all-gather of updated params)',
            'exchange': {'bucket_mib': buckets_mb, 'NCCL_ALGO': os.environ.get('NCCL_ALGO', 'default'), 'NCCL_PROTO': os.environ.get('NCCL_PROTO', 'default'),
                         'NCCL_MAX_NCHANNELS': os.environ.get('NCCL_MAX_NCHANNELS', 'unset = RCCL default (chosen: profiles/r05_rccl_contention.md section 4)'),
                         # one-GPU stand-in for RCCL's channel workgroups (tools/micro/rccl_shadow_lab.py): forward + backward stretch beside C resident streaming workgroups
                         'contention_model': {'source': 'profiles/r05_rccl_contention.md', 'fwd_bwd_stretch_by_channel_workgroups': {'1': 1.13, '8': 1.23, '16': 1.28, '32': 1.34, '64': 1.46},
                                              'both_streams_cu_masked': 1.13, 'predicted_fwd_bwd_ms_at_8_gpus': [16.4, 17.7]},
                         'gradient_bytes_per_rank': 3570e6 if world > 1 else 0, **exchange_info},
            'gflop_per_rank_step': round(fl / 1e9, 1), 'mfma_frac': round(fl * world * steps / dt / (world * 2.5e15), 4),
            'fwd_bwd_mfma_frac': round(fl / (fwd_bwd_ms * 1e-3) / 2.5e15, 4), **side}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='both', choices=['vla_chunk', 'sft', 'both'])
    ap.add_argument('--sft-steps', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-8b', action='store_true', help='skip the Vlaser-8B x 13-tile side numbers (BASELINE configs[3])')

    ap.add_argument('--dry-run', action='store_true', help='rendezvous only (gloo, no GPU work): checks the N-rank launch path')
    a = ap.parse_args()

    if a.gpus > 1 and 'RANK' not in os.environ:
        # `python bench.py --gpus N`: this parent never touches the GPU; it starts N ranks (one process per GPU) and relays rank 0's line
        sys.exit(_spawn_ranks(a.gpus))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != a.gpus:
        raise SystemExit(f'bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus} (or run `python bench.py --gpus N`)')
    if a.dry_run:
        return _dry_run(rank, world)
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or os.environ.get('VLASER_FORCE_DP') == '1':      # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        for k, v in (('MASTER_ADDR', '127.0.0.1'), ('MASTER_PORT', '29533'), ('RANK', '0'), ('WORLD_SIZE', '1')):
            os.environ.setdefault(k, v)
        dist.init_process_group('nccl')      # RCCL on ROCm: barrier, max-over-ranks time, and the SFT gradient exchange

    torch.set_grad_enabled(False)
    sft_line = None
    if a.workload == 'sft':
        sft_line = sft_bench(rank, world, local, dist, a.sft_steps)
        if rank == 0:
            sft_line.update({'n_gpus': world, 'warmup': 2, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16',
                             'data': 'synthetic', 'config': {'workload': 'Vlaser-2B SFT, per-GPU micro-batch 1, S=560 (BASELINE configs[4])'}})
        _finish(dist, sft_line if rank == 0 else None)
        return
    from vlaser_amd import config as C, synth
    from vlaser_amd.pizero import PiZeroInference
    vla = C.VLAConfig(base=C.vlaser_2b())
    dev = f'cuda:{local}'
    sd = synth.vla_state_dict(vla, device=dev, dtype=torch.bfloat16)
    model = PiZeroInference(vla, device=dev, max_batch=1)
    model.load_state_dict(sd)
    del sd
    torch.cuda.empty_cache()
    ids, pv, proprio, noise = make_inputs(vla.base, 1, seed=rank)
    # inputs resident in HBM before the timed region.  The timed call is the REFERENCE's call (pizero_internvl.py:798-808): the eight tensors exactly as
    # Vlaser_VLA/Simpler/src/agent/eval.py:110-130 builds them per control step and moves them to the device -- dense image_text_proprio_mask / action_mask,
    # three position-id tensors -- plus the explicit noise; the result is a fresh tensor per call, as the reference returns
    ids_d, pv_d, pro_d, noise_d = ids.to(dev), pv.to(dev).to(torch.bfloat16), proprio.to(dev), noise.to(dev)
    valid = (ids != vla.base.pad_token_id).sum(-1).to(dev)
    mask, vp, pp, ap = model.build_causal_mask_and_position_ids((ids != vla.base.pad_token_id).long(), torch.bfloat16)
    m1, m2 = model.split_full_mask_into_submasks(mask)
    m1_d, m2_d, vp_d, pp_d, ap_d = (t.to(dev) for t in (m1, m2, vp, pp, ap))
    call = lambda: model.infer_action(ids_d, pv_d, m1_d, m2_d, vp_d, pp_d, ap_d, pro_d, noise=noise_d)
    call_ext = lambda: model.infer_action(ids_d, pv_d, proprios=pro_d, noise=noise_d, valid_len=valid)      # the `valid_len=` extension (no dense masks)

    for _ in range(max(a.warmup, 1)):
        out = call()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = call()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    assert torch.isfinite(out).all()
    model.check_errors()                                             # the device-side mask check of every timed call came back clean

    def _time_calls(fn, n):
        fn(); torch.cuda.synchronize()
        t0_ = time.perf_counter()
        for _ in range(n):
            o_ = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0_) / n * 1e3, o_

    line = None
    if rank == 0:
        # side timings of the same chunk through the two other call forms (never `value`): the valid_len extension with a fresh result, and with the
        # zero-copy result ring switched on (output_ring = 4: a view that is overwritten four calls later)
        ext_ms, out_ext = _time_calls(call_ext, a.steps)
        model.output_ring = 4
        ring_ms, _ = _time_calls(call_ext, a.steps)
        model.output_ring = 0
        assert torch.equal(out_ext, out)
        line = {
            'metric': 'action_chunks_per_sec', 'value': round(world * a.steps / dt, 3), 'unit': 'action-chunks/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'Vlaser-2B-VLA action-chunk inference: 1x448px image, 384-token prompt (277 valid), '
                                   '7-DoF x 4-step chunk, 10 Euler steps, batch 1 per GPU (BASELINE configs[2])',
                       'parallelism': f'replicas x{world} (no collective)', 'weights': 'random-init, true architecture',
                       'call': 'reference signature: infer_action(input_ids, pixel_values, image_text_proprio_mask, action_mask, vlm / proprio / action position ids, '
                               'proprios) + noise, all on the device (eval.py:110-130); fresh result tensor per call',
                       'reference_mode': 'use_bf16=True (bf16 storage / fp32 accumulate: the mode the reference trains in, slurm/train_internvl.sh:35; its shipped eval '
                                         'config runs fp32, bridge_internvl_448.yaml:36)'},
        }
        line['reference_signature'] = {'ms_per_chunk': round(dt / a.steps * 1e3, 4), 'is_headline': True,
                                       'valid_len_extension_ms': round(ext_ms, 4), 'valid_len_extension_output_ring4_ms': round(ring_ms, 4),
                                       'note': 'the dense masks are checked and valid_len counted ON the device by the one staging launch (vlaser_vla_stage, ABI 6); '
                                               'no host copy, no sync inside the call'}
        if not a.no_roofline:
            iso_ms, byts, n_iso = _probe(model)
            phases, inchain_us, n = _phases(model)
            avg_ms = inchain_us * 1e-3
            ach = byts / (avg_ms * 1e-3) / 1e9
            line['phases'] = phases
            phases['call_overhead_ms'] = round(dt / a.steps * 1e3 - phases['chunk_graph_ms'], 4)     # staging launch + graph launch gap per infer_action call
            # whole-chunk floor: 1 726 GFLOP of ViT + projector + joint prefill on MFMA (2.5 PFLOP/s dense bf16) + 10 Euler steps x
            # 1.310 GB of expert weights (+ 11.2 MB of K/V each) over HBM (8 TB/s) = 0.690 + 1.652 ms
            floor_ms = 1726e9 / 2.5e15 * 1e3 + 10 * (1.310e9 + 11.2e6) / 8e12 * 1e3
            line['chunk_roofline'] = {'floor_ms': round(floor_ms, 3), 'ms_per_chunk': round(dt / a.steps * 1e3, 3), 'frac': round(floor_ms / (dt / a.steps * 1e3), 4),
                                      'note': 'MFMA-bound part at 2.5 PFLOP/s + HBM-bound Euler part at 8 TB/s; the Euler phase is a chain of 1 400 dependent launches '
                                              '(5 per layer-step), see DESIGN.md section 3'}
            traffic, traffic_src = _pmc_traffic()
            # two clocks for the same launch: the IN-CHAIN figure measured live (graph with the kernel - graph without it: what the chunk pays for it,
            # its boundary included) and the kernel's own duration as rocprofv3 reports it in the committed kernel trace of this tree.  Neither bounds the other (the
            # in-chain difference can come out BELOW the rocprof duration: removing the kernel also removes the cache state it leaves its successor); `frac` is quoted
            # on the LARGER of the two, i.e. the lower fraction
            rp_us, rp_src = _rocprof_duration()
            us_used = max(avg_ms * 1e3, rp_us or 0.0)
            ach = byts / (us_used * 1e-6) / 1e9
            line['roofline'] = {'bound': 'hbm', 'kernel': 'chain_gu_kernel (csrc/chain.hip: action-expert gate/up GEMV, residual + split-K reduce + RMSNorm -> SwiGLU, N=17920 K=768, M=4; '
                                                          '280 launches per chunk)',
                                'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4),
                                'traffic': traffic, 'traffic_source': traffic_src,
                                'bytes_per_launch': byts, 'us_per_launch': round(us_used, 3), 'launches_timed': n,
                                'us_per_launch_in_chain': round(avg_ms * 1e3, 3), 'us_per_launch_rocprof': rp_us, 'rocprof_source': rp_src,
                                'timing': 'us_per_launch = max(in-chain, rocprof).  IN-CHAIN: (Euler-phase graph with the kernel - the same graph without it) / launches, HIP events '
                                          'on the launch stream, measured in this run.  ROCPROF: average duration of the kernel in the committed `rocprofv3 --kernel-trace --stats` '
                                          'summary of this tree (profiles/).  The two differ by a few per cent in either direction',
                                'us_per_launch_isolated': round(iso_ms * 1e3, 3), 'achieved_isolated': round(byts / (iso_ms * 1e-3) / 1e9, 1)}
        if world == 1 and not a.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(vla)
    # ---- SFT line (BASELINE configs[4]) AFTER the headline has been timed: the gradient exchange is the only part of this file that
    # has never run on more than one physical GPU, so it must not be able to take the headline down with it.  Any rank that fails
    # skips ahead; rank 0 runs a watchdog that prints the headline line and leaves if the exchange hangs on a missing peer.
    if a.workload == 'both':
        del model
        torch.cuda.empty_cache()
        watchdog = None
        if world > 1:
            import threading

            def _bail():
                if rank == 0:
                    line['sft'] = {'error': 'SFT sub-bench did not finish within 420 s (a rank failed or the exchange hung); headline line printed by the watchdog'}
                    print(json.dumps(line), flush=True)
                os._exit(3)                                      # the headline is out, but a hung exchange / dead peer is a FAILED run: non-zero on every rank
            watchdog = threading.Timer(420.0 if rank == 0 else 450.0, _bail)
            watchdog.daemon = True
            watchdog.start()
        try:
            sft_line = sft_bench(rank, world, local, dist, a.sft_steps)
        except Exception as e:                                   # noqa: BLE001  (reported in the line, never fatal for the headline)
            sft_line = {'error': f'{type(e).__name__}: {e}'[:400]}
        if watchdog is not None:
            watchdog.cancel()
    if rank == 0:
        if sft_line is not None:
            if world == 1 and dist is None and 'error' not in sft_line and os.environ.get('VLASER_BENCH_NO_FORCED_DP') != '1':
                # both exchange modes: torch's ProcessGroupNCCL, and RCCL's C API on a CU-masked stream with the compute streams on the complement
                sft_line['forced_dp_world1_ms'] = {m: _forced_dp_world1(a.sft_steps, m) for m in ('pg', 'capi')}
            if world == 1 and not a.no_cpu_baseline and 'error' not in sft_line:
                sft_line['cpu_baseline'] = sft_cpu_baseline(vla.base)
            line['sft'] = sft_line
        if world == 1 and a.workload == 'both':
            line['batched'] = batched_chunks(vla, dev, a.steps)
            line['two_in_flight'] = two_in_flight(vla, dev, a.steps)
            line['qa'] = qa_bench(local)
            if not a.no_8b:
                line['qa_8b'] = qa8b_bench(local)
            if os.environ.get('VLASER_BENCH_NO_SFT_SIDE') != '1':
                try:
                    line['vla_train'] = vla_train_bench(local)
                except Exception as e:      # a side number must not take the line down
                    line['vla_train'] = {'error': str(e)[:200]}
    _finish(dist, line if rank == 0 else None)
    if isinstance(sft_line, dict) and 'error' in sft_line:
        sys.exit(3)                                              # the headline line is out; a failed SFT sub-bench is still a failed run


def _forced_dp_world1(steps, mode='pg'):
    """ms per SFT step with the ZeRO-1 exchange forced on at world size 1 (`VLASER_FORCE_DP=1`: every bucket goes through RCCL's reduce_scatter / all_gather with
    itself, the optimizer runs on the comm stream in front of each all-gather -- what an N-GPU rank executes, with N = 1): a CHILD bench process (the process group
    cannot be added to this one after the fact), its own line parsed; None (with the reason) when the child fails."""
    env = dict(os.environ, VLASER_FORCE_DP='1', VLASER_DP_EXCHANGE=mode, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--workload', 'sft', '--sft-steps', str(steps), '--no-cpu-baseline'], env=env, capture_output=True,
                           text=True, timeout=600)
        ln = next((x for x in reversed(r.stdout.splitlines()) if x.startswith('{')), None)
        if ln is None:
            return {'error': f'child rc {r.returncode}: ' + (r.stderr.strip().splitlines()[-1][:160] if r.stderr.strip() else 'no line')}
        d = json.loads(ln)
        return {'ms_per_step': d['ms_per_step'], 'fwd_bwd_ms': d['fwd_bwd_ms'], 'rccl_version': d['exchange'].get('rccl_version'),
                'exchange': {k: d['exchange'].get(k) for k in ('mode', 'comm_cus', 'compute_cus', 'cu_masks') if k in d['exchange']},
                'note': 'the whole model is this rank\'s "shard": all of AdamW sits in front of the all-gathers (1/N of it at N ranks)'}
    except Exception as e:          # noqa: BLE001 -- a side number, never a reason to lose the line
        return {'error': f'{type(e).__name__}: {e}'[:200]}


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _spawn_ranks(n):
    """Parent of `python bench.py --gpus N` (N > 1, no RANK in the environment): start N worker ranks through
    torch.distributed.run as a CHILD process (never an exec: see the GPU-box rules), relay rank 0's JSON line, and return
    the launcher's exit code (non-zero if any rank failed or the SFT watchdog fired, even when the line was printed).  The parent itself makes no GPU call."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout:                                           # streamed: the line is relayed the moment rank 0 prints it
        ln = ln.rstrip('\n')
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
            print(line, flush=True)
        else:
            print(ln, file=sys.stderr)
    rc = p.wait()
    if line is None:
        print(f'bench.py: {n}-rank run failed (rc {rc})', file=sys.stderr)
        return rc or 1
    if rc != 0:
        print(f'bench.py: {n}-rank run printed its line but exited with rc {rc}', file=sys.stderr)
    return rc


def _dry_run(rank, world):
    """No GPU: rendezvous over gloo, count the ranks, print the line's skeleton (tests/test_bench_launch.py)."""
    n = world
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('gloo')
        t = torch.ones(1)
        dist.all_reduce(t)
        n = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({'metric': 'action_chunks_per_sec', 'value': 0.0, 'n_gpus': world, 'ranks_seen': n, 'dry_run': True}), flush=True)
    rc = int(os.environ.get('VLASER_BENCH_DRYRUN_EXIT', '0'))      # test hook: a rank that fails AFTER the line was printed
    if rc and rank == world - 1:
        sys.exit(rc)


def two_in_flight(vla, dev, steps):
    """Side number (never `value`): TWO independent batch-1 requests in flight -- two PiZeroInference instances (own workspaces, KV caches and graphs; same weights' values)
    on two streams, calls alternating -- the serving situation of two robots / environments on one GPU.  The launch-bound Euler phase of one request (few CUs busy)
    runs under the MFMA-bound ViT + prefill of the other.  Throughput only: each request's latency roughly doubles the per-chunk figure."""
    from vlaser_amd import synth
    from vlaser_amd.pizero import PiZeroInference
    sd = synth.vla_state_dict(vla, device=dev, dtype=torch.bfloat16)
    models, streams, inputs = [], [], []
    for i in range(2):
        m = PiZeroInference(vla, device=dev, max_batch=1)
        m.load_state_dict(sd)
        models.append(m)
        streams.append(torch.cuda.Stream(device=dev))
        ids, pv, proprio, noise = make_inputs(vla.base, 1, seed=200 + i)
        inputs.append((ids.to(dev), pv.to(dev).to(torch.bfloat16), proprio.to(dev), noise.to(dev), (ids != vla.base.pad_token_id).sum(-1).to(dev)))
    del sd

    def call(i):
        ids, pv, pro, noise, valid = inputs[i]
        with torch.cuda.stream(streams[i]):
            return models[i].infer_action(ids, pv, proprios=pro, noise=noise, valid_len=valid)

    for i in range(2):
        for _ in range(3):
            call(i)
    torch.cuda.synchronize()
    n = 2 * max(steps // 2, 10)
    t0 = time.perf_counter()
    for k in range(n):
        out = call(k % 2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    del models
    torch.cuda.empty_cache()
    return {'requests_in_flight': 2, 'action_chunks_per_sec': round(n / dt, 1), 'ms_per_chunk': round(dt / n * 1e3, 3),
            'note': 'two independent batch-1 requests on two streams (two model instances); throughput of a 2-client serving loop, NOT the headline: one request at a time is `value`'}


def batched_chunks(vla, dev, steps):
    """Side number (never `value`): the same chunk workload with 4 observations per infer_action call -- the largest batch
    whose 4 x 4 action rows fit the 16-row weight-streaming kernels, i.e. the expert's weights are streamed once for 4 chunks."""
    from vlaser_amd import synth
    from vlaser_amd.pizero import PiZeroInference
    B = 4
    sd = synth.vla_state_dict(vla, device=dev, dtype=torch.bfloat16)
    model = PiZeroInference(vla, device=dev, max_batch=B)
    model.load_state_dict(sd)
    del sd
    ids, pv, proprio, noise = make_inputs(vla.base, B, seed=100)
    ids_d, pv_d, pro_d, noise_d = ids.to(dev), pv.to(dev).to(torch.bfloat16), proprio.to(dev), noise.to(dev)
    valid = (ids != vla.base.pad_token_id).sum(-1).to(dev)
    call = lambda: model.infer_action(ids_d, pv_d, proprios=pro_d, noise=noise_d, valid_len=valid)
    for _ in range(3):
        out = call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = call()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    del model
    torch.cuda.empty_cache()
    return {'batch': B, 'action_chunks_per_sec': round(B * steps / dt, 2), 'ms_per_call': round(dt / steps * 1e3, 3)}


def qa_bench(local):
    """BASELINE configs[1] side numbers ('report QA forward latency as prefill ms + decode tokens/s'): Vlaser-2B,
    one 448 px tile + 256-token prompt (S = 560), greedy decode of 32 tokens, batch 1 and batch 8."""
    from vlaser_amd import config as C, synth
    from vlaser_amd.internvl_chat import InternVLChatModel
    dev = f'cuda:{local}'
    cfg = C.vlaser_2b()
    sd = synth.vlm_state_dict(cfg, device=dev, dtype=torch.bfloat16)
    m = InternVLChatModel(cfg, device=dev, max_seq_len=640, max_batch=8)
    m.load_state_dict(sd)
    del sd
    m.img_context_token_id = cfg.img_context_token_id
    out = {}
    for B in (1, 8):
        g = torch.Generator().manual_seed(7)
        pv = torch.randn(B, 3, 448, 448, generator=g).to(dev).to(torch.bfloat16)
        ids = torch.cat([torch.randint(0, 151643, (B, 41), generator=g), torch.full((B, 256), cfg.img_context_token_id),
                         torch.randint(0, 151643, (B, 263), generator=g)], 1)
        ts = []
        for n_new in (1, 33):                      # time(33 tokens) - time(1 token) = 32 decode steps
            m.generate(pv, ids, max_new_tokens=n_new, min_new_tokens=n_new)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                m.generate(pv, ids, max_new_tokens=n_new, min_new_tokens=n_new)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 3)
        step_ms = (ts[1] - ts[0]) / 32 * 1e3
        llm = cfg.llm
        NQ = (llm.num_attention_heads + 2 * llm.num_key_value_heads) * llm.head_dim
        # bytes one decode step streams: every layer's weights + the lm_head once (shared by the B rows), each row's K / V^T cache (S + the new tokens)
        w_bytes = 2 * (llm.num_hidden_layers * (NQ * llm.hidden_size + llm.hidden_size * llm.num_attention_heads * llm.head_dim + 3 * llm.intermediate_size * llm.hidden_size)
                       + llm.vocab_size * llm.hidden_size)
        kv_bytes = B * llm.num_hidden_layers * 2 * llm.num_key_value_heads * llm.head_dim * (ids.shape[1] + 16) * 2
        out[f'batch{B}'] = {'prefill_ms': round(ts[0] * 1e3, 2), 'decode_tokens_per_s': round(B * 32 / (ts[1] - ts[0]), 1),
                            'decode_ms_per_step': round(step_ms, 3), 'decode_us_per_layer_step': round(step_ms * 1e3 / llm.num_hidden_layers, 2),
                            'decode_roofline': {'bound': 'hbm', 'gbytes_per_step': round((w_bytes + kv_bytes) / 1e9, 3),
                                                'achieved_gbs': round((w_bytes + kv_bytes) / (step_ms * 1e-3) / 1e9, 1),
                                                'frac_of_8000': round((w_bytes + kv_bytes) / (step_ms * 1e-3) / 8e12, 4)}}
    out['config'] = 'Vlaser-2B, 1 tile + 256-token prompt (S=560), greedy, 32 new tokens; prefill_ms = ViT + prefill + first token; decode steps replayed from one HIP graph (device-resident slot / key-count state)'
    del m
    torch.cuda.empty_cache()
    return out


def qa8b_bench(local):
    """BASELINE configs[3]: Vlaser-8B dynamic-high-res grounding request -- 13 tiles (12 + thumbnail) of 448 px, S = 13*256 + 48 + 32 =
    3408 prompt tokens, greedy decode of 32 tokens, batch 1, random-init weights of the true architecture.  Prefill is MFMA-bound
    (58.7 TFLOP: ViT 9 407 + mlp1 183 + LLM 44 476 + attention 4 662 GFLOP), decode is HBM-bound (14.14 GB of weights
    + 28 x 2 x S x 4 x 128 x 2 B of KV per token) on the chunked-K weight-streaming kernels."""
    from vlaser_amd import config as C, synth
    from vlaser_amd.internvl_chat import InternVLChatModel
    dev = f'cuda:{local}'
    cfg = C.vlaser_8b()
    sd = synth.vlm_state_dict(cfg, device=dev, dtype=torch.bfloat16)
    m = InternVLChatModel(cfg, device=dev, max_tiles=13, max_seq_len=3456, max_batch=1)
    m.load_state_dict(sd)
    del sd
    torch.cuda.empty_cache()
    m.img_context_token_id = cfg.img_context_token_id
    g = torch.Generator().manual_seed(9)
    pv = torch.randn(13, 3, 448, 448, generator=g).to(dev).to(torch.bfloat16)
    ids = torch.cat([torch.randint(0, 151643, (1, 41), generator=g), torch.full((1, 13 * 256), cfg.img_context_token_id),
                     torch.randint(0, 151643, (1, 39), generator=g)], 1)
    S = ids.shape[1]
    ts = []
    for n_new in (1, 33):
        m.generate(pv, ids, max_new_tokens=n_new, min_new_tokens=n_new)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            m.generate(pv, ids, max_new_tokens=n_new, min_new_tokens=n_new)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 2)
    del m
    torch.cuda.empty_cache()
    # the request's image preparation (load_image, eval_example.py:38-82): a 12-megapixel frame -> the same 13 tiles, on the device (csrc/image.hip, Pillow-exact) with the
    # decoded frame already in HBM, and through Pillow on this box's host as the reference does it
    prep_side = None
    try:
        import numpy as np
        from PIL import Image
        from vlaser_amd import prep
        from vlaser_amd.image import ImagePrep
        frame = np.random.default_rng(9).integers(0, 256, (3024, 4032, 3), dtype=np.uint8)
        d = torch.from_numpy(frame).to(dev)
        ip = ImagePrep(dev)
        for _ in range(3):
            out = ip.load_image(d, max_num=12)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            out = ip.load_image(d, max_num=12)
        torch.cuda.synchronize()
        dev_us = (time.perf_counter() - t0) / 10 * 1e6
        pil = Image.fromarray(frame)
        t0 = time.perf_counter()
        ref = prep.load_image(pil, 448, 12)
        host_ms = (time.perf_counter() - t0) * 1e3
        prep_side = {'frame': '4032x3024 uint8', 'tiles': int(out.shape[0]), 'device_us': round(dev_us, 1), 'pillow_host_ms': round(host_ms, 1),
                     'bit_exact_vs_pillow': bool(torch.equal(out.cpu(), ref.to(torch.bfloat16))),
                     'note': 'load_image from the decoded frame on: bicubic resize to the 4 x 3 grid + thumbnail, tile cut, ToTensor + Normalize; device time with the frame resident in HBM '
                             '(its PCIe copy is 36.6 MB ~ 0.6-0.9 ms), host time = the reference path through Pillow on this box (one run)'}
    except Exception as e:      # a side number must not take the line down
        prep_side = {'error': str(e)[:200]}
    l = cfg.llm
    per_tok = 2.0 * (l.num_hidden_layers * (l.hidden_size * (l.num_attention_heads + 2 * l.num_key_value_heads) * l.head_dim
                                           + l.num_attention_heads * l.head_dim * l.hidden_size + 3 * l.hidden_size * l.intermediate_size))
    pre_flop = 13 * 723.6e9 + 13 * 14.09e9 + per_tok * S + l.num_hidden_layers * 4.0 * S * S * l.num_attention_heads * l.head_dim + 2.0 * l.vocab_size * l.hidden_size
    dec_bytes = per_tok + 2.0 * l.vocab_size * l.hidden_size + l.num_hidden_layers * 2 * (S + 16) * l.num_key_value_heads * l.head_dim * 2
    step = (ts[1] - ts[0]) / 32
    return {'config': f'Vlaser-8B, 13 tiles, S={S}, greedy, 32 new tokens, batch 1 (BASELINE configs[3]); prefill_ms = ViT + prefill + first token',
            'prefill_ms': round(ts[0] * 1e3, 2), 'decode_tokens_per_s': round(1.0 / step, 1), 'decode_ms_per_step': round(step * 1e3, 3), 'image_prep': prep_side,
            'roofline': {'prefill': {'bound': 'mfma', 'achieved': round(pre_flop / ts[0] / 1e12, 1), 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': round(pre_flop / ts[0] / 2.5e15, 4)},
                         'decode': {'bound': 'hbm', 'achieved': round(dec_bytes / step / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(dec_bytes / step / 1e9 / HBM_PEAK_GBS, 4)}}}


def vla_train_bench(local, steps=6):
    """SURVEY 8f-1 (the other half of "fine-tune" for the VLA): one flow-matching training step of Vlaser-2B-VLA at full depth, per-device batch 1 -- `PiZero.forward`
    (pizero_internvl.py:1064-1197) + backward + the optimizer update of `TrainAgent.run` (train.py:345-636): the action-expert parameter group (the reference's default), and with
    `train_vlm: True` the VLM group as well (ViT + projector + LLM, second optimizer, one clip).  Random-init weights, synthetic sample; fp32 AdamW moments (the reference: bnb 8-bit)."""
    from vlaser_amd import config as C, synth
    from vlaser_amd.vla_train import VLATrainer
    dev = f'cuda:{local}'
    vla = C.VLAConfig(base=C.vlaser_2b())
    sd = synth.vla_state_dict(vla, device=dev, dtype=torch.bfloat16)
    ids, pv, pro, _ = make_inputs(vla.base, 1, seed=5)
    g = torch.Generator().manual_seed(6)
    smp = dict(input_ids=ids, pixel_values=pv, proprios=pro, actions=torch.rand(1, vla.num_action_tokens, vla.action_dim, generator=g) * 2 - 1,
               t=torch.rand(1, generator=g) * 0.999, x0=torch.randn(1, vla.num_action_tokens, vla.action_dim, generator=g))
    out = {'config': 'Vlaser-2B-VLA flow-matching training step, full depth, per-device batch 1, 384-token prompt (277 valid), 4 action tokens; ms per optimizer step'}
    for key, kw in (('action_expert_group_ms', {}), ('with_vlm_group_ms', {'train_vlm': True})):
        m = VLATrainer(vla, device=dev, lr=5e-5, max_grad_norm=1.0, **kw)
        m.load_state_dict(sd)
        for _ in range(2):
            r = m.step([smp])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = m.step([smp])
        torch.cuda.synchronize()
        out[key] = round((time.perf_counter() - t0) / steps * 1e3, 2)
        out[key.replace('_ms', '_last_loss')] = round(float(r.loss), 4)
        del m
        torch.cuda.empty_cache()
    return out


def _pmc_traffic():
    """HBM bytes per launch of the dominant kernel from hardware counters -> (bytes or None, source dict).  rocprofv3 --pmc cannot run under torch on this
    image, so the counters come from the torch-free harness tools/pmc/skinny_pmc (same kernel, same shape, 28 weight buffers cycled), collected as the
    guide prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE `--pmc` passes with `--kernel-trace` only, FETCH_SIZE x 2 on gfx950 (128-byte requests of wide
    coalesced reads are tallied at 64 B).  Measured IN THIS RUN when the harness and rocprofv3 are there (each pass is a child process of its own: the
    profiler never wraps this torch process; ~5 s); otherwise the committed collection profiles/r05g_pmc_dominant_kernel.json."""
    root = os.path.dirname(os.path.abspath(__file__))
    how = 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) on tools/pmc/skinny_pmc (torch-free harness, same kernel and shape); FETCH_SIZE x 2 (gfx950)'
    exe = os.path.join(root, 'tools', 'pmc', 'skinny_pmc')
    import shutil
    import tempfile
    rocprof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if os.path.exists(exe) and os.path.exists(rocprof) and os.environ.get('VLASER_BENCH_NO_LIVE_PMC') != '1':
        try:
            import csv
            import glob
            means = {}
            for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
                with tempfile.TemporaryDirectory(dir='/tmp') as d:
                    # a clean environment for the child profiler: when this process itself runs under rocprofv3 its preload / tool variables must not leak in
                    env = {k: v for k, v in os.environ.items() if not (k.startswith(('ROCP', 'ROCPROF', 'HSA_TOOLS', 'ROCTRACER', 'ROCTX')) or k == 'LD_PRELOAD')}
                    env['TMPDIR'] = '/tmp'
                    subprocess.run([rocprof, '--pmc', ctr, '--kernel-trace', '--output-format', 'csv', '-d', d, '--', exe, '2'], cwd='/tmp', env=env, timeout=180,
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
                    f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)[0]
                    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'chain_gu_kernel' in r['Kernel_Name'] and r['Counter_Name'] == ctr]
                    vals = vals[len(vals) // 3:]                # drop the warm-up round
                    means[ctr] = sum(vals) / len(vals)
            traffic = int(means['FETCH_SIZE'] * 1024 * 2 + means['WRITE_SIZE'] * 1024)
            return traffic, {'measured_in_run': True, 'fetch_size_kb': round(means['FETCH_SIZE'], 1), 'write_size_kb': round(means['WRITE_SIZE'], 1), 'how': how}
        except Exception as e:                                  # noqa: BLE001 -- the counters are evidence, never a reason to lose the line
            err = f'{type(e).__name__}: {e}'[:200]
    else:
        err = 'harness or rocprofv3 not found'
    path = os.path.join(root, 'profiles', 'r05g_pmc_dominant_kernel.json')
    try:
        return json.load(open(path))['traffic_bytes_per_launch'], {'file': 'profiles/r05g_pmc_dominant_kernel.json', 'measured_in_run': False, 'live_attempt': err, 'how': how}
    except (OSError, KeyError, ValueError):
        return None, {'measured_in_run': False, 'live_attempt': err}


def _rocprof_duration():
    """Average rocprofv3 duration (us) of the dominant kernel from the newest committed record profiles/*_dominant_kernel_rocprof.json (written by
    tools/rocprof_dominant.py from a `rocprofv3 --kernel-trace --stats` run of the chunk workload) -> (us or None, source)."""
    import glob
    root = os.path.dirname(os.path.abspath(__file__))
    recs = sorted(glob.glob(os.path.join(root, 'profiles', '*_dominant_kernel_rocprof.json')))
    if not recs:
        return None, 'no profiles/*_dominant_kernel_rocprof.json'
    try:
        d = json.load(open(recs[-1]))
        return float(d['avg_us']), {'file': 'profiles/' + os.path.basename(recs[-1]), 'calls': d.get('calls'), 'from': d.get('from')}
    except (OSError, KeyError, ValueError) as e:
        return None, f'{type(e).__name__}: {e}'[:120]


def _finish(dist, line):
    """Tear the process group down first and flush C stdio (RCCL prints its version banner there), so that the JSON line is
    the last thing rank 0 writes."""
    if dist is not None:
        dist.destroy_process_group()
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if line is not None:
        print(json.dumps(line), flush=True)


def _graph_ms(fn, reps=10):
    """Milliseconds per replay of a HIP graph capturing fn(), HIP events on the launch stream (torch's current stream)."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _phases(model):
    """Per-phase milliseconds of one chunk, each phase replayed from its own HIP graph (the chunk's graph is their concatenation), and the
    IN-CHAIN time of the dominant kernel: Euler phase with its launches minus the same chain without them, per launch -- i.e. what
    the kernel costs where it actually runs, behind its producer (o_proj) and in front of its consumer (down_proj), boundary included.
    Returns (phases dict, in-chain us per launch, launches per chunk)."""
    vit = _graph_ms(lambda: model._run_vit(1))
    pre = _graph_ms(lambda: model._run_prefill(1))
    eul = _graph_ms(lambda: model._run_euler(1))
    eul_wo = _graph_ms(lambda: model._run_euler(1, skip=('gu',)))
    # the captured chunk graph itself, replayed back to back without any per-call staging: ms_per_step minus this = what infer_action adds around the graph
    gchunk = model._graphs[1]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for g_ in gchunk:
        g_.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        for g_ in gchunk:
            g_.replay()
    e1.record()
    torch.cuda.synchronize()
    chunk_graph = e0.elapsed_time(e1) / 10
    # the three phases IN the chain: one HIP graph each, replayed back to back with an event between them -- where the difference between `sum_ms` (each phase
    # replayed alone, 10 x in a row) and the chunk graph sits (r04 attributed it to "cache state" without a measurement)
    in_chain = None
    try:
        gs = []
        for fn in (lambda: model._run_vit(1), lambda: model._run_prefill(1), lambda: model._run_euler(1)):
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_):
                fn()
            gs.append(g_)
        for g_ in gs:
            g_.replay()
        torch.cuda.synchronize()
        reps = 10
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps)]
        for r in range(reps):
            evs[r][0].record()
            for i, g_ in enumerate(gs):
                g_.replay()
                evs[r][i + 1].record()
        torch.cuda.synchronize()
        d = [sum(evs[r][i].elapsed_time(evs[r][i + 1]) for r in range(reps)) / reps for i in range(3)]
        in_chain = {'vit_projector_scatter_ms': round(d[0], 4), 'joint_prefill_ms': round(d[1], 4), 'euler_ms': round(d[2], 4), 'sum_ms': round(sum(d), 4),
                    'how': 'the same three graphs replayed back to back (ViT, prefill, Euler, ViT, ...) with a HIP event between them: each phase behind its real predecessor'}
        del gs
    except Exception as e:          # noqa: BLE001 -- a diagnostic, never a reason to lose the line
        in_chain = {'error': f'{type(e).__name__}: {e}'[:160]}
    nL, ns = model.cfg.expert.num_hidden_layers, model.num_inference_steps
    n = nL * ns
    ph = {'vit_projector_scatter_ms': round(vit, 4), 'joint_prefill_ms': round(pre, 4), 'euler_ms': round(eul, 4), 'sum_ms': round(vit + pre + eul, 4),
          'chunk_graph_ms': round(chunk_graph, 4),
          'euler_us_per_layer_step': round(eul * 1e3 / n, 3), 'euler_ms_without_dominant_kernel': round(eul_wo, 4),
          'mfma_part': {'gflop': 1726.0, 'achieved_tflops': round(1726.0 / (vit + pre), 1), 'frac_of_2500': round(1726e9 / ((vit + pre) * 1e-3) / 2.5e15, 4)},
          'euler_part': {'gbytes': 13.21, 'achieved_gbs': round(13.21e9 / (eul * 1e-3) / 1e9, 1), 'frac_of_8000': round(13.21e9 / (eul * 1e-3) / 8e12, 4)},
          'in_chain': in_chain,
          'how': 'HIP events around 10 replays of one HIP graph per phase; the chunk graph is the three phases back to back'}
    return ph, (eul - eul_wo) * 1e3 / n, n


def _probe(model):
    """ISOLATED per-launch time of the dominant kernel -- the action expert's gate/up weight-streaming GEMV
    (chain_gu_kernel since r05, 27.5 MB of packed weights per launch): HIP events around replays of a HIP graph holding the 28
    layers' launches back to back (each launch streams a different layer's weights, i.e. HBM-cold as in the real chunk), graph
    boundaries included.  With no producer kernel in front of each launch this comes out slightly BELOW the kernel's duration inside
    the real chain (r02: 8.2 vs 8.6 us in rocprof); the roofline uses the in-chain figure of `_phases`, this one is reported beside it."""
    from vlaser_amd import ops, _lib as L
    ex, sb, cfg = model.expert, model.sb_act, model.cfg
    M = cfg.num_action_tokens
    llm = ex.llm

    def seq():
        for lw in ex.layers:
            a, _ = ops.skinny_args(sb.hA, lw.sk_gu, M, partials=sb.part_o, n_partials=ex.ks_o, norm_w=lw.ln_post, eps=llm.rms_norm_eps, h_out=sb.hB, out=sb.act,
                                   ldo=llm.intermediate_size)
            if 'chain' in ex.opts and ops.chain_gu_supported(M, lw.sk_gu.N, llm.hidden_size, ex.ks_o, lw.sk_gu.tpu):
                ops.launch_chain_gu(a)
            else:
                ops.launch_skinny(L.PRO_NORM, L.SK_SWIGLU, a)
    ms = _graph_ms(seq, reps=20)
    w = ex.layers[0].sk_gu
    # algorithmic bytes per launch: packed gate/up weights once + residual/partials in + SwiGLU activations out
    byts = w.n_valid * w.K * 2 + M * w.K * 2 + ex.ks_o * M * w.K * 4 + M * (w.n_valid // 2) * 2
    return ms / len(ex.layers), byts, 20 * len(ex.layers)


if __name__ == '__main__':
    main()
