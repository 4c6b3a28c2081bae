"""Where the SFT step's time goes (BASELINE configs[4], 1 GPU): whole step (pipelined, as bench.py measures it), host enqueue time per step,
and the serialised phases forward+backward / gradient norm + AdamW with a device sync between them.   python tools/micro/sft_phases.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from vlaser_amd import config as C, synth
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
    sync = torch.cuda.synchronize
    for _ in range(3):
        m.step(pv, ids, labels)
    sync()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        m.step(pv, ids, labels)
    t_host = (time.perf_counter() - t0) / n * 1e3
    sync()
    t_all = (time.perf_counter() - t0) / n * 1e3
    print(f'pipelined step: {t_all:.2f} ms   host enqueue: {t_host:.2f} ms')
    fb, opt, fb_host = [], [], []
    for _ in range(6):
        m.wait_optimizer() if hasattr(m, 'wait_optimizer') else None
        sync()
        t0 = time.perf_counter()
        m.forward_backward(pv, ids, labels)
        t1 = time.perf_counter()
        sync()
        t2 = time.perf_counter()
        m.optimizer_step()
        m.wait_optimizer() if hasattr(m, 'wait_optimizer') else None
        sync()
        t3 = time.perf_counter()
        fb.append((t2 - t0) * 1e3); fb_host.append((t1 - t0) * 1e3); opt.append((t3 - t2) * 1e3)
    med = lambda x: sorted(x)[len(x) // 2]
    print(f'serialised: forward+backward {med(fb):.2f} ms (host enqueue {med(fb_host):.2f}), norm + AdamW {med(opt):.2f} ms, sum {med(fb) + med(opt):.2f}')
    # forward only / backward split: events around the ViT + LLM forward are not exposed, so time a forward-only pass through the loss
    if hasattr(m, 'forward_loss'):
        sync(); t0 = time.perf_counter(); m.forward_loss(pv, ids, labels); sync()
        print(f'forward only: {(time.perf_counter() - t0) * 1e3:.2f} ms')


if __name__ == '__main__':
    main()
