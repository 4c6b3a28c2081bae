"""Same-box A/B of the Vlaser-2B greedy decode step (S = 560 prompt, 32 new tokens, batch 1 and 8) over VLASER_DECODE_OPTS settings, interleaved:
    python tools/micro/decode_ab.py qkv16,gu16 qkv16,chain qkv16,gu16,chain
Prints ms per decode step (time(33 tokens) - time(1 token)) / 32 and checks that every setting generates the ids of the first."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vlaser_amd import _lib as L, config as C, synth            # noqa: E402
from vlaser_amd.internvl_chat import InternVLChatModel  # noqa: E402


def main():
    settings = sys.argv[1:] or ['qkv16,gu16', 'qkv16,chain', 'qkv16,gu16,chain']
    dev = 'cuda:0'
    torch.set_grad_enabled(False)
    cfg = C.vlaser_2b()
    sd = synth.vlm_state_dict(cfg, device=dev, dtype=torch.bfloat16)
    models = []
    for s in settings:
        os.environ['VLASER_DECODE_OPTS'] = s
        m = InternVLChatModel(cfg, device=dev, max_seq_len=640, max_batch=8)
        m.load_state_dict(sd)
        m.img_context_token_id = cfg.img_context_token_id
        models.append(m)
    ref_ids = {}
    for B in (1, 8):
        g = torch.Generator().manual_seed(7)
        pv = torch.randn(B, 3, 448, 448, generator=g).to(dev).to(torch.bfloat16)
        ids = torch.cat([torch.randint(0, 151643, (B, 41), generator=g), torch.full((B, 256), cfg.img_context_token_id), torch.randint(0, 151643, (B, 263), generator=g)], 1)
        res = {s: [] for s in settings}
        for rep in range(3):
            for s, m in zip(settings, models):
                ts = []
                L.lib().vlaser_chain_qkv_set_waves(1 if 'qkv1w' in s.split(',') else 0)      # (pseudo-option: the one-wave q/k/v kernel at hidden 1536; captured with the model's graphs)
                for n_new in (1, 33):
                    out = m.generate(pv, ids, max_new_tokens=n_new, min_new_tokens=n_new)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        out = m.generate(pv, ids, max_new_tokens=n_new, min_new_tokens=n_new)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) / 3)
                res[s].append((ts[1] - ts[0]) / 32 * 1e3)
                if B not in ref_ids:
                    ref_ids[B] = out.cpu()
                same = int((out.cpu() == ref_ids[B]).all())
                if rep == 0:
                    print(f'  batch {B} {s:24s}: ids == first setting: {bool(same)}')
        for s in settings:
            print(f'batch {B} {s:24s}: decode ms per step {sorted(res[s])[1]:.4f} (runs {[round(x, 4) for x in res[s]]})')


if __name__ == '__main__':
    main()
