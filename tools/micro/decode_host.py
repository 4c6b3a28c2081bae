"""Is greedy decode host-bound?  Host issue time vs GPU time of one decode step (Vlaser-2B, batch 1, S=560)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vlaser_amd import config as C, synth
from vlaser_amd.internvl_chat import InternVLChatModel
torch.set_grad_enabled(False)
cfg = C.vlaser_2b()
sd = synth.vlm_state_dict(cfg, device='cuda', dtype=torch.bfloat16)
m = InternVLChatModel(cfg, max_seq_len=640, max_batch=1); m.load_state_dict(sd); del sd
m.img_context_token_id = cfg.img_context_token_id
g = torch.Generator().manual_seed(7)
pv = torch.randn(1, 3, 448, 448, generator=g).cuda().to(torch.bfloat16)
ids = torch.cat([torch.randint(0, 151643, (1, 41), generator=g), torch.full((1, 256), cfg.img_context_token_id), torch.randint(0, 151643, (1, 263), generator=g)], 1)
m.generate(pv, ids, max_new_tokens=2)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for i in range(N):
    m._decode_step(1, 561 + i)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_total = time.perf_counter() - t0
print(f'per decode step: host issue {t_issue / N * 1e3:.3f} ms, issue+drain {t_total / N * 1e3:.3f} ms')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(N):
    m._decode_step(1, 561 + i)
e1.record(); torch.cuda.synchronize()
print(f'GPU-side span per step: {e0.elapsed_time(e1) / N:.3f} ms')
