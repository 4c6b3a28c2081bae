"""Which producer's sum-of-squares slot disagrees with the gradient buffer?  (truncated Vlaser-2B of the golden fixtures, 3 SFT steps; per step the fused norm,
the buffer norm of the same gradients and every per-tensor pair that differs by > 1e-5 relative)"""
import os, sys, math
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vlaser_amd import config as C, synth
from vlaser_amd.sft import SFTModel
torch.set_grad_enabled(False)
cfg = C.truncated(C.vlaser_2b(), 2, 2)
sd = synth.vla_state_dict(C.VLAConfig(base=cfg), with_head=True)
d = np.load(os.path.join(ROOT, 'tests', 'golden', 'g5g6_vlm.npz'))
ids = torch.from_numpy(d['input_ids'])
pv = torch.randn(1, 3, 448, 448, generator=torch.Generator().manual_seed(0))
labels = torch.full_like(ids, -100); labels[0, -16:] = ids[0, -16:]
m = SFTModel(cfg, max_seq_len=ids.shape[1], lr=1e-3, weight_decay=0.05, max_grad_norm=1.0)
m.load_state_dict(sd)
for step in range(3):
    o = m.step(pv, ids, labels)
    m.wait_optimizer(); torch.cuda.synchronize()
    g = m.fp.g.float()
    print(f'step {step}: fused norm {o.grad_norm.item():.6f}  buffer norm {g.norm().item():.6f}  loss {o.loss.item():.6f}')
    parts = m.norm_parts
    for name, shape, off in m.fp.specs:
        if name not in m.norm_slot:
            continue
        lo, cap = m.norm_slot[name]
        a = parts[lo:lo + cap].double().sum().item()
        b = g[off:off + math.prod(shape)].double().pow(2).sum().item()
        if abs(a - b) > 1e-5 * max(b, 1e-12):
            print(f'   {name:14s} slots {a:.6e}  buffer {b:.6e}  rel {abs(a - b) / max(b, 1e-30):.2e}')
    # the chunked small tensors, bucket by bucket
    for (lo, hi, c0, tab) in m.norm_plan:
        if tab is None:
            continue
        for j, (off, n) in enumerate(tab.tolist()):
            a = parts[c0 + j].item(); b = g[off:off + n].double().pow(2).sum().item()
            if abs(a - b) > 1e-5 * max(b, 1e-12):
                print(f'   chunk at {off} (+{n}) slot {a:.6e} buffer {b:.6e}')
