"""GPU-box probe: model-level errors of the HIP path vs the fp32 CPU oracle / golden fixtures (prints numbers)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vlaser_amd import config as C, synth
from vlaser_amd.internvl_chat import InternVLChatModel
from vlaser_amd.pizero import PiZeroInference
from oracle import vit as ovit, vlm as ovlm, vla as ovla

torch.set_grad_enabled(False)
G = os.path.join(ROOT, 'tests', 'golden')
cfg = C.truncated(C.vlaser_2b(), 2, 2)
vla = C.VLAConfig(base=cfg)
sd = synth.vla_state_dict(vla, with_head=True)

def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / b.abs().max()).item(), ((a - b).norm() / b.norm()).item()

d = np.load(os.path.join(G, 'g5g6_vlm.npz'))
g = torch.Generator().manual_seed(0)
pv = torch.randn(1, 3, 448, 448, generator=g)
ids = torch.from_numpy(d['input_ids'])
m = InternVLChatModel(cfg, max_seq_len=512)
m.load_state_dict(sd)
m.img_context_token_id = 151667
feat, layers = m.vit.forward(m._to_bf16(pv), return_layers=True)
oh, olayers = ovit.vision_forward(sd, cfg.vision, pv, return_layers=True)
print('vit emb', rel(layers[0].view(1, 1025, 1024), ovit.embeddings(sd, cfg.vision, pv)))
for i, (a, b) in enumerate(zip(layers[1:], olayers)):
    print('vit layer', i, rel(a.view(1, 1025, 1024), b))
of = ovit.extract_feature(sd, cfg, pv)
print('feat', rel(m.extract_feature(pv), of))
out = m.forward(pv, ids, image_flags=torch.ones(1, 1, dtype=torch.long))
ol = ovlm.forward_logits(sd, cfg, pv, ids)
print('logits', rel(out.logits, ol), 'top8 ids', out.logits[0, -1].topk(8).indices.tolist(), d['last_top_ids'].tolist())
labels = torch.full_like(ids, -100); labels[0, -16:] = ids[0, -16:]
out2 = m.forward(pv, ids, image_flags=torch.ones(1, 1, dtype=torch.long), labels=labels)
print('loss', out2.loss.item(), float(d['sft_loss']))
gen, lg = m.generate(pv, ids, max_new_tokens=8, return_logits=True)
print('greedy', gen.tolist(), d['greedy_ids'].tolist(), 'margins', d['greedy_margin'])
print('greedy top vals', lg[0].topk(2, dim=-1).values.cpu().numpy()[:, :], d['greedy_top_vals'][:, :2])

d7 = np.load(os.path.join(G, 'g7_vla.npz'))
pz = PiZeroInference(vla, max_batch=1)
pz.load_state_dict(sd)
for case in ('a', 'b'):
    seed = int(d7[f'{case}_seed'])
    g = torch.Generator().manual_seed(seed)
    pv = torch.randn(1, 3, 448, 448, generator=g)
    ids = torch.from_numpy(d7[f'{case}_input_ids'])
    am = (ids != 151643).long()
    mask, vp, pp, ap = pz.build_causal_mask_and_position_ids(am, torch.float32)
    m1, m2 = pz.split_full_mask_into_submasks(mask)
    for rep in range(2):
        t0 = time.time()
        act = pz.infer_action(ids, pv, m1, m2, vp, pp, ap, torch.from_numpy(d7[f'{case}_proprio']), noise=torch.from_numpy(d7[f'{case}_noise']))
        torch.cuda.synchronize()
        print('infer_action', case, rep, 'time', time.time() - t0, 'max abs err', (act.cpu() - torch.from_numpy(d7[f'{case}_action'])).abs().max().item())
    print(act.cpu().numpy().round(4)); print(d7[f'{case}_action'].round(4))
