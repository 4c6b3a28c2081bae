"""Lab (r06): what serving general additive masks costs -- PiZero(general_masks=True) (VL_ATTN_DENSE attention, the proprio row in its own pass) beside the default path and
beside the default path with ride_proprio=False, full-depth Vlaser-2B-VLA, batch 1, the reference's eight-tensor call, HIP-graph replay.   python tools/micro/general_masks_ab.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vlaser_amd import config as C, synth
from vlaser_amd.pizero import PiZeroInference
from bench import make_inputs

torch.set_grad_enabled(False)
vla = C.VLAConfig(base=C.vlaser_2b())
sd = synth.vla_state_dict(vla, device='cuda', dtype=torch.bfloat16)
ids, pv, pro, noise = make_inputs(vla.base, 1)
print('| path | ms per chunk |\n|---|---|')
for name, kw in [('default (descriptors, proprio row riding in Euler step 0)', {}), ('default, ride_proprio=False', {'ride_proprio': False}),
                 ('general_masks=True (dense additive masks served)', {'general_masks': True})]:
    m = PiZeroInference(vla, max_batch=1, **kw)
    m.load_state_dict(sd)
    mask, vp, pp, ap = m.build_causal_mask_and_position_ids((ids != vla.base.pad_token_id).long(), torch.float32)
    m1, m2 = m.split_full_mask_into_submasks(mask)
    dev = [t.cuda() for t in (ids, pv.to(torch.bfloat16), m1, m2, vp, pp, ap, pro)]
    nz = noise.cuda()
    for _ in range(3):
        out = m.infer_action(*dev, noise=nz)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        out = m.infer_action(*dev, noise=nz)
    torch.cuda.synchronize()
    print(f'| {name} | {(time.perf_counter() - t0) / 30 * 1e3:.3f} |', flush=True)
    m.check_errors()
    del m
    torch.cuda.empty_cache()
