"""Same-process A/B of the SFT step's exchange modes at world size 1 (VLASER_FORCE_DP=1): ProcessGroupNCCL vs RCCL's C API without / with CU masks of several widths.
At world 1 RCCL launches no kernel, so this prices what the MASKS cost the compute side on their own (forward + backward, whole step).
    python tools/micro/capi_ab.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29611', VLASER_FORCE_DP='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
import torch.distributed as dist
from vlaser_amd import config as C, synth, ops
from vlaser_amd.sft import SFTModel

torch.set_grad_enabled(False)
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
cfg = C.vlaser_2b()
g = torch.Generator().manual_seed(1000)
S, R = 560, 128
ids = torch.cat([torch.randint(1, 151643, (41,), generator=g), torch.full((256,), cfg.img_context_token_id), torch.randint(1, 151643, (S - 41 - 256,), generator=g)])[None]
labels = torch.full_like(ids, -100); labels[0, -R:] = ids[0, -R:]
pv = torch.randn(1, 3, 448, 448, generator=g).cuda().to(torch.bfloat16)
print('| exchange | comm CUs | compute CUs (budget) | forward + backward ms | step ms |\n|---|---|---|---|---|')
for mode, cus in (('pg', 0), ('capi', 0), ('capi', 8), ('capi', 16), ('capi', 32), ('pg', 0)):
    os.environ.update(VLASER_DP_EXCHANGE=mode, VLASER_DP_COMM_CUS=str(cus))
    ops.set_cu_budget(256)
    torch.cuda.set_stream(torch.cuda.default_stream())      # (a CU-masked model makes its main stream the thread's current stream: undo it between configurations)
    sd = synth.vlm_state_dict(cfg, device='cuda', dtype=torch.bfloat16)
    m = SFTModel(cfg, max_seq_len=576, process_group=dist.group.WORLD)
    m.load_state_dict(sd)
    del sd
    for _ in range(3):
        m.step(pv, ids, labels)
    m.wait_optimizer(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        m.step(pv, ids, labels)
    m.wait_optimizer(); torch.cuda.synchronize()
    st = (time.perf_counter() - t0) / 10 * 1e3
    t0 = time.perf_counter()
    for _ in range(10):
        m.forward_backward(pv, ids, labels)
    torch.cuda.synchronize()
    fb = (time.perf_counter() - t0) / 10 * 1e3
    print(f'| {mode} | {cus} | {ops.get_cu_budget() if hasattr(ops, "get_cu_budget") else ""} | {fb:.2f} | {st:.2f} |', flush=True)
    if m.capi is not None:
        m.capi.destroy()
    del m
    torch.cuda.empty_cache()
dist.destroy_process_group()
