"""Where do the +6 ms of the CU-masked SFT step come from?  One model (capi, 32 comm CUs), forward + backward timed with the stream roles varied."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29612', VLASER_FORCE_DP='1', HSA_ENABLE_IPC_MODE_LEGACY='0', VLASER_DP_EXCHANGE='capi', VLASER_DP_COMM_CUS='32')
import torch.distributed as dist
from vlaser_amd import config as C, synth, ops, rccl_capi
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
sd = synth.vlm_state_dict(cfg, device='cuda', dtype=torch.bfloat16)
m = SFTModel(cfg, max_seq_len=576, process_group=dist.group.WORLD)
m.load_state_dict(sd)
del sd
for _ in range(3):
    m.step(pv, ids, labels)
m.wait_optimizer(); torch.cuda.synchronize()
masked_main, masked_wg = m.main_stream, m.wgrad_stream
plain_wg = torch.cuda.Stream()

def timed(label, outer=None):
    torch.cuda.synchronize()
    for _ in range(2):
        m.forward_backward(pv, ids, labels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if outer is not None:
        with torch.cuda.stream(outer):
            for _ in range(10):
                m.forward_backward(pv, ids, labels)
    else:
        for _ in range(10):
            m.forward_backward(pv, ids, labels)
    torch.cuda.synchronize()
    print(f'| {label} | {(time.perf_counter() - t0) / 10 * 1e3:.2f} |', flush=True)

print('| forward + backward, cu budget %d | ms |\n|---|---|' % ops.get_cu_budget())
timed('main masked (per-call switch), wgrad masked')
m.wgrad_stream = plain_wg
timed('main masked (per-call switch), wgrad on a plain stream')
m.wgrad_stream = None
timed('main masked (per-call switch), no wgrad stream')
m.main_stream = None
m.wgrad_stream = masked_wg
timed('caller on the default stream (no main mask), wgrad masked')
timed('caller inside ONE `with stream(masked main)`, wgrad masked', outer=masked_main)
m.wgrad_stream = plain_wg
timed('everything on plain streams, budget 224')
ops.set_cu_budget(256)
timed('everything on plain streams, budget 256')
m.wgrad_stream = masked_wg
timed('caller inside ONE `with stream(masked main)`, wgrad masked, budget 256', outer=masked_main)
dist.destroy_process_group()
