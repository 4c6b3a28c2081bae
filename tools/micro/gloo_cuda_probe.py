"""Probe: can two processes on ONE GPU exchange CUDA tensors over gloo (all_reduce / all_gather)?  Lets the world-2 SFT path run on a 1-GPU box."""
import os, sys
import torch, torch.distributed as dist, torch.multiprocessing as mp


def w(rank, world, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    t = torch.full((8,), float(rank + 1), device='cuda')
    dist.all_reduce(t)
    parts = [torch.empty(4, device='cuda') for _ in range(world)]
    dist.all_gather(parts, torch.full((4,), float(rank), device='cuda'))
    b = torch.full((4,), float(rank), device='cuda', dtype=torch.bfloat16)
    try:
        dist.all_reduce(b); ok_bf16 = b.float().tolist()
    except Exception as e:
        ok_bf16 = repr(e)[:80]
    print(rank, t.tolist()[:2], [p.tolist()[0] for p in parts], ok_bf16, flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    mp.spawn(w, args=(2, 29577), nprocs=2)
