"""GPU-box lab: AdamW kernel time vs grid size (VLASER_ADAMW_BLOCKS) on one 198 M-parameter bucket (the SFT step's bucket size)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import torch
    from vlaser_amd import ops
    n = 198 * 1024 * 1024
    p = torch.zeros(n, dtype=torch.bfloat16, device='cuda'); g = torch.full((n,), 1e-3, dtype=torch.bfloat16, device='cuda')
    ma = torch.zeros(n, device='cuda'); m = torch.zeros(n, device='cuda'); v = torch.zeros(n, device='cuda')
    gn = torch.ones(1, device='cuda')
    f = lambda s: ops.adamw_clipped(p, ma, m, v, g, 1e-4, 0.9, 0.999, 1e-8, 0.05, 1.0, gn, 1.0, s)
    for s in range(1, 4): f(s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for s in range(4, 24): f(s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f'blocks {os.environ.get("VLASER_ADAMW_BLOCKS", "default"):>7s}: {ms * 1e3:8.1f} us  {28.0 * n / ms / 1e9:6.2f} TB/s')
else:
    for b, t in (('4096', '256'), ('256', '256'), ('256', '512'), ('256', '1024'), ('128', '512'), ('128', '1024'), ('512', '128')):
        print('threads', t, end=' ', flush=True)
        subprocess.run([sys.executable, __file__, 'run'], env=dict(os.environ, VLASER_ADAMW_BLOCKS=b, VLASER_ADAMW_THREADS=t))
