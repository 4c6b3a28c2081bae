"""Lab: does pulling the action expert's weights into the Infinity Cache from a SECOND stream (two hardware queues do overlap: two_in_flight_lab.py) shorten the Euler phase?
Stream A replays the Euler-phase graph; stream B replays a graph of paced `touch` launches over the same packed weights in layer order, unsynchronised, at several paces.
    python tools/micro/prefetch_lab.py"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LAB = os.path.join(ROOT, 'tools', 'micro', 'lab_build')
os.makedirs(LAB, exist_ok=True)
so = os.path.join(LAB, 'libprefetch_lab.so')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', so, os.path.join(ROOT, 'tools', 'micro', 'prefetch_lab.hip')])
lab = C.CDLL(so)
lab.touch.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_void_p]
import bench  # noqa: E402
from vlaser_amd import config as Cfg, synth  # noqa: E402
from vlaser_amd.pizero import PiZeroInference  # noqa: E402

torch.set_grad_enabled(False)
dev = 'cuda:0'
vla = Cfg.VLAConfig(base=Cfg.vlaser_2b())
m = PiZeroInference(vla, device=dev, max_batch=1)
m.load_state_dict(synth.vla_state_dict(vla, device=dev, dtype=torch.bfloat16))
ids, pv, proprio, noise = bench.make_inputs(vla.base, 1, seed=0)
valid = (ids != vla.base.pad_token_id).sum(-1).to(dev)
for _ in range(3):
    m.infer_action(ids.to(dev), pv.to(dev).to(torch.bfloat16), proprios=proprio.to(dev), noise=noise.to(dev), valid_len=valid)
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
with torch.cuda.stream(sa):
    m._run_euler(1)
    torch.cuda.synchronize()
    ga = torch.cuda.CUDAGraph()
    with torch.cuda.graph(ga, stream=sa):
        m._run_euler(1)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
layers = m.expert.layers
bufs = [[lw.sk_qkv.t, lw.sk_o.t, lw.sk_gu.t, lw.sk_down.t] for lw in layers]
nsteps = m.num_inference_steps


lab.bump.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
progress = torch.zeros(4, dtype=torch.int32, device=dev)
from vlaser_amd import ops, _lib as L  # noqa: E402
real_launch = ops.launch_skinny
state = {'n': 0}


def launch_with_progress(pro, epi, a, stream=None):
    real_launch(pro, epi, a, stream)
    if pro == L.PRO_PLAIN and epi == L.SK_PARTIAL:                # the layer-step's last launch (down_proj)
        state['n'] += 1
        lab.bump(progress.data_ptr(), state['n'], torch.cuda.current_stream().cuda_stream)


# Euler graph with a progress bump behind every layer-step (lab only: in a product version the down_proj kernel would store the counter itself)
ops.launch_skinny = launch_with_progress
with torch.cuda.stream(sa):
    gp = torch.cuda.CUDAGraph()
    state['n'] = 0
    with torch.cuda.graph(gp, stream=sa):
        lab.bump(progress.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        m._run_euler(1)
ops.launch_skinny = real_launch
n_items = state['n']
print('layer-steps with a bump:', n_items)
nL = len(layers)


def build_toucher(n_wg, ahead, which=(0, 1, 2, 3)):
    with torch.cuda.stream(sb):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=sb):
            for idx in range(n_items):
                lb = bufs[idx % nL]
                for j in which:
                    t = lb[j]
                    lab.touch(t.data_ptr(), t.numel() * t.element_size(), n_wg, sink.data_ptr(), progress.data_ptr() if j == which[0] else None, idx, ahead, 20000,
                              torch.cuda.current_stream().cuda_stream)
    return g


def time_pair(ga_, gb, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        progress.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(sa):
            e0.record()
        if gb is not None:
            with torch.cuda.stream(sb):
                gb.replay()
        with torch.cuda.stream(sa):
            ga_.replay()
            e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps


print(f'Euler phase alone: {time_pair(ga, None):.3f} ms;  with the progress bumps: {time_pair(gp, None):.3f} ms')
for which, name in (((2, 3), 'gate/up + down'), ((0, 1, 2, 3), 'all four')):
    for n_wg in (32, 64, 128):
        for ahead in (1, 2, 4):
            gb = build_toucher(n_wg, ahead, which)
            print(f'prefetch {name:15s} {n_wg:3d} workgroups, {ahead} layer-steps ahead: Euler phase (with bumps) beside it: {time_pair(gp, gb):.3f} ms')
print(f'Euler phase alone again: {time_pair(ga, None):.3f} ms')
