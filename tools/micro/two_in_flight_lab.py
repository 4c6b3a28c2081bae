"""Two independent batch-1 requests in flight (two PiZeroInference instances, two streams, calls alternating): does the launch-bound Euler phase of one
request hide the MFMA-bound ViT + prefill of the other?   python tools/micro/two_in_flight_lab.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vlaser_amd import config as C, synth  # noqa: E402
from vlaser_amd.pizero import PiZeroInference  # noqa: E402

torch.set_grad_enabled(False)
dev = 'cuda:0'
vla = C.VLAConfig(base=C.vlaser_2b())
sd = synth.vla_state_dict(vla, device=dev, dtype=torch.bfloat16)
N = int(os.environ.get('N_INST', 2))
models, streams, inputs = [], [], []
for i in range(N):
    m = PiZeroInference(vla, device=dev, max_batch=1)
    m.load_state_dict(sd)
    models.append(m)
    streams.append(torch.cuda.Stream(device=dev))
    ids, pv, proprio, noise = bench.make_inputs(vla.base, 1, seed=i)
    inputs.append((ids.to(dev), pv.to(dev).to(torch.bfloat16), proprio.to(dev), noise.to(dev), (ids != vla.base.pad_token_id).sum(-1).to(dev)))
del sd


def call(i):
    ids, pv, pro, noise, valid = inputs[i]
    with torch.cuda.stream(streams[i]):
        return models[i].infer_action(ids, pv, proprios=pro, noise=noise, valid_len=valid)


for i in range(N):
    for _ in range(3):
        call(i)
torch.cuda.synchronize()
# one at a time (reference point, same process)
t0 = time.perf_counter()
for k in range(40):
    call(0)
torch.cuda.synchronize()
one = (time.perf_counter() - t0) / 40 * 1e3
t0 = time.perf_counter()
for k in range(40 * N):
    out = call(k % N)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'one request at a time: {one:.3f} ms per chunk = {1e3 / one:.1f} chunks/s;   {N} in flight: {dt / (40 * N) * 1e3:.3f} ms per chunk = {40 * N / dt:.1f} chunks/s')
