"""What the SFT step's forward + backward costs while C CUs are held by a resident streaming kernel on another stream -- a one-GPU stand-in for RCCL's channel
workgroups during the gradient exchange (RCCL 2.26 launches NO kernel for a one-rank communicator, so `VLASER_FORCE_DP=1` at world 1 cannot show this:
profiles/r05_rccl_contention.md).  The backward's GEMM grids are single-round by construction (108-252 workgroups on 256 CUs): every CU a channel holds can push a
grid into a second round.

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/comm_shadow.hip -o tools/micro/libcomm_shadow.so     (cross-compiles without a GPU)
    python tools/micro/rccl_shadow_lab.py [<channels>[:stream|spin|read[:<cu budget>]] ...]
"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vlaser_amd import config as Cf, synth          # noqa: E402
from vlaser_amd.sft import SFTModel                 # noqa: E402


MODES = {'stream': 0, 'spin': 1, 'read': 2}


def main():
    """argv: cases of the form <channels>[:<mode>[:<cu budget>[:<masks>]]] (mode stream | spin | read; budget = ops.set_cu_budget for the timed loop; masks: 'c' = the
    forward + backward runs on a stream CU-masked to the first <budget> CUs, 's' = the shadow on a stream masked to the CUs above <budget>, 'cs' both)."""
    from vlaser_amd import ops
    cases = sys.argv[1:] or ['0', '8:spin', '1', '8', '16', '32', '0::248', '8::248', '0::248:c', '8::248:c', '8::248:s', '8::248:cs', '16::240:c', '16::240:cs', '32::224:c', '32::224:cs', '0']
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libcomm_shadow.so'))
    lib.comm_shadow.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    lib.comm_shadow_masked_stream.restype = C.c_void_p
    lib.comm_shadow_masked_stream.argtypes = [C.c_int, C.c_int]
    streams = {}

    def masked(first, n, k=0):
        if (first, n, k) not in streams:
            p_ = lib.comm_shadow_masked_stream(first, n)
            if not p_:
                sys.exit('hipExtStreamCreateWithCUMask failed')
            streams[(first, n, k)] = torch.cuda.ExternalStream(p_)
        return streams[(first, n, k)]
    dev = 'cuda:0'
    cfg = Cf.vlaser_2b()
    sd = synth.vlm_state_dict(cfg, device=dev, dtype=torch.bfloat16)
    model = SFTModel(cfg, device=dev, max_seq_len=576)
    model.load_state_dict(sd)
    del sd
    g = torch.Generator().manual_seed(1000)
    S, R = 560, 128
    ids = torch.cat([torch.randint(1, 151643, (41,), generator=g), torch.full((256,), cfg.img_context_token_id), torch.randint(1, 151643, (S - 41 - 256,), generator=g)])[None]
    labels = torch.full_like(ids, -100)
    labels[0, -R:] = ids[0, -R:]
    pv = torch.randn(1, 3, 448, 448, generator=g).to(dev).to(torch.bfloat16)
    for _ in range(3):
        model.step(pv, ids, labels)
    model.wait_optimizer()
    torch.cuda.synchronize()
    nbytes = 1 << 28
    stop = torch.zeros(1, dtype=torch.int32, device=dev)
    src, dst = torch.zeros(nbytes, dtype=torch.uint8, device=dev), torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    wgrad0 = model.wgrad_stream
    steps = 6
    print('| channel workgroups (256 threads) | what they do | CU budget of the GEMM heuristics | CU masks | forward + backward ms | stretch | GB/s moved by the shadow kernel alone |')
    print('|---|---|---|---|---|---|---|')
    base = None
    for case in cases:
        f = (case.split(':') + ['', '', ''])[:4]
        ch, mode, budget, masks = int(f[0]), f[1] or 'stream', int(f[2] or 256), f[3]
        ops.set_cu_budget(budget)
        main = masked(0, budget) if 'c' in masks else torch.cuda.current_stream()
        side_ = masked(budget, 256 - budget) if 's' in masks else side
        if model.wgrad_stream is not None:            # the weight gradients' side stream gets the compute mask too
            model.wgrad_stream = masked(0, budget, 1) if 'c' in masks else wgrad0
        with torch.cuda.stream(main):
            for _ in range(2):                        # (tile choices are made per launch; warm the variant's code objects)
                model.forward_backward(pv, ids, labels)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stop.zero_()
        torch.cuda.synchronize()
        if ch:
            lib.comm_shadow(src.data_ptr(), dst.data_ptr(), nbytes, ch, 1500, stop.data_ptr(), side_.cuda_stream, MODES[mode])   # runs until the flag is raised behind the timed loop
        time.sleep(0.002)
        t0 = time.perf_counter()
        with torch.cuda.stream(main):
            for _ in range(steps):
                model.forward_backward(pv, ids, labels)
            main.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        stop.fill_(1)
        torch.cuda.synchronize()
        rate = ''
        if ch and mode != 'spin':
            torch.cuda.synchronize()
            with torch.cuda.stream(side_):
                e0.record()
                lib.comm_shadow(src.data_ptr(), dst.data_ptr(), nbytes, ch, 2, None, side_.cuda_stream, MODES[mode])
                e1.record()
            torch.cuda.synchronize()
            rate = f'{(2 if mode == "stream" else 1) * 2 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9:.0f}'
        torch.cuda.synchronize()
        base = ms if base is None else base
        print(f'| {ch} | {mode if ch else ""} | {budget} | {dict(c="compute", s="shadow", cs="both").get(masks, "")} | {ms:.2f} | x{ms / base:.3f} | {rate} |', flush=True)
    ops.set_cu_budget(256)


if __name__ == '__main__':
    main()
