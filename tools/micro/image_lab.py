"""Lab (r06): the device image preparation (csrc/image.hip) on the largest case of the path -- a 12-megapixel frame onto the 13 tiles of BASELINE's 8B configuration -- per
launch and as a whole, beside Pillow on the host (the reference's way: eval_example.py:38-82).  Algorithmic bytes of a pass = its input image + its output image.
python tools/micro/image_lab.py"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from PIL import Image
from vlaser_amd import prep
from vlaser_amd.image import ImagePrep


def ev_time(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


ip = ImagePrep('cuda')
rng = np.random.default_rng(0)
print('| case | device us | algorithmic MB | GB/s | Pillow on the host ms |\n|---|---|---|---|---|')
for (H, W) in [(3024, 4032), (1080, 1920), (480, 640)]:
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    d = torch.from_numpy(img).cuda()
    cols, rows = prep.dynamic_grid(W, H, 1, 12, 448)
    tw, th = 448 * cols, 448 * rows
    pil = Image.fromarray(img)
    # the two passes alone
    tmp = torch.empty(H, tw, 3, dtype=torch.uint8, device='cuda')
    us_h = ev_time(lambda: ip.resize(d, tw, H, out=tmp))
    mb_h = (H * W * 3 + H * tw * 3) / 1e6
    print(f'| {W}x{H} -> {tw}x{H} horizontal pass | {us_h:.1f} | {mb_h:.1f} | {mb_h / us_h * 1e3:.0f} | |')
    big = torch.empty(th, tw, 3, dtype=torch.uint8, device='cuda')
    us_v = ev_time(lambda: ip.resize(tmp, tw, th, out=big))
    mb_v = (H * tw * 3 + th * tw * 3) / 1e6
    print(f'| {tw}x{H} -> {tw}x{th} vertical pass | {us_v:.1f} | {mb_v:.1f} | {mb_v / us_v * 1e3:.0f} | |')
    out = torch.empty(cols * rows, 3, 448, 448, dtype=torch.bfloat16, device='cuda')
    us_n = ev_time(lambda: ip.tiles_normalize(big, cols, rows, out))
    mb_n = (th * tw * 3 + th * tw * 3 * 2) / 1e6
    print(f'| {cols}x{rows} tiles crop + normalise | {us_n:.1f} | {mb_n:.1f} | {mb_n / us_n * 1e3:.0f} | |')
    us_all = ev_time(lambda: ip.load_image(d, max_num=12))
    t0 = time.time()
    for _ in range(3): ref = prep.load_image(pil, 448, 12)
    ms_host = (time.time() - t0) / 3 * 1e3
    t0 = time.time()
    for _ in range(3): pil.resize((tw, th))
    ms_resize = (time.time() - t0) / 3 * 1e3
    print(f'| {W}x{H} load_image whole ({cols * rows + (cols * rows > 1)} tiles, image already on the device) | {us_all:.1f} | | | {ms_host:.1f} (of it the big resize {ms_resize:.1f}) |')
    t0 = time.time()
    for _ in range(5):
        o = ip.load_image(img, max_num=12); torch.cuda.synchronize()
    print(f'| {W}x{H} load_image from a host array (PCIe copy of {H * W * 3 / 1e6:.1f} MB included, synchronised) | {(time.time() - t0) / 5 * 1e6:.0f} | | | |', flush=True)
