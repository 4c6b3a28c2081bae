"""Golden fixture G12: Pillow's own outputs for the image preparation of `load_image` (eval_example.py:38-82 -> dataset.py:276-310,830-866).  The resize arithmetic of
that path lives in Pillow (pinned by the reference at pillow==11.2.1, Vlaser_VLA/Simpler/requirements.txt:165), not under /root/reference; this script runs Pillow in the
build container and commits inputs + outputs as data (tests/golden/g12_resize.npz, < 200 KB).  `python tools/gen_golden_resize.py`"""
import os
import sys

import numpy as np
import PIL
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(37, 53, 20, 11), (37, 53, 111, 90), (64, 64, 64, 31), (64, 64, 17, 64), (5, 7, 96, 80), (1, 1, 8, 8), (90, 120, 56, 56), (9, 300, 112, 14), (48, 48, 48, 48), (131, 77, 56, 112)]


def main():
    rng = np.random.default_rng(20261003)
    out = {'pillow_version': np.array(PIL.__version__)}
    for i, (H, W, ow, oh) in enumerate(CASES):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        if i % 3 == 1:                                  # hard edges: exercises the clip at 0 / 255 (bicubic overshoot)
            img = np.where(rng.random((H, W, 1)) < 0.5, 0, 255).astype(np.uint8).repeat(3, axis=2)
        out[f'in{i}'] = img
        out[f'size{i}'] = np.array([ow, oh])
        out[f'out{i}'] = np.asarray(Image.fromarray(img).resize((ow, oh)))             # default filter = BICUBIC, as dynamic_preprocess calls it (dataset.py:849)
        assert np.array_equal(out[f'out{i}'], np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC)))
    # one whole load_image: a 150 x 260 image on a 2 x 1 grid of 56-px tiles + thumbnail (input_size 56 keeps the fixture small; the code path is size-independent)
    img = rng.integers(0, 256, (150, 260, 3), dtype=np.uint8)
    S, cols, rows = 56, 2, 1
    im = Image.fromarray(img)
    big = im.resize((S * cols, S * rows))
    tiles = [np.asarray(big.crop(((i % cols) * S, (i // cols) * S, (i % cols + 1) * S, (i // cols + 1) * S))) for i in range(cols * rows)]
    tiles.append(np.asarray(im.resize((S, S))))
    out['li_in'], out['li_grid'], out['li_tiles'] = img, np.array([cols, rows, S]), np.stack(tiles)
    path = os.path.join(ROOT, 'tests', 'golden', 'g12_resize.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), 'bytes; Pillow', PIL.__version__)


if __name__ == '__main__':
    sys.exit(main())
