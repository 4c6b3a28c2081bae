"""GPU: the device image preparation (csrc/image.hip through the C ABI, vlaser_amd/image.py) against the oracle's restatement of Pillow's resampler, against Pillow itself
and against the committed fixture G12 -- bit-exact (byte work).  Reference: load_image, eval_example.py:38-82 = dataset.py:276-310,830-866 (Pillow BICUBIC)."""
import os

import numpy as np
import pytest
import torch

from oracle import resize as R

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'g12_resize.npz')


@pytest.fixture(scope='module')
def ip():
    from vlaser_amd.image import ImagePrep
    return ImagePrep('cuda')


def _dev(ip, img, ow, oh):
    return ip.resize(torch.from_numpy(img).cuda(), ow, oh).cpu().numpy()


def test_resize_vs_golden_fixture(ip):
    g = np.load(GOLD)
    n = 0
    while f'in{n}' in g:
        ow, oh = (int(v) for v in g[f'size{n}'])
        assert np.array_equal(_dev(ip, g[f'in{n}'], ow, oh), g[f'out{n}']), f'case {n}'
        n += 1
    assert n == 10


@pytest.mark.parametrize('H,W,ow,oh', [(37, 53, 20, 11), (37, 53, 111, 90), (64, 64, 64, 31), (64, 64, 17, 64), (480, 640, 448, 448), (300, 500, 896, 448), (700, 500, 448, 896),
                                       (5, 7, 448, 448), (448, 448, 448, 448), (1, 1, 8, 8), (9, 1200, 448, 14), (333, 448, 448, 333), (50, 9000, 3, 50), (6000, 40, 40, 2)])
def test_resize_vs_oracle_and_pillow(ip, H, W, ow, oh):
    Image = pytest.importorskip('PIL.Image')
    rng = np.random.default_rng(H * 7919 + W)
    for kind in range(2):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        if kind:
            img = np.where(rng.random((H, W, 1)) < 0.5, 0, 255).astype(np.uint8).repeat(3, axis=2)      # overshoot: the clip at 0 / 255
        got = _dev(ip, img, ow, oh)
        assert np.array_equal(got, np.asarray(Image.fromarray(img).resize((ow, oh)))), 'device vs Pillow'
        if H * W <= 480 * 640:
            assert np.array_equal(got, R.resize_bicubic_u8(img, ow, oh)), 'device vs oracle'


def test_resize_strided_and_unaligned_views(ip):
    """A crop of a larger image (row stride != 3 W, first byte not dword-aligned): the byte paths of both passes."""
    Image = pytest.importorskip('PIL.Image')
    rng = np.random.default_rng(3)
    full = rng.integers(0, 256, (90, 131, 3), dtype=np.uint8)
    d = torch.from_numpy(full).cuda()
    for (y0, x0, hh, ww, ow, oh) in [(3, 7, 60, 101, 56, 56), (0, 1, 90, 130, 200, 45), (5, 0, 80, 131, 131, 33), (1, 3, 77, 99, 50, 77)]:
        view = d[y0:y0 + hh, x0:x0 + ww]
        want = np.asarray(Image.fromarray(np.ascontiguousarray(full[y0:y0 + hh, x0:x0 + ww])).resize((ow, oh)))
        assert np.array_equal(ip.resize(view, ow, oh).cpu().numpy(), want)


def test_full_size_photo_vs_pillow(ip):
    """A 12-megapixel frame onto the 4 x 3 grid of 448-px tiles (1792 x 1344) and onto the thumbnail -- the largest resize of BASELINE's 13-tile configuration."""
    Image = pytest.importorskip('PIL.Image')
    rng = np.random.default_rng(11)
    base = rng.integers(0, 256, (190, 252, 3), dtype=np.uint8)
    img = np.asarray(Image.fromarray(base).resize((4032, 3024), Image.NEAREST)).copy()
    img ^= rng.integers(0, 32, img.shape, dtype=np.uint8)                 # structure + noise
    pil = Image.fromarray(img)
    assert np.array_equal(_dev(ip, img, 1792, 1344), np.asarray(pil.resize((1792, 1344))))
    assert np.array_equal(_dev(ip, img, 448, 448), np.asarray(pil.resize((448, 448))))


def test_load_image_vs_host_path_and_fixture(ip):
    """ImagePrep.load_image == prep.load_image (the host path through Pillow: dynamic_preprocess + build_transform) rounded to bf16, tile for tile; and the uint8 tiles of the
    fixture's whole load_image case."""
    from PIL import Image
    from vlaser_amd import prep
    rng = np.random.default_rng(17)
    for (H, W) in [(600, 800), (448, 448), (300, 1300), (1500, 500), (1000, 1000)]:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        want = prep.load_image(Image.fromarray(img), 448, 12).to(torch.bfloat16)
        got = ip.load_image(img, max_num=12).cpu()
        assert got.shape == want.shape and torch.equal(got, want), (H, W, got.shape)
    from vlaser_amd.image import ImagePrep
    g = np.load(GOLD)
    cols, rows, S = (int(v) for v in g['li_grid'])
    small = ImagePrep('cuda', input_size=S)
    d = torch.from_numpy(g['li_in']).cuda()
    big = small.resize(d, S * cols, S * rows)
    tiles = [big[:, i * S:(i + 1) * S] for i in range(cols)] + [small.resize(d, S, S)]
    assert np.array_equal(torch.stack(tiles).cpu().numpy(), g['li_tiles'])
    # normalisation of those tiles: the fp32 expression of ToTensor + Normalize, one bf16 rounding
    out = torch.empty(cols * rows, 3, S, S, dtype=torch.bfloat16, device='cuda')
    small.tiles_normalize(big, cols, rows, out)
    t = torch.from_numpy(g['li_tiles'][:cols * rows]).permute(0, 3, 1, 2).float() / 255.0
    want = ((t - torch.tensor(prep.IMAGENET_MEAN).view(1, 3, 1, 1)) / torch.tensor(prep.IMAGENET_STD).view(1, 3, 1, 1)).to(torch.bfloat16)
    assert torch.equal(out.cpu(), want)


def test_load_image_from_file_like_the_reference(tmp_path):
    """`vlaser_amd.image.load_image(path)` == the reference-style host path (`prep.load_image(Image.open(path))`) in bf16, from a PNG on disk."""
    from PIL import Image
    from vlaser_amd import prep
    from vlaser_amd.image import load_image
    img = np.random.default_rng(23).integers(0, 256, (333, 777, 3), dtype=np.uint8)
    f = tmp_path / 'obs.png'
    Image.fromarray(img).save(f)
    got = load_image(str(f), 448, 12)
    want = prep.load_image(Image.open(f).convert('RGB'), 448, 12).to(torch.bfloat16)
    assert got.is_cuda and got.dtype == torch.bfloat16 and torch.equal(got.cpu(), want)


def test_resize_argument_errors(ip):
    from vlaser_amd import _lib as L
    d = torch.zeros(8, 8, 3, dtype=torch.uint8, device='cuda')
    o = torch.zeros(4, 4, 3, dtype=torch.uint8, device='cuda')
    with pytest.raises(L.VlaserHipError, match='weight tables'):
        L.check(L.lib().vlaser_resize_u8(d.data_ptr(), 8, 8, 24, None, 0, o.data_ptr(), 4, 4, 12, None, None, 0, None, None, 0, None), 'vlaser_resize_u8')
