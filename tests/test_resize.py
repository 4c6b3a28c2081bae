"""CPU: the oracle's restatement of Pillow's 8-bit bicubic resampler against Pillow itself and against the committed fixture (G12), and the library's HOST weight tables
(vlaser_resample_coeffs: no GPU work) against the oracle's.  The reference resizes through Pillow (dataset.py:276-310,830-866; pillow==11.2.1)."""
import os

import numpy as np
import pytest

from oracle import resize as R

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'g12_resize.npz')


def test_oracle_vs_golden_fixture():
    g = np.load(GOLD)
    n = 0
    while f'in{n}' in g:
        ow, oh = (int(v) for v in g[f'size{n}'])
        assert np.array_equal(R.resize_bicubic_u8(g[f'in{n}'], ow, oh), g[f'out{n}']), f'case {n}'
        n += 1
    assert n == 10
    cols, rows, S = (int(v) for v in g['li_grid'])
    assert np.array_equal(R.load_image_u8(g['li_in'], S, grid=(cols, rows)), g['li_tiles'])


@pytest.mark.parametrize('H,W,ow,oh', [(37, 53, 20, 11), (37, 53, 111, 90), (480, 640, 448, 448), (300, 500, 896, 448), (700, 500, 448, 896), (5, 7, 448, 448), (448, 448, 448, 448),
                                       (1, 1, 8, 8), (9, 1200, 448, 14), (640, 480, 448, 640), (333, 448, 448, 333)])
def test_oracle_vs_pillow(H, W, ow, oh):
    Image = pytest.importorskip('PIL.Image')
    rng = np.random.default_rng(H * 7919 + W)
    for kind in range(2):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        if kind:
            img = np.where(rng.random((H, W, 1)) < 0.5, 0, 255).astype(np.uint8).repeat(3, axis=2)
        assert np.array_equal(R.resize_bicubic_u8(img, ow, oh), np.asarray(Image.fromarray(img).resize((ow, oh))))


def test_host_weight_tables_match_oracle():
    """The product computes Pillow's weight tables on the host in C (csrc/image.hip, doubles, no contraction): same bounds, same fixed-point weights as the oracle."""
    from vlaser_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(5)
    pairs = [(37, 20), (53, 111), (64, 64), (480, 448), (640, 448), (3000, 448), (2000, 1344), (1, 8), (5, 448), (1200, 14), (4032, 1792), (3024, 1344), (448, 896), (7, 3), (1000, 999)]
    pairs += [(int(rng.integers(1, 3000)), int(rng.integers(1, 2000))) for _ in range(60)]
    for a, b in pairs:
        ks = lib.vlaser_resample_ksize(a, b)
        bounds, kk = np.empty(2 * b, np.int32), np.empty(ks * b, np.int32)
        assert lib.vlaser_resample_coeffs(a, b, bounds.ctypes.data, kk.ctypes.data) == ks
        ks_o, bo, ko = R.coeffs_8bpc(a, b)
        assert ks == ks_o and np.array_equal(bounds.reshape(b, 2), bo) and np.array_equal(kk.reshape(ks, b).T, ko), (a, b)
    assert lib.vlaser_resample_ksize(0, 5) == -1 and lib.vlaser_resample_coeffs(0, 5, bounds.ctypes.data, kk.ctypes.data) == -1
