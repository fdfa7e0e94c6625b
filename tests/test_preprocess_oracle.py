"""Input-pipeline oracle (oracle/resample_oracle.c) against Pillow's own outputs (tests/golden/preprocess.npz, made by
oracle/make_golden.py `preprocess`: Image.crop + Image.resize(BICUBIC) + torch ToTensor/Normalize arithmetic), and the
host-side parameter logic of mvlpt_amd/transforms.py.  CPU only.  Bar: bit-exact bytes, bit-exact fp32."""
import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as PO
from tests.golden_util import load_npz


@pytest.fixture(scope="module")
def golden():
    return load_npz("preprocess")


def test_oracle_equals_pillow_bytes_and_torch_floats(golden):
    g = golden
    for i in range(int(g["n"])):
        ct, cl, ch, cw, rh, rw, ot, ol, oh, ow, flip = [int(v) for v in g[f"c{i}_desc"]]
        u8, f32 = PO.preprocess(g[f"c{i}_src"], (ct, cl, ch, cw), (rh, rw), (ot, ol, oh, ow), flip, g["mean"], g["std"])
        assert np.array_equal(u8, g[f"c{i}_u8"]), f"case {i}: resized bytes differ from Pillow"
        assert np.array_equal(f32, g[f"c{i}_f32"]), f"case {i}: normalised floats differ from torch"


def test_oracle_equals_installed_pillow_on_random_boxes():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(1)
    for trial in range(60):
        H, W = int(rng.integers(1, 160)), int(rng.integers(1, 160))
        a = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
        ch, cw = int(rng.integers(1, H + 1)), int(rng.integers(1, W + 1))
        ct, cl = int(rng.integers(0, H - ch + 1)), int(rng.integers(0, W - cw + 1))
        rh, rw = int(rng.choice([ch, 8, 32, 50])), int(rng.choice([cw, 8, 32, 50]))
        ref = np.asarray(Image.fromarray(a).crop((cl, ct, cl + cw, ct + ch)).resize((rw, rh), Image.BICUBIC))
        got = PO.resample_crop_u8(a, (ct, cl, ch, cw), (rh, rw), (0, 0, rh, rw))
        assert np.array_equal(got, ref), (H, W, ct, cl, ch, cw, rh, rw)


def test_oracle_rejects_boxes_outside_the_image():
    a = np.zeros((10, 12, 3), np.uint8)
    with pytest.raises(ValueError):
        PO.resample_crop_u8(a, (0, 0, 11, 12), (8, 8), (0, 0, 8, 8))
    with pytest.raises(ValueError):
        PO.resample_crop_u8(a, (0, 0, 10, 12), (8, 8), (1, 0, 8, 8))


def test_resize_and_center_crop_integer_rules():
    from mvlpt_amd.transforms import center_crop_offsets, resize_shorter_side
    assert resize_shorter_side(375, 500, 224) == (224, 298)          # landscape: int(224 * 500 / 375) = 298
    assert resize_shorter_side(500, 375, 224) == (298, 224)
    assert resize_shorter_side(224, 224, 224) == (224, 224)
    assert center_crop_offsets(224, 298, 224, 224) == (0, 37)
    assert center_crop_offsets(225, 299, 224, 224) == (0, 38)         # round-half-even: 0.5 -> 0, 37.5 -> 38
    with pytest.raises(ValueError):
        center_crop_offsets(100, 300, 224, 224)


def test_random_resized_crop_params_stay_inside_and_are_reproducible():
    from mvlpt_amd.transforms import random_resized_crop_params
    g1, g2 = torch.Generator().manual_seed(7), torch.Generator().manual_seed(7)
    seen = set()
    for (h, w) in [(375, 500), (32, 32), (1, 50), (600, 20), (224, 224)]:
        for _ in range(50):
            a = random_resized_crop_params(h, w, generator=g1)
            b = random_resized_crop_params(h, w, generator=g2)
            assert a == b
            t, l, ch, cw = a
            assert 0 <= t and 0 <= l and ch >= 1 and cw >= 1 and t + ch <= h and l + cw <= w
            seen.add(a)
    assert len(seen) > 50


def test_describe_builds_consistent_descriptors():
    from mvlpt_amd.transforms import DeviceTransform
    shapes = [(375, 500), (500, 375), (224, 224), (300, 1000)]
    tr = DeviceTransform(engine=None, size=224, train=True, generator=torch.Generator().manual_seed(0))
    descs, total = tr.describe(shapes)
    assert total == sum(h * w * 3 for h, w in shapes)
    off = 0
    for d, (h, w) in zip(descs, shapes):
        assert d.offset == off and (d.height, d.width) == (h, w) and d.resize_height == d.resize_width == 224
        assert d.crop_top + d.crop_height <= h and d.crop_left + d.crop_width <= w and d.flip in (0, 1)
        off += h * w * 3
    ev = DeviceTransform(engine=None, size=224, train=False, center_crop=True)
    descs, _ = ev.describe(shapes)
    for d, (h, w) in zip(descs, shapes):
        assert (d.crop_height, d.crop_width) == (h, w) and min(d.resize_height, d.resize_width) == 224
        assert d.out_top + 224 <= d.resize_height and d.out_left + 224 <= d.resize_width and d.flip == 0
