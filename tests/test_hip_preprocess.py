"""Device input pipeline (`mvlpt_preprocess`, csrc/preprocess.hip) through the C ABI, -m gpu.
Bar: BIT-EXACT — resized bytes equal Pillow's (golden fixture) and the C oracle's, normalised fp32 values equal
torch's / the oracle's; f16 / bf16 outputs equal the exact fp32 value rounded once."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import preprocess_oracle as PO  # noqa: E402
from tests.golden_util import load_npz  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    from mvlpt_amd.engine import Engine
    from mvlpt_amd.weights import ARCHS
    return Engine(ARCHS["tiny"], "fp16")


def _descs(items):
    """items: (src array, crop, resize, window origin, flip) -> (packed uint8 tensor, ctypes descs)"""
    from mvlpt_amd import _lib
    descs = (_lib.MvlptImageDesc * len(items))()
    off, chunks = 0, []
    for d, (a, (ct, cl, ch, cw), (rh, rw), (ot, ol), flip) in zip(descs, items):
        d.offset, d.height, d.width = off, a.shape[0], a.shape[1]
        d.crop_top, d.crop_left, d.crop_height, d.crop_width = ct, cl, ch, cw
        d.resize_height, d.resize_width, d.out_top, d.out_left, d.flip = rh, rw, ot, ol, flip
        off += a.size
        chunks.append(np.ascontiguousarray(a).reshape(-1))
    return torch.from_numpy(np.concatenate(chunks)).cuda(), descs


def test_golden_cases_bit_exact(eng):
    g = load_npz("preprocess")
    by_size = {}
    for i in range(int(g["n"])):
        ct, cl, ch, cw, rh, rw, ot, ol, oh, ow, flip = [int(v) for v in g[f"c{i}_desc"]]
        by_size.setdefault((oh, ow), []).append((i, (g[f"c{i}_src"], (ct, cl, ch, cw), (rh, rw), (ot, ol), flip)))
    for (oh, ow), cases in by_size.items():              # one ragged batch per output size
        src, descs = _descs([c for _, c in cases])
        out, u8 = eng.preprocess(src, descs, (oh, ow), g["mean"], g["std"], torch.float32, want_u8=True)
        out16 = eng.preprocess(src, descs, (oh, ow), g["mean"], g["std"], torch.float16)
        for k, (i, _) in enumerate(cases):
            assert np.array_equal(u8[k].cpu().numpy(), g[f"c{i}_u8"]), f"case {i}: bytes differ from Pillow"
            assert np.array_equal(out[k].cpu().numpy(), g[f"c{i}_f32"]), f"case {i}: fp32 differs from torch"
            assert torch.equal(out16[k].cpu(), torch.from_numpy(g[f"c{i}_f32"]).half()), f"case {i}: f16 is not the rounded fp32"


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_ragged_batches_equal_the_oracle(eng, seed):
    rng = np.random.default_rng(seed)
    R = [32, 224, 48][seed]
    items, want = [], []
    for k in range(24):
        H, W = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        a = rng.integers(0, 256, (H, W, 3)).astype(np.uint8) if k % 2 else \
            np.clip(np.add.outer(np.arange(H) * 2, np.arange(W) * 3)[..., None] % 256 + rng.integers(-3, 4, (H, W, 3)), 0, 255).astype(np.uint8)
        ch, cw = int(rng.integers(1, H + 1)), int(rng.integers(1, W + 1))
        ct, cl = int(rng.integers(0, H - ch + 1)), int(rng.integers(0, W - cw + 1))
        if k % 5 == 0:                                   # eval style: window inside a larger resized image
            rh, rw = R + int(rng.integers(0, 40)), R + int(rng.integers(0, 40))
            ot, ol = int(rng.integers(0, rh - R + 1)), int(rng.integers(0, rw - R + 1))
        else:
            rh = rw = R
            ot = ol = 0
        flip = int(rng.integers(0, 2))
        items.append((a, (ct, cl, ch, cw), (rh, rw), (ot, ol), flip))
        want.append(PO.preprocess(a, (ct, cl, ch, cw), (rh, rw), (ot, ol, R, R), flip, (0.5, 0.4, 0.3), (0.2, 0.25, 0.3)))
    src, descs = _descs(items)
    out, u8 = eng.preprocess(src, descs, R, (0.5, 0.4, 0.3), (0.2, 0.25, 0.3), torch.float32, want_u8=True)
    for k, (w8, w32) in enumerate(want):
        assert np.array_equal(u8[k].cpu().numpy(), w8), f"image {k}: bytes"
        assert np.array_equal(out[k].cpu().numpy(), w32), f"image {k}: floats"


def test_device_transform_train_and_eval_paths(eng):
    from mvlpt_amd.transforms import CLIP_MEAN, CLIP_STD, DeviceTransform
    rng = np.random.default_rng(3)
    imgs = [rng.integers(0, 256, (h, w, 3)).astype(np.uint8) for h, w in [(120, 160), (160, 120), (64, 64), (50, 200)]]
    for train in (True, False):
        tr = DeviceTransform(eng, size=32, train=train, generator=torch.Generator().manual_seed(11))
        descs, _ = DeviceTransform(eng, size=32, train=train, generator=torch.Generator().manual_seed(11)).describe([im.shape[:2] for im in imgs])
        out, u8 = tr(imgs, want_u8=True)
        assert out.shape == (4, 3, 32, 32) and out.dtype == torch.float32
        for k, (im, d) in enumerate(zip(imgs, descs)):
            w8, w32 = PO.preprocess(im, (d.crop_top, d.crop_left, d.crop_height, d.crop_width), (d.resize_height, d.resize_width),
                                    (d.out_top, d.out_left, 32, 32), d.flip, CLIP_MEAN, CLIP_STD)
            assert np.array_equal(u8[k].cpu().numpy(), w8) and np.array_equal(out[k].cpu().numpy(), w32)


def test_output_feeds_the_image_tower(eng):
    """The produced batch is what `mvlpt_image_fwd` consumes (fp32 or the compute dtype)."""
    from mvlpt_amd.transforms import DeviceTransform
    from mvlpt_amd.weights import ARCHS, make_state_dict
    from mvlpt_amd.engine import Engine
    rng = np.random.default_rng(4)
    imgs = [rng.integers(0, 256, (40 + 7 * i, 50 + 3 * i, 3)).astype(np.uint8) for i in range(5)]
    E = Engine.from_state_dict(make_state_dict(ARCHS["tiny"], seed=2), "fp16")
    x32 = DeviceTransform(E, size=32, train=False)(imgs)
    x16 = DeviceTransform(E, size=32, train=False, out_dtype=torch.float16)(imgs)
    f32, f16 = E.image_fwd(x32), E.image_fwd(x16)
    assert torch.isfinite(f32).all() and float((f32 - f16).abs().max()) <= 2e-2 * float(f32.abs().max())


def test_loud_failures(eng):
    a = np.zeros((10, 12, 3), np.uint8)
    for bad in [((0, 0, 11, 12), (8, 8), (0, 0)), ((0, 0, 10, 12), (8, 8), (1, 0)), ((0, 0, 10, 12), (0, 8), (0, 0))]:
        src, descs = _descs([(a, bad[0], bad[1], bad[2], 0)])
        with pytest.raises(RuntimeError, match="descriptor 0"):
            eng.preprocess(src, descs, 8, (0, 0, 0), (1, 1, 1))
    src, descs = _descs([(a, (0, 0, 10, 12), (8, 8), (0, 0), 0)])
    with pytest.raises(RuntimeError):                     # image extends past the end of src
        eng.preprocess(src[:100], descs, 8, (0, 0, 0), (1, 1, 1))
    with pytest.raises(RuntimeError, match="no CPU path"):
        eng.preprocess(src.cpu(), descs, 8, (0, 0, 0), (1, 1, 1))
