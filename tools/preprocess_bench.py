"""Throughput of the device input pipeline (mvlpt_preprocess) on an ImageNet-like batch, with its HBM roofline and the
CPU baselines (the C oracle and, when importable, Pillow itself) on one host core.  GPU box only.
Usage: python tools/preprocess_bench.py [batch] [height] [width]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mvlpt_amd.engine import Engine
from mvlpt_amd.transforms import CLIP_MEAN, CLIP_STD, DeviceTransform
from mvlpt_amd.weights import ARCHS

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = int(sys.argv[2]) if len(sys.argv) > 2 else 375
W = int(sys.argv[3]) if len(sys.argv) > 3 else 500
R = 224
eng = Engine(ARCHS["tiny"], "fp16")
rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, (H, W, 3)).astype(np.uint8) for _ in range(B)]
res = {}
for name, train, dt in [("train random_resized_crop+flip -> f16", True, torch.float16), ("train -> fp32", True, torch.float32),
                        ("eval Resize+CenterCrop -> f16", False, torch.float16)]:
    tr = DeviceTransform(eng, size=R, train=train, out_dtype=dt, generator=torch.Generator().manual_seed(1))
    descs, total = tr.describe([(H, W)] * B)
    host, n = tr.pack(imgs)
    src = host[:n].cuda()
    fn = lambda: eng.preprocess(src, descs, R, CLIP_MEAN, CLIP_STD, dt)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    # algorithmic bytes: crop box read once, 8-bit intermediate (rows the vertical pass needs ~ crop rows) written + read,
    # output written
    osz = 2 if dt == torch.float16 else 4
    by = sum(d.crop_height * d.crop_width * 3 + 2 * d.crop_height * R * 3 + 3 * R * R * osz for d in descs)
    # end to end incl. the pinned H2D copy of the decoded images
    t0 = time.perf_counter()
    for _ in range(5):
        host[:n].cuda(non_blocking=True); fn()
    torch.cuda.synchronize()
    e2e = (time.perf_counter() - t0) / 5 * 1e3
    res[name] = {"ms_per_batch": round(ms, 4), "images_per_s": round(B / ms * 1e3), "algorithmic_GB": round(by / 1e9, 4),
                 "achieved_GBps": round(by / ms / 1e6, 1), "hbm_frac_of_8TBps": round(by / ms / 1e6 / 8000, 4),
                 "with_pcie_h2d_ms": round(e2e, 3), "src_MB": round(n / 1e6, 1)}
# CPU baselines on one core, 16 images
from oracle import preprocess_oracle as PO
tr = DeviceTransform(eng, size=R, train=True, generator=torch.Generator().manual_seed(1))
descs, _ = tr.describe([(H, W)] * 16)
t0 = time.perf_counter()
for im, d in zip(imgs, descs):
    PO.preprocess(im, (d.crop_top, d.crop_left, d.crop_height, d.crop_width), (R, R), (0, 0, R, R), d.flip, CLIP_MEAN, CLIP_STD)
res["cpu_oracle_images_per_s_1core"] = round(16 / (time.perf_counter() - t0), 1)
try:
    from PIL import Image
    mean, std = torch.tensor(CLIP_MEAN)[:, None, None], torch.tensor(CLIP_STD)[:, None, None]
    torch.set_num_threads(1)
    t0 = time.perf_counter()
    for im, d in zip(imgs, descs):
        p = Image.fromarray(im).crop((d.crop_left, d.crop_top, d.crop_left + d.crop_width, d.crop_top + d.crop_height)).resize((R, R), Image.BICUBIC)
        if d.flip: p = p.transpose(Image.FLIP_LEFT_RIGHT)
        t = (torch.from_numpy(np.asarray(p).copy()).permute(2, 0, 1).float().div(255) - mean) / std
    res["cpu_pillow_images_per_s_1core"] = round(16 / (time.perf_counter() - t0), 1)
except ImportError:
    pass
print(json.dumps({"batch": B, "source": [H, W], "out": R, **res}, indent=1))
