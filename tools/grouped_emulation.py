"""Kernel-level estimate of the 'grouped launch' (text-tower GEMM tiles appended to an image GEMM's persistent
tile list), with the kernels that exist: the persistent 256x256 kernel's time is measured at M = 50 432 (image MLP up, 2 364
tiles of 12 K-stages) and at M = 54 528 (+192 tiles of 12 K-stages = the 186 256x256 tiles of the text QKV GEMM, 7 700 x 1536 at
12 K-stages); the difference is what hosting the text tiles costs, against the text GEMM launched by itself and against the two
launches back to back.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd import engine as E


def mk(M, N, K):
    return (torch.randn(M, K, device="cuda").half(), (torch.randn(N, K, device="cuda") * K ** -0.5).half(), torch.randn(N, device="cuda"))


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


img = mk(50432, 3072, 768)
host = mk(50432 + 16 * 256, 3072, 768)
txt = mk(7700, 1536, 768)
f_img = lambda: E.op_gemm(img[0], img[1], 1, bias=img[2])
f_host = lambda: E.op_gemm(host[0], host[1], 1, bias=host[2])
f_txt = lambda: E.op_gemm(txt[0], txt[1], 0, bias=txt[2])
def f_both():
    f_img(); f_txt()
for rep in range(3):
    a, h, t, b = timed(f_img), timed(f_host), timed(f_txt), timed(f_both)
    print(f"image MLP up alone {a:7.1f} us | +192 hosted tiles {h:7.1f} us (+{h - a:5.1f}) | text QKV-sized GEMM alone {t:6.1f} us | "
          f"back to back {b:7.1f} us (+{b - a:5.1f}) | hosted / alone = {(h - a) / t:.2f}")
