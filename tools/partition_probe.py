"""CU-partitioned streams (mvlpt_stream_create_cus) measured one thing at a time. GPU box only.
  a) one GEMM shape on the default stream and on partitions of 256 / 224 / 192 / 128 / 64 CUs (time and launch overhead),
  b) the image tower alone and the text tower alone on partitions,
  c) both towers concurrently: shared streams vs disjoint partitions."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd import engine as E
from mvlpt_amd.model import FrozenCLIP, build_prompt_layout
from mvlpt_amd.weights import ARCHS, make_state_dict

dev = torch.device("cuda:0")
total = E.device_cus(dev)
print("device CUs", total)


def timed(fn, stream, iters=20, warm=3):
    with torch.cuda.stream(stream):
        for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def streams():
    out = [("default", torch.cuda.current_stream()), ("plain side stream", torch.cuda.Stream())]
    for first, n in [(0, total), (0, 224), (32, 224), (0, 192), (64, 192), (0, 128), (128, 128), (0, 64), (0, 32)]:
        out.append((f"cus[{first},{first + n})", E.partition_stream(dev, first, n)))
    return out


sel = sys.argv[1] if len(sys.argv) > 1 else "abc"
if "a" in sel:
    for (M, N, K, epi) in [(50432, 3072, 768, 1), (50432, 768, 768, 2), (7700, 512, 2048, 2), (256, 768, 768, 0)]:
        A = torch.randn(M, K, device=dev).half(); Bt = (torch.randn(N, K, device=dev) * K ** -0.5).half()
        bias = torch.randn(N, device=dev); resid = torch.randn(M, N, device=dev) if epi == 2 else None
        for name, st in streams():
            ms = timed(lambda: E.op_gemm(A, Bt, epi, bias=bias, resid=resid), st)
            print(f"gemm {M}x{N}x{K} epi{epi} on {name:22s}: {ms * 1e3:8.1f} us")

arch = ARCHS["ViT-B/16"]
if "b" in sel or "c" in sel:
    eng = FrozenCLIP(make_state_dict(arch, 1)).engine
    x = torch.randn(256, 3, 224, 224, device=dev).half()
    C, L, n = 100, 77, 16
    nl = [1 + (i % 3) for i in range(C)]
    layout = build_prompt_layout(nl, n, L, "middle").cuda()
    eot = torch.tensor([n + v + 2 for v in nl], dtype=torch.int32).cuda()
    pre = torch.randn(C, 1, 512, device=dev) * 0.02; suf = torch.randn(C, L - 1 - n, 512, device=dev) * 0.02
    ctx = torch.randn(n, 512, device=dev) * 0.02; dfeat = torch.randn(C, 512, device=dev) * 1e-3

    def image(): eng.image_fwd(x)

    def text():
        eng.text_fwd(pre, suf, ctx, layout, eot, save_for_bwd=True)
        eng.text_bwd(dfeat)
if "b" in sel:
    for name, st in streams():
        print(f"image tower on {name:22s}: {timed(image, st, 10):8.3f} ms    text fwd+bwd: {timed(text, st, 10):8.3f} ms")
if "c" in sel:
    def both(si, stx, iters=10):
        def go(k):
            with torch.cuda.stream(si):
                for _ in range(k): image()
            with torch.cuda.stream(stx):
                for _ in range(k): text()
        go(2); torch.cuda.synchronize()
        t0 = time.perf_counter(); go(iters); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3
    print(f"both towers, two plain streams: {both(torch.cuda.Stream(), torch.cuda.Stream()):.3f} ms per (image + text)")
    for t in (32, 64, 96, 128):
        print(f"both towers, text on cus[0,{t}) image on cus[{t},{total}): "
              f"{both(E.partition_stream(dev, t, total - t), E.partition_stream(dev, 0, t)):.3f} ms")
    print(f"both towers, image plain stream + text on cus[0,64): {both(torch.cuda.Stream(), E.partition_stream(dev, 0, 64)):.3f} ms")
