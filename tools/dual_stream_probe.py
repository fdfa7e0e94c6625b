"""Experiment: image tower forward (ViT-B/16, B=256, fast mode) as ONE launch sequence vs TWO half-batches on two HIP
streams (two engine handles = two workspaces).  Tests whether cross-stream overlap hides kernel tails / HBM-bound phases."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd.model import FrozenCLIP
from mvlpt_amd.weights import ARCHS, make_state_dict
arch = ARCHS["ViT-B/16"]
B = 256
sd = make_state_dict(arch, 1)
e1 = FrozenCLIP(sd).engine
e2 = FrozenCLIP(sd).engine
x = torch.randn(B, 3, 224, 224, device="cuda").half()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"single stream B=256: {t(lambda: e1.image_fwd(x)):.3f} ms")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for parts in (2, 4):
    xs = x.chunk(parts)
    engs = [e1, e2]
    def dual():
        main = torch.cuda.current_stream()
        s1.wait_stream(main); s2.wait_stream(main)
        for i, xi in enumerate(xs):
            with torch.cuda.stream(s1 if i % 2 == 0 else s2):
                engs[i % 2].image_fwd(xi)
        main.wait_stream(s1); main.wait_stream(s2)
    print(f"two streams, {parts} chunks of {B // parts}: {t(dual):.3f} ms")
print(f"sequential halves one stream: {t(lambda: (e1.image_fwd(x[:128]), e1.image_fwd(x[128:]))):.3f} ms")
