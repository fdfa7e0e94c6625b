import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlpt_amd import engine as E
from oracle import clip_oracle as O
g = torch.Generator().manual_seed(9)
M, N, K = 391, 512, 256
dtype = torch.float16
A = torch.randn(M, K, generator=g)
Bt = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype)
bias = torch.randn(N, generator=g)
acc = (A.double() @ Bt.double().t()).float()
A2 = E.split_pair(A, dtype).cuda()
a_pair, u16 = E.op_gemm_split(A2, Bt.cuda(), E._lib.EPI_GELU_SPLIT, bias=bias.cuda(), out2=True)
ref = O.quick_gelu(acc + bias)
got = E.join_pair(a_pair).cpu()
hi = a_pair[:, :N].float().cpu(); lo = a_pair[:, N:].float().cpu()
err = (got - ref).abs()
print("max err", err.max(), "at", divmod(int(err.argmax()), N), "ref max", ref.abs().max())
i, j = divmod(int(err.argmax()), N)
print("ref", ref[i, j], "hi", hi[i, j], "lo", lo[i, j], "u", (acc + bias)[i, j], "u16", u16[i, j])
print("hi-only err", (hi - ref).abs().max())
# error as function of column block / row block
e2 = err.reshape(M, N)
print("err by 64-col block", [float(e2[:, c:c+64].max()) for c in range(0, N, 64)])
print("err by row block", [float(e2[r:r+64].max()) for r in range(0, M, 64)])
u32 = E.op_gemm_split(A2, Bt.cuda(), E._lib.EPI_STORE32, bias=bias.cuda()).cpu()
print("u32 err", (u32 - (acc + bias)).abs().max())
ref2 = O.quick_gelu(u32)
print("vs gelu(device u32)", (got - ref2).abs().max())
