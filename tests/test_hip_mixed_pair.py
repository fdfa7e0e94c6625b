"""Kernel-level parity (-m gpu) of the MIXED PAIR, the default split format of the towers that carry a gradient
(include/mvlpt_hip.h, "mixed pair"): A = hi (16 bit) + one e5m2 residual byte, multiplied as hi*W16 on the 16-bit MFMA plus
residual*W8 (e4m3 copy of the frozen weight) on the block-scaled fp8 MFMA.  Checked against double-precision products of the
same seeded inputs: the mixed product has to sit an order of magnitude below the single-operand rounding error, and every
producer (cast, LayerNorm forward/backward, the GELU / GELU' epilogues, the attention core's outputs) has to write the format
the GEMM reads."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import clip_oracle as O  # noqa: E402


def _eng():
    from mvlpt_amd import engine
    return engine


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)


DTYPES = [torch.float16, torch.bfloat16]
# value carried by (hi, residual byte): hi has 11 / 8 significant bits, the e5m2 byte adds 3 -> 2^-15 / 2^-12 worst case
PAIR_TOL = {torch.float16: 2.0 ** -14, torch.bfloat16: 2.0 ** -11}
# the product additionally sees the weight's e4m3 copy on the residual term: measured ~1e-5 (fp16), bounded here at
GEMM_TOL = {torch.float16: 4e-5, torch.bfloat16: 4e-4}


@pytest.mark.parametrize("dtype", DTYPES)
def test_cast_mixed_carries_the_value(dtype):
    E = _eng()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(300, 256, generator=g) * torch.logspace(-3, 2, 300).view(300, 1)      # rows of very different magnitude
    p = E.op_cast_mixed(x.cuda(), dtype)
    got = E.join_mixed(p).cpu()
    assert p.shape == (300, 512)
    assert bool(torch.equal(p[:, :256].cpu(), x.to(dtype)))                                  # hi plane = round16(x), bit-exact
    rel_rows = ((got - x).abs().amax(1) / x.abs().amax(1))
    assert float(rel_rows.max()) < PAIR_TOL[dtype], float(rel_rows.max())                  # per row: no tensor-level scale involved


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (300, 256, 256), (77, 128, 512), (1000, 768, 3072), (4096, 2304, 768), (7700, 512, 2048),
                                   (7700, 512, 512), (7700, 512, 1536), (250, 768, 3072)])     # one-tile-per-CU problems: the kernel with dedicated data-movement waves
def test_gemm_mixed_operand(dtype, M, N, K):
    E = _eng()
    g = torch.Generator().manual_seed(M + N + K + 1)
    A = torch.randn(M, K, generator=g)
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype).float()                      # frozen weights are exactly 16-bit
    bias = torch.randn(N, generator=g)
    ref = A.double() @ W.double().t() + bias.double()
    A2 = E.op_cast_mixed(A.cuda(), dtype)
    Wp, e8 = E.op_pack_weight_mixed(W.cuda(), dtype)
    assert Wp.shape == (N, K + K // 2) and bool(torch.equal(Wp[:, :K].float().cpu(), W))
    out = E.op_gemm_mixed(A2, Wp, e8, E._lib.EPI_STORE32, bias=bias.cuda())
    err = relerr(out, ref)
    single = relerr(A.to(dtype).double() @ W.double().t() + bias.double(), ref)             # what one 16-bit operand would give
    assert err < GEMM_TOL[dtype] and err < single / 6, (err, single)
    resid = torch.randn(M, N, generator=g)
    outr = E.op_gemm_mixed(A2, Wp, e8, E._lib.EPI_RESID32, bias=bias.cuda(), resid=resid.cuda())
    assert relerr(outr, ref + resid.double()) < GEMM_TOL[dtype]
    # the dX orientation: same weight, transposed pack
    G = torch.randn(M, N, generator=g)
    Wt, e8t = E.op_pack_weight_mixed(W.cuda(), dtype, transposed=True)
    assert Wt.shape == (K, N + N // 2) and e8t == e8
    dx = E.op_gemm_mixed(E.op_cast_mixed(G.cuda(), dtype), Wt, e8t, E._lib.EPI_STORE32)
    assert relerr(dx, G.double() @ W.double()) < GEMM_TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_mixed_weight_exponent_follows_the_weight(dtype):
    """Weights of very different magnitude: the fp8 plane's exponent is per tensor, the product keeps its accuracy."""
    E = _eng()
    g = torch.Generator().manual_seed(3)
    M, N, K = 200, 128, 256
    A = torch.randn(M, K, generator=g)
    for scale in (1e-3, 1.0, 30.0):
        W = (torch.randn(N, K, generator=g) * scale).to(dtype).float()
        Wp, e8 = E.op_pack_weight_mixed(W.cuda(), dtype)
        out = E.op_gemm_mixed(E.op_cast_mixed(A.cuda(), dtype), Wp, e8, E._lib.EPI_STORE32)
        assert relerr(out, A.double() @ W.double().t()) < GEMM_TOL[dtype], scale
        assert 128.0 <= float(W.abs().max()) * 2.0 ** e8 < 256.0


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_mixed_epilogues(dtype):
    E = _eng()
    g = torch.Generator().manual_seed(9)
    M, N, K = 391, 512, 256
    A = torch.randn(M, K, generator=g)
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype).float()
    bias = torch.randn(N, generator=g)
    acc = (A.double() @ W.double().t()).float()
    A2 = E.op_cast_mixed(A.cuda(), dtype)
    Wp, e8 = E.op_pack_weight_mixed(W.cuda(), dtype)
    a_mixed, u16 = E.op_gemm_mixed(A2, Wp, e8, E._lib.EPI_GELU_SPLIT, bias=bias.cuda(), out2=True)
    assert a_mixed.shape == (M, 2 * N)
    assert relerr(E.join_mixed(a_mixed), O.quick_gelu(acc + bias)) < 2 * PAIR_TOL[dtype]
    assert relerr(u16, acc + bias) < (2e-3 if dtype == torch.float16 else 1.6e-2)
    u = torch.randn(M, N, generator=g).to(dtype)
    d_mixed = E.op_gemm_mixed(A2, Wp, e8, E._lib.EPI_GELUBWD_SPLIT, aux=u.cuda())
    assert relerr(E.join_mixed(d_mixed), acc * O.quick_gelu_grad(u.float())) < 2 * PAIR_TOL[dtype]
    # the pair that feeds the attention core stays a 16-bit hi|lo pair
    s_pair = E.op_gemm_mixed(A2, Wp, e8, E._lib.EPI_STORE_SPLIT, bias=bias.cuda())
    assert s_pair.shape == (M, 2 * N) and relerr(E.join_pair(s_pair), acc + bias) < GEMM_TOL[dtype]
    # and a mixed output is directly the next GEMM's A operand
    W2 = (torch.randn(128, N, generator=g) * N ** -0.5).to(dtype).float()
    W2p, e82 = E.op_pack_weight_mixed(W2.cuda(), dtype)
    y = E.op_gemm_mixed(a_mixed, W2p, e82, E._lib.EPI_STORE32)
    assert relerr(y, O.quick_gelu(acc + bias).double() @ W2.double().t()) < 3 * GEMM_TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("d,rows", [(128, 203), (512, 4099), (768, 20011), (1024, 300)])
def test_layernorm_mixed_outputs(dtype, d, rows):
    E = _eng()
    g = torch.Generator().manual_seed(d + 1)
    x = torch.randn(rows, d, generator=g) * 3 + 0.5
    gamma = 1 + 0.1 * torch.randn(d, generator=g)
    beta = 0.1 * torch.randn(d, generator=g)
    y_ref, (xhat, rstd) = O.layernorm_fwd(x, gamma, beta)
    y2 = E.op_layernorm_fwd_mixed(x.cuda(), gamma.cuda(), beta.cuda(), dtype)
    assert y2.shape == (rows, 2 * d) and relerr(E.join_mixed(y2), y_ref) < PAIR_TOL[dtype]
    dy = torch.randn(rows, d, generator=g)
    resid = torch.randn(rows, d, generator=g)
    dx_ref = resid + O.layernorm_bwd(dy, xhat, rstd, gamma)
    dx32, dx2 = E.op_layernorm_bwd_mixed(dy.cuda(), x.cuda(), gamma.cuda(), dtype, resid.cuda())
    assert relerr(dx32, dx_ref) < 2e-5
    assert relerr(E.join_mixed(dx2), dx32) < PAIR_TOL[dtype]


@pytest.mark.parametrize("L,causal", [(5, True), (17, False), (77, True), (80, False), (81, True), (197, False), (205, False), (257, True), (581, False)])
def test_attention32_mixed_outputs(L, causal):
    """Same kernels as test_attention32_fwd_bwd with the GEMM-side tensors (O; dQ|dK|dV) written as mixed pairs: the values
    have to agree with the 16-bit-pair outputs to the mixed pair's own resolution, and delta = rowsum(dO * O) read from a mixed O
    must not move the gradients."""
    E = _eng()
    N, H = 3, 2
    d = H * 64
    dtype = torch.float16
    g = torch.Generator().manual_seed(L * 2 + int(causal) + 300)
    qkv = torch.randn(N * L, 3 * d, generator=g)
    qp = E.split_pair(qkv.cuda(), dtype)
    o_pair, lse = E.op_attention32_fwd_pair(qp, N, L, H, causal)
    o_mix, lse_m = E.op_attention32_fwd_mixed(qp, N, L, H, causal)
    assert bool(torch.equal(lse, lse_m))
    assert bool(torch.equal(o_pair[:, :d], o_mix[:, :d]))                          # same hi plane
    assert relerr(E.join_mixed(o_mix), E.join_pair(o_pair)) < PAIR_TOL[dtype]
    dout = E.split_pair(torch.randn(N * L, d, generator=g).cuda(), dtype)
    ref = E.join_pair(E.op_attention32_bwd_pair(qp, o_pair, dout, lse, N, L, H, causal))
    got = E.join_mixed(E.op_attention32_bwd_mixed(qp, o_mix, dout, lse, N, L, H, causal))
    for i, nm in enumerate("qkv"):
        r_i = ref[:, i * d:(i + 1) * d]
        if float(r_i.abs().max()) < 1e-6:
            continue
        e = relerr(got[:, i * d:(i + 1) * d], r_i)
        assert e < 3 * PAIR_TOL[dtype], f"d{nm}: {e}"


def test_attention32_persistent_mixed_output():
    """The persistent resident forward (enough heads to give every workgroup several) writes the same hi plane and lse in both output
    formats, and the mixed O carries the pair's value to the mixed pair's resolution."""
    E = _eng()
    N, H, L = 43, 12, 205
    d = H * 64
    dtype = torch.float16
    qkv = torch.randn(N * L, 3 * d, generator=torch.Generator().manual_seed(77))
    qp = E.split_pair(qkv.cuda(), dtype)
    o_pair, lse = E.op_attention32_fwd_pair(qp, N, L, H, False)
    o_mix, lse_m = E.op_attention32_fwd_mixed(qp, N, L, H, False)
    assert bool(torch.equal(lse, lse_m)) and bool(torch.equal(o_pair[:, :d], o_mix[:, :d]))
    assert relerr(E.join_mixed(o_mix), E.join_pair(o_pair)) < PAIR_TOL[dtype]

