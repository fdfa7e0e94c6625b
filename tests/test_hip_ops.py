"""Kernel-level parity (-m gpu): every hand-written HIP kernel, called through the C ABI, against the CPU
oracle (oracle/clip_oracle.py) on the same seeded inputs.  Tolerances are relative to the reference's max
magnitude; inputs are rounded to the 16-bit compute type first so only the kernel's own arithmetic differs."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import clip_oracle as O  # noqa: E402


def _eng():
    from mvlpt_amd import engine
    return engine


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)


DTYPES = [torch.float16, torch.bfloat16]
TOL = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}   # output rounding of the 16-bit type dominates


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 256, 192), (77, 128, 128), (1000, 768, 3072), (4096, 2304, 768)])
def test_gemm_store16_bias(dtype, M, N, K):
    E = _eng()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dtype)
    Bt = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype)
    bias = torch.randn(N, generator=g)
    ref = A.float() @ Bt.float().t() + bias
    out = E.op_gemm(A.cuda(), Bt.cuda(), E._lib.EPI_STORE16, bias=bias.cuda())
    assert relerr(out, ref) < TOL[dtype]
    out_nb = E.op_gemm(A.cuda(), Bt.cuda(), E._lib.EPI_STORE16)
    assert relerr(out_nb, ref - bias) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogues(dtype):
    E = _eng()
    g = torch.Generator().manual_seed(5)
    M, N, K = 391, 512, 256
    A = torch.randn(M, K, generator=g).to(dtype)
    Bt = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype)
    bias = torch.randn(N, generator=g)
    acc = A.float() @ Bt.float().t()
    # fp32 store: only accumulation-order error
    out32 = E.op_gemm(A.cuda(), Bt.cuda(), E._lib.EPI_STORE32, bias=bias.cuda())
    assert relerr(out32, acc + bias) < 2e-5
    # residual
    resid = torch.randn(M, N, generator=g)
    outr = E.op_gemm(A.cuda(), Bt.cuda(), E._lib.EPI_RESID32, bias=bias.cuda(), resid=resid.cuda())
    assert relerr(outr, acc + bias + resid) < 2e-5
    # in-place residual (out aliases resid) is what the towers use
    # QuickGELU + saved pre-activation
    a16, u16 = E.op_gemm(A.cuda(), Bt.cuda(), E._lib.EPI_GELU, bias=bias.cuda(), out2=True)
    assert relerr(u16, acc + bias) < TOL[dtype]
    assert relerr(a16, O.quick_gelu(acc + bias)) < TOL[dtype]
    # GELU backward epilogue
    u = torch.randn(M, N, generator=g).to(dtype)
    outg = E.op_gemm(A.cuda(), Bt.cuda(), E._lib.EPI_GELUBWD, aux=u.cuda())
    assert relerr(outg, acc * O.quick_gelu_grad(u.float())) < TOL[dtype]


@pytest.mark.parametrize("d,rows", [(128, 203), (512, 203), (768, 203), (1024, 203),
                                    # >= 4096 rows: the grid-stride kernels (next-row prefetch; 8192 rows in flight on 256 CUs,
                                    # so 20011 rows make every wave loop 2-3 times and end on a ragged tail)
                                    (512, 4099), (768, 20011), (1024, 9000), (1280, 4500)])
def test_layernorm_fwd_bwd(d, rows):
    E = _eng()
    g = torch.Generator().manual_seed(d)
    x = torch.randn(rows, d, generator=g) * 3 + 0.5
    gamma = 1 + 0.1 * torch.randn(d, generator=g)
    beta = 0.1 * torch.randn(d, generator=g)
    y_ref, (xhat, rstd) = O.layernorm_fwd(x, gamma, beta)
    y32 = E.op_layernorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), torch.float32)
    assert relerr(y32, y_ref) < 1e-5
    for dtype in DTYPES:
        y16 = E.op_layernorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), dtype)
        assert relerr(y16, y_ref) < TOL[dtype]
        dy = torch.randn(rows, d, generator=g).to(dtype)
        resid = torch.randn(rows, d, generator=g)
        dx_ref = resid + O.layernorm_bwd(dy.float(), xhat, rstd, gamma)
        dx32, dx16 = E.op_layernorm_bwd(dy.cuda(), x.cuda(), gamma.cuda(), resid.cuda())
        assert relerr(dx32, dx_ref) < 2e-5
        assert relerr(dx16, dx_ref) < TOL[dtype]


def _attn_ref(qkv, N, L, H, causal):
    d = H * 64
    q, k, v = (t.reshape(N, L, H, 64).permute(0, 2, 1, 3) for t in qkv.float().reshape(N, L, 3 * d).split(d, dim=-1))
    o, p = O.attention_fwd(q, k, v, causal)
    return q, k, v, o, p


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("L,causal", [(5, False), (5, True), (24, True), (50, False), (77, True), (197, False), (205, False), (256, False),
                                      (261, False), (581, False)])   # ViT-L/14 @224 / @336 (+4 prompts): online-softmax blocks
def test_attention_fwd_bwd(dtype, L, causal):
    E = _eng()
    N, H = 3, 2
    d = H * 64
    g = torch.Generator().manual_seed(L * 2 + int(causal))
    qkv = torch.randn(N * L, 3 * d, generator=g).to(dtype)
    q, k, v, o, p = _attn_ref(qkv, N, L, H, causal)
    out, lse = E.op_attention_fwd(qkv.cuda(), N, L, H, causal)
    o_ref = o.permute(0, 2, 1, 3).reshape(N * L, d)
    assert relerr(out, o_ref) < TOL[dtype] * 1.5
    # log-sum-exp of the scaled scores
    s = torch.matmul(q, k.transpose(-1, -2)) / 8.0
    if causal:
        s = s + torch.full((L, L), float("-inf")).triu_(1)
    lse_ref = torch.logsumexp(s, -1).reshape(-1)
    assert float((lse.cpu() - lse_ref).abs().max()) < 2e-2
    # backward
    dout = torch.randn(N * L, d, generator=g).to(dtype)
    do = dout.float().reshape(N, L, H, 64).permute(0, 2, 1, 3)
    dq, dk, dv = O.attention_bwd(do, q, k, v, p)
    dqkv_ref = torch.cat([t.permute(0, 2, 1, 3).reshape(N * L, d) for t in (dq, dk, dv)], dim=-1)
    dqkv = E.op_attention_bwd(qkv.cuda(), out, dout.cuda(), lse, N, L, H, causal)
    for i, nm in enumerate("qkv"):
        e = relerr(dqkv[:, i * d:(i + 1) * d], dqkv_ref[:, i * d:(i + 1) * d])
        assert e < TOL[dtype] * 3, f"d{nm}: {e}"


def test_attention_is_not_transposed():
    """Asymmetric probe: one query attends to one spiked key; V rows are distinct ramps."""
    E = _eng()
    N, L, H = 1, 40, 1
    qkv = torch.zeros(L, 192)
    qkv[:, 128:] = torch.arange(L).float().view(L, 1) * 0.01 + torch.arange(64).float().view(1, 64) * 1e-3
    qkv[7, 0] = 8.0       # q_7
    qkv[23, 64] = 8.0     # k_23  -> score(7,23) = 64/8 = 8
    out, _ = E.op_attention_fwd(qkv.half().cuda(), N, L, H, False)
    _, _, _, o, _ = _attn_ref(qkv.half(), N, L, H, False)
    assert relerr(out, o.reshape(L, 64)) < 3e-3


@pytest.mark.parametrize("M,N,K,epi", [(16640, 768, 3072, 2),     # phased 256x128 kernel (two wave groups one phase apart)
                                       (30000, 2304, 768, 0),     # 256x256 geometry, ragged last M tile
                                       (25600, 768, 768, 2),      # 256x128 3-stage ring
                                       (7700, 512, 2048, 2)])     # 128x128, ragged M
def test_gemm_race_screen(M, N, K, epi):
    """The pipelined GEMMs keep LDS-DMA in flight across barriers: a mis-placed wait shows up as rare wrong tiles that
    depend on timing.  Run every geometry many times: all runs must be BITWISE identical and match the reference."""
    E = _eng()
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=g).half().cuda()
    Bt = (torch.randn(N, K, generator=g) * K ** -0.5).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    resid = torch.randn(M, N, generator=g).cuda() if epi == 2 else None
    ref = A.float() @ Bt.float().t() + bias + (resid if resid is not None else 0)
    first = None
    for it in range(25):
        if it % 5 == 4:   # perturb timing: run something else in between
            torch.randn(4096, 4096, device="cuda").sum()
        out = E.op_gemm(A, Bt, epi, bias=bias, resid=resid)
        if first is None:
            first = out.clone()
            assert relerr(out, ref) < (2e-5 if epi == 2 else 2e-3)
        else:
            assert torch.equal(out, first), f"run {it} differs from run 0 (race)"


# ---------------------------------------------------------------------------------------------- split-precision mode
# (hi+lo 16-bit operand pairs for the GEMM A operand and in attention: DESIGN.md "Precision modes").  The weights are
# exactly 16-bit (as CLIP checkpoints are); the activations are arbitrary fp32: the results must match an fp32 matmul to
# ~2^-21 relative, i.e. three orders of magnitude better than the single-operand kernels above.
SPLIT_TOL = {torch.float16: 3e-6, torch.bfloat16: 6e-5}    # pair = 22 / 16 significant bits


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (391, 512, 256), (77, 128, 128), (1000, 768, 3072), (4096, 2304, 768), (30000, 768, 768),
                                   (7700, 512, 512), (7700, 512, 2048)])      # dedicated data-movement waves
def test_gemm_split_operand(dtype, M, N, K):
    E = _eng()
    g = torch.Generator().manual_seed(M + N + K + 1)
    A = torch.randn(M, K, generator=g)                                   # fp32 activations, NOT pre-rounded
    Bt = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype)
    bias = torch.randn(N, generator=g)
    ref = (A.double() @ Bt.double().t() + bias.double()).float()
    A2 = E.split_pair(A, dtype).cuda()
    assert relerr(E.join_pair(A2), A) < SPLIT_TOL[dtype] * 0.5
    out = E.op_gemm_split(A2, Bt.cuda(), E._lib.EPI_STORE32, bias=bias.cuda())
    assert relerr(out, ref) < SPLIT_TOL[dtype]
    resid = torch.randn(M, N, generator=g)
    outr = E.op_gemm_split(A2, Bt.cuda(), E._lib.EPI_RESID32, bias=bias.cuda(), resid=resid.cuda())
    assert relerr(outr, ref + resid) < SPLIT_TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_split_epilogues(dtype):
    E = _eng()
    g = torch.Generator().manual_seed(9)
    M, N, K = 391, 512, 256
    A = torch.randn(M, K, generator=g)
    Bt = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype)
    bias = torch.randn(N, generator=g)
    acc = (A.double() @ Bt.double().t()).float()
    A2 = E.split_pair(A, dtype).cuda()
    a_pair, u16 = E.op_gemm_split(A2, Bt.cuda(), E._lib.EPI_GELU_SPLIT, bias=bias.cuda(), out2=True)
    assert a_pair.shape == (M, 2 * N)
    # the device's QuickGELU uses v_exp/v_rcp (~1 ulp each): fp32-level agreement, not pair-level
    assert relerr(E.join_pair(a_pair), O.quick_gelu(acc + bias)) < 2e-6 * (1 if dtype == torch.float16 else 40)
    assert relerr(u16, acc + bias) < TOL[dtype]
    s_pair = E.op_gemm_split(A2, Bt.cuda(), E._lib.EPI_STORE_SPLIT, bias=bias.cuda())
    assert s_pair.shape == (M, 2 * N) and relerr(E.join_pair(s_pair), acc + bias) < SPLIT_TOL[dtype]
    u = torch.randn(M, N, generator=g).to(dtype)
    d_pair = E.op_gemm_split(A2, Bt.cuda(), E._lib.EPI_GELUBWD_SPLIT, aux=u.cuda())
    assert relerr(E.join_pair(d_pair), acc * O.quick_gelu_grad(u.float())) < 2e-6 * (1 if dtype == torch.float16 else 40)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("d,rows", [(128, 203), (512, 4099), (768, 20011), (1024, 300)])
def test_layernorm_split_outputs(dtype, d, rows):
    E = _eng()
    g = torch.Generator().manual_seed(d + 1)
    x = torch.randn(rows, d, generator=g) * 3 + 0.5
    gamma = 1 + 0.1 * torch.randn(d, generator=g)
    beta = 0.1 * torch.randn(d, generator=g)
    y_ref, (xhat, rstd) = O.layernorm_fwd(x, gamma, beta)
    y2 = E.op_layernorm_fwd_split(x.cuda(), gamma.cuda(), beta.cuda(), dtype)
    assert y2.shape == (rows, 2 * d) and relerr(E.join_pair(y2), y_ref) < max(2e-6, SPLIT_TOL[dtype])
    dy = torch.randn(rows, d, generator=g)
    resid = torch.randn(rows, d, generator=g)
    dx_ref = resid + O.layernorm_bwd(dy, xhat, rstd, gamma)
    dx32, dx2 = E.op_layernorm_bwd_split(dy.cuda(), x.cuda(), gamma.cuda(), dtype, resid.cuda())
    assert relerr(dx32, dx_ref) < 2e-5
    assert relerr(E.join_pair(dx2), dx32) < SPLIT_TOL[dtype]          # the pair carries the fp32 result


@pytest.mark.parametrize("L,causal", [(5, False), (5, True), (1, False), (16, True), (17, False), (24, True), (64, True), (65, False), (77, True),
                                      (80, True), (80, False), (81, True), (81, False), (128, False), (129, True), (197, False), (205, False),
                                      (256, False), (257, True), (261, False), (581, False)])
def test_attention32_fwd_bwd(L, causal):
    """Attention core of the split-precision mode (pair operands, three-term products): fp32-level agreement with the oracle."""
    E = _eng()
    N, H = 3, 2
    d = H * 64
    g = torch.Generator().manual_seed(L * 2 + int(causal) + 100)
    qkv = torch.randn(N * L, 3 * d, generator=g)
    q, k, v, o, p = _attn_ref(qkv, N, L, H, causal)
    out2, lse = E.op_attention32_fwd(qkv.cuda(), N, L, H, causal)
    o_ref = o.permute(0, 2, 1, 3).reshape(N * L, d)
    assert relerr(E.join_pair(out2), o_ref) < 5e-6
    s = torch.matmul(q, k.transpose(-1, -2)) / 8.0
    if causal:
        s = s + torch.full((L, L), float("-inf")).triu_(1)
    assert float((lse.cpu() - torch.logsumexp(s, -1).reshape(-1)).abs().max()) < 1e-5
    dout = torch.randn(N * L, d, generator=g)
    do = dout.reshape(N, L, H, 64).permute(0, 2, 1, 3)
    dq, dk, dv = O.attention_bwd(do, q, k, v, p)
    dqkv_ref = torch.cat([t.permute(0, 2, 1, 3).reshape(N * L, d) for t in (dq, dk, dv)], dim=-1)
    dqkv = E.join_pair(E.op_attention32_bwd(qkv.cuda(), out2, dout.cuda(), lse, N, L, H, causal))
    for i, nm in enumerate("qkv"):
        got_i, ref_i = dqkv[:, i * d:(i + 1) * d].cpu(), dqkv_ref[:, i * d:(i + 1) * d]
        if float(ref_i.abs().max()) < 1e-6:          # L = 1: softmax over one key, dQ = dK = 0 exactly in the reference
            assert float(got_i.abs().max()) < 1e-5, f"d{nm}"
            continue
        e = relerr(got_i, ref_i)
        assert e < 1e-5, f"d{nm}: {e}"
    # CLS-only forward (last image block): only query 0 of every sequence is produced, the rest stays untouched
    if not causal and L > 1:
        o1, _ = E.op_attention32_fwd(qkv.cuda(), N, L, H, causal, q_rows=1)
        got = E.join_pair(o1).cpu().reshape(N, L, d)
        assert relerr(got[:, 0], o_ref.reshape(N, L, d)[:, 0]) < 5e-6 and float(got[:, 1:].abs().max()) == 0.0
        # ... and the backward over the whole sequence (dO = 0 off the CLS rows) must see P = 0 on the rows that were never
        # computed, whatever their lse buffer held before: the op marks them with lse = +huge
        lse1 = torch.full((N * H * L,), float("nan"), device="cuda")
        qp = E.split_pair(qkv.cuda(), torch.float16)
        o1 = torch.zeros(N * L, 2 * d, device="cuda", dtype=torch.float16)
        E._lib.check(E.lib.mvlpt_op_attention32_fwd(1, qp.data_ptr(), o1.data_ptr(), lse1.data_ptr(), N, L, H, 0, 1,
                                                    torch.cuda.current_stream().cuda_stream), None, "fwd")
        l1 = lse1.cpu().reshape(N, H, L)
        assert bool(torch.isfinite(l1).all()) and float(l1[:, :, 1:].min()) > 1e37
        d_cls = torch.zeros(N, L, d)
        d_cls[:, 0] = dout.reshape(N, L, d)[:, 0]
        dq, dk, dv = O.attention_bwd(d_cls.reshape(N, L, H, 64).permute(0, 2, 1, 3), q, k, v, p)
        ref1 = torch.cat([t.permute(0, 2, 1, 3).reshape(N * L, d) for t in (dq, dk, dv)], dim=-1)
        got1 = E.join_pair(E.op_attention32_bwd_pair(qp, o1, E.split_pair(d_cls.reshape(N * L, d).cuda(), torch.float16), lse1, N, L, H, False))
        assert bool(torch.isfinite(got1).all()) and relerr(got1, ref1) < 1e-5


@pytest.mark.parametrize("N,H,L", [(43, 12, 205), (64, 8, 197), (90, 6, 100), (171, 3, 81)])
def test_attention32_fwd_persistent_heads(N, H, L):
    """The persistent resident forward (80 < L <= 208, at least two (sequence, head) items per compute unit: each workgroup walks
    several heads and prefetches the next one's K / V through its three-buffer rotation) against the fp32 oracle, a ragged number of
    heads per workgroup included, for the pair and the mixed-pair output; two runs are bit-identical."""
    E = _eng()
    d = H * 64
    g = torch.Generator().manual_seed(N * 1000 + L)
    qkv = torch.randn(N * L, 3 * d, generator=g)
    q, k, v, o, p = _attn_ref(qkv, N, L, H, False)
    o_ref = o.permute(0, 2, 1, 3).reshape(N * L, d)
    out2, lse = E.op_attention32_fwd(qkv.cuda(), N, L, H, False)
    assert relerr(E.join_pair(out2), o_ref) < 5e-6
    s = torch.matmul(q, k.transpose(-1, -2)) / 8.0
    assert float((lse.cpu() - torch.logsumexp(s, -1).reshape(-1)).abs().max()) < 1e-5
    again, lse2 = E.op_attention32_fwd(qkv.cuda(), N, L, H, False)
    assert torch.equal(again, out2) and torch.equal(lse2, lse)
    # the backward consumes what the persistent forward left (out pair, lse)
    dout = torch.randn(N * L, d, generator=g)
    do = dout.reshape(N, L, H, 64).permute(0, 2, 1, 3)
    dq, dk, dv = O.attention_bwd(do, q, k, v, p)
    dqkv_ref = torch.cat([t.permute(0, 2, 1, 3).reshape(N * L, d) for t in (dq, dk, dv)], dim=-1)
    dqkv = E.join_pair(E.op_attention32_bwd(qkv.cuda(), out2, dout.cuda(), lse, N, L, H, False))
    assert relerr(dqkv.cpu(), dqkv_ref) < 1e-5


@pytest.mark.parametrize("N,H,L,causal", [(100, 8, 77, True), (67, 8, 77, True), (130, 4, 23, True), (75, 7, 80, False), (300, 2, 50, False)])
def test_attention32_short_many_heads(N, H, L, causal):
    """The short resident kernels (L <= 80) at the text tower's scale — 100 classes x 8 heads = 800 workgroups, more than the 768 resident
    slots, so a second ragged round runs — against the fp32 oracle: sequences with fewer than five tiles (idle waves keep staging and
    meeting the barriers), pair and mixed-pair output, bit-identical from run to run; the backward (own key / value rows out of the
    staged images, delta from the own dO fragments) consumes what the forward leaves.  (A persistent variant of the forward — two
    workgroups per CU walking heads with a second image set — was bit-identical and 14 % slower: NOTES round 6.)"""
    E = _eng()
    d = H * 64
    g = torch.Generator().manual_seed(N * 1000 + L + int(causal))
    qkv = torch.randn(N * L, 3 * d, generator=g)
    q, k, v, o, p = _attn_ref(qkv, N, L, H, causal)
    o_ref = o.permute(0, 2, 1, 3).reshape(N * L, d)
    out2, lse = E.op_attention32_fwd(qkv.cuda(), N, L, H, causal)
    assert relerr(E.join_pair(out2), o_ref) < 5e-6
    s = torch.matmul(q, k.transpose(-1, -2)) / 8.0
    if causal:
        s = s + torch.full((L, L), float("-inf")).triu_(1)
    assert float((lse.cpu() - torch.logsumexp(s, -1).reshape(-1)).abs().max()) < 1e-5
    for _ in range(3):
        again, lse2 = E.op_attention32_fwd(qkv.cuda(), N, L, H, causal)
        assert torch.equal(again, out2) and torch.equal(lse2, lse)
    qp = E.split_pair(qkv.cuda(), torch.float16)
    om, lsem = E.op_attention32_fwd_mixed(qp, N, L, H, causal)
    assert torch.equal(lsem, lse) and torch.equal(om[:, :d], out2[:, :d]) and relerr(E.join_mixed(om), o_ref) < 2.0 ** -13   # (one e5m2 byte of residual)
    dout = torch.randn(N * L, d, generator=g)
    do = dout.reshape(N, L, H, 64).permute(0, 2, 1, 3)
    dq, dk, dv = O.attention_bwd(do, q, k, v, p)
    dqkv_ref = torch.cat([t.permute(0, 2, 1, 3).reshape(N * L, d) for t in (dq, dk, dv)], dim=-1)
    dqkv = E.join_pair(E.op_attention32_bwd(qkv.cuda(), out2, dout.cuda(), lse, N, L, H, causal))
    assert relerr(dqkv.cpu(), dqkv_ref) < 1e-5


def test_attention32_is_not_transposed():
    E = _eng()
    N, L, H = 1, 40, 1
    qkv = torch.zeros(L, 192)
    qkv[:, 128:] = torch.arange(L).float().view(L, 1) * 0.01 + torch.arange(64).float().view(1, 64) * 1e-3
    qkv[7, 0] = 8.0
    qkv[23, 64] = 8.0
    out2, _ = E.op_attention32_fwd(qkv.cuda(), N, L, H, False)
    _, _, _, o, _ = _attn_ref(qkv, N, L, H, False)
    assert relerr(E.join_pair(out2), o.reshape(L, 64)) < 5e-6


@pytest.mark.parametrize("M,N,K,epi", [(50432, 768, 3072, "resid"),      # MLP down-projection of the headline: 591 tiles of 256x256
                                       (50432, 3072, 768, "gelu"),       # MLP up-projection: 2364 tiles
                                       (17920, 512, 512, "store32"),     # 128x128 tiles, two workgroups per CU: 560 tiles
                                       (33280, 768, 768, "store16"),     # 256x128 tiles: 780
                                       (52480, 768, 3072, "resid")])     # 205 row tiles (B = 256 with 8 prompt tokens)
def test_gemm_ragged_last_round_full_size(M, N, K, epi):
    """Full-size GEMMs of the BASELINE configurations whose persistent grid ends in a ragged round (591 / 2364 / 560 / 780 / 615
    tiles on 256 CUs), every epilogue family and geometry: right answer, bit-identical from run to run.  (Round 3 also built a
    deterministic stream-K split of that last round — commit 8445596 — and measured it neutral: DESIGN.md §5.)"""
    E = _eng()
    dtype = torch.float16
    g = torch.Generator().manual_seed(M % 1000 + N + K)
    A = torch.randn(M, K, generator=g).to(dtype).cuda()
    Bt = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype).cuda()
    bias = torch.randn(N, generator=g).cuda()
    ref = A.float() @ Bt.float().t() + bias            # on the GPU through torch (fp32 accumulate): 1e-5-level agreement expected
    kw = dict(bias=bias)
    if epi == "resid":
        resid = torch.randn(M, N, generator=g).cuda()
        run = lambda: E.op_gemm(A, Bt, E._lib.EPI_RESID32, resid=resid, **kw)
        want, tol = ref + resid, 3e-5
    elif epi == "gelu":
        run = lambda: E.op_gemm(A, Bt, E._lib.EPI_GELU, **kw)
        want, tol = O.quick_gelu(ref.cpu()), TOL[dtype]
    elif epi == "store32":
        run = lambda: E.op_gemm(A, Bt, E._lib.EPI_STORE32, **kw)
        want, tol = ref, 3e-5
    else:
        run = lambda: E.op_gemm(A, Bt, E._lib.EPI_STORE16, **kw)
        want, tol = ref, TOL[dtype]
    out0 = run()
    assert relerr(out0, want) < tol
    # the LAST rows are the ones the split round produces: check them on their own as well
    assert relerr(out0[-2048:], want[-2048:]) < tol
    # ... and against the ORACLE, not only the library matmul: the last 4096 rows (the ragged round) in fp64 on the host
    sl = slice(M - 4096, M)
    o64 = A[sl].double().cpu() @ Bt.double().cpu().t() + bias.double().cpu()
    if epi == "resid":
        o64 = o64 + resid[sl].double().cpu()
    elif epi == "gelu":
        o64 = O.quick_gelu(o64)
    assert relerr(out0[sl].cpu().double(), o64) < tol
    for _ in range(5):
        assert torch.equal(run(), out0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("L", [197, 205])
def test_attention_fwd_persistent_heads(dtype, L):
    """attn_fwdp_kernel (13 key tiles, >= 2 heads per compute unit: one workgroup per CU walks the (image, head) list with three K / V
    buffers in rotation): 48 images x 12 heads = 576 heads on 256 CUs, every head against the oracle — the first head of a workgroup
    (prologue staging), the middle ones (prefetched K / V / Q) and the ragged last round (workgroups without a third head)."""
    E = _eng()
    N, H = 48, 12
    d = H * 64
    g = torch.Generator().manual_seed(L)
    qkv = torch.randn(N * L, 3 * d, generator=g).to(dtype)
    out, lse = E.op_attention_fwd(qkv.cuda(), N, L, H, False)
    # without the log-sum-exp output (the headline's forward-only tower): the counted-wait path that leaves the previous head's stores
    # in flight — many launches back to back, bit-identical to the run that waits for everything
    for _ in range(20):
        out2, _ = E.op_attention_fwd(qkv.cuda(), N, L, H, False, want_lse=False)
        assert torch.equal(out, out2)
    worst = 0.0
    for n0 in range(0, N, 8):                       # the oracle in slices of 8 images
        q, k, v, o, p = _attn_ref(qkv[n0 * L:(n0 + 8) * L], 8, L, H, False)
        o_ref = o.permute(0, 2, 1, 3).reshape(8 * L, d)
        worst = max(worst, relerr(out[n0 * L:(n0 + 8) * L], o_ref))
        s = torch.matmul(q, k.transpose(-1, -2)) / 8.0
        assert float((lse[n0 * H * L:(n0 + 8) * H * L].cpu() - torch.logsumexp(s, -1).reshape(-1)).abs().max()) < 2e-2
    assert worst < TOL[dtype] * 1.5, worst
