"""Packed residual stream (-m gpu; include/mvlpt_hip.h: mvlpt_set_resid_packed, mvlpt_op_respk_pack / respk_unpack / fold_weight /
gemm_residp): the prompt-free, gradient-free fp16 image tower carries `x = x + ...` (clip/model.py:185-188) as hi = round16(x) + one
byte and feeds hi straight into the GEMM behind each LayerNorm (gamma folded into the frozen weight).

* format: pack / unpack bit-exact against oracle/packed_stream.py (the numpy statement of the encoding);
* kernel level: residual update + LayerNorm + linear [+ QuickGELU] against fp64 torch for every tile geometry, ragged M, in place;
* tower level: image features of the full-size CoOp fixtures' towers, packed vs fp32 stream vs the fp32 oracle, and the fixtures
  themselves (logits / loss / gradients at the north_star 1e-3 bound) with the packed tower forced on;
* determinism: bit-identical run to run."""
import numpy as np
import pytest
import torch

from oracle import packed_stream as P
from tests.golden_util import load_npz

pytestmark = pytest.mark.gpu


def _E():
    from mvlpt_amd import engine
    return engine


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)


def _values(rows, d, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, d, generator=g) * 2.0 + 0.3 * torch.randn(rows, 1, generator=g)
    x[:, 7] += 25.0
    x[:, 11] *= 1e-4                                 # a channel in the fp16 subnormal range
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 2048.0 + 1.0, 1.0 + 2.0 ** -11, 2.0 ** -14, 60000.0])
    return x


@pytest.mark.parametrize("rows,d", [(5, 768), (1000, 512), (50432, 768), (333, 1024)])
def test_pack_unpack_bit_exact(rows, d):
    E = _E()
    x = _values(rows, d, rows + d)
    hi, lo, part = E.op_respk_pack(x.cuda(), ntp=8)
    rh, rl = P.pack(x.numpy())
    assert np.array_equal(hi.cpu().numpy(), rh)
    assert np.array_equal(lo.cpu().numpy(), rl)
    y = E.op_respk_unpack(hi, lo)
    assert np.array_equal(y.cpu().numpy(), P.unpack(rh, rl))
    s1, s2 = P.row_stats(x.numpy())
    p = part.double().cpu().numpy()
    assert np.all(p[:, 1:] == 0.0)
    assert np.abs(p[:, 0, 0] - s1).max() <= 1e-5 * np.abs(x.numpy()).sum(-1).max()
    assert np.abs(p[:, 0, 1] - s2).max() <= 1e-5 * s2.max()
    # strided rows (the CLS row of every sequence)
    if rows % 5 == 0:
        z = E.op_respk_unpack(hi, lo, row_mul=5)
        assert np.array_equal(z.cpu().numpy(), P.unpack(rh, rl)[::5])


def test_pack_saturates_values_beyond_the_fp16_range():
    """|x| > 65504: hi stays finite, the stored value is +-65504, bit for bit as oracle/packed_stream.py states."""
    E = _E()
    x = torch.zeros(4, 512)
    x[0, :8] = torch.tensor([7e4, -7e4, 1e9, -3e38, 65504.0, 65519.9, float("inf"), float("-inf")])
    x[1] = torch.randn(512) * 3e4
    hi, lo, _ = E.op_respk_pack(x.cuda(), ntp=8)
    rh, rl = P.pack(x.numpy())
    assert np.array_equal(hi.cpu().numpy(), rh) and np.array_equal(lo.cpu().numpy(), rl)
    assert bool(torch.isfinite(hi.float()).all())
    assert np.array_equal(E.op_respk_unpack(hi, lo).cpu().numpy(), P.unpack(rh, rl))


def test_fold_weight_bit_exact():
    E = _E()
    g = torch.Generator().manual_seed(1)
    W = (torch.randn(2304, 768, generator=g) * 768 ** -0.5).half()
    Wp = torch.zeros(2304, 768 + 384, dtype=torch.float16)       # the engine's packed rows have a pitch of 3K/2
    Wp[:, :768] = W
    gamma = 1.0 + 0.2 * torch.randn(768, generator=g)
    Wg, cs = E.op_fold_weight(Wp.cuda(), 768, gamma.cuda())
    rg, rs = P.fold_weight(W.numpy(), gamma.numpy())
    assert np.array_equal(Wg.cpu().numpy(), rg)
    assert np.abs(cs.double().cpu().numpy() - rs).max() < 1e-5 * np.abs(rg.astype(np.float64)).sum(-1).max()


def _problem(M, N1, K1, N2, seed):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, K1, generator=g).half()
    W1 = (torch.randn(N1, K1, generator=g) * K1 ** -0.5).half()
    b1 = torch.randn(N1, generator=g) * 0.1
    resid = _values(M, N1, seed + 1)
    resid[0, 7] = 0.5                               # (keep row 0 inside the fp16 range after the update)
    gamma = 1.0 + 0.2 * torch.randn(N1, generator=g)
    beta = 0.1 * torch.randn(N1, generator=g)
    W2 = (torch.randn(N2, N1, generator=g) * N1 ** -0.5).half()
    b2 = torch.randn(N2, generator=g) * 0.1
    return A, W1, b1, resid, gamma, beta, W2, b2


# (M, N1 = stream width, K1, N2): out-projection / MLP-down shapes of ViT-B and ViT-L towers, every geometry of launch_one
SHAPES = [(300, 768, 768, 2304), (4096, 768, 768, 3072), (12608, 768, 3072, 2304), (50432, 768, 768, 3072), (50432, 768, 3072, 2304),
          (9000, 1024, 4096, 3072)]


@pytest.mark.parametrize("M,N1,K1,N2", SHAPES)
@pytest.mark.parametrize("epi_name", ["store16", "gelu"])
def test_packed_update_and_folded_consumer(M, N1, K1, N2, epi_name):
    E = _E()
    L = E._lib
    A, W1, b1, resid, gamma, beta, W2, b2 = _problem(M, N1, K1, N2, M + N2)
    dev = "cuda"
    hi0, lo0, _ = E.op_respk_pack(resid.to(dev))
    x0 = torch.from_numpy(P.unpack(*P.pack(resid.numpy())))                          # what the stream holds
    xref = A.double() @ W1.double().t() + b1.double() + x0.double()
    hi, lo, part, nt = E.op_gemm_residp(A.to(dev), W1.to(dev), b1.to(dev), hi0, lo0)
    x1 = E.op_respk_unpack(hi, lo)
    assert relerr(x1, xref) < 2e-6 * K1 ** 0.5 + 2.0 ** -18
    # the stored planes are the packing of an fp32 value next to the reference: hi is its fp16 rounding (ties / last-bit
    # differences of the fp32 accumulation may move single elements by one ulp)
    # (the fp32 accumulation over K1 moves elements that sit near a rounding boundary to the neighbouring fp16 value)
    d_hi = (hi.float().cpu() - xref.float().half().float()).abs()
    assert float((d_hi > 0).float().mean()) < 1e-3 * K1 ** 0.5
    big = xref.abs() >= 0.25
    assert float((d_hi[big] / xref[big].abs().float()).max()) < 2.0 ** -9
    assert torch.equal(hi.cpu(), x1.cpu().half()) or float((hi.float().cpu() - x1.cpu()).abs().max()) <= float(x1.abs().max()) * 2.0 ** -10
    # statistics of the fp32 rows
    bn = N1 // nt
    got = part[:, :nt].double().cpu()
    o = xref.view(M, nt, bn)
    assert float((got[..., 0] - o.sum(-1)).abs().max()) <= 1e-4 * float(o.abs().sum(-1).max())
    assert float((got[..., 1] - (o * o).sum(-1)).abs().max()) <= 1e-5 * float((o * o).sum(-1).max())
    # consumer: LN(x) W2^T + b2 from the hi plane and the gamma-folded weight
    W2p = W2.to(dev)
    Wg, cs = E.op_fold_weight(W2p, N1, gamma.to(dev))
    _, bias2 = E.op_fold_vectors(W2p, N1, gamma.to(dev), beta.to(dev), b2.to(dev))
    epi = {"store16": L.EPI_STORE16, "gelu": L.EPI_GELU}[epi_name]
    res = E.op_gemm_folded(hi, Wg, cs, bias2, part, nt, epi=epi, out2=(epi == L.EPI_GELU))
    y = torch.nn.functional.layer_norm(xref, (N1,), gamma.double(), beta.double(), 1e-5) @ W2.double().t() + b2.double()
    tol = 3e-3
    if epi == L.EPI_GELU:
        out, u = res
        assert relerr(u, y) < tol
        assert relerr(out, y * torch.sigmoid(1.702 * y)) < tol
    else:
        assert relerr(res, y) < tol
        # ... as close to the reference as the fp32 stream's folded path (round16(x * gamma), plain weight)
        out32, x16, part32, nt32 = E.op_gemm_ln_producer(A.to(dev), W1.to(dev), b1.to(dev), x0.to(dev), gamma.to(dev))
        cs32, _ = E.op_fold_vectors(W2p, N1, gamma.to(dev), beta.to(dev), b2.to(dev))
        plain = E.op_gemm_folded(x16, W2p, cs32, bias2, part32, nt32)
        assert relerr(res, y) < 1.5 * relerr(plain, y) + 1e-4


def test_packed_update_in_place_and_deterministic():
    E = _E()
    dev = "cuda"
    A, W1, b1, resid, gamma, beta, W2, b2 = _problem(20000, 768, 3072, 2304, 9)
    hi0, lo0, _ = E.op_respk_pack(resid.to(dev))
    ref = None
    for k in range(4):
        if k < 2:
            hi, lo, part, nt = E.op_gemm_residp(A.to(dev), W1.to(dev), b1.to(dev), hi0, lo0)
        else:
            h, l = hi0.clone(), lo0.clone()
            hi, lo, part, nt = E.op_gemm_residp(A.to(dev), W1.to(dev), b1.to(dev), h, l, in_place=True)
        cur = (hi.clone(), lo.clone(), part.clone())
        if ref is not None:
            assert all(torch.equal(a, b) for a, b in zip(ref, cur))
        ref = cur


def test_packed_update_refuses_other_formats():
    E = _E()
    dev = "cuda"
    A, W1, b1, resid, *_ = _problem(300, 768, 768, 768, 2)
    hi0, lo0, _ = E.op_respk_pack(resid.to(dev))
    with pytest.raises(RuntimeError):
        E.op_gemm_residp(A.to(dev), W1.to(dev), b1.to(dev), hi0, lo0, ntp=4)          # 6 slots needed for 768 columns


# ------------------------------------------------------------------------------------------------ tower level
_clips = {}


def _clip(arch_name):
    from mvlpt_amd.model import FrozenCLIP
    from mvlpt_amd.weights import ARCHS, make_state_dict
    if arch_name not in _clips:
        _clips.clear()
        sd = make_state_dict(ARCHS[arch_name], 2, include_token_embedding=True)
        _clips[arch_name] = (FrozenCLIP(sd, compute_dtype="fp16"), sd)
    return _clips[arch_name]


@pytest.mark.parametrize("arch_name,B", [("ViT-B/16", 24), ("ViT-B/32", 96), ("ViT-L/14", 20)])
def test_image_features_packed_vs_fp32_stream_vs_oracle(arch_name, B):
    """B * L >= 4096 token rows: the towers fold their LayerNorms by default; the packed stream is switched on for the first two runs."""
    from mvlpt_amd.weights import ARCHS
    from oracle import clip_oracle as O
    clip, sd = _clip(arch_name)
    arch = ARCHS[arch_name]
    g = torch.Generator().manual_seed(5)
    image = torch.randn(B, 3, arch.image_resolution, arch.image_resolution, generator=g)
    eng = clip.engine
    try:
        eng.set_resid_packed(True)
        f_packed = eng.image_fwd(image.cuda().half()).float().cpu()
        f_again = eng.image_fwd(image.cuda().half()).float().cpu()
        eng.set_resid_packed(False)
        f_plain = eng.image_fwd(image.cuda().half()).float().cpu()
    finally:
        eng.set_resid_packed(True)
    assert torch.equal(f_packed, f_again)
    nref = min(B, 6)
    torch.set_num_threads(8)
    ref, _ = O.image_encoder_fwd({k: v.float() for k, v in sd.items()}, image[:nref], None, None, heads=arch.vision_heads, need_bwd=False)
    e_packed, e_plain = relerr(f_packed[:nref], ref), relerr(f_plain[:nref], ref)
    assert e_packed < 1e-3 and e_plain < 1e-3, (e_packed, e_plain)          # north_star bound on the tower output
    assert e_packed < 6e-4 and e_packed < 2.0 * e_plain + 1e-4, (e_packed, e_plain)
    assert not torch.equal(f_packed, f_plain)                                # (the two paths are different kernels)


@pytest.mark.parametrize("arch_name,name", [("ViT-B/32", "full_vitb32_coop_end"), ("ViT-B/16", "full_vitb16_coop_middle")])
@pytest.mark.parametrize("packed", [True, False])
def test_coop_fixtures_with_the_packed_tower(arch_name, name, packed):
    """The CoOp fixtures (BASELINE configs[0..1]) have B = 4: force the folded towers on (min_rows = 1) so that the packed image
    tower runs, and hold logits / loss / context gradients / inference logits to the bounds of tests/test_hip_model.py."""
    from mvlpt_amd.weights import ARCHS
    from tests.golden_util import full_case_inputs
    from tests.test_hip_model import GRAD_TOL_FP16, TOL_FP16, _check_inference, _inference_logits, build_model, run_case
    clip, sd = _clip(arch_name)
    clip.engine.set_ln_fold(2, 1)
    clip.engine.set_resid_packed(packed)
    try:
        case = load_npz(name)
        res = ARCHS[arch_name].image_resolution
        image, pre, suf = full_case_inputs(case, sd, res)
        model = build_model(case, clip, res, pre, suf)
        run_case(case, model, image, TOL_FP16, GRAD_TOL_FP16)
        _check_inference(case, _inference_logits(case, build_model(case, clip, res, pre, suf), image), name)
    finally:
        clip.engine.set_ln_fold(2, 1024)
        clip.engine.set_resid_packed(True)
