"""Does the row pitch of the A / Bt operand matter for the 256x256 GEMM (memory-channel aliasing)?  Debug build only
(-DMVLPT_GEMM_TRACE reads MVLPT_DBG_LDA / MVLPT_DBG_LDB): MVLPT_HIP_LIB=$PWD/mvlpt_amd/libvar_trace.so python tools/pitch_probe.py"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    M, N, K, epi, lda, ldb = [int(v) for v in sys.argv[1:7]]
    os.environ["MVLPT_DBG_LDA"], os.environ["MVLPT_DBG_LDB"] = str(lda), str(ldb)
    import torch
    from mvlpt_amd import engine as E
    from mvlpt_amd._lib import lib
    L = E._lib
    A = torch.randn(M, lda, device="cuda").half()
    Bt = (torch.randn(N, ldb, device="cuda") * K ** -0.5).half()
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    def run():
        L.check(lib.mvlpt_op_gemm(1, epi, E._ptr(A), E._ptr(Bt), M, N, K, E._ptr(bias), None, None, E._ptr(out), None, E._stream()), None, "op")
    run()
    ref = A[:, :K].float() @ Bt[:, :K].float().t() + bias
    err = float((out.float() - ref).abs().max()) / float(ref.abs().max())
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): run()
    e.record(); torch.cuda.synchronize()
    print(f"M={M} N={N} K={K} lda={lda} ({lda*2} B) ldb={ldb} ({ldb*2} B): {s.elapsed_time(e)/20*1e3:7.1f} us  err {err:.1e}")
    sys.exit(0)
cases = []
for lda in (768, 800, 832, 896, 1024): cases.append((50432, 2304, 768, 0, lda, 768))
for ldb in (800, 832, 1152): cases.append((50432, 2304, 768, 0, 768, ldb))
cases.append((50432, 2304, 768, 0, 832, 832))
for lda in (3072, 3136, 3200, 3328, 4096): cases.append((50432, 768, 3072, 0, lda, 3072))
for ldb in (3136, 4608): cases.append((50432, 768, 3072, 0, 3072, ldb))
cases.append((50432, 768, 3072, 0, 3136, 3136))
for c in cases:
    subprocess.run([sys.executable, __file__] + [str(v) for v in c])
