"""TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package).  numpy restatement of the PACKED residual-stream format of the HIP engine (mvlpt_amd/csrc/common.h respk_*,
include/mvlpt_hip.h mvlpt_op_respk_* / mvlpt_op_gemm_residp) and of the arithmetic a packed residual update performs.

The reference keeps the residual stream `x = x + attention(ln_1(x))`, `x = x + mlp(ln_2(x))` (clip/model.py:185-188) in the model's
dtype; the engine's gradient-free fp16 image tower carries it as

    x  = clamp(x, -65504, 65504)                     a fp16 tower cannot carry more: hi stays finite (NaN stays NaN)
    hi = round16(x)                                  fp16 — at the same time the 16-bit operand of the GEMM behind the LayerNorm
    lo = clamp((bits(x) - bits(float(hi))) >> 5, -128, 127)  int8 — the next 8 bits of x (arithmetic shift; bits() = the fp32 pattern)
    x' = bits(float(hi)) + (lo << 5)                 the value a later kernel reads back

i.e. x to 2^-9 of an fp16 ulp.  This is a storage format of the engine, not an algorithm of the reference: the oracle for its
VALUES stays oracle/clip_oracle.py; this file pins the encoding bit for bit."""
import numpy as np


def pack(x: np.ndarray):
    """fp32 array -> (hi float16, lo int8)."""
    x = np.clip(np.ascontiguousarray(x, dtype=np.float32), np.float32(-65504.0), np.float32(65504.0))      # (np.clip keeps NaN)
    hi = x.astype(np.float16)
    d = x.view(np.int32).astype(np.int64) - hi.astype(np.float32).view(np.int32).astype(np.int64)
    return hi, np.clip(d >> 5, -128, 127).astype(np.int8)      # (only fp16 subnormals and round-to-even ties ever clamp)


def unpack(hi: np.ndarray, lo: np.ndarray) -> np.ndarray:
    b = hi.astype(np.float32).view(np.int32).astype(np.int64) + (lo.astype(np.int64) << 5)
    return b.astype(np.int32).view(np.float32)


def row_stats(x: np.ndarray):
    """{sum, sum of squares} of every row in fp64 (what the partial-sum slots of a row add up to)."""
    x = x.astype(np.float64)
    return x.sum(-1), (x * x).sum(-1)


def fold_weight(W16: np.ndarray, gamma: np.ndarray):
    """Wg = round16(W16 * gamma) and its row sums (the LayerNorm's gamma moved into the consumer's weight)."""
    Wg = (W16.astype(np.float32) * gamma.astype(np.float32)).astype(np.float16)
    return Wg, Wg.astype(np.float64).sum(-1)
