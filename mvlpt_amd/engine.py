"""Thin torch-facing wrapper over the C ABI: device pointers + the current HIP stream in, tensors out.

PyTorch is plumbing here (device memory, streams); all compute is in libmvlpt_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from ._lib import DT_BF16, DT_F16, DT_F32, lib
from .weights import ClipArch, arch_from_state_dict

_TORCH2DT = {torch.float32: DT_F32, torch.float16: DT_F16, torch.bfloat16: DT_BF16}
_DT2TORCH = {DT_F16: torch.float16, DT_BF16: torch.bfloat16, DT_F32: torch.float32}


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _on_device(fn):
    """Run an Engine method with the engine's GPU current: `_stream()` then returns THAT device's current stream and the
    library's internal hipMalloc / kernel launches land on it (one process per GPU, but rank r's GPU is cuda:r, not 0)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        if torch.cuda.current_device() == self.device.index:
            return fn(self, *a, **k)
        with torch.cuda.device(self.device):
            return fn(self, *a, **k)
    return wrapped


_PARTITION_STREAMS: Dict[Tuple[int, int, int], "torch.cuda.ExternalStream"] = {}


def partition_stream(device: torch.device, cu_first: int, cu_count: int) -> "torch.cuda.Stream":
    """A torch view of a stream that owns logical compute units [cu_first, cu_first + cu_count) of `device`
    (mvlpt_stream_create_cus, include/mvlpt_hip.h).  One stream per (device, range), kept for the life of the process."""
    key = (device.index, cu_first, cu_count)
    st = _PARTITION_STREAMS.get(key)
    if st is None:
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.mvlpt_stream_create_cus(cu_first, cu_count, C.byref(h)), None, "stream_create_cus")
        st = _PARTITION_STREAMS[key] = torch.cuda.ExternalStream(h.value, device=device)
    return st


def device_cus(device: torch.device) -> int:
    with torch.cuda.device(device):
        return int(lib.mvlpt_stream_cus(None))


def _req(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA/HIP tensor: mvlpt_amd has no CPU path")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


class Engine:
    """One handle per process per GPU (include/mvlpt_hip.h)."""

    def __init__(self, arch: ClipArch, compute_dtype: str = "fp16", device: Optional[torch.device] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("mvlpt_amd.Engine needs a HIP device (no CPU fallback)")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.index is None:
            self.device = torch.device(f"cuda:{torch.cuda.current_device()}")
        self.arch = arch
        self.dt = {"fp16": DT_F16, "bf16": DT_BF16}[compute_dtype]
        self.torch_dtype = _DT2TORCH[self.dt]
        a = _lib.MvlptArch(arch.image_resolution, arch.vision_patch_size, arch.vision_width, arch.vision_layers,
                           arch.vision_heads, arch.context_length, arch.transformer_width, arch.transformer_layers,
                           arch.transformer_heads, arch.embed_dim, self.dt)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.mvlpt_create(C.byref(a), C.byref(h)), None, "mvlpt_create")
        self.h = h
        self.precision = _lib.PREC_SPLIT_GRAD
        self._keep: List[torch.Tensor] = []     # tensors the library reads asynchronously / later
        self._img_state = None
        self._txt_state = None
        self._head_state = None

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        if getattr(self, "h", None):
            lib.mvlpt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_precision(self, mode) -> None:
        """"fast" | "split_grad" (default) | "split_all" — see MVLPT_PREC_* in include/mvlpt_hip.h."""
        code = {"fast": _lib.PREC_FAST, "split_grad": _lib.PREC_SPLIT_GRAD, "split_all": _lib.PREC_SPLIT_ALL}.get(mode, mode)
        _lib.check(lib.mvlpt_set_precision(self.h, int(code)), self.h, "set_precision")
        self.precision = int(code)

    def set_vpt_dropout(self, masks: Optional[torch.Tensor]) -> None:
        """Per-image dropout masks of the visual prompt rows for the next image_fwd / image_bwd pair (include/mvlpt_hip.h:
        mvlpt_set_vpt_dropout): fp32 [n_layers, B, n_vpt, width], or None to clear.  One-shot: the next image_fwd consumes the setting
        (and keeps the tensor alive for its backward); a forward without a new call runs without dropout."""
        if masks is not None:
            masks = _req(masks, torch.float32, "vpt dropout masks")
            if masks.dim() != 4 or masks.shape[-1] != self.arch.vision_width:
                raise ValueError("vpt dropout masks must be [n_layers, B, n_vpt, vision_width]")
        self._vpt_masks = masks
        sh = (0, 0, 0, 0) if masks is None else tuple(int(v) for v in masks.shape)
        _lib.check(lib.mvlpt_set_vpt_dropout(self.h, _ptr(masks), *sh), self.h, "set_vpt_dropout")

    def debug_checksums(self, enable: bool = True):
        """mvlpt_debug_checksums: the stage fingerprints of the last image_fwd (list of ints) and the new on / off state."""
        buf = (C.c_uint64 * 256)()
        n = lib.mvlpt_debug_checksums(self.h, int(bool(enable)), buf, 256)
        if n < 0:
            raise RuntimeError(_lib.last_error(self.h))
        return [int(buf[i]) for i in range(n)]

    def set_ln_fold(self, mode: int, min_rows: int = 1024) -> None:
        """LayerNorm folding (include/mvlpt_hip.h: mvlpt_set_ln_fold): 0 off, 1 image tower, 2 both towers."""
        _lib.check(lib.mvlpt_set_ln_fold(self.h, int(mode), int(min_rows)), self.h, "set_ln_fold")

    def set_resid_packed(self, on: bool) -> None:
        """Packed residual stream of the prompt-free, gradient-free fp16 image tower (include/mvlpt_hip.h: mvlpt_set_resid_packed)."""
        _lib.check(lib.mvlpt_set_resid_packed(self.h, int(bool(on))), self.h, "set_resid_packed")

    @_on_device
    def trim(self) -> None:
        """Release workspace blocks that were outgrown (epoch boundary: synchronises the device)."""
        _lib.check(lib.mvlpt_trim(self.h), self.h, "trim")

    @classmethod
    def from_state_dict(cls, sd: Dict[str, torch.Tensor], compute_dtype: str = "fp16", device=None,
                        arch: Optional[ClipArch] = None) -> "Engine":
        eng = cls(arch or arch_from_state_dict(sd), compute_dtype, device)
        eng.load_frozen(sd)
        return eng

    def load_frozen(self, sd: Dict[str, torch.Tensor]) -> None:
        """Pack the frozen CLIP tensors (keys of clip.model.CLIP.state_dict())."""
        with torch.cuda.device(self.device):
            for name, t in sd.items():
                if name in ("logit_scale", "token_embedding.weight", "input_resolution", "context_length", "vocab_size"):
                    continue
                if t.dtype not in _TORCH2DT:
                    t = t.float()
                tg = t.to(self.device).contiguous()
                shape = (C.c_int64 * max(tg.dim(), 1))(*tg.shape)
                _lib.check(lib.mvlpt_load_frozen(self.h, name.encode(), _ptr(tg), _TORCH2DT[tg.dtype], shape, tg.dim(),
                                                 _stream()), self.h, f"load_frozen({name})")
                torch.cuda.current_stream().synchronize()   # tg may be freed right after
            _lib.check(lib.mvlpt_frozen_ready(self.h), self.h, "frozen_ready")

    # ------------------------------------------------------------------ towers
    @_on_device
    def image_fwd(self, image: torch.Tensor, vpt: Optional[torch.Tensor] = None, vpt_deep: Optional[torch.Tensor] = None,
                  save_for_bwd: bool = False) -> torch.Tensor:
        if image.dtype not in _TORCH2DT:
            image = image.float()
        image = image.contiguous()
        if not image.is_cuda:
            raise RuntimeError("image must be on the GPU")
        B = image.shape[0]
        n_vpt = n_deep = 0
        if vpt is not None:
            vpt = _req(vpt, torch.float32, "vpt").reshape(-1, self.arch.vision_width)
            n_vpt = vpt.shape[0]
        m = getattr(self, "_vpt_masks", None)
        if m is not None and (m.shape[1] != B or m.shape[2] != n_vpt):
            raise ValueError(f"vpt dropout masks are {tuple(m.shape)} but the batch has {B} images and {n_vpt} prompt tokens")
        if vpt_deep is not None:
            vpt_deep = _req(vpt_deep, torch.float32, "vpt_deep")
            n_deep = vpt_deep.shape[0]
        feat = torch.empty(B, self.arch.embed_dim, device=image.device, dtype=torch.float32)
        _lib.check(lib.mvlpt_image_fwd(self.h, _ptr(image), _TORCH2DT[image.dtype], _ptr(vpt), _ptr(vpt_deep), n_vpt, n_deep, B,
                                       _ptr(feat), int(save_for_bwd), _stream()), self.h, "image_fwd")
        self._img_state = (n_vpt, n_deep, B) if save_for_bwd else None
        # the setting was one-shot: this forward consumed it; its backward reads the buffer, so the tensor stays alive until then
        self._vpt_masks_saved, self._vpt_masks = (m if save_for_bwd else None), None
        return feat

    @_on_device
    def image_bwd(self, dfeat: torch.Tensor) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
        if self._img_state is None:
            raise RuntimeError("image_bwd without image_fwd(save_for_bwd=True)")
        n_vpt, n_deep, B = self._img_state
        dfeat = _req(dfeat, torch.float32, "dfeat")
        dv = self.arch.vision_width
        dvpt = torch.empty(n_vpt, dv, device=dfeat.device, dtype=torch.float32) if n_vpt else None
        ddeep = torch.empty(n_deep, n_vpt, dv, device=dfeat.device, dtype=torch.float32) if n_deep else None
        _lib.check(lib.mvlpt_image_bwd(self.h, _ptr(dfeat), _ptr(dvpt), _ptr(ddeep), _stream()), self.h, "image_bwd")
        return dvpt, ddeep

    @_on_device
    def text_fwd(self, prefix: torch.Tensor, suffix: torch.Tensor, ctx: Optional[torch.Tensor], layout: torch.Tensor,
                 eot: torch.Tensor, save_for_bwd: bool = False) -> torch.Tensor:
        prefix = _req(prefix, torch.float32, "token_prefix")
        suffix = _req(suffix, torch.float32, "token_suffix")
        layout = _req(layout, torch.int32, "layout")
        eot = _req(eot, torch.int32, "eot")
        Cn, L = layout.shape
        per_class, n_ctx = 0, 0
        if ctx is not None:
            ctx = _req(ctx, torch.float32, "ctx")
            per_class = int(ctx.dim() == 3)
            n_ctx = ctx.shape[-2]
        feat = torch.empty(Cn, self.arch.embed_dim, device=prefix.device, dtype=torch.float32)
        _lib.check(lib.mvlpt_text_fwd(self.h, _ptr(prefix), _ptr(suffix), _ptr(ctx), per_class, n_ctx, _ptr(layout), _ptr(eot),
                                      Cn, L, _ptr(feat), int(save_for_bwd), _stream()), self.h, "text_fwd")
        self._txt_state = (tuple(ctx.shape) if ctx is not None else None, layout, eot) if save_for_bwd else None
        return feat

    @_on_device
    def text_bwd(self, dfeat: torch.Tensor) -> torch.Tensor:
        if self._txt_state is None or self._txt_state[0] is None:
            raise RuntimeError("text_bwd without text_fwd(save_for_bwd=True) with context tokens")
        dfeat = _req(dfeat, torch.float32, "dfeat")
        dctx = torch.empty(self._txt_state[0], device=dfeat.device, dtype=torch.float32)
        _lib.check(lib.mvlpt_text_bwd(self.h, _ptr(dfeat), _ptr(dctx), _stream()), self.h, "text_bwd")
        return dctx

    # ------------------------------------------------------------------ head
    @_on_device
    def logits_fwd(self, img_feat, txt_feat, logit_scale_exp: float, task_lo=None, task_hi=None) -> torch.Tensor:
        img_feat = _req(img_feat, torch.float32, "img_feat")
        txt_feat = _req(txt_feat, torch.float32, "txt_feat")
        if task_lo is not None:
            task_lo = _req(task_lo, torch.int32, "task_lo")
            task_hi = _req(task_hi, torch.int32, "task_hi")
        B, Cn = img_feat.shape[0], txt_feat.shape[0]
        logits = torch.empty(B, Cn, device=img_feat.device, dtype=torch.float32)
        _lib.check(lib.mvlpt_logits_fwd(self.h, _ptr(img_feat), _ptr(txt_feat), float(logit_scale_exp), _ptr(task_lo),
                                        _ptr(task_hi), B, Cn, _ptr(logits), _stream()), self.h, "logits_fwd")
        self._head_state = (task_lo, task_hi, B, Cn)   # keep the mask tensors alive until logits_bwd
        return logits

    @_on_device
    def logits_bwd(self, dlogits, need_img: bool = True, need_txt: bool = True):
        if self._head_state is None:
            raise RuntimeError("logits_bwd without logits_fwd")
        _, _, B, Cn = self._head_state
        dlogits = _req(dlogits, torch.float32, "dlogits")
        e = self.arch.embed_dim
        dimg = torch.empty(B, e, device=dlogits.device, dtype=torch.float32) if need_img else None
        dtxt = torch.empty(Cn, e, device=dlogits.device, dtype=torch.float32) if need_txt else None
        _lib.check(lib.mvlpt_logits_bwd(self.h, _ptr(dlogits), _ptr(dimg), _ptr(dtxt), _stream()), self.h, "logits_bwd")
        return dimg, dtxt

    @_on_device
    def cross_entropy(self, logits, label, need_grad: bool = True):
        """Returns (loss[1], dlogits or None, ncorrect[1]) — device tensors, no host sync."""
        logits = _req(logits, torch.float32, "logits")
        B, Cn = logits.shape
        if label.dtype in (torch.int64, torch.int32, torch.int16, torch.uint8):
            label = _req(label, torch.int64, "label")
            kind = _lib.LABEL_INT64
            if label.shape != (B,):
                raise ValueError("integer labels must have shape [B]")
        else:
            label = _req(label, torch.float32, "label")
            kind = _lib.LABEL_PROB_F32
            if label.shape != (B, Cn):
                raise ValueError("probability labels must have shape [B, C]")
        loss = torch.empty(1, device=logits.device, dtype=torch.float32)
        nc = torch.empty(1, device=logits.device, dtype=torch.float32)
        dl = torch.empty_like(logits) if need_grad else None
        _lib.check(lib.mvlpt_cross_entropy(self.h, _ptr(logits), _ptr(label), kind, B, Cn, _ptr(loss), _ptr(dl), _ptr(nc),
                                           _stream()), self.h, "cross_entropy")
        return loss, dl, nc

    # ------------------------------------------------------------------ input pipeline
    @_on_device
    def preprocess(self, src: torch.Tensor, descs, out_size, mean, std, out_dtype=torch.float32, want_u8: bool = False):
        """src: uint8 device tensor holding the packed decoded HWC images; descs: `_lib.MvlptImageDesc` ctypes array.
        Returns [B,3,h,w] normalised images (and the resized 8-bit images [B,h,w,3] when `want_u8`)."""
        if not src.is_cuda or src.dtype != torch.uint8:
            raise RuntimeError("src must be a uint8 CUDA/HIP tensor: mvlpt_amd has no CPU path")
        src = src.contiguous()
        B = len(descs)
        oh, ow = (out_size, out_size) if isinstance(out_size, int) else tuple(out_size)
        out = torch.empty(B, 3, oh, ow, device=src.device, dtype=out_dtype)
        u8 = torch.empty(B, oh, ow, 3, device=src.device, dtype=torch.uint8) if want_u8 else None
        m = (C.c_float * 3)(*[float(v) for v in mean])
        sd = (C.c_float * 3)(*[float(v) for v in std])
        _lib.check(lib.mvlpt_preprocess(self.h, _ptr(src), src.numel(), descs, B, oh, ow, m, sd, _ptr(out), _TORCH2DT[out_dtype],
                                        _ptr(u8), _stream()), self.h, "preprocess")
        return (out, u8) if want_u8 else out

    # ------------------------------------------------------------------ profiling
    @_on_device
    def profile_begin(self, all_kernels: bool = False):
        _lib.check(lib.mvlpt_profile_begin(self.h, int(all_kernels)), self.h, "profile_begin")

    @_on_device
    def profile_pause(self, paused: bool):
        _lib.check(lib.mvlpt_profile_pause(self.h, int(paused)), self.h, "profile_pause")

    @_on_device
    def profile_end(self) -> Dict[str, dict]:
        arr = (_lib.MvlptKernelStat * 64)()
        n = lib.mvlpt_profile_end(self.h, arr, 64)
        if n < 0:
            raise RuntimeError("profile_end failed")
        return {arr[i].name.decode(): dict(launches=arr[i].launches, ms=arr[i].ms, flops=arr[i].flops, bytes=arr[i].bytes,
                                           busy_ms=arr[i].busy_ms, flops_executed=arr[i].flops_executed)
                for i in range(n)}


# ---------------------------------------------------------------------- kernel-level ops (parity tests)
def op_gemm(A, Bt, epi=_lib.EPI_STORE16, bias=None, aux=None, resid=None, out2=False):
    dt = _TORCH2DT[A.dtype]
    M, K = A.shape
    N = Bt.shape[0]
    out_dtype = torch.float32 if epi in (_lib.EPI_RESID32, _lib.EPI_STORE32) else A.dtype
    out = torch.empty(M, N, device=A.device, dtype=out_dtype)
    o2 = torch.empty(M, N, device=A.device, dtype=A.dtype) if out2 else None
    _lib.check(lib.mvlpt_op_gemm(dt, epi, _ptr(A.contiguous()), _ptr(Bt.contiguous()), M, N, K, _ptr(bias), _ptr(aux),
                                 _ptr(resid), _ptr(out), _ptr(o2), _stream()), None, "op_gemm")
    return (out, o2) if out2 else out


def split_pair(x: torch.Tensor, dtype) -> torch.Tensor:
    """fp32 [M,K] -> 16-bit hi|lo pair [M,2K] (the layout of GemmArgs::a_split; host-side helper for tests)."""
    hi = x.to(dtype)
    lo = (x - hi.float()).to(dtype)
    return torch.cat([hi, lo], dim=1).contiguous()


def join_pair(p: torch.Tensor) -> torch.Tensor:
    k = p.shape[1] // 2
    return p[:, :k].float() + p[:, k:].float()


def op_gemm_split(A2, Bt, epi=_lib.EPI_STORE32, bias=None, aux=None, resid=None, out2=False):
    """A2: 16-bit pair [M,2K]."""
    dt = _TORCH2DT[A2.dtype]
    M, K = A2.shape[0], A2.shape[1] // 2
    N = Bt.shape[0]
    if epi in (_lib.EPI_RESID32, _lib.EPI_STORE32):
        out = torch.empty(M, N, device=A2.device, dtype=torch.float32)
    elif epi in (_lib.EPI_GELU_SPLIT, _lib.EPI_GELUBWD_SPLIT, _lib.EPI_STORE_SPLIT):
        out = torch.empty(M, 2 * N, device=A2.device, dtype=A2.dtype)
    else:
        out = torch.empty(M, N, device=A2.device, dtype=A2.dtype)
    o2 = torch.empty(M, N, device=A2.device, dtype=A2.dtype) if out2 else None
    _lib.check(lib.mvlpt_op_gemm_split(dt, epi, _ptr(A2.contiguous()), _ptr(Bt.contiguous()), M, N, K, _ptr(bias), _ptr(aux),
                                       _ptr(resid), _ptr(out), _ptr(o2), _stream()), None, "op_gemm_split")
    return (out, o2) if out2 else out


# ---- mixed pair: [hi (d x 16 bit) | residual bytes (d, e5m2 of (x - hi) * 2^LO8_EXP) | unused], pitch 2d 16-bit elements
LO8_EXP = {torch.float16: 10, torch.bfloat16: 7}


def join_mixed(p: torch.Tensor) -> torch.Tensor:
    """Value of a mixed-pair tensor [rows, 2d] (host-side decoder for tests)."""
    d = p.shape[1] // 2
    lo8 = p[:, d:d + d // 2].contiguous().view(torch.uint8).view(torch.float8_e5m2).float()
    return p[:, :d].float() + lo8 * 2.0 ** -LO8_EXP[p.dtype]


def op_cast_mixed(x32: torch.Tensor, dtype) -> torch.Tensor:
    rows, d = x32.shape
    out = torch.zeros(rows, 2 * d, device=x32.device, dtype=dtype)
    _lib.check(lib.mvlpt_op_cast_mixed(_TORCH2DT[dtype], _ptr(x32.contiguous()), _ptr(out), rows, d, _stream()), None, "op_cast_mixed")
    return out


def op_pack_weight_mixed(w32: torch.Tensor, dtype, transposed=False):
    """nn.Linear weight [out, in] fp32 -> (packed [R, 3K/2] 16-bit elements = [W16 | e4m3(W * 2^e8) bytes], e8)."""
    rows, cols = w32.shape
    R, K = (cols, rows) if transposed else (rows, cols)
    out = torch.zeros(R, K + K // 2, device=w32.device, dtype=dtype)
    e8 = C.c_int(0)
    _lib.check(lib.mvlpt_op_pack_weight_mixed(_TORCH2DT[dtype], _ptr(w32.contiguous()), rows, cols, int(transposed), _ptr(out),
                                              C.byref(e8), _stream()), None, "op_pack_weight_mixed")
    return out, e8.value


def op_gemm_mixed(A2, Wp, e8, epi=_lib.EPI_STORE32, bias=None, aux=None, resid=None, out2=False):
    """A2: mixed pair [M, 2K]; Wp: packed weight [N, 3K/2] from op_pack_weight_mixed."""
    dt = _TORCH2DT[A2.dtype]
    M, K = A2.shape[0], A2.shape[1] // 2
    N = Wp.shape[0]
    if epi in (_lib.EPI_RESID32, _lib.EPI_STORE32):
        out = torch.empty(M, N, device=A2.device, dtype=torch.float32)
    else:
        out = torch.zeros(M, 2 * N, device=A2.device, dtype=A2.dtype)
    o2 = torch.empty(M, N, device=A2.device, dtype=A2.dtype) if out2 else None
    _lib.check(lib.mvlpt_op_gemm_mixed(dt, epi, _ptr(A2.contiguous()), _ptr(Wp), Wp.shape[1], e8, M, N, K, _ptr(bias), _ptr(aux),
                                       _ptr(resid), _ptr(out), _ptr(o2), _stream()), None, "op_gemm_mixed")
    return (out, o2) if out2 else out


def op_layernorm_fwd_mixed(x, gamma, beta, out_dtype):
    rows, d = x.shape
    y = torch.zeros(rows, 2 * d, device=x.device, dtype=out_dtype)
    _lib.check(lib.mvlpt_op_layernorm_fwd_mixed(_TORCH2DT[out_dtype], _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), rows, d, _stream()),
               None, "op_layernorm_fwd_mixed")
    return y


def op_layernorm_bwd_mixed(dy32, x, gamma, dtype, resid=None):
    rows, d = x.shape
    out32 = torch.empty(rows, d, device=x.device, dtype=torch.float32)
    out16 = torch.zeros(rows, 2 * d, device=x.device, dtype=dtype)
    _lib.check(lib.mvlpt_op_layernorm_bwd_mixed(_TORCH2DT[dtype], _ptr(dy32), _ptr(x), _ptr(gamma), _ptr(resid), _ptr(out32),
                                                _ptr(out16), rows, d, _stream()), None, "op_layernorm_bwd_mixed")
    return out32, out16


def op_attention32_fwd_mixed(qkv_pair, N, L, H, causal, q_rows=0):
    """qkv: 16-bit pair [N*L, 6d] -> (O as a mixed pair [N*L, 2d], lse)."""
    dtype = qkv_pair.dtype
    out = torch.zeros(N * L, 2 * H * 64, device=qkv_pair.device, dtype=dtype)
    lse = torch.zeros(N * H * L, device=qkv_pair.device, dtype=torch.float32)
    _lib.check(lib.mvlpt_op_attention32_fwd_mixed(_TORCH2DT[dtype], _ptr(qkv_pair), _ptr(out), _ptr(lse), N, L, H, int(causal), q_rows,
                                                  _stream()), None, "op_attention32_fwd_mixed")
    return out, lse


def op_attention32_bwd_mixed(qkv_pair, out_mixed, dout_pair, lse, N, L, H, causal):
    """-> dqkv as a mixed pair [N*L, 6d] (hi plane 3d wide)."""
    dtype = out_mixed.dtype
    dqkv = torch.zeros(N * L, 6 * H * 64, device=qkv_pair.device, dtype=dtype)
    delta = torch.empty(N * H * L, device=qkv_pair.device, dtype=torch.float32)
    _lib.check(lib.mvlpt_op_attention32_bwd_mixed(_TORCH2DT[dtype], _ptr(qkv_pair), _ptr(out_mixed), _ptr(dout_pair), _ptr(lse),
                                                  _ptr(delta), _ptr(dqkv), N, L, H, int(causal), _stream()), None, "op_attention32_bwd_mixed")
    return dqkv


def op_layernorm_fwd_split(x, gamma, beta, out_dtype):
    rows, d = x.shape
    y = torch.empty(rows, 2 * d, device=x.device, dtype=out_dtype)
    _lib.check(lib.mvlpt_op_layernorm_fwd_split(_TORCH2DT[out_dtype], _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), rows, d, _stream()),
               None, "op_layernorm_fwd_split")
    return y


def op_layernorm_bwd_split(dy32, x, gamma, dtype, resid=None):
    rows, d = x.shape
    out32 = torch.empty(rows, d, device=x.device, dtype=torch.float32)
    out16 = torch.empty(rows, 2 * d, device=x.device, dtype=dtype)
    _lib.check(lib.mvlpt_op_layernorm_bwd_split(_TORCH2DT[dtype], _ptr(dy32), _ptr(x), _ptr(gamma), _ptr(resid), _ptr(out32),
                                                _ptr(out16), rows, d, _stream()), None, "op_layernorm_bwd_split")
    return out32, out16


def op_attention32_fwd_pair(qkv_pair, N, L, H, causal, q_rows=0):
    """Split-precision attention core: qkv hi|lo pair [N*L, 6d] -> (O pair [N*L, 2d], lse)."""
    dtype = qkv_pair.dtype
    out = torch.zeros(N * L, 2 * H * 64, device=qkv_pair.device, dtype=dtype)
    lse = torch.zeros(N * H * L, device=qkv_pair.device, dtype=torch.float32)
    _lib.check(lib.mvlpt_op_attention32_fwd(_TORCH2DT[dtype], _ptr(qkv_pair), _ptr(out), _ptr(lse), N, L, H, int(causal), q_rows,
                                            _stream()), None, "op_attention32_fwd")
    return out, lse


def op_attention32_bwd_pair(qkv_pair, out_pair, dout_pair, lse, N, L, H, causal):
    dtype = out_pair.dtype
    dqkv = torch.empty(N * L, 6 * H * 64, device=qkv_pair.device, dtype=dtype)
    delta = torch.empty(N * H * L, device=qkv_pair.device, dtype=torch.float32)
    _lib.check(lib.mvlpt_op_attention32_bwd(_TORCH2DT[dtype], _ptr(qkv_pair), _ptr(out_pair), _ptr(dout_pair), _ptr(lse),
                                            _ptr(delta), _ptr(dqkv), N, L, H, int(causal), _stream()), None, "op_attention32_bwd")
    return dqkv


def op_attention32_fwd(qkv32, N, L, H, causal, dtype=torch.float16, q_rows=0):
    """Same on fp32 inputs, split to pairs here (as the QKV GEMM's epilogue does in the engine)."""
    return op_attention32_fwd_pair(split_pair(qkv32, dtype), N, L, H, causal, q_rows)


def op_attention32_bwd(qkv32, out_pair, dout32, lse, N, L, H, causal):
    dtype = out_pair.dtype
    return op_attention32_bwd_pair(split_pair(qkv32, dtype), out_pair, split_pair(dout32, dtype), lse, N, L, H, causal)


def op_layernorm_fwd(x, gamma, beta, out_dtype):
    rows, d = x.shape
    y = torch.empty(rows, d, device=x.device, dtype=out_dtype)
    _lib.check(lib.mvlpt_op_layernorm_fwd(_TORCH2DT[out_dtype], _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), rows, d, _stream()),
               None, "op_layernorm_fwd")
    return y


def op_layernorm_bwd(dy, x, gamma, resid=None, want16=True):
    rows, d = x.shape
    out32 = torch.empty(rows, d, device=x.device, dtype=torch.float32)
    out16 = torch.empty(rows, d, device=x.device, dtype=dy.dtype) if want16 else None
    _lib.check(lib.mvlpt_op_layernorm_bwd(_TORCH2DT[dy.dtype], _ptr(dy), _ptr(x), _ptr(gamma), _ptr(resid), _ptr(out32),
                                          _ptr(out16), rows, d, _stream()), None, "op_layernorm_bwd")
    return out32, out16


def op_attention_fwd(qkv, N, L, H, causal, want_lse=True):
    out = torch.empty(N * L, H * 64, device=qkv.device, dtype=qkv.dtype)
    lse = torch.empty(N * H * L, device=qkv.device, dtype=torch.float32) if want_lse else None
    _lib.check(lib.mvlpt_op_attention_fwd(_TORCH2DT[qkv.dtype], _ptr(qkv), _ptr(out), _ptr(lse), N, L, H, int(causal), _stream()),
               None, "op_attention_fwd")
    return out, lse


def op_attention_bwd(qkv, out, dout, lse, N, L, H, causal):
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(N * H * L, device=qkv.device, dtype=torch.float32)
    _lib.check(lib.mvlpt_op_attention_bwd(_TORCH2DT[qkv.dtype], _ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _ptr(delta),
                                          _ptr(dqkv), N, L, H, int(causal), _stream()), None, "op_attention_bwd")
    return dqkv


# ---- LayerNorm folding at kernel level (include/mvlpt_hip.h: mvlpt_op_fold_vectors / gemm_ln_producer / gemm_folded)
def op_fold_vectors(W16: torch.Tensor, K: int, gamma, beta, b):
    """W16: packed 16-bit weight [N, ld >= K] -> (colsum = W gamma, bias2 = b + W beta), fp32 [N]."""
    N, ld = W16.shape
    cs = torch.empty(N, device=W16.device, dtype=torch.float32)
    b2 = torch.empty(N, device=W16.device, dtype=torch.float32)
    _lib.check(lib.mvlpt_op_fold_vectors(_TORCH2DT[W16.dtype], _ptr(W16), ld, _ptr(gamma), _ptr(beta), _ptr(b), _ptr(cs), _ptr(b2), N, K,
                                         _stream()), None, "op_fold_vectors")
    return cs, b2


def op_gemm_ln_producer(A, Bt, bias, resid, gamma, a_split=0, x16_split=0, ldb=0, w8_exp=0, ntp=8):
    """-> (out32 [M,N], x16 (format x16_split), part [M, ntp, 2], nt)."""
    dtype = A.dtype
    M = A.shape[0]
    K = A.shape[1] // (2 if a_split else 1)
    N = Bt.shape[0]
    out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    x16 = torch.zeros(M, N * (2 if x16_split else 1), device=A.device, dtype=dtype)
    part = torch.zeros(M, ntp, 2, device=A.device, dtype=torch.float32)
    nt = C.c_int(0)
    _lib.check(lib.mvlpt_op_gemm_ln_producer(_TORCH2DT[dtype], _ptr(A.contiguous()), a_split, _ptr(Bt), ldb, w8_exp, M, N, K, _ptr(bias),
                                             _ptr(resid), _ptr(gamma), x16_split, _ptr(out), _ptr(x16), _ptr(part), ntp, C.byref(nt),
                                             _stream()), None, "op_gemm_ln_producer")
    return out, x16, part, nt.value


# ---- packed residual stream at kernel level (include/mvlpt_hip.h: mvlpt_op_fold_weight / respk_pack / respk_unpack / gemm_residp)
def op_fold_weight(W16: torch.Tensor, K: int, gamma):
    """W16 fp16 [N, ld >= K] -> (Wg = round16(W16 * gamma) [N, K], colsum = row sums of Wg, fp32 [N])."""
    N, ld = W16.shape
    Wg = torch.empty(N, K, device=W16.device, dtype=torch.float16)
    cs = torch.empty(N, device=W16.device, dtype=torch.float32)
    _lib.check(lib.mvlpt_op_fold_weight(_ptr(W16), ld, _ptr(gamma), _ptr(Wg), K, _ptr(cs), N, K, _stream()), None, "op_fold_weight")
    return Wg, cs


def op_respk_pack(x: torch.Tensor, ntp: int = 0):
    """fp32 [rows, d] -> (hi fp16 [rows, d], lo int8 [rows, d], part [rows, ntp, 2] or None)."""
    rows, d = x.shape
    hi = torch.empty(rows, d, device=x.device, dtype=torch.float16)
    lo = torch.empty(rows, d, device=x.device, dtype=torch.int8)
    part = torch.full((rows, ntp, 2), float("nan"), device=x.device, dtype=torch.float32) if ntp else None
    _lib.check(lib.mvlpt_op_respk_pack(_ptr(x.contiguous()), _ptr(hi), _ptr(lo), _ptr(part), ntp, rows, d, _stream()), None, "op_respk_pack")
    return hi, lo, part


def op_assemble_packed(pe: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, g: torch.Tensor, b: torch.Tensor, batch: int, ntp: int = 6):
    """patch embeddings fp32 [batch * G2, d] -> the packed tower entry: (hi fp16 [batch * (1 + G2), d], lo int8, part [rows, ntp, 2])."""
    d = pe.shape[1]
    G2 = pe.shape[0] // batch
    rows = batch * (1 + G2)
    hi = torch.empty(rows, d, device=pe.device, dtype=torch.float16)
    lo = torch.empty(rows, d, device=pe.device, dtype=torch.int8)
    part = torch.empty(rows, ntp, 2, device=pe.device, dtype=torch.float32)
    _lib.check(lib.mvlpt_op_assemble_packed(_ptr(pe), _ptr(cls), _ptr(pos), _ptr(g), _ptr(b), _ptr(hi), _ptr(lo), _ptr(part), ntp, batch, G2, d,
                                            _stream()), None, "op_assemble_packed")
    return hi, lo, part


def op_respk_unpack(hi: torch.Tensor, lo: torch.Tensor, row_mul: int = 1):
    rows, d = hi.shape[0] // row_mul, hi.shape[1]
    out = torch.empty(rows, d, device=hi.device, dtype=torch.float32)
    _lib.check(lib.mvlpt_op_respk_unpack(_ptr(hi), _ptr(lo), row_mul, _ptr(out), rows, d, _stream()), None, "op_respk_unpack")
    return out


def op_gemm_residp(A, Bt, bias, hi, lo, ldb=0, ntp=8, in_place=False):
    """(hi', lo') = pack(A Bt^T + bias + unpack(hi, lo)); -> (hi', lo', part [M, ntp, 2], nt)."""
    M, K = A.shape
    N = Bt.shape[0]
    ho = hi if in_place else torch.empty_like(hi)
    lo_o = lo if in_place else torch.empty_like(lo)
    part = torch.zeros(M, ntp, 2, device=A.device, dtype=torch.float32)
    nt = C.c_int(0)
    _lib.check(lib.mvlpt_op_gemm_residp(_ptr(A.contiguous()), _ptr(Bt), ldb, M, N, K, _ptr(bias), _ptr(hi), _ptr(lo), _ptr(ho), _ptr(lo_o),
                                        _ptr(part), ntp, C.byref(nt), _stream()), None, "op_gemm_residp")
    return ho, lo_o, part, nt.value


def op_gemm_folded(x16, Bt, colsum, bias2, part, nt, epi=_lib.EPI_STORE16, a_split=0, ldb=0, w8_exp=0, out2=False):
    dtype = x16.dtype
    M = x16.shape[0]
    K = x16.shape[1] // (2 if a_split else 1)
    N = Bt.shape[0]
    wide = epi in (_lib.EPI_GELU_SPLIT, _lib.EPI_STORE_SPLIT)
    out = torch.zeros(M, N * (2 if wide else 1), device=x16.device, dtype=dtype)
    o2 = torch.empty(M, N, device=x16.device, dtype=dtype) if out2 else None
    _lib.check(lib.mvlpt_op_gemm_folded(_TORCH2DT[dtype], epi, _ptr(x16), a_split, _ptr(Bt), ldb, w8_exp, M, N, K, _ptr(colsum), _ptr(bias2),
                                        _ptr(part), part.shape[1], nt, _ptr(out), _ptr(o2), _stream()), None, "op_gemm_folded")
    return (out, o2) if out2 else out
