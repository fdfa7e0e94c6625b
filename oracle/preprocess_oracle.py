"""TEST INFRASTRUCTURE — ctypes front end of oracle/resample_oracle.c (the CPU restatement of Pillow's 8-bit bicubic
resample + ToTensor + Normalize; see the C file's header for the reference call sites).  Builds the shared object with
gcc on first use.  Only tests/, __graft_entry__.smoke() and CPU-baseline legs may import this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libresample_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "resample_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", _SO, src, "-lm"])
    return _SO


_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.resample_crop_u8.argtypes = [C.c_void_p] + [C.c_int] * 12 + [C.c_void_p]
        _lib.resample_crop_u8.restype = C.c_int
        _lib.to_tensor_normalize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.to_tensor_normalize.restype = None
    return _lib


def resample_crop_u8(src: np.ndarray, crop, resize, window) -> np.ndarray:
    """src uint8 [H,W,3]; crop (top,left,h,w); resize (rh,rw); window (top,left,h,w) of the resized image -> uint8 [h,w,3]"""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    H, W, _ = src.shape
    ct, cl, ch, cw = [int(v) for v in crop]
    rh, rw = [int(v) for v in resize]
    ot, ol, oh, ow = [int(v) for v in window]
    out = np.empty((oh, ow, 3), np.uint8)
    rc = _load().resample_crop_u8(src.ctypes.data, H, W, ct, cl, ch, cw, rh, rw, ot, ol, oh, ow, out.ctypes.data)
    if rc != 0:
        raise ValueError("resample_crop_u8: box / window outside the image")
    return out


def to_tensor_normalize(img_u8: np.ndarray, flip: bool, mean, std) -> np.ndarray:
    img_u8 = np.ascontiguousarray(img_u8, dtype=np.uint8)
    h, w, _ = img_u8.shape
    m, s = np.asarray(mean, np.float32), np.asarray(std, np.float32)
    out = np.empty((3, h, w), np.float32)
    _load().to_tensor_normalize(img_u8.ctypes.data, h, w, int(bool(flip)), m.ctypes.data, s.ctypes.data, out.ctypes.data)
    return out


def preprocess(src, crop, resize, window, flip, mean, std):
    """-> (uint8 [h,w,3] AFTER the flip, float32 [3,h,w])"""
    u8 = resample_crop_u8(src, crop, resize, window)
    f32 = to_tensor_normalize(u8, flip, mean, std)
    return (u8[:, ::-1].copy() if flip else u8), f32
