"""The product library must not contain the instruction sequence of the gfx950 hazard found in round 6 (NOTES_experiments.md round 6,
tools/pkfma_hazard.hip): a packed-fp32 VALU op with an op_sel source selection issued straight behind a partial `s_waitcnt lgkmcnt(n >= 1)`.
hipcc emits it on its own whenever it packs a broadcast of a value that has just been read from LDS — the round-5 fold arithmetic
(656 sites in 40 kernels of that binary, `profiles/r06_isa_audit.txt`) — so the audit runs on every build: CPU-only, deterministic."""
import os

import pytest

from tools import isa_audit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mvlpt_amd", "libmvlpt_hip.so")


def test_audit_recognises_the_failing_sequence():
    text = """
0000000000001000 <_Zkernel>:
	ds_read_b64 v[64:65], v60 offset:2048
	ds_read_b128 v[60:63], v1 offset:1088
	s_waitcnt lgkmcnt(1)
	v_pk_fma_f32 v[66:67], v[38:39], v[64:65], v[34:35] op_sel:[0,1,0]
	s_waitcnt lgkmcnt(0)
	v_pk_fma_f32 v[60:61], v[64:65], v[60:61], v[66:67] op_sel_hi:[0,1,1]
	s_waitcnt lgkmcnt(2)
	s_nop 0
	v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1]
	s_waitcnt lgkmcnt(1)
	v_pk_fma_f32 v[8:9], v[10:11], v[12:13], v[14:15]
"""
    found = isa_audit.audit_text(text, distance=1)
    assert [(k, d) for k, _, _, d in found] == [("_Zkernel", 0)] and "op_sel:[0,1,0]" in found[0][2]
    assert len(isa_audit.audit_text(text, distance=2)) == 2                      # ... one instruction further: the v_pk_mul as well
    assert len(isa_audit.audit_text(text, distance=1, all_packed=True)) == 2     # ... or packed ops without op_sel


@pytest.mark.skipif(not os.path.isfile(LIB), reason="libmvlpt_hip.so not built")
def test_product_library_is_free_of_the_sequence():
    assert len(isa_audit.code_objects(LIB)) >= 8                                # every translation unit's gfx950 code object is there
    found = isa_audit.audit_library(LIB, distance=2)                             # one instruction of margin over what the hardware needs
    assert not found, "\n".join(f"{k}: {w} -> [{d}] {p}" for k, w, p, d in found[:20])
    # the second form of the hazard (VALU producer one slot ahead of the op_sel consumer: the packed tower entry of round 5) has no
    # window a disassembly scan could bound, so the build emits NO packed fp32 instruction at all (Makefile NOPK)
    per = isa_audit.count_packed_fp32(LIB)
    assert not per, f"packed fp32 VALU instructions in the product library: {sorted(per.items(), key=lambda kv: -kv[1])[:5]}"
