"""ISA audit of libmvlpt_hip.so for the gfx950 hazard found in round 6 (NOTES_experiments.md, round 6; tools/pkfma_hazard.hip).

The hazard: a packed fp32 VALU instruction whose LOW lane selects the HIGH register of a source pair (op_sel) can read 0 for that source
in lanes 48-63 when the pair was written immediately before — by an LDS return that a partial `s_waitcnt lgkmcnt(n >= 1)` has just
released, or by a VALU instruction one slot earlier — while another wave's MFMA occupies the SIMD.  hipcc (ROCm 7.2) emits such
sequences on its own (the round-5 fold arithmetic; the packed tower entry of round 5's residual stream) and does not pad them.
The product build therefore compiles WITHOUT packed fp32 instructions (`--packed` checks that); the windowed audit below is what found
and sized the problem in the older binaries.

The audit extracts every gfx950 code object of the library (clang offload bundles in .hip_fatbin), disassembles it with llvm-objdump and
reports, per kernel, every packed-fp32 instruction with a non-default op_sel / op_sel_hi whose DISTANCE (in instructions) behind a
partial `s_waitcnt` (lgkmcnt(n >= 1), or vmcnt(n >= 1) with --vmcnt) is below --distance (default 2: the micro-reproducer is clean from
one intervening instruction on).  Exit code 1 when anything is found.  `tests/test_isa_audit.py` runs it on the product library.

Usage: python tools/isa_audit.py [path/to/lib.so] [--distance N] [--vmcnt] [--all-packed] [-v]"""
import argparse
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    """gfx950 code objects (bytes) of every offload bundle in the file"""
    blob = open(path, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), blob):
        p = m.start()
        n = struct.unpack_from("<Q", blob, p + 24)[0]
        q = p + 32
        for _ in range(n):
            off, size, ts = struct.unpack_from("<QQQ", blob, q)
            q += 24
            triple = blob[q:q + ts].decode()
            q += ts
            if "gfx950" in triple and size:
                out.append(blob[p + off:p + off + size])
    return out


def disassemble(obj_bytes):
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(obj_bytes)
        name = f.name
    try:
        return subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "--no-leading-addr", name], capture_output=True, text=True, check=True).stdout
    finally:
        os.unlink(name)


PK = re.compile(r"^\s*(v_pk_(?:fma|mul|add)_f32)\b(.*)$")
WAIT = re.compile(r"^\s*s_waitcnt\b(.*)$")
LABEL = re.compile(r"^(?:[0-9a-fA-F]+\s+)?<([^>]+)>:\s*$")


def partial_wait(args, with_vmcnt):
    m = re.search(r"lgkmcnt\((\d+)\)", args)
    if m and int(m.group(1)) >= 1:
        return True
    if with_vmcnt:
        m = re.search(r"vmcnt\((\d+)\)", args)
        if m and int(m.group(1)) >= 1:
            return True
    return False


def audit_text(text, distance=2, with_vmcnt=False, all_packed=False):
    """-> list of (kernel, wait instruction, packed instruction, distance)"""
    found = []
    kernel, since, wait_txt = "?", None, ""
    for line in text.splitlines():
        lm = LABEL.match(line.strip())
        if lm and not line.startswith(("\t", " ")):
            kernel, since = lm.group(1), None
            continue
        ins = line.strip()
        if not ins or ins.startswith((";", "//", ".")) or ins.endswith(":"):
            continue
        ins = ins.split("//")[0].strip()
        w = WAIT.match(ins)
        if w:
            if partial_wait(w.group(1), with_vmcnt):
                since, wait_txt = 0, ins
            else:
                since = None          # a full wait (or one that only names other counters) ends the window
            continue
        if since is not None:          # `since` instructions lie between the partial wait and this one (0 = straight behind)
            p = PK.match(ins)
            if p and (all_packed or "op_sel" in p.group(2)):
                found.append((kernel, wait_txt, ins, since))
            since += 1
            if since >= distance:
                since = None
    return found


PACKED_F32 = re.compile(r"^\s*(v_pk_(?:fma|mul|add)_f32|v_pk_mov_b32)\b")


def count_packed_fp32(path):
    """-> {kernel: number of packed fp32 VALU instructions}: the product build has none at all (Makefile: -target-feature -packed-fp32-ops)"""
    per = {}
    for obj in code_objects(path):
        kernel = "?"
        for line in disassemble(obj).splitlines():
            lm = LABEL.match(line.strip())
            if lm and not line.startswith(("\t", " ")):
                kernel = lm.group(1)
            elif PACKED_F32.match(line):
                per[kernel] = per.get(kernel, 0) + 1
    return per


def audit_library(path, distance=2, with_vmcnt=False, all_packed=False):
    found = []
    for obj in code_objects(path):
        found += audit_text(disassemble(obj), distance, with_vmcnt, all_packed)
    return found


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib", nargs="?", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mvlpt_amd", "libmvlpt_hip.so"))
    ap.add_argument("--distance", type=int, default=2, help="flag packed ops fewer than this many instructions behind the partial wait (1 = straight behind)")
    ap.add_argument("--vmcnt", action="store_true", help="also treat a partial vmcnt wait as opening the window")
    ap.add_argument("--all-packed", action="store_true", help="flag packed fp32 ops without op_sel too")
    ap.add_argument("--packed", action="store_true", help="count EVERY packed fp32 instruction (the product library must have none)")
    ap.add_argument("-v", action="store_true")
    a = ap.parse_args()
    if a.packed:
        per = count_packed_fp32(a.lib)
        print(f"{a.lib}: {sum(per.values())} packed fp32 VALU instruction(s) (v_pk_fma/mul/add_f32, v_pk_mov_b32) in {len(per)} kernel(s)")
        for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:20 if a.v else 0]:
            print(f"  {v:5d}  {k[:150]}")
        return 1 if per else 0
    found = audit_library(a.lib, a.distance, a.vmcnt, a.all_packed)
    per = {}
    for k, w, p, d in found:
        per.setdefault(k, []).append((w, p, d))
    print(f"{a.lib}: {len(found)} packed-fp32 op_sel instruction(s) within {a.distance} instruction(s) of a partial s_waitcnt, in {len(per)} kernel(s)")
    for k, v in sorted(per.items(), key=lambda kv: -len(kv[1])):
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
        print(f"  {len(v):4d}  {name[:150]}")
        if a.v:
            for w, p, d in v[:6]:
                print(f"          {w}  ->  [{d}] {p}")
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
