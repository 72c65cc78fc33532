#!/usr/bin/env python3
"""Build gate: scan the gfx950 code objects inside the given .o / .so files for instruction forms this library must
not ship.  Run by pangnn_amd/csrc/Makefile after every link; a hit fails the build.

Rule 1 — packed-f32 instruction whose LOW result takes the HIGH dword of src1 (`v_pk_{mul,add,fma,...}_f32 ... op_sel:[x,1...]`).
  On MI355X (gfx950) such an instruction computes its low result with src1 = 0 in lanes 48-63 whenever another wave of
  the same SIMD is issuing matrix instructions (v_mfma_*): tools/pk_opsel_probe.hip reproduces it in isolation (1e7
  wrong results per 5e10 with v_mfma_f32_16x16x32_bf16 beside it; none with one wave per SIMD, with a VALU-only
  neighbour, with op_sel on src0 / src2, with op_sel_hi, or with plain v_mul_f32) — profiles/r03_pk_opsel_probe.txt.
  hipcc -O3 emits the form from SLP-vectorised f32 math (a broadcast of the odd element of a register pair); it is
  what made the SLP build of csrc/decoder16.hip return dL/dh1 = v * (+0) for edge 13 / 29 of a tile in round 2
  (tools/slp_probe.py, tools/slp_probe_rows.py).  The library is built so that the form does not occur
  (-fno-slp-vectorize where SLP produced it) and this gate keeps a compiler or source change from bringing it back.

usage: check_isa.py file.o [file.o ...]        (exit status 1 and a listing on a hit)"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
# op_sel:[a,b] / op_sel:[a,b,c]: the element for src1 is the second one
BAD = re.compile(r"\bv_pk_[a-z0-9_]*_f32\b.*\bop_sel:\[[01],1")
KERNEL = re.compile(r"^[0-9a-f]+ <([^>]+)>:")


def scan(path):
    hits = []
    tmp = tempfile.mkdtemp(prefix="check_isa_")
    try:
        local = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        bundles = [f for f in os.listdir(tmp) if "amdgcn" in f]
        if not bundles:
            raise RuntimeError(f"{path}: no amdgcn code object found inside")
        for b in bundles:
            out = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, b)], check=True, capture_output=True, text=True).stdout
            kernel = "?"
            for line in out.splitlines():
                m = KERNEL.match(line)
                if m:
                    kernel = m.group(1)
                elif BAD.search(line):
                    hits.append((kernel, line.split("//")[0].strip()))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return hits


def main(argv):
    bad = 0
    for path in argv:
        for kernel, ins in scan(path):
            bad += 1
            print(f"{path}: {kernel}: {ins}")
    if bad:
        print(f"check_isa: {bad} packed-f32 instruction(s) take the high dword of src1 for the low result — wrong in lanes 48-63 "
              f"beside another wave's MFMAs on gfx950 (tools/check_isa.py, rule 1).  Build that file with -fno-slp-vectorize "
              f"or restructure the source.", file=sys.stderr)
        return 1
    print(f"check_isa: {len(argv)} file(s) clean")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
