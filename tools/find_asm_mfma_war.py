"""Scan gfx950 ISA (hipcc -S --cuda-device-only) for inline-asm vector instructions whose DESTINATION register is a
source operand (A, B or C) of an MFMA issued at most `window` instructions earlier.

Why: LLVM's hazard recognizer pads VALU writes that would overtake the operand reads of an in-flight MFMA, but it
does not look inside inline asm.  An asm statement with a pure output ("=v") can therefore be placed right behind an
MFMA that still reads that VGPR; the MFMA's later passes then see the new value.  Symptom in this repo: round 1's
decoder_bwd_x3_kernel built with AGPR-form MFMAs gave intermittently wrong gradients in the partial tile (8 hits in
that build, 0 in the -amdgpu-mfma-vgpr-form=1 build it was shipped with); the first version of decoder16.hip zeroed
rows 12-15 (the last MFMA pass) of its second product now and then (9-13 hits per kernel instance).

usage: python tools/find_asm_mfma_war.py file.s [window]     exit code 1 if any hit"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1):
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def scan(path, window=40):
    src = open(path).read().split("\n")
    hits, inasm, func = [], False, "?"
    for i, line in enumerate(src):
        s = line.strip()
        if re.match(r'^[_A-Za-z][\w.$]*:\s*(;.*)?$', line) and not s.startswith(".L"):
            func = s.split(":")[0]
        if s.startswith(";;#ASMSTART"):
            inasm = True
            continue
        if s.startswith(";;#ASMEND"):
            inasm = False
            continue
        if not (inasm and s.startswith("v_")):
            continue
        dst = regs(s) if s.startswith("v_permlane") else regs(s.split(",")[0])
        cnt, j = 0, i - 1
        while j > 0 and cnt < window:
            t = src[j].strip()
            if t and not t.startswith(";") and not t.startswith("."):
                cnt += 1
                if t.startswith("v_mfma"):
                    ops = t.split(None, 1)[1].split(",")
                    srcs = set()
                    for o in ops[1:]:
                        srcs |= regs(o)
                    if dst & srcs:
                        hits.append((func, i + 1, cnt, s, t))
                        break
            j -= 1
    return hits


if __name__ == "__main__":
    h = scan(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
    for func, line, dist, a, m in h:
        print(f"{func[:60]} line {line}: `{a}` {dist} instructions after `{m}`")
    print(f"{len(h)} inline-asm definitions overlap the sources of a recently issued MFMA")
    sys.exit(1 if h else 0)
