"""Instruction-class sequence of a kernel's main loop (see isa_loop_stats.py for how the loop is found):
M = v_mfma 16x16, W = v_mfma 32x32, v = other VALU, L = LDS, g = global/buffer, s = SALU, _ = s_waitcnt, n = s_nop,
P = s_setprio, B = branch / barrier, | = label.  Shows whether vector work sits BETWEEN matrix instructions.
usage: python tools/isa_sequence.py file.s <kernel-name-substring>"""
import re
import sys


def main(path, name):
    src = open(path).read().split("\n")
    start = [i for i, l in enumerate(src) if re.match(r"^_Z\w*:", l) and name in l][0]
    end = [i for i in range(start, len(src)) if src[i].strip().startswith("s_endpgm")][0]
    body = src[start:end]
    mf = [i for i, l in enumerate(body) if l.strip().startswith("v_mfma")]
    lab = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    cands = []
    for i, l in enumerate(body):
        m = re.match(r"\s*s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < mf[0] and i > mf[-1]:
            cands.append((lab[m.group(1)], i))
    a, b = max(cands, key=lambda x: x[0])
    out = []
    for l in body[a:b + 1]:
        s = l.strip()
        if not s or s[0] in ";.":
            continue
        if s.endswith(":"):
            out.append("|")
            continue
        op = s.split()[0]
        if op.startswith("v_mfma"):
            out.append("M" if "16x16" in op else "W")
        elif op.startswith("v_"):
            out.append("v")
        elif op.startswith("ds_"):
            out.append("L")
        elif op.startswith(("global_", "buffer_", "scratch_")):
            out.append("g")
        elif op.startswith("s_waitcnt"):
            out.append("_")
        elif op.startswith("s_nop"):
            out.append("n")
        elif op.startswith("s_setprio"):
            out.append("P")
        elif op.startswith(("s_barrier", "s_cbranch", "s_branch")):
            out.append("B")
        else:
            out.append("s")
    t = "".join(out)
    for i in range(0, len(t), 150):
        print(t[i:i + 150])


if __name__ == "__main__":
    main(*sys.argv[1:3])
