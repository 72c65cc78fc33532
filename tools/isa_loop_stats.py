"""Static instruction mix of the main loop of a kernel in gfx950 ISA text (hipcc -S --cuda-device-only).
usage: python tools/isa_loop_stats.py file.s <kernel-name-substring> [mfma-mnemonic-prefix]
The loop is taken as the innermost backward branch that encloses every MFMA of the kernel."""
import collections
import re
import sys


def main(path, name, mf_prefix="v_mfma"):
    src = open(path).read().split("\n")
    start = [i for i, l in enumerate(src) if re.match(r"^_Z\w*:", l) and name in l][0]
    end = [i for i in range(start, len(src)) if src[i].strip().startswith("s_endpgm")][0]
    body = src[start:end]
    mf = [i for i, l in enumerate(body) if l.strip().startswith(mf_prefix)]
    lab = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            lab[m.group(1)] = i
    cands = []
    for i, l in enumerate(body):
        m = re.match(r"\s*s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < mf[0] and i > mf[-1]:
            cands.append((lab[m.group(1)], i))
    a, b = max(cands, key=lambda x: x[0])
    cnt, ops = collections.Counter(), collections.Counter()
    for l in body[a:b + 1]:
        s = l.strip()
        if not s or s[0] in ";." or s.endswith(":"):
            continue
        op = s.split()[0]
        if op.startswith("v_mfma"):
            cnt["mfma"] += 1
        elif op.startswith("v_"):
            cnt["valu"] += 1
            ops[op] += 1
        elif op.startswith("ds_"):
            cnt["lds"] += 1
        elif op.split("_")[0] in ("global", "buffer", "scratch", "flat"):
            cnt["vmem"] += 1
            if op.startswith("scratch"):
                cnt["scratch"] += 1
        elif op.startswith("s_waitcnt"):
            cnt["waitcnt"] += 1
        elif op.startswith("s_nop"):
            cnt["s_nop"] += 1
        elif op.startswith("s_"):
            cnt["salu"] += 1
    print(f"{name}: loop lines {a}..{b} of {len(body)}", dict(cnt))
    print("  valu:", ops.most_common(24))
    for l in src[end:end + 120]:
        if any(k in l for k in ("NumVgprs", "ScratchSize", "Occupancy", "LDSByteSize")):
            print("  ", l.strip())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], *(sys.argv[3:4]))
