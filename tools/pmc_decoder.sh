#!/bin/bash
# SQ counters of the training decoder kernel (two rocprofv3 --pmc passes over tools/time_decoder_ab.py).
# usage (on the GPU box, from the repo root): bash tools/pmc_decoder.sh <out_dir> [kernel-name substring]
set -e
OUT=${1:-gpurun_out/pmc_dec}
PAT=${2:-decoder_}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
  --kernel-trace -d /tmp/pmcA -o a --output-format csv -- python3 "$ROOT/tools/time_decoder_ab.py" > "$ROOT/$OUT/passA.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD \
  --kernel-trace -d /tmp/pmcB -o b --output-format csv -- python3 "$ROOT/tools/time_decoder_ab.py" > "$ROOT/$OUT/passB.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
  --kernel-trace -d /tmp/pmcC -o c --output-format csv -- python3 "$ROOT/tools/time_decoder_ab.py" > "$ROOT/$OUT/passC.log" 2>&1 || true
# round 5: what else gfx950's counter set offers towards a stall breakdown (rocprofv3 --list-avail: the only wait counters are
# SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_INST_LDS — no VMEM / "other" split of SQ_WAIT_INST_ANY exists): per-class instruction
# cycles, per-class VALU instruction counts, LDS / TA back-pressure
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH SQ_CYCLES \
  --kernel-trace -d /tmp/pmcD -o d --output-format csv -- python3 "$ROOT/tools/time_decoder_ab.py" > "$ROOT/$OUT/passD.log" 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM SQ_INSTS_BRANCH \
  --kernel-trace -d /tmp/pmcE -o e --output-format csv -- python3 "$ROOT/tools/time_decoder_ab.py" > "$ROOT/$OUT/passE.log" 2>&1 || true
rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR \
  --kernel-trace -d /tmp/pmcF -o f --output-format csv -- python3 "$ROOT/tools/time_decoder_ab.py" > "$ROOT/$OUT/passF.log" 2>&1 || true
cd "$ROOT"
for p in A B C D E F; do
  f=$(find /tmp/pmc$p -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && { head -1 "$f"; grep "$PAT" "$f" || true; } > "$OUT/pass${p}_counters.csv"
done
python3 - "$OUT" <<'PY'
import csv, sys, collections, glob
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/pass*_counters.csv")):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, cs in acc.items():
        fh.write(k + "\n")
        for c, v in sorted(cs.items()):
            fh.write(f"  {c:32s} mean {sum(v)/len(v):.6g}  (n={len(v)})\n")
print(open(out + "/summary.txt").read())
PY
