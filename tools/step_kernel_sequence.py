"""The ordered kernel launches of ONE step from a rocprofv3 --kernel-trace CSV directory: every launch between the last two
launches of a marker kernel (default: collate_padded_kernel, the first kernel of a replayed fresh mini-batch step), with its
duration and the idle gap in front of it.  usage: step_kernel_sequence.py <dir> [marker-substring]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
marker = sys.argv[2] if len(sys.argv) > 2 else "collate_padded_kernel"
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t_prev = int(rows[a - 1]["End_Timestamp"]) if a else int(rows[a]["Start_Timestamp"])
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t_prev) / 1e3:7.1f} us gap  {(e - s) / 1e3:7.1f} us  {r['Kernel_Name'][:120]}")
    busy += e - s
    t_prev = e
span = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
print(f"{b - a} launches, {span / 1e3:.1f} us from marker to marker, {busy / 1e3:.1f} us inside kernels")
