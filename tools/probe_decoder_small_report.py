"""mean / min duration of the S and T kernels per variant of tools/probe_decoder_small.py from a rocprofv3 kernel trace"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
for name in ("decoder_train16_kernel", "decoder_dgrad16_kernel"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]]
    tags = [r["Kernel_Name"].split("(")[0][-40:] for r in rows if name in r["Kernel_Name"]]
    per = 40 if name.startswith("decoder_train") else None
    if per is None:
        per = len(d) // 6 if len(d) % 6 == 0 else 40
    for i in range(0, len(d), per):
        g = d[i:i + per]
        print(f"{name:26s} launches {i:4d}..{i + len(g) - 1:4d}  {tags[i]:42s} mean {sum(g) / len(g):6.1f} us  min {min(g):6.1f} us")
