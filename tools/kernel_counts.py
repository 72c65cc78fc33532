"""Per-kernel launch counts and durations from a rocprofv3 --kernel-trace --stats CSV directory (launch-bound regimes:
how many launches a step is made of).  usage: kernel_counts.py <dir> <steps+warmup> [--by-time]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(f)))
tot_calls = sum(int(r["Calls"]) for r in rows)
tot_ns = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"{tot_calls} launches, {tot_ns / 1e6:.3f} ms of kernel time; per step ({steps:g}): {tot_calls / steps:.1f} launches, "
      f"{tot_ns / 1e3 / steps:.1f} us")
by_time = "--by-time" in sys.argv
for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs" if by_time else "Calls"])):
    if by_time:
        print(f'{int(r["TotalDurationNs"]) / 1e3 / steps:9.1f} us/step  {int(r["Calls"]) / steps:6.2f}/step  avg '
              f'{float(r["AverageNs"]) / 1e3:9.1f} us  {r["Name"][:110]}')
        continue
    print(f'{int(r["Calls"]) / steps:7.2f}/step  avg {float(r["AverageNs"]) / 1e3:8.1f} us  {r["Name"][:110]}')
