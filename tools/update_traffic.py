"""profiles/traffic.json from the rocprofv3 counter rows tools/pmc_traffic.sh collected.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads and is doubled
(MI355X_MICROARCH.md §HBM); both count L2 fabric requests, Infinity-Cache hits included.  The doubling is calibrated
for 16-byte-per-lane streaming reads only: for the dgrad kernel's random 32-byte record gathers the RAW counter
(64 B per edge) is already the sector traffic, so `bytes_fetch_raw` is the figure to read there.

Every entry records what it was measured on — full kernel name, workload, GPU count, similarity-edge count of the
profiled run (read from the bench line that run printed) — and bench.py only quotes an entry whose four match its own
run (`"traffic": null` otherwise) and whose `kernel_source_sha16` (hash of csrc/decoder16.hip, spmm.hip, common.h at
collection time; `collected_at_head` = $PANGNN_HEAD, the commit the collecting call was made from) equals the hash of the
running tree's sources.  usage: PANGNN_HEAD=$(git rev-parse --short HEAD) python tools/update_traffic.py <out_dir> <tag> [workload]"""
import collections
import csv
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "cfg4"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import TRAFFIC_KERNELS, kernel_source_sha16  # noqa: E402  (one table of kernel-name substrings / one source hash for the collector and the reader)


def bench_line(path):
    """the JSON line bench.py printed during a counter pass (stdout and stderr share the log)"""
    if not os.path.exists(path):
        return None
    for ln in open(path, errors="replace"):
        ln = ln.strip()
        if ln.startswith('{"metric"'):
            try:
                return json.loads(ln)
            except Exception:
                pass
    return None


acc = collections.defaultdict(lambda: collections.defaultdict(list))
names = {}
for c in ("fetch_size", "write_size"):
    f = os.path.join(out, f"{tag}_pmc_{c}.csv")
    for r in csv.DictReader(open(f)):
        for short, sub in TRAFFIC_KERNELS.items():
            if sub in r["Kernel_Name"]:
                acc[short][c].append(float(r["Counter_Value"]))
                names[short] = r["Kernel_Name"]
lines = [bench_line(os.path.join(out, f"pass_{c}.log")) for c in ("FETCH_SIZE", "WRITE_SIZE")]
edges = {ln["config"]["sim_edges"] for ln in lines if ln}
gpus = {ln["n_gpus"] for ln in lines if ln}
if len(edges) != 1 or len(gpus) != 1:
    sys.exit(f"the two counter passes must have printed bench lines of one and the same graph (edge counts {edges})")
e_sim, n_gpus = edges.pop(), gpus.pop()

path = os.path.join(root, "profiles", "traffic.json")
d = json.load(open(path)) if os.path.exists(path) else {}
d = {"_doc": "L2-fabric traffic per launch from rocprofv3 PMC (separate --pmc passes; FETCH_SIZE / WRITE_SIZE in KiB; on "
             "gfx950 FETCH_SIZE is doubled for streaming reads per MI355X_MICROARCH.md, HBM section: bytes_fetch_doubled; "
             "bytes_fetch_raw = undoubled, the right figure for random 32-byte gathers).  bench.py quotes an entry only when "
             "workload, n_gpus, sim_edges and the kernel name match its own run.",
     "_source": f"profiles/{tag}_pmc_fetch_size.csv + {tag}_pmc_write_size.csv",
     "entries": [e for e in d.get("entries", []) if not (e.get("workload") == workload and e.get("n_gpus") == n_gpus)]}
for short, v in acc.items():
    if not v["fetch_size"] or not v["write_size"]:
        continue
    fetch = sum(v["fetch_size"]) / len(v["fetch_size"])
    write = sum(v["write_size"]) / len(v["write_size"])
    ent = {"key": short, "kernel": names[short], "workload": workload, "n_gpus": n_gpus, "sim_edges": e_sim,
           "launches_averaged": len(v["fetch_size"]), "fetch_kib_raw": fetch, "write_kib": write,
           "bytes_fetch_doubled": int((2 * fetch + write) * 1024), "bytes_fetch_raw": int((fetch + write) * 1024),
           "source": d["_source"],
           # what the counters were collected ON: bench.py prints "traffic": null once the kernel sources differ
           "kernel_source_sha16": kernel_source_sha16(), "collected_at_head": os.environ.get("PANGNN_HEAD", "unknown")}
    d["entries"].append(ent)
    print(short, "fetch KiB (raw)", fetch, "write KiB", write, "=> bytes", ent["bytes_fetch_doubled"], "E =", e_sim)
json.dump(d, open(path, "w"), indent=1)
for c in ("fetch_size", "write_size"):
    src = os.path.join(out, f"{tag}_pmc_{c}.csv")
    dst = os.path.join(root, "profiles", f"{tag}_pmc_{c}.csv")
    if os.path.abspath(src) != os.path.abspath(dst):
        open(dst, "w").write(open(src).read())
