"""profiles/traffic.json from the rocprofv3 counter rows tools/pmc_traffic.sh collected.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads and is doubled
(MI355X_MICROARCH.md §HBM); both count L2 fabric requests, Infinity-Cache hits included.  The doubling is calibrated
for 16-byte-per-lane streaming reads only: for the dgrad kernel's random 32-byte record gathers the RAW counter
(64 B per edge) is already the sector traffic, so `_bytes_fetch_undoubled` is the figure to read there."""
import csv, json, os, sys, collections

out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = {"decoder_train16": "decoder_train", "decoder_dgrad16": "decoder_dgrad", "spmm_row_kernel<64, 4, false, false>": "spmm_fwd"}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("fetch_size", "write_size"):
    f = os.path.join(out, f"{tag}_pmc_{c}.csv")
    for r in csv.DictReader(open(f)):
        for k, short in names.items():
            if k in r["Kernel_Name"]:
                acc[short][c].append(float(r["Counter_Value"]))
path = os.path.join(root, "profiles", "traffic.json")
d = json.load(open(path)) if os.path.exists(path) else {}
d["_source"] = f"profiles/{tag}_pmc_fetch_size.csv + {tag}_pmc_write_size.csv"
for short, v in acc.items():
    if not v["fetch_size"] or not v["write_size"]:
        continue
    fetch = sum(v["fetch_size"]) / len(v["fetch_size"])
    write = sum(v["write_size"]) / len(v["write_size"])
    d[f"cfg4_n1_{short}_bytes"] = int((2 * fetch + write) * 1024)
    d[f"cfg4_n1_{short}_bytes_fetch_undoubled"] = int((fetch + write) * 1024)
    d[f"cfg4_n1_{short}_fetch_kib_raw"] = fetch
    d[f"cfg4_n1_{short}_write_kib"] = write
    print(short, "fetch KiB (raw)", fetch, "write KiB", write, "=> bytes", d[f"cfg4_n1_{short}_bytes"])
json.dump(d, open(path, "w"), indent=1)
for c in ("fetch_size", "write_size"):
    src = os.path.join(out, f"{tag}_pmc_{c}.csv")
    dst = os.path.join(root, "profiles", f"{tag}_pmc_{c}.csv")
    if os.path.abspath(src) != os.path.abspath(dst):
        open(dst, "w").write(open(src).read())
