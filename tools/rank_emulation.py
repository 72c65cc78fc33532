"""Per-rank compute of the 8-way split on ONE GPU: `bench.py --emulate-rank r --of 8` for r = 0..7 in
turn — each rank's true shard (rank-local generation, balanced node ranges), its true halo plan, the exchange served by a
self-exchange of the same row counts over RCCL — collected into profiles/r05_rank_emulation.json with the balance figures.
The compute side only: NOT a scaling curve (no xGMI transfer, no skew between ranks, no all-reduce wait).
usage: python tools/rank_emulation.py [--workloads cfg4 cfg5] [--of 8] [--steps 10] [--out profiles/r05_rank_emulation.json]"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--workloads", nargs="+", default=["cfg4", "cfg5"])
ap.add_argument("--of", type=int, default=8)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_rank_emulation.json"))
args = ap.parse_args()
res = {"_doc": "bench.py --emulate-rank r --of W, r = 0..W-1 one after the other on ONE MI355X: the compute side of the "
               "destination-partitioned step per rank (true shard, true halo plan, self-exchange of the true row counts over "
               "RCCL).  NOT a scaling curve: no xGMI transfer, no inter-rank skew."}
for wl in args.workloads:
    rows = []
    for r in range(args.of):
        env = dict(os.environ, MASTER_PORT=str(29600 + r))
        t0 = time.time()
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--emulate-rank", str(r), "--of",
                            str(args.of), "--steps", str(args.steps), "--warmup", "3", "--no-cpu-baseline"],
                           capture_output=True, text=True, env=env, timeout=900)
        line = None
        for ln in p.stdout.splitlines():
            if ln.startswith("{"):
                line = json.loads(ln)
        if line is None:
            print(f"{wl} rank {r}: failed\n{p.stderr[-2000:]}", flush=True)
            sys.exit(1)
        line["wall_s"] = round(time.time() - t0, 1)
        rows.append(line)
        print(f"{wl} rank {r}/{args.of}: {line['ms_per_step']:.3f} ms/step, E_local {line['sim_edges_local']}, halo rows "
              f"{line['halo_rows']}, S launches {[round(x, 3) for x in line['decoder_S_launch_ms']]} ms, T {line['decoder_T_ms_per_step']:.3f} ms "
              f"({line['wall_s']} s)", flush=True)
    ms = [x["ms_per_step"] for x in rows]
    ed = [x["sim_edges_local"] for x in rows]
    res[wl] = {"ranks": rows, "ms_per_step": ms, "max_over_mean_ms": max(ms) / (sum(ms) / len(ms)),
               "max_over_mean_edges": max(ed) / (sum(ed) / len(ed)), "sum_local_edges": sum(ed),
               "slowest_rank_ms": max(ms), "edges_per_s_if_every_rank_took_the_slowest": sum(ed) / (max(ms) * 1e-3)}
    print(f"{wl}: max / mean ms {res[wl]['max_over_mean_ms']:.4f}, max / mean edges {res[wl]['max_over_mean_edges']:.4f}", flush=True)
json.dump(res, open(args.out, "w"), indent=1)
print("wrote", args.out)
