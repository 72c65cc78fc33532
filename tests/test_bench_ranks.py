"""bench.py's N > 1 branch end to end (graph agreement check, partition, per-rank timers, max-over-ranks, ONE JSON
line from rank 0) — rehearsed with several ranks on the single GPU of the test box over gloo (rehearsal hooks in
bench.py: PANGNN_BENCH_BACKEND / PANGNN_BENCH_ONE_GPU); the scaling run proper is one rank per GPU over RCCL."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]


def _run(cmd, env_extra):
    env = dict(os.environ, **env_extra)
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                    # exactly one line on stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 4])
def test_bench_multi_rank_line(world):
    common = ["--workload", "cfg2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"]
    one = _run([sys.executable, "bench.py"] + common, {})
    many = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                 "--master-addr", "127.0.0.1", "--master-port", str(29640 + world), "bench.py", "--gpus", str(world)]
                + common, {"PANGNN_BENCH_BACKEND": "gloo", "PANGNN_BENCH_ONE_GPU": "1"})
    for d in (one, many):
        assert all(k in d for k in REQUIRED), d.keys()
        assert d["value"] > 0 and d["steps"] == 4 and d["warmup"] == 2
    assert one["n_gpus"] == 1 and many["n_gpus"] == world
    assert many["config"]["sim_edges"] == one["config"]["sim_edges"]          # the whole job's edges, not a shard's
    assert "destination-partitioned" in many["config"]["partition"]
    # same seed, same initial weights, same six steps: the partitioned job follows the single-GPU loss
    assert abs(many["config"]["final_loss"] - one["config"]["final_loss"]) < 2e-3
    assert many["roofline"]["avg_launch_ms"] > 0


def test_bench_single_gpu_line_carries_roofline_step_bytes_and_cpu_baseline():
    """the N = 1 line of the measurement contract on a small workload: every required key, the roofline object of the
    dominant kernel with its live launch time, the step-level byte accounting whose terms add up, the strict-fp32 extra,
    and the CPU baseline (the oracle's train step on a bounded sample)"""
    d = _run([sys.executable, "bench.py", "--workload", "cfg2", "--steps", "3", "--warmup", "1", "--cpu-genes", "100",
              "--cpu-steps", "1"], {})
    assert all(k in d for k in REQUIRED), d.keys()
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "edges/s" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert abs(d["value"] - d["config"]["sim_edges"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms"):
        assert k in r, k
    # `bound` names what binds the kernel (vector-instruction issue); achieved / peak / frac stay the byte figures
    assert r["bound"] == "hbm" and r["limiter"] == "issue" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    gf = d["general_features"]                        # the same step with the * propagate and its transpose inside it
    assert gf["ms_per_step"] > 0 and gf["propagate_fwd_ms"] > 0 and gf["propagate_bwd_ms"] > 0
    assert 0 < gf["propagate_share_of_step"] < 1 and abs(gf["final_loss"] - d["config"]["final_loss"]) < 0.5
    assert r["traffic"] is None                       # profiles/traffic.json holds cfg-4 entries only: no stale figure
    assert abs(sum(d["step_alg_bytes_terms"].values()) - d["step_alg_bytes"]) < 1.0
    assert 0 < d["step_hbm_frac"] < 1 and d["strict_fp32"]["ms_per_step"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "oracle" in c["sample"]
    assert d["value"] > 10 * c["value"]


def test_bench_emulated_rank_line():
    """bench.py --emulate-rank r --of W (the per-rank compute side of the W-way split on one GPU; tools/rank_emulation.py):
    one JSON line with the shard's sizes, its halo rows per owner, the S launches of the own-source and halo-source ranges"""
    common = ["--workload", "cfg2", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"]
    lines = [_run([sys.executable, "bench.py", "--emulate-rank", str(r), "--of", "4"] + common, {"MASTER_PORT": str(29660 + r)})
             for r in (0, 2)]
    for r, d in zip((0, 2), lines):
        assert d["emulated_rank"] == r and d["of"] == 4 and d["ms_per_step"] > 0 and d["steps"] == 3
        assert d["sim_edges_local"] == d["own_source_edges"] + d["halo_source_edges"] > 0
        assert sum(d["halo_rows_by_owner"]["sim"]) == d["halo_rows"]["sim"] and d["halo_rows_by_owner"]["sim"][r] == 0
        assert len(d["decoder_S_launch_ms"]) == 2 and d["per_step_exchange_bytes"]["gradient_all_reduce"] == 4 * 29249   # node_dim 64, hidden_dim 64: 29 249 parameters
    assert lines[0]["node_range"][0] == 0 and lines[1]["node_range"][0] > lines[0]["node_range"][1] - 1
