"""Distances of every HIP parameter gradient of the model tests to the fp64 oracle, as a JSON-lines file — the evidence behind
tests/test_hip_parity.py's OBSERVED_FLIPS (which (graph, flags, tensor) cases keep the relu-flip band instead of the
direct FP64_DIRECT bound).  Diagnostics, not a test: every tensor is checked against the band only, nothing is skipped.

  python tools/grad_distances.py gpurun_out/r04a/grad_fp64.jsonl
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(out):
    os.environ["PANGNN_GRAD_LOG"] = out
    import test_hip_parity as T
    names = [("sim_200x4", (64, 128)), ("cfg1_2genomes", (64, 128)), ("cfg2_sim_1000x5", (64, 64)), ("cfg3_5genomes", (64, 128))]
    over = []
    for name, dims in names:
        for flags in T.FLAG_SETS:
            fid = "-".join(f"{k}={v}" for k, v in flags.items()) or "default"
            g, gd, oracle, model = T._pair(name, dims, dict(flags))
            keys = {k for k, _ in model.named_parameters()}
            _, _, worst = T._check_logits_loss_grads_against_oracle(g, gd, oracle, model, tag=f"{name}/{fid}", flips=keys)
            for k, (a, b) in worst.items():
                if a > T.FP64_DIRECT:
                    over.append((name, fid, k, a, b))
    print("cases above the direct bound (graph, flags, tensor, HIP distance, fp32-oracle distance):")
    for o in over:
        print("   ", o)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "grad_fp64.jsonl")
