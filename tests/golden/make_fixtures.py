#!/usr/bin/env python3
"""Generate golden graph-construction fixtures from the reference's OWN construction code.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (needs /root/reference, which never
travels to the GPU box); the `.npz` files it writes under tests/golden/ are the only thing the
test-suite reads.

What is pinned: the pure-Python host pipeline of the reference
  src/simulate.py:83-230, src/preprocessing.py:73-156,264-325,370-548, src/helper.py:327-433,
  src/dataset.py:58-158,222-395
i.e. raw similarity scores -> remove_trivial_cases -> normalize_sim_scores -> edge_index /
edge_attr / y / neighbour_edge_index, for the whole graph and for the per-ortholog-group
sub-graphs.  The GCNConv arithmetic is NOT in the reference tree (third-party torch_geometric,
absent here) and is therefore not pinned by these fixtures (DESIGN.md "parity unpinned").

How: the reference modules import `torch_geometric` (absent) and `seaborn` (absent) at module
level but only use `Data`/`Dataset` as attribute holders on this path.  We put an arithmetic-free
holder stub for those names on sys.path from a temp dir OUTSIDE the repo, import `src.dataset`
from /root/reference, seed every RNG (the reference never seeds), replace multiprocessing.Pool by
a serial map, and dump tensors.  One subprocess per config because `src.setup` parses
`sys.argv` at import time (setup.py:53).  PYTHONHASHSEED=0 pins str-set iteration order
(preprocessing.py:463, helper.py:340-362).

Usage:  python tests/golden/make_fixtures.py            # all configs
        python tests/golden/make_fixtures.py --only sim_200x4
"""
import os, sys, subprocess, tempfile, json, argparse, textwrap

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

STUB = {
    "torch_geometric/__init__.py": "",
    "torch_geometric/data/__init__.py": textwrap.dedent('''
        class Data:
            """attribute holder only (no arithmetic)"""
            def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, **kw):
                self.x, self.edge_index, self.edge_attr, self.y = x, edge_index, edge_attr, y
                for k, v in kw.items():
                    setattr(self, k, v)
        class Dataset:
            def __init__(self, root=None, transform=None, pre_transform=None, pre_filter=None):
                pass
    '''),
    "torch_geometric/utils/__init__.py": "def to_networkx(*a, **k):\n    raise NotImplementedError\n",
    "torch_geometric/utils/convert.py": "def to_scipy_sparse_matrix(*a, **k):\n    raise NotImplementedError\n",
    "torch_geometric/transforms/__init__.py": "class RemoveDuplicatedEdges:\n    pass\n",
    "seaborn/__init__.py": "",
}

CONFIGS = {
    # name: (argv for the reference, max number of sub-graphs kept)
    # NOTE: data/dummy_dataset cannot be built by the reference itself (ZeroDivisionError at
    # dataset.py:319: its single group yields no positive edge), so it is not a fixture.
    # BASELINE config 1: default 2-genome CPU --train
    "cfg1_2genomes": (["--train", "-@", "1",
                       "-a", f"{REF}/data/Cga_08-1274-3_RENAMED.gff", f"{REF}/data/Cga_12-4358_RENAMED.gff",
                       "-s", f"{REF}/data/mmseq2_result.csv",
                       "-r", f"{REF}/data/holy_python_ribap_95.csv"], 10**9),
    # BASELINE config 3: all 5 bundled Chlamydia GFFs
    "cfg3_5genomes": (["--train", "-@", "1",
                       "-a"] + [f"{REF}/data/{g}_RENAMED.gff" for g in
                                ("Cav_10DC88", "Cav_11DC096", "Cga_08-1274-3", "Cga_12-4358", "Ctr_A-HAR-13")] +
                      ["-s", f"{REF}/data/mmseq2_result.csv",
                       "-r", f"{REF}/data/holy_python_ribap_95.csv"], 96),
    # small simulated graph (fast oracle sizes)
    "sim_200x4": (["--train", "-@", "1", "--simulate_dataset", "200", "4", "0.3", "10", "2"], 64),
    # BASELINE config 2
    "cfg2_sim_1000x5": (["--train", "-@", "1", "--simulate_dataset", "1000", "5", "0.3", "10", "2"], 64),
}


def child(name, out_path, max_sub):
    """Runs inside the per-config subprocess (cwd = scratch dir, stub dir + REF on sys.path)."""
    import random
    import numpy as np
    import torch
    random.seed(0); np.random.seed(0); torch.manual_seed(0)

    import src.setup as setup            # parses sys.argv (setup.py:53)
    import src.preprocessing as prep
    import src.dataset as ds

    class SerialPool:                    # dataset.py:140 uses Pool(...).map
        def __init__(self, *a, **k): pass
        def __enter__(self): return self
        def __exit__(self, *a): return False
        def map(self, fn, it): return [fn(x) for x in it]
    ds.Pool = SerialPool

    captured = {}
    real_rtc = prep.remove_trivial_cases

    def spy_rtc(d):                      # the earliest integer-representable stage of both pipelines
        captured["raw_before_trivial"] = {k: dict(v) for k, v in d.items()}
        return real_rtc(d)
    ds.remove_trivial_cases = spy_rtc
    prep.remove_trivial_cases = spy_rtc  # load_similarity_score resolves the module global

    real_gsg = ds.UnionGraphDataset.generate_sub_graphs

    def spy_gsg(self, groups):           # keep local->global node ids (split_data deletes gene_lst)
        res = real_gsg(self, groups)
        for g in res[0]:
            g._golden_gene_idx = [self.gene_id_position_dict[x] for x in g.gene_lst]
        return res
    ds.UnionGraphDataset.generate_sub_graphs = spy_gsg

    args = setup.args
    if args.simulate_dataset:
        dset = ds.UnionGraphDataset(calculate_baseline=True, split=(0.7, 0.15, 0.01),
                                    categorical_nodes=False)
    else:
        dset = ds.UnionGraphDataset(args.annotation, args.similarity, args.ribap_groups,
                                    split=(0.7, 0.15, 0.01), categorical_nodes=False,
                                    calculate_baseline=True)
    whole = dset.generate_graphs()       # what inference mode / simulate test uses (dataset.py:325)

    genes = list(dset.gene_str_ids_lst)  # node id == list position (gene_id_position_dict)
    pos = dset.gene_id_position_dict
    genome_names = sorted({g.split('_')[0] for g in genes})
    genome_of = np.array([genome_names.index(g.split('_')[0]) for g in genes], dtype=np.int32)

    def dict_to_coo(d):
        s, t, w = [], [], []
        for a, inner in d.items():
            for b, sc in inner.items():
                # genes that are not in any loaded GFF keep id -1 (dropped by build_edge_index)
                s.append(pos.get(a, -1)); t.append(pos.get(b, -1)); w.append(float(sc))
        return (np.array(s, dtype=np.int64), np.array(t, dtype=np.int64), np.array(w, dtype=np.float64))

    raw_s, raw_t, raw_w = dict_to_coo(captured["raw_before_trivial"])
    flt_s, flt_t, flt_w = dict_to_coo(dset.sim_score_dict_raw)
    nrm_s, nrm_t, nrm_w = dict_to_coo(dset.sim_score_dict)

    # ortholog ("RIBAP") relation as directed gene-id pairs (key -> member), for label parity
    gp_s, gp_t = [], []
    for k, members in dset.ribap_groups_dict.items():
        if k not in pos:
            continue
        for m in members:
            if m in pos:
                gp_s.append(pos[k]); gp_t.append(pos[m])

    out = dict(
        num_nodes=np.int64(len(genes)),
        genome_of=genome_of,
        raw_src=raw_s, raw_dst=raw_t, raw_score=raw_w,
        flt_src=flt_s, flt_dst=flt_t, flt_score=flt_w,
        nrm_src=nrm_s, nrm_dst=nrm_t, nrm_weight=nrm_w,
        grp_src=np.array(gp_s, dtype=np.int64), grp_dst=np.array(gp_t, dtype=np.int64),
        whole_edge_index=whole.edge_index.numpy().astype(np.int64),
        whole_edge_attr=whole.edge_attr.numpy().astype(np.float32),
        whole_y=whole.y.numpy().astype(np.float32),
        whole_neighbour_edge_index=whole.neighbour_edge_index.numpy().astype(np.int64),
        whole_x=whole.x.numpy().astype(np.float32),
        class_balance_whole=np.float64(float(dset.class_balance)),
    )

    # per-ortholog-group sub-graphs in the order the (seeded) split_data shuffle left them:
    # train first, then val (dataset.py:172-213)
    subs = list(dset.train) + list(dset.val)
    subs = subs[:max_sub]
    n_off, e_off, nb_off = [0], [0], [0]
    ei, ea, yy, nb, gl = [], [], [], [], []
    for g in subs:
        gl.append(np.array(g._golden_gene_idx, dtype=np.int64))
        n_off.append(n_off[-1] + g.x.shape[0])
        e_off.append(e_off[-1] + g.edge_index.shape[1])
        nb_off.append(nb_off[-1] + g.neighbour_edge_index.shape[1])
        ei.append(g.edge_index.numpy()); ea.append(g.edge_attr.numpy()); yy.append(g.y.numpy())
        nb.append(g.neighbour_edge_index.numpy())
    if subs:
        out.update(
            sub_node_off=np.array(n_off, dtype=np.int64), sub_edge_off=np.array(e_off, dtype=np.int64),
            sub_nb_off=np.array(nb_off, dtype=np.int64),
            sub_edge_index=np.concatenate(ei, axis=1).astype(np.int64),
            sub_edge_attr=np.concatenate(ea).astype(np.float32),
            sub_y=np.concatenate(yy).astype(np.float32),
            sub_neighbour_edge_index=np.concatenate(nb, axis=1).astype(np.int64),
            sub_global_node=np.concatenate(gl),
            n_train=np.int64(len(dset.train)), n_val=np.int64(len(dset.val)),
            class_balance_train=np.float64(float(dset.class_balance)),
        )
    np.savez_compressed(out_path, **out)
    meta = {k: (list(v.shape) if hasattr(v, "shape") else None) for k, v in out.items()}
    print(json.dumps({"config": name, "N": int(len(genes)), "E_sim": int(whole.edge_index.shape[1]),
                      "E_nb": int(whole.neighbour_edge_index.shape[1]),
                      "pos_frac": float(whole.y.mean()), "n_sub": len(subs)}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--child", default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--max_sub", type=int, default=64)
    a, rest = ap.parse_known_args()
    if a.child:
        sys.argv = ["pangnn.py"] + rest
        child(a.child, a.out, a.max_sub)
        return
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    with tempfile.TemporaryDirectory(prefix="pangnn_golden_") as tmp:
        stub_dir = os.path.join(tmp, "stubs")
        for rel, body in STUB.items():
            p = os.path.join(stub_dir, rel)
            os.makedirs(os.path.dirname(p), exist_ok=True)
            with open(p, "w") as f:
                f.write(body)
        for name, (argv, max_sub) in CONFIGS.items():
            if a.only and a.only != name:
                continue
            work = os.path.join(tmp, name)
            os.makedirs(work)
            env = dict(os.environ, PYTHONHASHSEED="0", PYTHONPATH=f"{stub_dir}:{REF}",
                       MPLBACKEND="Agg", COLUMNS="200")
            out_path = os.path.join(HERE, f"{name}.npz")
            cmd = [sys.executable, os.path.abspath(__file__), "--child", name, "--out", out_path,
                   "--max_sub", str(max_sub)] + argv
            r = subprocess.run(cmd, cwd=work, env=env, capture_output=True, text=True)
            tail = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(name, "rc=", r.returncode, tail[-1] if tail else r.stderr[-2000:])


if __name__ == "__main__":
    main()
