"""Where the S kernel's waves spend their cycles AT FULL SIZE (cfg-4 graph, the headline instance): a diagnostic build of
decoder16.hip with -DPANGNN_D16_CYC (loaded through PANGNN_HIP_LIB) makes every wave add the shader-clock cycles between six
marks of its loop body into per-phase sums over ALL of its half tiles; the eight waves of workgroup 0 write them out.

    cd pangnn_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DPANGNN_D16_CYC -c decoder16.hip \
        -o ../../build_variants/decoder16_cyc.o && hipcc --offload-arch=gfx950 -shared -fPIC spmm.o edge_ops.o graph_build.o \
        decoder.o ../../build_variants/decoder16_cyc.o linear.o -o ../../build_variants/libpangnn_hip_cyc.so
    PANGNN_HIP_LIB=build_variants/libpangnn_hip_cyc.so python tools/probe_decoder_cycles.py

A wave's own clock runs while its SIMD partner (two waves per SIMD) executes, so a wave's cycles per half tile are about TWICE
the SIMD time per wave and half tile that kernel duration / half tiles per SIMD gives; the table prints both."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangnn_amd import _lib, functional as PF    # noqa: E402
from pangnn_amd import simulate                  # noqa: E402
from pangnn_amd.graph import structure_of        # noqa: E402

PHASES = ["0 next ids requested, gathered rows of this half awaited, h1 = relu(p + q)",
          "1 P1: operand splits, 48 x 16x16x32, relu, w3 dot, lane sums -> logit",
          "2 next rows requested, loss / dL/dlogit, m2 operands, g_e h1 splits + images, P3 (12 x 32x32x16), records",
          "3 P2 (24 x 16x16x32), mask factors, run sums by source",
          "4 wave barrier that frees the tile images",
          "5 loop / chunk bookkeeping between half tiles"]
dev = torch.device("cuda")
lib = _lib.load()
if not hasattr(lib, "pangnn_debug_set_cyc"):
    sys.exit("load a -DPANGNN_D16_CYC build through PANGNN_HIP_LIB")
lib.pangnn_debug_set_cyc.argtypes = [ctypes.c_void_p]
cyc = torch.zeros(64, dtype=torch.int64, device=dev)
assert lib.pangnn_debug_set_cyc(cyc.data_ptr()) == 0
g = simulate.simulate_graph(50000, 20, 0.2, 100, 20, seed=0, device=dev)
n, e = g.num_nodes, g.edge_index.shape[1]
st = structure_of(g.edge_index, n, g, "sim")
torch.manual_seed(0)
pq = torch.randn(n, 128, device=dev)
par = [torch.randn(64, 64, device=dev) / 8, torch.randn(64, device=dev) * 0.1, torch.randn(64, device=dev) / 8,
       torch.randn(1, device=dev)]
PF.KERNEL_TIMER = {"dec.bwd": [], "dec.dgrad": []}
rows = []
for it in range(6):
    cyc.zero_()
    PF.decoder_loss_pq(pq, st, None, None, *par, g.y, g.class_balance, e)
    torch.cuda.synchronize()
    rows.append(cyc.view(8, 8).cpu().double())
ms = sorted(a.elapsed_time(b) for a, b in PF.KERNEL_TIMER["dec.bwd"][1:])
ms_med = ms[len(ms) // 2]
t = torch.stack(rows[1:]).mean(0)                      # [wave][0..5 phases, 6 halves, 7 entry -> loop exit]
halves = t[:, 6]
per_half = t[:, :6] / halves[:, None]
clk = torch.cuda.get_device_properties(0).clock_rate * 1e3 if hasattr(torch.cuda.get_device_properties(0), "clock_rate") else 2.4e9
cus = torch.cuda.get_device_properties(0).multi_processor_count
half_tiles = (e + 15) // 16
simd_cycles_per_half = ms_med * 1e-3 * 2.4e9 / (half_tiles / (cus * 4.0))
print(f"S kernel, cfg 4 (E = {e}), instrumented build: {ms_med:.3f} ms median of {len(ms)} launches "
      f"(product build on the same box: see the line tools/time_decoder_ab.py prints)")
print(f"half tiles per SIMD {half_tiles / (cus * 4.0):.0f}; kernel time x 2.4 GHz / that = {simd_cycles_per_half:.0f} cycles of SIMD time per "
      f"(wave, half tile) — two waves share a SIMD, so a wave's own clock sees about twice that per half tile")
print(f"workgroup 0: half tiles per wave {halves.tolist()}")
print(f"cycles from kernel entry to loop exit per wave: {[int(x) for x in t[:, 7].tolist()]}  (s_memtime counts at "
      f"{float(t[:, 7].mean()) / (ms_med * 1e-3):.3g} Hz against the kernel duration)")
mean = per_half.mean(0)
tot = float(mean.sum())
print("\nmean over the 8 waves of workgroup 0, wave-clock cycles per half tile (and scaled to the SIMD-time total):")
for i, name in enumerate(PHASES):
    print(f"  {float(mean[i]):8.0f}  {100 * float(mean[i]) / tot:5.1f} %  -> {simd_cycles_per_half * float(mean[i]) / tot:7.0f}   phase {name}")
print(f"  {tot:8.0f}  100.0 %  -> {simd_cycles_per_half:7.0f}   sum")
print("\nper wave (cycles per half tile, phases 0..5):")
for w in range(8):
    print(f"  wave {w}: " + "  ".join(f"{float(x):7.0f}" for x in per_half[w]))
