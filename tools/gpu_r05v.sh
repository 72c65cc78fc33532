#!/bin/bash
# round 5, call v: float16 P | Q tables in the decoder (decoder16_f16.o) — the f16 tests, the accelerate loops, the decoder tests
set -o pipefail
O=gpurun_out/r05v; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_f16_rows.py tests/test_accelerate_loop.py -q -m gpu -x -s > $O/f16.log 2>&1 || { tail -60 $O/f16.log | cut -c1-240; exit 1; }
grep -h "fp16 loop" $O/f16.log; tail -2 $O/f16.log
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_dist_gpu.py -q -m gpu -x -k "decoder or bf16 or S_and_T or dist or partition" > $O/dec.log 2>&1 || { tail -40 $O/dec.log | cut -c1-220; exit 1; }
tail -2 $O/dec.log
