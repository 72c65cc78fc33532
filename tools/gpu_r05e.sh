#!/bin/bash
# round 5, call e: the six SQ counter passes of the decoder kernels (A-C as round 4, D-F new)
set -o pipefail
O=gpurun_out/r05e; mkdir -p $O
timeout -k 10 900 bash tools/pmc_decoder.sh $O decoder_ > $O/pmc.log 2>&1 || { tail -20 $O/pmc.log; exit 1; }
sed -n 1,45p $O/summary.txt
