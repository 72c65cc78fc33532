#!/bin/bash
# round 5, call d: S-kernel stall attribution — available counters, same-box S / T / inference times, the cycle-instrumented
# build's per-phase table, the SQ counter passes
set -o pipefail
O=gpurun_out/r05d; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit $rc; fi; return $rc; }
(cd /tmp && export TMPDIR=/tmp && run 120 rocprofv3 --list-avail > $OLDPWD/$O/list_avail.txt 2>&1) || true
grep -c . $O/list_avail.txt
run 300 python tools/time_decoder_ab.py > $O/ab_intree.txt 2>&1 || { tail -5 $O/ab_intree.txt; exit 1; }
cat $O/ab_intree.txt
PANGNN_HIP_LIB=$PWD/build_variants/libpangnn_hip_cyc.so run 300 python tools/probe_decoder_cycles.py > $O/s_cycles.txt 2>&1 || { tail -20 $O/s_cycles.txt; exit 1; }
cat $O/s_cycles.txt
run 900 bash tools/pmc_decoder.sh $O decoder_ > $O/pmc.log 2>&1 || { tail -20 $O/pmc.log; exit 1; }
tail -5 $O/pmc.log
