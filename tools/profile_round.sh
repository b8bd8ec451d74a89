#!/bin/bash
# Profiles of one round, run on the GPU box from the repository root:  bash tools/profile_round.sh r03
# 1. rocprofv3 --kernel-trace --stats of the default bench command (headline on f32 + the emulated block);
# 2. three counter passes of `bench.py --steps 1 --warmup 0` (separate --pmc runs, as the guide's HBM section
#    prescribes): FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE.
# Results: gpurun_out/<tag>_kernel_stats.csv, <tag>_pmc_{fetch_size,write_size,mfma_busy}.csv,
# <tag>_bench_under_rocprof.json; copy them to profiles/ and run profiles/summarize_pmc.py <tag> 49152 there.
set -u
TAG=${1:-r03}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
biggest() { ls -S "$1"/*/*"$2" 2>/dev/null | head -1; }
rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG/stats --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra \
    > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/prof_${TAG}_stats.err || echo "stats run failed"
FAILED=0
# a pass that produced no CSV is reported and skipped (cp with an empty argument would fail anyway), and the script exits non-zero
keep() { if [ -n "$1" ] && [ -f "$1" ]; then cp "$1" "$2"; else echo "no CSV for $2: pass failed" >&2; FAILED=1; fi; }
keep "$(biggest $OUT/prof_$TAG/stats _kernel_stats.csv)" $OUT/${TAG}_kernel_stats.csv
for pass in "fetch_size FETCH_SIZE" "write_size WRITE_SIZE" "mfma_busy SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    set -- $pass
    name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" -d $OUT/prof_$TAG/pmc_$name --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra \
        > $OUT/prof_${TAG}_pmc_$name.json 2> $OUT/prof_${TAG}_pmc_$name.err || echo "pmc $name run failed"
    keep "$(biggest $OUT/prof_$TAG/pmc_$name _counter_collection.csv)" $OUT/${TAG}_pmc_$name.csv
    echo "pass $name done"
done
ls -la $OUT/${TAG}_*
exit $FAILED
