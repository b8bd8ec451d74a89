#!/bin/bash
# Profiles of one round, run on the GPU box from the repository root:
#     bash tools/profile_round.sh r05 [bench] [cfg2] [cfg3] [calib]        (no target = all four)
# Targets (the program always directly behind `--`, never a shell or env hop):
#   bench  python3 bench.py ...               the headline launch shape (49152 chunks) -> <tag>_kernel_stats.csv, <tag>_pmc_*.csv
#   cfg2   python3 tools/fft_batch_rate.py    BASELINE config 2: window + rFFT-320 + |X| at 1024 and 2^20 frames
#   cfg3   python3 tools/cfg3_loop.py         BASELINE config 3's literal 82-chunk batch, 204 calls back to back
#   calib  tools/fetch_calib                  known byte counts in this library's read shapes (what the counters mean)
# Per target: one `--kernel-trace --stats` run, then SEPARATE `--pmc` runs (the guide's HBM section: never with --stats):
#   fetch_size  FETCH_SIZE                      write_size  WRITE_SIZE
#   rdreq       TCC_EA0_RDREQ_sum + the 32B / 64B / 128B request counts (bytes = 32 a + 64 b + 128 c: no correction factor)
#   wrreq       TCC_EA0_WRREQ_sum + TCC_EA0_WRREQ_64B_sum
#   mfma_busy   SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
#   sq          SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU
# Results land in gpurun_out/; copy the <tag>_* files to profiles/ and run, there,
#     python3 profiles/summarize_pmc.py <tag> 49152          (bench)      python3 profiles/summarize_target.py <tag> cfg2|cfg3|calib
set -u
TAG=${1:-r05}
shift || true
TARGETS=${*:-bench cfg2 cfg3 calib}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
FAILED=0
biggest() { ls -S "$1"/*/*"$2" 2>/dev/null | head -1; }
# a pass that produced no CSV is reported and skipped, and the script exits non-zero
keep() { if [ -n "$1" ] && [ -f "$1" ]; then cp "$1" "$2"; else echo "no CSV for $2: pass failed" >&2; FAILED=1; fi; }

PASSES=(
  "fetch_size FETCH_SIZE"
  "write_size WRITE_SIZE"
  "rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
  "wrreq TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
  "mfma_busy SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
  "sq SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU"
)

# profile <prefix> <stats-args...> -- <pmc-args...>: both argument lists start with the program itself
run_target() {
    local prefix=$1; shift
    local stats_cmd=() pmc_cmd=() seen=0
    for a in "$@"; do
        if [ "$a" = "--" ]; then seen=1; continue; fi
        if [ $seen = 0 ]; then stats_cmd+=("$a"); else pmc_cmd+=("$a"); fi
    done
    rocprofv3 --kernel-trace --stats -d $OUT/prof_$prefix/stats --output-format csv -- "${stats_cmd[@]}" \
        > $OUT/${prefix}_under_rocprof.out 2> $OUT/prof_${prefix}_stats.err || echo "$prefix: stats run failed"
    keep "$(biggest $OUT/prof_$prefix/stats _kernel_stats.csv)" $OUT/${prefix}_kernel_stats.csv
    for pass in "${PASSES[@]}"; do
        set -- $pass
        local name=$1; shift
        rocprofv3 --kernel-trace --pmc "$@" -d $OUT/prof_$prefix/pmc_$name --output-format csv -- "${pmc_cmd[@]}" \
            > $OUT/prof_${prefix}_pmc_$name.out 2> $OUT/prof_${prefix}_pmc_$name.err || echo "$prefix: pmc $name run failed"
        keep "$(biggest $OUT/prof_$prefix/pmc_$name _counter_collection.csv)" $OUT/${prefix}_pmc_$name.csv
        echo "$prefix: pass $name done"
    done
}

for t in $TARGETS; do
    case $t in
    bench)
        run_target $TAG python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra
        cp $OUT/${TAG}_under_rocprof.out $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null ;;
    cfg2)
        run_target ${TAG}_cfg2 python3 tools/fft_batch_rate.py -- python3 tools/fft_batch_rate.py ;;
    cfg3)
        run_target ${TAG}_cfg3 python3 tools/cfg3_loop.py -- python3 tools/cfg3_loop.py ;;
    calib)
        run_target ${TAG}_calib tools/fetch_calib -- tools/fetch_calib ;;
    *) echo "unknown target $t" >&2; FAILED=1 ;;
    esac
done
ls -la $OUT/${TAG}_* 2>/dev/null
exit $FAILED
