#!/bin/bash
# Runs on the GPU box (gpurun): the rocprofv3 evidence of one round, written under gpurun_out/prof_$1/ as small CSV / JSON files
# (the result databases are deleted: they are hundreds of MB).  Copy what is to be judged into profiles/ afterwards.
#   bash tools/profile_round.sh r03
# Passes (PMC passes separate from the kernel trace and from each other, as MI355X_MICROARCH.md prescribes):
#   1 kernel trace of the DEFAULT bench command            -> ${R}_bench_kernel_stats.csv, ${R}_bench_under_rocprof.json
#   2 kernel trace of `bench.py --serialize`               -> ${R}_gemm_serialized_kernel_stats.csv
#   3 --pmc FETCH_SIZE | 4 --pmc WRITE_SIZE | 5 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE   -> ${R}_pmc_*.csv
set -e    # a pass that fails or is killed ends the script: no further GPU step is started behind it
R=${1:-r03}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python3 bench.py $ARGS > $OUT/${R}_bench_under_rocprof.json 2> $OUT/kt.err
python3 tools/kstats.py $(ls $OUT/kt/*.db | head -1) 12 $OUT/${R}_bench_kernel_stats.csv > $OUT/kt_top.txt 2>&1 || true
python3 tools/stream_stats.py $(ls $OUT/kt/*.db | head -1) 12 > $OUT/${R}_streams.txt 2>&1 || true
rm -rf $OUT/kt
echo "pass 1 done"
rocprofv3 --kernel-trace -d $OUT/ks -o ks -- python3 bench.py $ARGS --serialize > $OUT/${R}_bench_serialized_under_rocprof.json 2> $OUT/ks.err
python3 tools/kstats.py $(ls $OUT/ks/*.db | head -1) 12 $OUT/${R}_gemm_serialized_kernel_stats.csv > $OUT/ks_top.txt 2>&1 || true
rm -rf $OUT/ks
echo "pass 2 done"
PARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pf -o pf -- python3 bench.py $PARGS > /dev/null 2> $OUT/pf.err
python3 tools/pmc_summary.py $(ls $OUT/pf/*.db | head -1) $OUT/${R}_pmc_fetch_size.csv || true; rm -rf $OUT/pf
echo "pass 3 done"
rocprofv3 --pmc WRITE_SIZE -d $OUT/pw -o pw -- python3 bench.py $PARGS > /dev/null 2> $OUT/pw.err
python3 tools/pmc_summary.py $(ls $OUT/pw/*.db | head -1) $OUT/${R}_pmc_write_size.csv || true; rm -rf $OUT/pw
echo "pass 4 done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pm -o pm -- python3 bench.py $PARGS > /dev/null 2> $OUT/pm.err
python3 tools/pmc_summary.py $(ls $OUT/pm/*.db | head -1) $OUT/${R}_pmc_mfma.csv || true; rm -rf $OUT/pm
echo "pass 5 done"
ls -la $OUT
