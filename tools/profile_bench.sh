#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box through gpurun):  tools/profile_bench.sh r04 [N]
# N (optional): BENCH_FAKE_WORLD=N - one rank's share of an N-rank run (the launch an N = 8 rank really makes: the split tail).
# kernel-trace stats, then the PMC passes in runs of their own (never combined with a trace domain other than
# --kernel-trace).  tools/summarize_prof.py <round> turns the CSVs into the committed summaries under profiles/.
set +e
RND=${1:-r05}
if [ -n "$2" ]; then export BENCH_FAKE_WORLD=$2; fi
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$RND
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 40 --warmup 10 --no-cpu-baseline --headline-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $RND -- python3 $R/bench.py $ARGS > $OUT/trace_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $RND -- python3 $R/bench.py $ARGS > $OUT/pmc_fetch_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $RND -- python3 $R/bench.py $ARGS > $OUT/pmc_write_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o $RND -- python3 $R/bench.py $ARGS > $OUT/pmc_sq_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc_misc -o $RND -- python3 $R/bench.py $ARGS > $OUT/pmc_misc_stdout.txt 2>&1 || true
grep -h "^{" $OUT/trace_stdout.txt | tail -1 > $OUT/bench_line_under_rocprof.json
find $OUT -name "*.csv" | head -50
