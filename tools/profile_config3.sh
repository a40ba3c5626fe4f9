#!/bin/bash
# rocprofv3 kernel-trace stats of BASELINE config 3 (strongest_field_propagation_reps on G15's boxunion cloud, the reference's
# 369 patches / 93 411 representatives; tools/gpu_reps_probe.py = 4 driver calls): which kernels a call launches and what each
# costs.  tools/profile_config3.sh r05  ->  gpurun_out/prof_<round>_config3/  (summary copied to profiles/<round>_config3_kernels.txt)
set +e
RND=${1:-r05}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_${RND}_config3
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o c3 -- python3 $R/tools/gpu_reps_probe.py > $OUT/stdout.txt 2>&1
find $OUT -name "*kernel_stats.csv" | head -3
