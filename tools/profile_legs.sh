#!/bin/bash
# rocprofv3 --kernel-trace of every leg of the product, one process per leg (tools/gpu_leg.py), for the per-kernel table of the
# final tree (profiles/SUMMARY_<round>.md, tools/make_summary.py):   tools/profile_legs.sh r05 [leg ...]
set +e
RND=${1:-r05}; shift
LEGS=${@:-"config4_driver config3_reps config2_fandisk allpairs_100k config5_reference_field potential_lattice config1_points config1_points_f64 config4_driver_f64 config2_fandisk_f64 allpairs_100k_f64 config5_reference_field_f64 potential_lattice_f64 xie_order xie_order_f64 xie_knn prep_partition"}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_${RND}_legs
mkdir -p $OUT
for LEG in $LEGS; do
  rm -rf $OUT/$LEG; mkdir -p $OUT/$LEG
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$LEG -o t -- python3 $R/tools/gpu_leg.py $LEG $OUT/$LEG.json > $OUT/$LEG.stdout.txt 2>&1
  grep "ms per call" $OUT/$LEG.stdout.txt
  find $OUT/$LEG -name "*kernel_trace.csv" -exec mv {} $OUT/$LEG.kernel_trace.csv \;
  rm -rf $OUT/$LEG
done
ls $OUT | head -40
