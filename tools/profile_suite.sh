#!/bin/bash
# rocprofv3 kernel-trace stats of the whole GPU test suite: which native kernels ran, how often, how long.
set +e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_suite
mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o suite -- python3 -m pytest tests -q -m gpu > $OUT/stdout.txt 2>&1
tail -3 $OUT/stdout.txt
