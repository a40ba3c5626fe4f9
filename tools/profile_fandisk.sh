#!/bin/bash
# rocprofv3 view of BASELINE config 2 (fandisk all-pairs through field_utils.field_grad, 50 calls): kernel-trace
# stats, then two PMC groups in runs of their own.
set +e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_fd
rm -rf $OUT; mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o fd -- python3 tools/gpu_k1_probe.py fandisk > $OUT/trace_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o fd -- python3 tools/gpu_k1_probe.py fandisk > $OUT/pmc_sq_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_misc -o fd -- python3 tools/gpu_k1_probe.py fandisk > $OUT/pmc_misc_stdout.txt 2>&1
find $OUT -name "*.csv" | head
