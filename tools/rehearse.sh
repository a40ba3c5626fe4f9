#!/bin/bash
# BENCH_BACKEND=gloo rehearsal of bench.py --gpus 2 / 4 on one GPU (the whole N > 1 line: in-order headline, the pipelined
# loop - which falls back to the in-order gather under gloo, BENCH_PIPELINED=1 forces the loop -, the sharded driver leg),
# then one rank's share of an N-rank run (BENCH_FAKE_WORLD) for N = 1, 2, 4, 8: the evidence under
# profiles/r04_bench_gloo_rehearsal.txt and profiles/r04_rank_share_timing.txt.
set +e
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/rehearsal.txt
: > $OUT
for N in 2 4; do
  echo "## torch.distributed.run --nproc-per-node $N bench.py --gpus $N --steps 5 --warmup 2  (BENCH_BACKEND=gloo)" >> $OUT
  BENCH_BACKEND=gloo BENCH_PIPELINED=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2950$N bench.py --gpus $N --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/rehearsal_err_$N.txt | grep '^{' > gpurun_out/rehearsal_line_$N.json
  cat gpurun_out/rehearsal_line_$N.json >> $OUT
  python -c "
import json
b=json.load(open('gpurun_out/rehearsal_line_$N.json'))
print('# keys: value', b['value'], 'ms_per_step', b['ms_per_step'], '| value_pipelined', b.get('value_pipelined'), 'ms_per_step_pipelined', b.get('ms_per_step_pipelined'), 'pipelined_matches_in_order', b.get('pipelined_matches_in_order'), '|', b.get('pipelined_note'))
print('# sharded_driver', b.get('sharded_driver'))
print('# signs_ok', b['signs_ok'], 'trace_matches_reference_G19', b['trace_matches_reference_G19'], '| precheck', b.get('precheck'), 'collective_timeout_s', b.get('collective_timeout_s'))" >> $OUT
  echo >> $OUT
done
# a peer that dies before its first collective: rank 0 must end with an "error" line within the (shortened) collective timeout
echo "## BENCH_TEST_DIE_RANK=1 BENCH_COLLECTIVE_TIMEOUT_S=15: --nproc-per-node 2 (gloo), rank 1 leaves before its first collective" >> $OUT
T0=$(date +%s)
BENCH_BACKEND=gloo BENCH_TEST_DIE_RANK=1 BENCH_COLLECTIVE_TIMEOUT_S=15 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29507 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --headline-only 2>gpurun_out/rehearsal_err_die.txt | grep '^{' | cut -c1-600 >> $OUT
echo "# exit after $(( $(date +%s) - T0 )) s" >> $OUT
# the host-side deadline: a run that cannot finish in time ends with an "error" line and a non-zero exit
echo "## BENCH_DEADLINE_S=0.4 python bench.py (one GPU): the watchdog fires" >> $OUT
BENCH_DEADLINE_S=0.4 timeout -k 10 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | cut -c1-400 >> $OUT
echo "# exit code ${PIPESTATUS[0]}" >> $OUT
echo >> $OUT
S=gpurun_out/rank_share.txt
: > $S
for N in 1 2 4 8; do
  BENCH_FAKE_WORLD=$N timeout -k 10 200 python bench.py --no-cpu-baseline --headline-only --steps 50 --warmup 10 2>/dev/null | grep '^{' | python -c "
import json,sys
b=json.loads(sys.stdin.read()); r=b['roofline']; print($N, round(b['ms_per_step'],4), round(r['launch_ms'],4), round(r['launch_ms_median'],4), round(r['launch_ms_min'],4), r['source_split'], b['step_parts']['interactions_kernel_ms'])" >> $S
done
cat $S
# the asynchronous all-gather of the pipelined loop through RCCL itself, with the one rank a one-GPU box allows
echo "## BENCH_ONE_RANK_RCCL=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --headline-only  (one-rank nccl group)" >> $OUT
BENCH_ONE_RANK_RCCL=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --headline-only 2>gpurun_out/rehearsal_err_rccl1.txt | grep '^{' | python -c "
import json,sys
b=json.loads(sys.stdin.read())
print('# keys: value', b['value'], 'ms_per_step', b['ms_per_step'], '| value_pipelined', b.get('value_pipelined'), 'ms_per_step_pipelined', b.get('ms_per_step_pipelined'), 'pipelined_matches_in_order', b.get('pipelined_matches_in_order'), '|', b.get('pipelined_note'), '|', b.get('rehearsal'))" >> $OUT
