#!/bin/bash
# One GPU session: `bash scripts/gpu_job.sh <tag> [steps...]` on the GPU box (gpurun).  Steps:
#   tests   python -m pytest tests -m gpu            -> gpurun_out/<tag>/pytest_gpu.log
#   bench   python bench.py                          -> gpurun_out/<tag>/bench.json
#   bench20 python bench.py --steps 20 --warmup 5 (the driver's invocation) -> bench_steps20.json
#   calib   scripts/valu_calib.bin                   -> gpurun_out/<tag>/valu_calib.json
#   stats   rocprofv3 --kernel-trace --stats of a short bench run
#   pmc     three separate PMC passes (FETCH_SIZE | WRITE_SIZE | SQ counters), --kernel-trace only; NAVHIP_HANDOVER=events:
#           counter collection serialises kernels, and a kernel that waits for another queue's kernel never ends then
#   smoke   __graft_entry__.smoke()
#   ranks2  bench.py --gpus 2 under torchrun, both ranks on the one GPU, gloo (a plumbing check, not a measurement)
#   fuzz    tests/tools/fuzz_gpu.py
#   fuzzmore  the same sweeps over cases the committed logs have not seen (FUZZ_FIRST, FUZZ_N)
#   statepass tests/tools/bench_state_pass.py (the state half of the movement tick through the binding) + its kernel stats
#   hostov  scripts/host_overhead.py (enqueue time per tick: python / c / c + graph) + rank_cost_probe --strong
#   ticktests  tests/test_tick_gpu.py only
#   aux     scripts/bench_aux.py (host-buffer rates, LOS)  + scripts/cp_unit_hist.py when its build is there
TAG=$1; shift
STEPS=${@:-tests bench calib stats}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
for s in $STEPS; do
case $s in
tests) timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log ;;
testsall) timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -25 $OUT/pytest_gpu.log ;;
bench) timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err ;;
bench20) timeout 600 python bench.py --steps 20 --warmup 5 --no-crowded > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; tail -c 600 $OUT/bench_steps20.json ;;
benchq) timeout 600 python bench.py --no-cpu-baseline --no-crowded > $OUT/bench_quick.json 2> $OUT/bench_quick.err; tail -c 2500 $OUT/bench_quick.json; tail -3 $OUT/bench_quick.err ;;
calib) timeout 300 scripts/valu_calib.bin > $OUT/valu_calib.json 2>&1; cat $OUT/valu_calib.json ;;
stats) timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-crowded > $OUT/bench_under_prof.json 2> $OUT/prof.err
       f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 $f ;;
pmc) for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES"; do
       n=$(echo $c | cut -d' ' -f1)
       NAVHIP_HANDOVER=events timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$n -o p --output-format csv -- python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-crowded > $OUT/pmc_$n.json 2> $OUT/pmc_$n.err
       tail -c 300 $OUT/pmc_$n.err
     done ;;
aux) timeout 600 python scripts/bench_aux.py > $OUT/bench_aux.json 2> $OUT/bench_aux.err; tail -c 1200 $OUT/bench_aux.json
     [ -f build_prof/libnavhip_cphist.so ] && timeout 300 python scripts/cp_unit_hist.py > $OUT/cp_unit_hist.json 2> /dev/null ;;
cfgs) for c in 0 1 3 4; do timeout 600 python bench.py --config $c --no-crowded > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err; tail -c 400 $OUT/bench_cfg$c.json; tail -2 $OUT/bench_cfg$c.err; done ;;
stats20) timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats20 -o s --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-crowded > $OUT/bench_under_prof20.json 2> $OUT/prof20.err
       f=$(find $OUT/stats20 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 $f ;;
avail) timeout 120 rocprofv3 -L > $OUT/avail.txt 2>&1; grep -c . $OUT/avail.txt ;;
pmccp) i=0; for c in "SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
       i=$((i+1))
       NAVHIP_HANDOVER=events timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmccp_$i -o p --output-format csv -- python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-crowded > $OUT/pmccp_$i.json 2> $OUT/pmccp_$i.err
       tail -c 200 $OUT/pmccp_$i.err
       NAVHIP_HANDOVER=events timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmccpc_$i -o p --output-format csv -- python bench.py --crowded --steps 20 --warmup 3 --no-cpu-baseline > $OUT/pmccpc_$i.json 2> $OUT/pmccpc_$i.err
       tail -c 200 $OUT/pmccpc_$i.err
     done ;;
cpstats) timeout 400 python scripts/cp_stats.py > $OUT/cp_stats.json 2> $OUT/cp_stats.err; tail -c 300 $OUT/cp_stats.err
         timeout 400 python scripts/cp_stats.py --crowd > $OUT/cp_stats_crowd.json 2>> $OUT/cp_stats.err; tail -c 600 $OUT/cp_stats_crowd.json ;;
secondary) timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/secondary -o s --output-format csv -- python scripts/bench_secondary.py > $OUT/bench_secondary.json 2> $OUT/secondary.err
       tail -c 1500 $OUT/bench_secondary.json; f=$(find $OUT/secondary -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 $f | cut -c1-160 ;;
hostov) timeout 600 python scripts/host_overhead.py > $OUT/host_overhead.txt 2>&1; cat $OUT/host_overhead.txt | tail -12
        # (every world in a process of its own, then all four in one process: the two must agree since round 6)
        for w in 1 2 4 8; do timeout 300 python scripts/rank_cost_probe.py --strong $w; done > $OUT/rank_cost_strong_per_process.txt 2>&1; grep world $OUT/rank_cost_strong_per_process.txt
        timeout 600 python scripts/rank_cost_probe.py --strong 1 2 4 8 > $OUT/rank_cost_strong.txt 2>&1; grep world $OUT/rank_cost_strong.txt ;;
ticktests) timeout 900 python -m pytest tests/test_tick_gpu.py -m gpu -q > $OUT/pytest_tick.log 2>&1; tail -15 $OUT/pytest_tick.log ;;
fieldsab) # where in tick t the field builds of tick t+1 start: behind the neighbour walk (default) or at the start
      for m in neighbours start neighbours start; do NAVTICK_FIELDS_AFTER=$m timeout 300 python scripts/queue_probe.py --config 2 --reps 2 --ticks 60 2>&1 | grep rep | sed "s/^/fields after $m: /"; done > $OUT/fields_after_ab.txt; cat $OUT/fields_after_ab.txt ;;
hoab) # hand-overs through device memory (NAVHIP_HANDOVER: one bit per hand-over, 0 = events as before)
      for m in ${HO_MASKS:-0 15 0 15}; do for c in 2 0 2of8; do NAVHIP_HANDOVER=$m timeout 300 python scripts/queue_probe.py --config $c --reps 2 --ticks 60 2>&1 | grep rep | sed "s/^/handover $m: /"; done; done > $OUT/handover_ab.txt; cat $OUT/handover_ab.txt ;;
timeline) timeout 300 rocprofv3 --kernel-trace -d /tmp/tl2 -o t --output-format csv -- python scripts/queue_probe.py --config 2 --reps 1 --ticks 30 > /dev/null 2>&1
      python scripts/tick_timeline.py /tmp/tl2 12 2 > $OUT/timeline_cfg2.txt 2>&1; head -40 $OUT/timeline_cfg2.txt ;;
crowded) for i in 1 2 3; do timeout 300 python bench.py --crowded --steps 40 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('crowded world: %.4f ms per tick (median %.4f)' % (d['ms_per_step'], d['summary']['ms_per_step_median']))"; done > $OUT/crowded.txt; cat $OUT/crowded.txt ;;
fieldcus) # the compute units the field builds of tick t+1 may use beside the step of tick t (default 160 of 256)
      for n in 160 128 192 224 256 160 96; do NAVTICK_FIELD_CUS=$n timeout 300 python scripts/queue_probe.py --config 2 --reps 2 --ticks 60 2>&1 | grep rep | sed "s/^/field CUs $n: /"; done > $OUT/field_cus_ab.txt; cat $OUT/field_cus_ab.txt ;;
smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log ;;
ranks2) NAVHIP_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 > $OUT/bench_2ranks_gloo.json 2> $OUT/bench_2ranks_gloo.err; tail -c 400 $OUT/bench_2ranks_gloo.json ;;
fuzz) timeout 900 python tests/tools/fuzz_gpu.py > $OUT/fuzz.log 2>&1; tail -3 $OUT/fuzz.log
      timeout 600 python tests/tools/fuzz_gpu.py --state > $OUT/fuzz_state.log 2>&1; tail -2 $OUT/fuzz_state.log ;;
fuzzmore) # fresh cases: FUZZ_FIRST (default 24) .. +FUZZ_N (default 72) of the world sweep, seeds 124+.. of the settle sweep
      timeout 240 python tests/tools/fuzz_gpu.py ${FUZZ_N:-72} --first ${FUZZ_FIRST:-24} > $OUT/fuzz_more.log 2>&1; tail -2 $OUT/fuzz_more.log
      timeout 180 python tests/tools/fuzz_gpu.py ${FUZZ_N:-72} --first $(( 100 + ${FUZZ_FIRST:-24} )) --state > $OUT/fuzz_state_more.log 2>&1; tail -2 $OUT/fuzz_state_more.log ;;
statetests) timeout 900 python -m pytest tests/test_state_gpu.py tests/test_state_binding_gpu.py tests/test_binding_gpu.py -m gpu -q > $OUT/pytest_state.log 2>&1; tail -8 $OUT/pytest_state.log ;;
statepass16) timeout 400 python tests/tools/bench_state_pass.py --threads 16 > $OUT/state_pass_16threads.json 2> $OUT/state_pass_16threads.err; tail -c 700 $OUT/state_pass_16threads.json ;;
statepass) timeout 400 python tests/tools/bench_state_pass.py > $OUT/state_pass.json 2> $OUT/state_pass.err; tail -c 900 $OUT/state_pass.json; tail -2 $OUT/state_pass.err
       timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/statepass -o s --output-format csv -- python tests/tools/bench_state_pass.py --reps 20 > $OUT/state_pass_under_prof.json 2> $OUT/statepass_prof.err
       f=$(find $OUT/statepass -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|k_heading_gate|k_state_update|k_state_aux|k_settled_count|k_arrival_settle|k_spatial" $f | cut -c1-200 ;;
esac
done
