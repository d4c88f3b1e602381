#!/bin/bash
# usage (on the GPU box): bash tools/collect_profiles.sh <tag>
# -> gpurun_out/<tag>/{bench.log, bench_kernel_stats.csv, bench_kernel_stats_summary.txt, bench_kernels_by_shape.txt,
#                      step_kernels.txt, pmc_traffic.txt, traffic.json}   (copy the ones to be judged into profiles/)
set -e
T=${1:-prof}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
STEPS=100
# 1. kernel trace + stats of the bench command itself (no CPU baseline: it is not GPU work) ...
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python3 bench.py --steps $STEPS --warmup 20 --no-cpu-baseline > $O/bench_prof.log 2>&1
cp $(find /tmp/pb -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv
python tools/prof_summary.py /tmp/pb $STEPS 40 > $O/bench_kernel_stats_summary.txt
# ... and of its timed region alone (tools/step_only.py brackets the K steps with marker launches): true per-step numbers
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o s -- python3 tools/step_only.py 40 10 > $O/step_prof.log 2>&1
python tools/step_trace_summary.py /tmp/ps 40 60 > $O/step_kernels.txt
# 2./3. HBM traffic counters of the roofline kernels: separate --pmc passes
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o f -- python3 tools/roofline_kernels.py both 10 > $O/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o w -- python3 tools/roofline_kernels.py both 10 > $O/pmc_write.log 2>&1
python tools/pmc_to_json.py /tmp/pf /tmp/pw $O/traffic.json > $O/pmc_traffic.txt
# 4. the plain bench line (with the CPU baseline)
timeout -k 10 400 python bench.py > $O/bench.log 2>&1
tail -1 $O/bench.log
