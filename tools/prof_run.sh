#!/bin/bash
# usage (on the GPU box): bash tools/prof_run.sh <tag>   -> gpurun_out/<tag>/{phase.log,by_shape_d.txt,seq_d.txt,by_shape_all.txt}
set -e
T=${1:-rX}; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 120 python tools/phase_bench.py graph both 40 > $O/phase.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pd -o p -- python tools/phase_bench.py graph d 20 > $O/prof_d.log 2>&1
python tools/trace_by_shape.py /tmp/pd 25 60 > $O/by_shape_d.txt
python tools/trace_seq.py /tmp/pd > $O/seq_d.txt
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -o p -- python tools/phase_bench.py graph g 20 > $O/prof_g.log 2>&1
python tools/trace_by_shape.py /tmp/pg 25 70 > $O/by_shape_g.txt
python tools/trace_seq.py /tmp/pg > $O/seq_g.txt
tail -3 $O/phase.log
