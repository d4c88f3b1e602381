set -e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_lab2
rm -rf $OUT; mkdir -p $OUT
BIN=$GRAFT_REPO_ROOT/tools/micro/spmm_lab.bin
cd /tmp
for w in 7 6 5 4 3; do
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum GRBM_GUI_ACTIVE -d $OUT/w$w --output-format csv -- $BIN 128 4 8 21 $w > $OUT/w$w.log 2>&1
grep "nt=1" $OUT/w$w.log | sed "s/^/wg$w /"
done
