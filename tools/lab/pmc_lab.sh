# PMC passes over the gather-mean lab (separate rocprofv3 runs, counters only with --kernel-trace).
set -e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_lab
rm -rf $OUT; mkdir -p $OUT
BIN=$GRAFT_REPO_ROOT/tools/micro/spmm_lab.bin
ARGS="${LAB_ARGS:-128 4 8 0,21,11 8}"
cd /tmp
i=0
for C in \
 "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_sum TCC_BUSY_sum" ; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --kernel-trace --pmc $C -d $OUT/p$i --output-format csv -- $BIN $ARGS > $OUT/p$i.log 2>&1
  echo "pass $i done"
done
ls -R $OUT | head -30
