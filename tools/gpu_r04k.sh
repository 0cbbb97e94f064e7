#!/bin/bash
# round 4, call k: what bounds the flat pooling backward -- SQ counters + texture-addresser / L1 counters of pool3_bwd_kernel
export TMPDIR=/tmp
O=gpurun_out/r04k; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TA|TCP|TD|TCC)_[A-Za-z0-9_]+|MemUnit[A-Za-z]+|WriteUnit[A-Za-z]+|L2CacheHit|FetchSize|WriteSize|LDSBankConflict|VALUBusy|SALUBusy|MemWrites32B|FETCH_SIZE|WRITE_SIZE" | sort -u > $O/avail_mem_counters.txt
wc -l $O/avail_mem_counters.txt
bash tools/pmc_kernel.sh pool_flat pool3_bwd $O/pmc_sq_pool_flat > /dev/null 2>&1; cat $O/pmc_sq_pool_flat/summary.txt
i=3
for P in "TA_BUSY_avr TA_BUSY_max TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
         "MemUnitBusy MemUnitStalled WriteUnitStalled"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $P --output-format csv -d $O/q$i -o pmc -- python tools/prof_kernel.py pool_flat > $O/q$i.log 2>&1
  python tools/pmc_summary.py $O/q$i pool3_bwd 2>&1 | tail -12
  tail -2 $O/q$i.log
  rm -rf $O/q$i
done
