#!/bin/bash
# Shared by pmc_groups.sh / pmc_probe.sh / profile_round.sh: one counter group per rocprofv3 pass, each pass under its own
# `timeout -k`, python3 directly behind `--` (no env / bash -c hop: the profiler's preload has the GPU initialised).
#
# Counter names are checked BEFORE any pass starts: a counter that is not on the list of names that have completed a pass on gfx950
# with this image needs PFC_PMC_FORCE=1 (and then still runs under the timeout, so a bad group costs one pass, not the call).
# The TA_* groups: in round 2 a pass with them aborted rocprofv3 and hung the evidence run (log not kept).  Round 4 ran ONE pass with
# TA_FLAT_READ_WAVEFRONTS_sum + TA_BUSY_avr, everything kept (scripts/ta_probe.sh, profiles/r04_ta_probe.txt): it completed -- exit code
# 0, plausible per-dispatch values for every kernel, nothing in rocprofv3's log.  The profiler does not reject the derived counters
# and the device does not fault on them; the round-2 abort is not reproducible with this image, so TA_* is no longer refused by name:
# the two names that passed are on the list, the other TA_* names are treated like any untested counter.
PMC_KNOWN="FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM \
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU \
SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 \
SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_LDS_BANK_CONFLICT \
SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum \
TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE GRBM_COUNT SQ_CYCLES \
SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_INSTS_EXP_GDS \
SQ_INSTS_VALU_IOPS SQ_INSTS_VMEM TCC_HIT TCC_MISS TCC_EA0_RDREQ SQ_ITEMS SQ_WAVES_EQ_64 SQ_LEVEL_WAVES SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_GDS LDSBankConflict \
TA_FLAT_READ_WAVEFRONTS_sum TA_BUSY_avr"
pmc_check_names() {   # pmc_check_names "<CTR1 CTR2 ...>" -> 0 ok, 1 refused (message on stderr)
  local c
  for c in $1; do
    if [ "${PFC_PMC_FORCE:-0}" != "1" ] && ! echo " $PMC_KNOWN " | grep -q " $c "; then
      echo "pmc: counter $c has not completed a pass on this image before; set PFC_PMC_FORCE=1 to try it (under the timeout)" >&2
      return 1
    fi
  done
  return 0
}
PMC_TIMEOUT=${PMC_TIMEOUT:-240}
pmc_pass() {   # pmc_pass <outdir> <logfile> "<counters>" <bench args...>
  local out=$1 log=$2 grp=$3; shift 3
  timeout -k 10 $PMC_TIMEOUT rocprofv3 --pmc $grp --output-format csv -d $out -- python3 $R/bench.py "$@" > $log 2>&1
}
