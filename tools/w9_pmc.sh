#!/bin/bash
# SQ counters (MFMA-busy, waits) and clock of the 9-tap weight gradient kernels per layer shape
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for l in "$@"; do
  rm -rf /tmp/w9pmc /tmp/w9clk
  REPS=3 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/w9pmc -- python3 $R/tools/bench_wgrad9.py $l > /dev/null 2>&1
  echo "== $l"
  python3 $R/tools/pmc_sq.py /tmp/w9pmc wgrad9
  python3 $R/tools/pmc_sq.py /tmp/w9pmc tv_transpose | tail -1
  REPS=3 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/w9clk -- python3 $R/tools/bench_wgrad9.py $l > /dev/null 2>&1
  python3 $R/tools/pmc_clock.py /tmp/w9clk 2>/dev/null | grep -i "wgrad9\|GHz\|kernel" | head -4
done
