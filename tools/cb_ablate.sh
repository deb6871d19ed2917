#!/bin/bash
# conv_pc phase ablations at the l9-l10 shape: AGCN_CB_DBG bits 1 = consumers idle (barriers only), 2 = producers idle,
# 4 = no window staging, 8 = no weight DMA; for 8 and 4 consumer waves
R=${GRAFT_REPO_ROOT:-.}
for nwc in 4; do
  for d in 0 1 2 3 4 8 12; do
    echo -n "nwc=$nwc dbg=$d: "
    AGCN_CONV_NWC=$nwc AGCN_CB_DBG=$d python3 $R/tools/bench_conv9.py 2>/dev/null | grep "l9-10"
  done
done
