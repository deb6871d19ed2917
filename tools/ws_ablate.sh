#!/bin/bash
# Ablation of gcn_ws_kernel at the l2 shape: AGCN_WS_DBG bits 1 = no matrix work, 2 = no x loads after the first tile,
# 4 = no row-store.  One process per variant (the switch is read per call, but keep the timings independent).
for d in 0 1 2 4 3 5 6 7; do
  echo -n "dbg=$d  "
  AGCN_WS_DBG=$d REPS=20 python tools/bench_gcn.py fwdn,bwdx l2 2>/dev/null | tail -1
done
