#!/bin/bash
# rocprofv3 kernel durations of the l2-shape chain forward for a list of AGCN_WS_DBG values
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in "$@"; do
  rm -rf /tmp/wsprof
  AGCN_WS_DBG=$d REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wsprof -- python3 $R/tools/bench_gcn.py ${WHICH:-fwdn} ${LAYER:-l2} > /dev/null 2>&1
  echo "== dbg=$d"
  python3 $R/tools/prof_summary.py /tmp/wsprof 1 6 | grep -v "^total"
done
