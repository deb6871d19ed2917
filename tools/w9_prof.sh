#!/bin/bash
# rocprofv3 kernel durations of the 9-tap weight gradient per layer shape (the transposes, the main kernel, the slab sum)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for l in "$@"; do
  rm -rf /tmp/w9prof
  REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/w9prof -- python3 $R/tools/bench_wgrad9.py $l > /dev/null 2>&1
  echo "== $l"
  python3 $R/tools/prof_summary.py /tmp/w9prof 1 5 | grep -v "^total"
done
