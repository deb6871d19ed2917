#!/bin/bash
# L1->L2 read requests of the persistent chain forward at l2-l4 (are the 16-byte x-fragment requests amplified on the way?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/wsl2
REPS=3 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/wsl2 -- python3 $R/tools/bench_gcn.py fwdn l2 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/wsl2/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen=set()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if (r['Dispatch_Id']) not in seen: seen.add(r['Dispatch_Id']); n[k]+=1
for k in acc:
    if 'gcn_ws' in k or 'gcn_chain' in k:
        print(k, {c: round(v / n[k] / 1e6, 2) for c, v in acc[k].items()}, '(millions per launch)')
PY
