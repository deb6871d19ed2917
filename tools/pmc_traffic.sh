#!/bin/bash
# Regenerate profiles/pmc_traffic.json: HBM bytes per launch of the kernels tools/pmc_roofline.py runs, from two
# rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE need separate passes: 4 TCC slots), keyed by the fingerprint of the
# kernel sources so that bench.py can refuse a stale file.  Run on the GPU box from the repo root:
#     bash tools/pmc_traffic.sh
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_traffic
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/pmc_roofline.py > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 tools/pmc_roofline.py > $OUT/write.log 2>&1
python3 tools/pmc_parse.py $OUT/fetch $OUT/write gpurun_out/pmc_traffic.json
# in place for the bench runs that follow in the same box session (and copied back with gpurun_out/)
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
