#!/bin/bash
# Collect the measurement artefacts of a build on the GPU box (run from the repo root): bench lines of every workload,
# rocprofv3 kernel stats of the headline step, SQ (MFMA-busy / stall) counters.  Outputs under gpurun_out/final/.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/final
rm -rf $O && mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --workload ntu_aagcn --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_ntu_aagcn.json 2> /dev/null
python3 bench.py --workload ntu_aagcn_bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_ntu_aagcn_bf16.json 2> /dev/null
python3 bench.py --workload kinetics_agcn --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_kinetics_agcn.json 2> /dev/null
AGCN_GEMM=f32 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_f32mfma.json 2> /dev/null
AGCN_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof.log 2>&1
python3 tools/prof_summary.py $O/prof 8 70 > $O/kernel_stats.txt
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
AGCN_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_sq.log 2>&1
python3 tools/pmc_sq.py $O/pmc_sq > $O/pmc_sq_summary.txt
AGCN_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_clk -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_clk.log 2>&1
python3 tools/pmc_clock.py $O/pmc_clk > $O/pmc_clock_summary.txt
rm -rf $O/prof/*/*kernel_trace.csv $O/pmc_sq/*/*kernel_trace.csv $O/pmc_clk/*/*kernel_trace.csv
head -c 600 $O/bench_default.json; echo; head -20 $O/kernel_stats.txt
