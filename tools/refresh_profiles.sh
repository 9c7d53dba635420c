#!/bin/bash
# Regenerates the evidence behind bench.py's headline line on a GPU box (run via gpurun):
#   bash tools/refresh_profiles.sh r03
# default bench line, rocprofv3 kernel stats of the same command, three separate PMC passes
# (SQ counters / FETCH_SIZE / WRITE_SIZE: never combined, never with a trace), latency tables.
set -e
R=${1:-refresh}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
python3 $B > $OUT/bench_default.json 2> $OUT/bench.err
python3 $B --cpu-sample 0 --modes 0 --octaves 0 --kernel k_harris_strip > $OUT/bench_harris.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o runc -- python3 $B --cpu-sample 0 --modes 0 > $OUT/prof.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmcA -o r -- python3 $B --frames 64 --steps 1 --warmup 1 --cpu-sample 0 --modes 0 > $OUT/pmcA.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmcB -o r -- python3 $B --frames 64 --steps 1 --warmup 1 --cpu-sample 0 --modes 0 > $OUT/pmcB.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmcC -o r -- python3 $B --frames 64 --steps 1 --warmup 1 --cpu-sample 0 --modes 0 > $OUT/pmcC.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $OUT/pmcA $OUT/pmcB $OUT/pmcC > $OUT/pmc_f64.json
python3 tools/bench_sizes.py > $OUT/sizes_latency.txt 2>> $OUT/bench.err
python3 tools/bench_rows.py > $OUT/rows_latency.json 2>> $OUT/bench.err
python3 tools/bench_hostfed.py > $OUT/hostfed_python.json 2>> $OUT/bench.err
# the C++ host of the throughput path (one rank over RCCL): device-resident and host-fed
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 ./visualslam_amd/bin/Stream --mode device --batches 30 --warmup 6 2>/dev/null | tail -1 > $OUT/stream_device.json
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 ./visualslam_amd/bin/Stream --mode hostfed --batches 40 --warmup 6 2>/dev/null | tail -1 > $OUT/stream_hostfed.json
python3 tools/mfma_probe.py > $OUT/mfma_probe.txt 2>> $OUT/bench.err || true
./tools/hbm_probe > $OUT/hbm_probe.json 2>> $OUT/bench.err || true
./tools/pcie_probe > $OUT/pcie_probe.json 2>> $OUT/bench.err || true
./tools/ubench_valu2 > $OUT/ubench_valu2.txt 2>&1 || true
cut -c1-300 $OUT/bench_default.json
