#!/bin/bash
# Regenerates the evidence behind bench.py's headline line on a GPU box (run via gpurun):
#   bash tools/refresh_profiles.sh r04
# default bench line, rocprofv3 kernel stats of the same command, three separate PMC passes
# (SQ counters / FETCH_SIZE / WRITE_SIZE: never combined, never with a trace), latency tables; since round 4 the same for
# the opt-in matrix path (bench.py --matrix-path 1: profiling only, never the headline).
set -e
R=${1:-refresh}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
python3 $B > $OUT/bench_default.json 2> $OUT/bench.err
python3 $B --cpu-sample 0 --modes 0 --octaves 0 --kernel k_harris_strip --mx 0 > $OUT/bench_harris.json 2>> $OUT/bench.err
Q="--cpu-sample 0 --modes 0 --mx 0 --cxx-host 0 --live-traffic 0"
S="--frames 64 --steps 1 --warmup 1 --roofline-pass 0 $Q"  # exactly two batch calls per PMC pass (tools/make_traffic.py divides by them)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o runc -- python3 $B $Q > $OUT/prof.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmcA -o r -- python3 $B $S > $OUT/pmcA.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmcB -o r -- python3 $B $S > $OUT/pmcB.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmcC -o r -- python3 $B $S > $OUT/pmcC.log 2>&1
# the matrix path (opt-in): the same command with --matrix-path 1
python3 $B $Q --matrix-path 1 --steps 10 > $OUT/bench_mx.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mx -o runc -- python3 $B $Q --matrix-path 1 > $OUT/prof_mx.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/mxA -o r -- python3 $B $S --matrix-path 1 > $OUT/mxA.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/mxB -o r -- python3 $B $S --matrix-path 1 > $OUT/mxB.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/mxC -o r -- python3 $B $S --matrix-path 1 > $OUT/mxC.log 2>&1
VSLAM_AUX_STREAMS=0 VSLAM_LIBRARY=$GRAFT_REPO_ROOT/visualslam_amd/lib/libvslam_diag.so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mx_serial -o runc -- python3 $B $Q --matrix-path 1 > $OUT/prof_mx_serial.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $OUT/pmcA $OUT/pmcB $OUT/pmcC > $OUT/pmc_f64.json
python3 tools/pmc_summary.py $OUT/mxA $OUT/mxB $OUT/mxC > $OUT/pmc_mx_f64.json
python3 tools/mx_alone.py --octaves 4 > $OUT/alone_dot.json 2>> $OUT/bench.err
VSLAM_MX=1 python3 tools/mx_alone.py --octaves 4 > $OUT/alone_mx.json 2>> $OUT/bench.err
python3 tools/bench_sizes.py > $OUT/sizes_latency.txt 2>> $OUT/bench.err
python3 tools/bench_rows.py > $OUT/rows_latency.json 2>> $OUT/bench.err
# the C++ host of the throughput path (one rank over RCCL): device-resident and host-fed
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 ./visualslam_amd/bin/Stream --mode device --batches 30 --warmup 6 2>/dev/null | tail -1 > $OUT/stream_device.json
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 ./visualslam_amd/bin/Stream --mode hostfed --batches 40 --warmup 6 2>/dev/null | tail -1 > $OUT/stream_hostfed.json
VSLAM_MX=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 ./visualslam_amd/bin/Stream --mode device --batches 30 --warmup 6 2>/dev/null | tail -1 > $OUT/stream_device_mx.json
VSLAM_MX=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 ./visualslam_amd/bin/Stream --mode hostfed --batches 40 --warmup 6 2>/dev/null | tail -1 > $OUT/stream_hostfed_mx.json
cut -c1-300 $OUT/bench_default.json
# the orientation and descriptor stages on the bench's `modes` content (timing at 256 frames, one PMC pass at 64)
bash tools/orient_pmc.sh $R/orient > $OUT/orient_stage.txt 2>&1 || true
bash tools/sift_pmc.sh $R/sift > $OUT/sift_stage.txt 2>&1 || true
