#!/bin/bash
# host-fed Stream under settings of one environment variable on one box: tools/hostfed_ab.sh VAR v1 v2 ...  ("-" = unset)
VAR=$1; shift
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
S=$GRAFT_REPO_ROOT/visualslam_amd/bin/Stream
pick='import json,sys; d=json.loads(sys.stdin.read()); print("%s whole %.0f steady %.0f f/s (%.2f ms)" % (d["mode"], d["frames_per_sec"], d["steady_frames_per_sec"], d["steady_ms_per_batch"]))'
echo -n "device: "; $S --mode device --batches 20 --warmup 6 2>/dev/null | tail -1 | python3 -c "$pick"
for v in "$@"; do
  if [ "$v" = "-" ]; then echo -n "$VAR unset: "; $S --mode hostfed --batches 40 --warmup 6 2>/dev/null | tail -1 | python3 -c "$pick"
  else echo -n "$VAR=$v: "; env $VAR=$v $S --mode hostfed --batches 40 --warmup 6 2>/dev/null | tail -1 | python3 -c "$pick"; fi
done
