#!/bin/bash
# Batched throughput at frame sizes next to the 1080p headline (profiles/rNN_sizes_throughput.txt).
echo "# python bench.py --rows R --cols C --frames N --steps 5 --cpu-sample 0 --modes 0 --cxx-host 0   (one MI355X, frames in HBM)"
for cfg in "2160 3840 64" "1080 1920 256" "720 1280 256" "480 640 512" "1234 2050 128"; do
  set -- $cfg
  python bench.py --rows $1 --cols $2 --frames $3 --steps 5 --cpu-sample 0 --modes 0 --cxx-host 0 --live-traffic 0 2>/dev/null | tail -n 1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['keypoints_per_step']
print('$1 $2 $3 frames/s %.1f ms/step %.2f list_overflow %s algorithmic GB/s %.0f kp/step %d %d' % (d['value'], d['ms_per_step'], k['list_overflow'], d['pipeline_hbm']['achieved_GBps'], k['harris'], k['dog']))"
done
