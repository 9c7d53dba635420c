#!/usr/bin/env bash
# Same-box A/B of library builds and environment switches on the bench (round 5): each line of the spec is
#   "<label> <library under visualslam_amd/lib/> <bench args...> [:: ENV=VALUE ...]"
# usage: bash tools/ab_libs.sh <reps> <spec file>
set -o pipefail
reps=${1:-2}; spec=$2
for rep in $(seq $reps); do
  while read -r label lib rest; do
    [ -z "$label" ] && continue
    args="${rest%%::*}"; envs=""; [[ "$rest" == *::* ]] && envs="${rest#*::}"
    env $envs VSLAM_LIBRARY=$PWD/visualslam_amd/lib/$lib python bench.py --steps 20 --warmup 5 --modes 1 --cxx-host 0 --cpu-sample 0 --live-traffic 0 --mx 0 $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={}
for e in d['roofline_by_kernel']: k[(e['kernel'],e['octave'])]=round(e['ms_per_step'],3)
kn='k_pyr_octave_mx' if d['config']['matrix_path'] else 'k_pyr_octave'
print('$label', 'strips', [k.get(('k_gauss_h_strip',o)) for o in (2,3)], 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'oct', [k.get((kn,o)) for o in range(4) if (kn,o) in k], 'harris', k.get(('k_harris_strip',None)))"
  done < $spec
done
