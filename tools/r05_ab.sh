set -o pipefail
python -m pytest tests/test_gpu_batch.py tests/test_gpu_large.py tests/test_gpu_parity.py tests/test_gpu_ref_images.py -m gpu -q -x > gpurun_out/r05_t9.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_t9.log
for rep in 1 2 3; do for L in ab/prev.so libvslam.so; do
  VSLAM_LIBRARY=$PWD/visualslam_amd/lib/$L python bench.py --steps 20 --warmup 5 --modes 1 --cxx-host 0 --cpu-sample 0 --live-traffic 0 --mx 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={ (e['kernel'],e['octave']): round(e['ms_per_step'],3) for e in d['roofline_by_kernel']}
print('$L', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'oct', [k.get(('k_pyr_octave',o)) for o in range(2)], 'frac', round(d['roofline']['frac'],4), 'alone', round(d['roofline']['alone']['frac'],4))"
done; done | tee gpurun_out/r05_default_stage_ab.txt
