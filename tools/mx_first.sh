set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_batch.py -m gpu -x -q -k "matrix_path_switch or matrix_core_octave" > gpurun_out/mx_test.log 2>&1 || { tail -40 gpurun_out/mx_test.log; exit 1; }
tail -5 gpurun_out/mx_test.log
timeout -k 10 300 python bench.py --steps 10 --warmup 6 > gpurun_out/mx_bench_default.json 2> gpurun_out/mx_bench_default.err
VSLAM_MX=1 timeout -k 10 300 python bench.py --steps 10 --warmup 6 > gpurun_out/mx_bench_mx.json 2> gpurun_out/mx_bench_mx.err
python - <<'PY'
import json
for f in ("gpurun_out/mx_bench_default.json","gpurun_out/mx_bench_mx.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d.get("roofline"))
PY
