#!/bin/bash
# gemm_wsx_kernel forms on ONE box: PNPP_WSX3=0 float32 MFMA, 1 (default) split dA product, 2 split dA and dW products.  The level tests run on
# the form named by $1 (default 2), then the step is timed in all three.
set -e -o pipefail
mkdir -p gpurun_out/wx3
PNPP_WSX3=${1:-2} timeout -k 10 500 python -m pytest tests/test_gpu_levels_routed.py tests/test_gpu_split_products.py tests/test_gpu_sa.py -x -q -m gpu > gpurun_out/wx3/tests.log 2>&1 || { tail -40 gpurun_out/wx3/tests.log; exit 1; }
tail -3 gpurun_out/wx3/tests.log
for i in 1 2; do
for v in 0 1 2; do
PNPP_WSX3=$v PNPP_BENCH_DUMP=gpurun_out/wx3/table_${v}_$i.txt timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-bf16-variant --no-mfma-variant > gpurun_out/wx3/bench_${v}_$i.json 2>/dev/null
done; done
grep -H "gemm_wsx" gpurun_out/wx3/table_[012]_*.txt
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/wx3/bench_[012]_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["final_loss"])
PY
