#!/bin/bash
# gemm_wsx_kernel with the split dA product (<64,1,S3>) against the float32 form on ONE box: the level tests, then the step both ways (PNPP_WSX3=0: float32 MFMA)
set -e -o pipefail
mkdir -p gpurun_out/wx3
timeout -k 10 500 python -m pytest tests/test_gpu_levels_routed.py tests/test_gpu_split_products.py tests/test_gpu_sa.py -x -q -m gpu > gpurun_out/wx3/tests.log 2>&1 || { tail -40 gpurun_out/wx3/tests.log; exit 1; }
tail -3 gpurun_out/wx3/tests.log
for i in 1 2; do
for v in 0 1; do
PNPP_WSX3=$v PNPP_BENCH_DUMP=gpurun_out/wx3/table_${v}_$i.txt timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-bf16-variant --no-mfma-variant > gpurun_out/wx3/bench_${v}_$i.json 2>/dev/null
done; done
grep -H "gemm_wsx" gpurun_out/wx3/table_[01]_*.txt
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/wx3/bench_[01]_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
PY
