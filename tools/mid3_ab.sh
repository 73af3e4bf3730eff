set -e -o pipefail
mkdir -p gpurun_out/mid3
timeout -k 10 400 python -m pytest tests/test_gpu_levels_routed.py tests/test_gpu_split_products.py -x -q -m gpu > gpurun_out/mid3/tests.log 2>&1 || { tail -30 gpurun_out/mid3/tests.log; exit 1; }
tail -3 gpurun_out/mid3/tests.log
for i in 1 2; do
PNPP_MID3=0 PNPP_BENCH_DUMP=gpurun_out/mid3/table_off_$i.txt timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-bf16-variant --no-mfma-variant > gpurun_out/mid3/bench_off_$i.json 2>/dev/null
PNPP_BENCH_DUMP=gpurun_out/mid3/table_on_$i.txt timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-bf16-variant --no-mfma-variant > gpurun_out/mid3/bench_on_$i.json 2>/dev/null
done
grep -h "gemm_mid" gpurun_out/mid3/table_*.txt
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/mid3/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
PY
