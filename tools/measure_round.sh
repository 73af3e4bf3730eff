#!/bin/bash
# One measurement pass on the GPU box (run through gpurun): writes raw rocprofv3 output under gpurun_out/<tag>/ and the
# condensed summaries (the files that are committed) under gpurun_out/<tag>/profiles/.  Usage: tools/measure_round.sh <tag> <git-rev>
# The profiled program comes directly after `--` (python3 bench.py ...): no env/bash wrapper, and counter passes are
# separate runs with --pmc only (no trace domains beside them).
set -e -o pipefail
TAG=${1:-round2}
export PNPP_GIT_REV=${2:-unknown}
OUT=gpurun_out/$TAG
P=$OUT/profiles
mkdir -p $P
export TMPDIR=/tmp
PART=${PNPP_MEASURE_PART:-ABC}   # a gpurun call is limited to 20 minutes: A = counters + bench line + trace, B = index kernels + sweep + SQ counters, C = other configs and side tools
BENCH="bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-mfma-variant"
if [[ $PART == *A* ]]; then
echo "[1/8] FETCH_SIZE of the step"; rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-graph --no-bf16-variant --no-mfma-variant > $OUT/fetch.log 2>&1
echo "[2/8] WRITE_SIZE of the step"; rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-graph --no-bf16-variant --no-mfma-variant > $OUT/write.log 2>&1
python3 tools/summarize_rocprof.py pmc $OUT/fetch $OUT/write $P/pmc_traffic.json $P/${TAG}_pmc_traffic.csv
python3 tools/summarize_rocprof.py pmc-all $OUT/fetch $OUT/write $P/${TAG}_pmc_traffic_all_kernels.json $P/${TAG}_pmc_traffic_all_kernels.csv
cp $P/pmc_traffic.json profiles/pmc_traffic.json   # the bench line below reads the counters that belong to THESE kernel sources
echo "[3/8] bench line"; python3 bench.py --steps 200 --warmup 20 > $P/${TAG}_bench.json 2> $OUT/bench.err
PNPP_BENCH_DUMP=$P/${TAG}_kernel_table_events.txt python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-bf16-variant --no-mfma-variant > /dev/null 2>> $OUT/bench.err
PNPP_BENCH_DUMP=$P/${TAG}_kernel_table_events_mfma.txt python3 bench.py --f32-products mfma --steps 50 --warmup 10 --no-cpu-baseline > $P/${TAG}_bench_mfma.json 2>> $OUT/bench.err
PNPP_BENCH_DUMP=$P/${TAG}_kernel_table_events_bf16.txt python3 bench.py --precision bf16 --steps 50 --warmup 10 --no-cpu-baseline > $P/${TAG}_bench_bf16.json 2>> $OUT/bench.err
echo "[4/8] kernel trace of the step"; rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $BENCH --no-roofline --no-bf16-variant > $OUT/trace.log 2>&1
python3 tools/summarize_rocprof.py stats $OUT/trace $P/${TAG}_kernel_stats.csv
fi
if [[ $PART == *B* ]]; then
echo "[5/8] index kernels: events, trace, counters"
python3 tools/bench_index_kernels.py --json $P/${TAG}_index_kernels.json > $P/${TAG}_index_kernels.txt 2> $OUT/idx.err
rocprofv3 --kernel-trace --stats -d $OUT/idx_trace -o idx --output-format csv -- python3 tools/bench_index_kernels.py --reps 5 > $OUT/idx_trace.log 2>&1
python3 tools/summarize_rocprof.py stats $OUT/idx_trace $P/${TAG}_index_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE -d $OUT/idx_fetch -o f --output-format csv -- python3 tools/bench_index_kernels.py --reps 3 > $OUT/idx_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/idx_write -o w --output-format csv -- python3 tools/bench_index_kernels.py --reps 3 > $OUT/idx_write.log 2>&1
python3 tools/summarize_rocprof.py pmc-all $OUT/idx_fetch $OUT/idx_write $P/${TAG}_index_pmc_traffic.json $P/${TAG}_index_pmc_traffic.csv
echo "[6/8] batch sweep"
python3 tools/batch_sweep.py > $P/${TAG}_batch_sweep.json 2> $OUT/sweep.err
echo "[7/8] SQ counters of every kernel (two --pmc passes, no trace domains)"
PNPP_SQ_FILTER= bash tools/sq_counters.sh ${TAG}_sq > /dev/null 2>&1 || echo "sq counters failed"
{ echo "# SQ counters per kernel launch (rocprofv3 --pmc, two passes; tools/sq_counters.sh), kernel sources of $PNPP_GIT_REV"; cat gpurun_out/${TAG}_sq/sq_p1.txt; echo; cat gpurun_out/${TAG}_sq/sq_p2.txt; } > $P/${TAG}_sq_counters.txt
fi
if [[ $PART == *C* ]]; then
echo "[8/8] the other BASELINE configs and the side tools"
python3 tools/bench_config.py --config 2 > $P/${TAG}_bench_config2.json 2> $OUT/cfg2.err
python3 tools/bench_config.py --config 3 > $P/${TAG}_bench_config3.json 2> $OUT/cfg3.err
python3 tools/bench_point_transformer.py > $P/${TAG}_point_transformer_step.json 2> $OUT/pt.err || true
python3 tools/bench_simple_pointnet.py > $P/${TAG}_simple_pointnet_step.json 2> $OUT/simple.err || true
python3 tools/script_throughput.py > $P/${TAG}_script_throughput.json 2> $OUT/script.err || true
python3 tools/convergence.py > $P/${TAG}_convergence.json 2> $OUT/conv.err || true
fi
# raw directories can be large: keep only the summaries for the merge back
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/idx_trace $OUT/idx_fetch $OUT/idx_write
ls -la $P
