#!/bin/bash
# The rocprofv3 evidence behind bench.py's `roofline` (run on the GPU box; summaries land in gpurun_out/prof_${ITX_PROF_ROUND:-r03}, the ones to
# keep are copied to profiles/ by hand): per-kernel times (--kernel-trace --stats), then HBM traffic from the PMC counters in
# two separate passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass; MI355X_MICROARCH.md, HBM section).
set -e
cd "$GRAFT_REPO_ROOT"
out=$PWD/gpurun_out/prof_${ITX_PROF_ROUND:-r03}
mkdir -p "$out"
export TMPDIR=/tmp
args="bench.py --steps 1 --warmup 0 --cpu-reads 0 --filter-steps 1 --replay-steps 10 --no-replay-check --no-settle"
python3 $args > "$out/bench_plain.json" 2> "$out/bench_plain.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- python3 $args > "$out/bench_stats.json" 2> "$out/bench_stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o bench -- python3 $args > "$out/bench_fetch.json" 2> "$out/bench_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -o bench -- python3 $args > "$out/bench_write.json" 2> "$out/bench_write.err"
python3 tools/profile_digest.py "$out" > "$out/digest.json"
find "$out" -name "*.csv" -size +3M -delete
cat "$out/digest.json"
