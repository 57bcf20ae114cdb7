#!/bin/bash
# round 3, GPU call 21: rehearsal of bench.py --gpus 2 on a one-GPU box (ranks share the card: gloo barriers, partials through files)
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3bb
mkdir -p $O
export OMP_NUM_THREADS=8
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
ITX_BENCH_SHARE_GPU=1 timeout -k 10 900 python bench.py --gpus 2 --steps 2 --warmup 1 --reads 40000000 --replay-reads 40000000 --replay-steps 3 > $O/bench_2ranks.json 2> $O/bench_2ranks.err
echo "rc $?"; tail -5 $O/bench_2ranks.err | cut -c1-300
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3bb/bench_2ranks.json"))
print(d["value"], d["n_gpus"], d["scaling"], d["ms_per_step"], d.get("strong_scaling"), d["checks"])
print(d["config"]["workload"][:300])
PY
