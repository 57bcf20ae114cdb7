#!/bin/bash
# round 3, GPU call 28: the decoder's device memory released right after the stream (default) or kept to the end (ITX_KEEP_DECODER=1): 10 back-to-back runs each
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3mm
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_cli_golden.py tests/test_cli_multi.py -x -q > $O/pytest.txt 2>&1
echo "pytest rc $?"; tail -3 $O/pytest.txt
timeout -k 10 500 python bench.py --steps 10 --warmup 2 --cpu-reads 0 --filter-steps 0 --replay-steps 0 > $O/release.json 2> $O/release.err
ITX_KEEP_DECODER=1 timeout -k 10 500 python bench.py --steps 10 --warmup 2 --cpu-reads 0 --filter-steps 0 --replay-steps 0 > $O/keep.json 2> $O/keep.err
python - <<'PY'
import json
for k in ("release", "keep"):
    d = json.load(open(f"gpurun_out/r3mm/{k}.json"))
    print(k, d["value"], d["step_wall_s"]["each"], [l for l in d["phases_last_step"] if "released" in l or "device decoder" in l])
PY
