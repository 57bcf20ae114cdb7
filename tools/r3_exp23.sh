#!/bin/bash
# round 3, GPU call 23: chunk buffers page-locked early (all five) or two early + the rest by the reader; records parsed ahead or not — 5 runs each, interleaved, one box
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3ee
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_cli_multi.py -x -q -k "parsed_ahead" > $O/pytest.txt 2>&1
echo "pytest rc $?"; tail -3 $O/pytest.txt
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 1000 python tools/ab_cli.py 500000000 100 5 \
  pin2:ITX_PIN_EARLY=2 \
  pin2_nopre:ITX_PIN_EARLY=2,ITX_NO_PREFETCH=1 \
  pin5_nopre:ITX_NO_PREFETCH=1 \
  > $O/cli_hiseq_500M.json 2> $O/cli_hiseq_500M.err
echo "rc $?"; tail -3 $O/cli_hiseq_500M.err
python - <<'PY'
import json, statistics
d = json.load(open("gpurun_out/r3ee/cli_hiseq_500M.json"))
for k in d["walls_s"]:
    w = d["walls_s"][k]
    print(k, w, "median", statistics.median(w), "mean", round(sum(w) / len(w), 3), [l for l in d["notes"][k] if "parsed ahead" in l or "HIP runtime" in l or "record loop" in l])
PY
