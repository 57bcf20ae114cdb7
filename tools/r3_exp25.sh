#!/bin/bash
# round 3, GPU call 25: pass 1's tables with a bank per lane (no LDS bank conflicts for the byte tables) — parity, kernels, the command
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3hh
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_bamwin.py tests/test_cli_golden.py -x -q > $O/pytest.txt 2>&1
echo "pytest rc $?"; tail -3 $O/pytest.txt
for c in legacy hiseq novaseq; do
  timeout -k 10 300 python tools/inflate_measure.py 8000000 100 2 content=$c cigar=mixed > $O/${c}_8M.txt 2>&1
  grep -v "^wrote\|^call" $O/${c}_8M.txt | tail -3
done
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 1000 python tools/ab_cli.py 500000000 100 4 > $O/cli_hiseq_500M.json 2> $O/cli_hiseq_500M.err
echo "rc $?"
python - <<'PY'
import json, statistics
d = json.load(open("gpurun_out/r3hh/cli_hiseq_500M.json"))
for k in d["walls_s"]:
    w = d["walls_s"][k]
    print(k, w, "median", statistics.median(w), [l for l in d["notes"][k] if "device decoder" in l or "record loop" in l])
PY
