#!/bin/bash
# round 3, GPU call 30: pass 1 with 26 KB of LDS per wave (six waves to a CU, was 40 KB and four): parity, the decoder alone, then the
# whole command with 4 / 6 compute lanes
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3pp
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout -k 10 400 python -m pytest tests/test_gpu_inflate.py -x -q -m gpu > $O/pytest_inflate.txt 2>&1 || { tail -20 $O/pytest_inflate.txt; exit 1; }
tail -2 $O/pytest_inflate.txt
timeout -k 10 200 python tools/inflate_measure.py 4000000 100 3 content=hiseq cigar=mixed > $O/inflate_hiseq.txt 2>&1 && tail -4 $O/inflate_hiseq.txt
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 900 python tools/ab_cli.py 500000000 100 5 \
  lanes6:ITX_LANES=6 \
  lanes8:ITX_LANES=8 \
  > $O/cli_hiseq_500M.json 2> $O/cli_hiseq_500M.err
echo "rc $?"; tail -3 $O/cli_hiseq_500M.err
python - <<'PY'
import json, statistics
d = json.load(open("gpurun_out/r3pp/cli_hiseq_500M.json"))
print(d["same_outputs_as_base"])
for k in d["walls_s"]:
    w = d["walls_s"][k]
    print(k, w, "median", statistics.median(w), "mean", round(sum(w) / len(w), 3), [l for l in d["notes"][k] if "device decoder" in l or "record loop" in l])
PY
