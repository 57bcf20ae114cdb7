#!/bin/bash
# round 3, GPU call 33: the two bigWig files written side by side — goldens, the command with / without, then the bench line
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3ss
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout -k 10 500 python -m pytest tests/test_cli_golden.py tests/test_bigwig_writer.py -x -q > $O/pytest.txt 2>&1 || { tail -20 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 600 python tools/ab_cli.py 500000000 100 5 \
  bw_serial:ITX_BW_SERIAL=1 \
  > $O/cli_hiseq_500M.json 2> $O/cli_hiseq_500M.err
echo "rc $?"; tail -3 $O/cli_hiseq_500M.err
python - <<'PY'
import json, statistics
d = json.load(open("gpurun_out/r3ss/cli_hiseq_500M.json"))
print(d["same_outputs_as_base"])
for k in d["walls_s"]:
    w = d["walls_s"][k]
    print(k, w, "median", statistics.median(w), "mean", round(sum(w) / len(w), 3), [l for l in d["notes"][k] if "bigWig" in l or "record loop" in l])
PY
bash tools/r3_bench.sh r3ss
