#!/bin/bash
# round 3, GPU call 24: five or six compute lanes behind the eight slots (the record loop is bound by 4 lanes x 35 ms per push; the link needs 1.0 s); new -R test
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3gg
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_dedup.py -x -q -k "host_has_to_read" > $O/pytest.txt 2>&1
echo "pytest rc $?"; tail -3 $O/pytest.txt
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 1000 python tools/ab_cli.py 500000000 100 4 \
  l5:ITX_LANES=5 \
  l6:ITX_LANES=6 \
  > $O/cli_hiseq_500M.json 2> $O/cli_hiseq_500M.err
echo "rc $?"; tail -3 $O/cli_hiseq_500M.err
python - <<'PY'
import json, statistics
d = json.load(open("gpurun_out/r3gg/cli_hiseq_500M.json"))
print(d["same_outputs_as_base"])
for k in d["walls_s"]:
    w = d["walls_s"][k]
    print(k, w, "median", statistics.median(w), "mean", round(sum(w) / len(w), 3), [l for l in d["notes"][k] if "device decoder" in l or "record loop" in l or "stream of" in l])
PY
