#!/bin/bash
# round 3, GPU call 40: the process placed on its GPU's memory node by itself (numa.c) against ITX_NUMA=0
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3ad
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
ls /dev/dri/ | tr '\n' ' '; echo
env | grep VISIBLE_DEVICES; ITX_NUMA_REPORT=1 iteres_amd/host/iteres stat 2>&1 | grep "itx numa"
timeout -k 10 300 python -m pytest tests/test_numa_placement.py -x -q -rs > $O/pytest.txt 2>&1 || { tail -20 $O/pytest.txt; exit 1; }
tail -4 $O/pytest.txt
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 900 python tools/ab_cli.py 500000000 100 6 \
  free:ITX_NUMA=0 \
  > $O/cli_hiseq_500M.json 2> $O/cli_hiseq_500M.err
echo "rc $?"; tail -3 $O/cli_hiseq_500M.err
python - <<'PY'
import json, statistics
d = json.load(open("gpurun_out/r3ad/cli_hiseq_500M.json"))
print(d["same_outputs_as_base"])
for k in d["walls_s"]:
    w = d["walls_s"][k]
    print(k, w, "median", statistics.median(w), "mean", round(sum(w) / len(w), 3), [l for l in d["notes"][k] if "record loop" in l or "load " in l])
PY
