#!/bin/bash
# round 3, GPU call 20: do idle OpenMP workers (spinning between regions) cost the 16-core share anything? OMP_WAIT_POLICY=passive, GOMP_SPINCOUNT=0
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3aa
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 1000 python tools/ab_cli.py 500000000 100 4 \
  passive:OMP_WAIT_POLICY=passive \
  spin0:GOMP_SPINCOUNT=0 \
  > $O/cli_hiseq_500M.json 2> $O/cli_hiseq_500M.err
echo "rc $?"; tail -3 $O/cli_hiseq_500M.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3aa/cli_hiseq_500M.json"))
print(d["same_outputs_as_base"])
for k in d["walls_s"]:
    print(k, d["walls_s"][k], d["scan_s"][k], [l for l in d["notes"][k] if "record loop" in l or "table build" in l or "load " in l or "bigWig" in l])
PY
