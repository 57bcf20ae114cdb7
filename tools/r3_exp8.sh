#!/bin/bash
# round 3, GPU call 8: where the fixed costs of the command go (table build, bigWig, start-up), 100 M reads
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3k
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
ITX_AB_MKBAM="content=hiseq cigar=mixed" ITX_TIMING_TABLE=1 ITX_TIMING_BW=1 ITX_TIMING_SETUP=1 timeout -k 10 600 python tools/ab_cli.py 100000000 100 2 > $O/cli.json 2> $O/cli.err
echo "rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3k/cli.json"))
print(d["walls_s"], d["scan_s"])
for l in d["notes"]["base"]:
    print("  ", l[:300])
PY
