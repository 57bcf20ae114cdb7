#!/bin/bash
# round 3, GPU call 12: pushes in flight now that the reader keeps ahead, 500 M reads of hiseq content
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3p
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 1000 python tools/ab_cli.py 500000000 100 2 \
  l6:ITX_PUSHES=6 \
  l8:ITX_PUSHES=8 \
  l8c256:ITX_PUSHES=8,ITX_BGZF_CHUNK=268435456,ITX_RESERVE_BLOCKS=8192 \
  > $O/cli_hiseq_500M.json 2> $O/cli_hiseq_500M.err
echo "rc $?"; tail -3 $O/cli_hiseq_500M.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3p/cli_hiseq_500M.json"))
for k in d["walls_s"]:
    print(k, d["walls_s"][k], d["scan_s"][k], [l for l in d["notes"][k] if "device decoder" in l or "record loop" in l or "BAM decode" in l or "HIP runtime" in l])
PY
