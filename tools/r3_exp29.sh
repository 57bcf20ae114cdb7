#!/bin/bash
# round 3, GPU call 29: pushes of 16 k blocks (512 MB chunks of this content) against 12 k (384 MB): the lanes are busy 36 ms per push whatever its size
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3oo
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 1000 python tools/ab_cli.py 500000000 100 5 \
  c512:ITX_BGZF_CHUNK=536870912,ITX_RESERVE_BLOCKS=16384 \
  c448:ITX_BGZF_CHUNK=469762048,ITX_RESERVE_BLOCKS=14336 \
  > $O/cli_hiseq_500M.json 2> $O/cli_hiseq_500M.err
echo "rc $?"; tail -3 $O/cli_hiseq_500M.err
python - <<'PY'
import json, statistics
d = json.load(open("gpurun_out/r3oo/cli_hiseq_500M.json"))
print(d["same_outputs_as_base"])
for k in d["walls_s"]:
    w = d["walls_s"][k]
    print(k, w, "median", statistics.median(w), "mean", round(sum(w) / len(w), 3), [l for l in d["notes"][k] if "device decoder" in l or "record loop" in l or "HIP runtime" in l])
PY
