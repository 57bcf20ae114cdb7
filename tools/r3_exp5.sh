#!/bin/bash
# round 3, GPU call 5: push sizes by block count (pass 1 wants ~64 k blocks in flight), ring of 8 windows
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3e
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 1000 python tools/ab_cli.py 200000000 100 3 \
  c256:ITX_BGZF_CHUNK=268435456,ITX_RESERVE_BLOCKS=8192 \
  c384:ITX_BGZF_CHUNK=402653184,ITX_RESERVE_BLOCKS=12288 \
  c480:ITX_BGZF_CHUNK=503316480,ITX_RESERVE_BLOCKS=16384 \
  c480w8:ITX_BGZF_CHUNK=503316480,ITX_RESERVE_BLOCKS=16384,ITX_RESERVE_WINDOWS=8 \
  c256w8:ITX_BGZF_CHUNK=268435456,ITX_RESERVE_BLOCKS=8192,ITX_RESERVE_WINDOWS=8 \
  c256w16:ITX_BGZF_CHUNK=268435456,ITX_RESERVE_BLOCKS=8192,ITX_RESERVE_WINDOWS=16 \
  > $O/cli_hiseq_200M.json 2> $O/cli_hiseq_200M.err
echo "rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3e/cli_hiseq_200M.json"))
for k in d["walls_s"]:
    print(k, d["walls_s"][k], d["scan_s"][k])
PY
