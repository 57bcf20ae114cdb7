#!/bin/bash
# round 3, GPU call 9: -R on the device (tests), fixed costs in detail, lanes / read slices
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3l
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_dedup.py tests/test_gpu_bamwin.py tests/test_gpu_inflate.py -x -q > $O/pytest.txt 2>&1
echo "pytest rc $?"; tail -15 $O/pytest.txt
bash tools/r3_exp8.sh
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 900 python tools/ab_cli.py 200000000 100 3 \
  p12:ITX_READ_PARTS=12 \
  l6:ITX_PUSHES=6 \
  l6p12:ITX_PUSHES=6,ITX_READ_PARTS=12 \
  l8p12:ITX_PUSHES=8,ITX_READ_PARTS=12 \
  > $O/cli_hiseq_200M.json 2> $O/cli_hiseq_200M.err
echo "rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3l/cli_hiseq_200M.json"))
for k in d["walls_s"]:
    print(k, d["walls_s"][k], d["scan_s"][k], [l for l in d["notes"][k] if "device decoder" in l or "record loop" in l or "BAM decode" in l])
PY
