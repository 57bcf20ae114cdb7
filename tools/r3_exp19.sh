#!/bin/bash
# round 3, GPU call 19: the mapped file with explicit page-locking ahead (ITX_MMAP=1) against the copying reader
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3z
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
ITX_MMAP=1 ITX_BGZF_CHUNK=3000000 timeout -k 10 600 python -m pytest tests/test_cli_golden.py tests/test_gpu_dedup.py -x -q > $O/pytest_mmap.txt 2>&1
echo "pytest (ITX_MMAP=1) rc $?"; tail -3 $O/pytest_mmap.txt
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 1000 python tools/ab_cli.py 500000000 100 3 \
  mmap:ITX_MMAP=1 \
  > $O/cli_hiseq_500M.json 2> $O/cli_hiseq_500M.err
echo "rc $?"; tail -3 $O/cli_hiseq_500M.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3z/cli_hiseq_500M.json"))
print(d["same_outputs_as_base"])
for k in d["walls_s"]:
    print(k, d["walls_s"][k], d["scan_s"][k], [l for l in d["notes"][k] if "device decoder" in l or "record loop" in l or "BAM decode" in l or "HIP runtime" in l or "table build" in l or "load " in l])
PY
