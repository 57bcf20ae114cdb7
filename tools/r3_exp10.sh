#!/bin/bash
# round 3, GPU call 10: lanes / read slices / host threads on 200 M reads of hiseq content; -R tests
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3m
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_dedup.py -x -q > $O/pytest.txt 2>&1
echo "pytest rc $?"; tail -5 $O/pytest.txt
nproc; python -c "import os; print(len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null
df -h /tmp | tail -1
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 900 python tools/ab_cli.py 200000000 100 3 \
  p12:ITX_READ_PARTS=12 \
  l6:ITX_PUSHES=6 \
  l6p12:ITX_PUSHES=6,ITX_READ_PARTS=12 \
  l8p12:ITX_PUSHES=8,ITX_READ_PARTS=12 \
  omp32:OMP_NUM_THREADS=32 \
  > $O/cli_hiseq_200M.json 2> $O/cli_hiseq_200M.err
echo "rc $?"; tail -5 $O/cli_hiseq_200M.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3m/cli_hiseq_200M.json"))
for k in d["walls_s"]:
    print(k, d["walls_s"][k], d["scan_s"][k], [l for l in d["notes"][k] if "device decoder" in l or "record loop" in l or "BAM decode" in l or "bigWig" in l])
PY
