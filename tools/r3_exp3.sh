#!/bin/bash
# round 3, GPU call 3: 4-byte tokens in one region + pass 2 in waves of bytes: parity, then the kernels on the three contents
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3c
mkdir -p $O
export OMP_NUM_THREADS=16
timeout -k 10 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_bamwin.py -x -q > $O/pytest_inflate.txt 2>&1
echo "pytest rc $?"; tail -3 $O/pytest_inflate.txt
for c in legacy hiseq novaseq; do
  timeout -k 10 300 python tools/inflate_measure.py 8000000 100 2 content=$c cigar=mixed > $O/${c}_8M.txt 2>&1
  grep -v "^wrote\|^call" $O/${c}_8M.txt | tail -3
done
timeout -k 10 300 python tools/inflate_measure.py 14000000 100 2 content=hiseq cigar=mixed > $O/hiseq_14M.txt 2>&1
grep -v "^wrote\|^call" $O/hiseq_14M.txt | tail -3
