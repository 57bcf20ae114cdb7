#!/bin/bash
# round 3, GPU call 11: the new tests (XA windows of several batches, -R through the command), then the bench line
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3o
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_xaveto.py tests/test_gpu_dedup.py -x -q > $O/pytest.txt 2>&1
echo "pytest rc $?"; tail -4 $O/pytest.txt
bash tools/r3_bench.sh r3o --steps 3 --warmup 1
