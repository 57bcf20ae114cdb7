#!/bin/bash
# the GPU suite as the driver runs it, output kept
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${1:-suite}
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
echo "rc $?"
tail -8 $O/pytest_gpu.txt
