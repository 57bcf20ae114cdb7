#!/bin/bash
# round 3, GPU call 32: the extra literal/length codes of a turn taken only when half the wave wants them (vote): decoder alone, then the whole GPU suite on the product build
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3rr
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
for L in 2 3; do
  for c in hiseq legacy novaseq; do
    ITX_LIB=$PWD/tools/vote$L.so timeout -k 10 200 python tools/inflate_measure.py 8000000 100 2 content=$c cigar=mixed > $O/vote${L}_${c}.txt 2>&1 || { tail -5 $O/vote${L}_${c}.txt; exit 1; }
    echo "vote LITS=$L $c: $(grep 'kernels only' $O/vote${L}_${c}.txt) $(grep -o 'equal zlib: [A-Za-z]*' $O/vote${L}_${c}.txt)"
  done
done
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
echo "pytest rc $?"; tail -8 $O/pytest_gpu.txt
