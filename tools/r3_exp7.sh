#!/bin/bash
# round 3, GPU call 7: what pass 1's stores cost (every refill waits with vmcnt(0), i.e. for the stores issued since the look-ahead load)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3g
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
bash tools/build_variant.sh exp_nostore -DITXI_EXP_NOSTORE > $O/build_variant.txt 2>&1
for c in legacy hiseq; do
  ITX_LIB=$PWD/tools/exp_nostore.so ITX_MEASURE_NOCHECK=1 timeout -k 10 300 python tools/inflate_measure.py 8000000 100 2 content=$c cigar=mixed > $O/nostore_$c.txt 2>&1
  tail -1 $O/nostore_$c.txt
done
