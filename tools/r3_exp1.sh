#!/bin/bash
# round 3, first GPU call: the decoder on the three content modes, how pass 1 scales with blocks per launch, what its stores cost
set -e
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3a
mkdir -p $O
export OMP_NUM_THREADS=16
for n in 4000000 8000000 16000000; do
  timeout -k 10 300 python tools/inflate_measure.py $n 100 2 > $O/legacy_$n.txt 2>&1 || true
  tail -2 $O/legacy_$n.txt
done
for n in 4000000 8000000 14000000; do
  timeout -k 10 300 python tools/inflate_measure.py $n 100 2 content=hiseq cigar=mixed > $O/hiseq_$n.txt 2>&1 || true
  tail -2 $O/hiseq_$n.txt
done
timeout -k 10 300 python tools/inflate_measure.py 8000000 100 2 content=novaseq cigar=mixed > $O/novaseq_8000000.txt 2>&1 || true
tail -2 $O/novaseq_8000000.txt
ITX_LIB=$PWD/tools/exp_nostore.so ITX_MEASURE_NOCHECK=1 timeout -k 10 300 python tools/inflate_measure.py 8000000 100 2 > $O/nostore_legacy.txt 2>&1 || true
tail -1 $O/nostore_legacy.txt
ITX_LIB=$PWD/tools/exp_nostore.so ITX_MEASURE_NOCHECK=1 timeout -k 10 300 python tools/inflate_measure.py 8000000 100 2 content=hiseq cigar=mixed > $O/nostore_hiseq.txt 2>&1 || true
tail -1 $O/nostore_hiseq.txt
# the whole command on 100 M reads of each content (3 runs each)
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 600 python tools/ab_cli.py 100000000 100 3 > $O/cli_hiseq_100M.json 2> $O/cli_hiseq_100M.err || true
tail -c 1500 $O/cli_hiseq_100M.json
