#!/bin/bash
# round 3, GPU call 13: what the device does during one command on the 500 M-read hiseq BAM (kernel + copy trace), 4 and 8 pushes in flight
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3q
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 600 python tools/ab_cli.py 500000000 100 1 > $O/warm.json 2> $O/warm.err
WD=/tmp/itx_bench_r500000000_s100_t5500000_c0_hiseq_mixed
timeout -k 10 300 bash tools/trace_cli.sh $WD $O/trace4 > $O/trace4.log 2>&1
python tools/trace_summary.py $O/trace4 > $O/trace4_summary.txt 2>&1
grep "itx timing" $O/trace4/stderr.txt >> $O/trace4_summary.txt
cat $O/trace4_summary.txt
timeout -k 10 300 bash tools/trace_cli.sh $WD $O/trace8 ITX_PUSHES=8 > $O/trace8.log 2>&1
python tools/trace_summary.py $O/trace8 > $O/trace8_summary.txt 2>&1
grep "itx timing" $O/trace8/stderr.txt >> $O/trace8_summary.txt
cat $O/trace8_summary.txt
find $O -name "*.csv" -size +2M -delete
