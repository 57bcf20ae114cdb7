#!/bin/bash
# round 3, GPU call 27: where the helper thread's start-up goes with the huge-page chunk buffers (ITX_TIMING_SETUP, ITX_TIMING_ALLOC)
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3ll
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
ITX_AB_MKBAM="content=hiseq cigar=mixed" python tools/ab_cli.py 200000000 100 1 > $O/warm.json 2> $O/warm.err
WD=/tmp/itx_bench_r200000000_s100_t5500000_c0_hiseq_mixed
for k in 1 2 3; do
  mkdir -p $O/run$k; cd $O/run$k
  ITX_TIMING=1 ITX_TIMING_SETUP=1 ITX_TIMING_ALLOC=1 ITX_GPUS=1 $GRAFT_REPO_ROOT/iteres_amd/host/iteres stat -w -o out $WD/chrom.sizes $WD/rep.sizes $WD/rmsk.txt $WD/reads.bam 2> stderr.txt
  grep "itx alloc\|inflater setup\|HIP runtime\|table build\|load \|parsed ahead" stderr.txt | cut -c1-250
  echo ---
  cd $GRAFT_REPO_ROOT
done
