#!/bin/bash
# round 3, GPU call 4 (after the container was replaced): the whole command on 200 M reads of hiseq content — where the time goes, chunk sizes
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3d
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 1000 python tools/ab_cli.py 200000000 100 3 \
  c256:ITX_BGZF_CHUNK=268435456,ITX_RESERVE_BLOCKS=8192 \
  c512:ITX_BGZF_CHUNK=536870912,ITX_RESERVE_BLOCKS=16000 \
  p8:ITX_READ_PARTS=8 \
  > $O/cli_hiseq_200M.json 2> $O/cli_hiseq_200M.err
echo "rc $?"
tail -c 4000 $O/cli_hiseq_200M.json
