#!/bin/bash
# round 3, GPU call 2: the whole command on hiseq content with bigger pushes (more blocks of pass 1 in flight)
set -e
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3b
mkdir -p $O
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 900 python tools/ab_cli.py 100000000 100 3 \
  c256:ITX_BGZF_CHUNK=268435456,ITX_RESERVE_BLOCKS=8192 \
  c512:ITX_BGZF_CHUNK=536870912,ITX_RESERVE_BLOCKS=16000 \
  c512w8:ITX_BGZF_CHUNK=536870912,ITX_RESERVE_BLOCKS=16000,ITX_RESERVE_WINDOWS=8 \
  > $O/cli_hiseq_100M_chunks.json 2> $O/cli_hiseq_100M_chunks.err || true
tail -c 3000 $O/cli_hiseq_100M_chunks.json
