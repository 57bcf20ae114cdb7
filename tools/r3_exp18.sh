#!/bin/bash
# round 3, GPU call 18: k_stream's per-partition key counts by a vote of the wave (-DITX_VOTE_PC) against one LDS add per key
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3x
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
# (the variants' code was taken out again after this measurement: profiles/r03_stream_variants.txt; the flags it was built with)
declare -A FLAGS=([vote_pc]="-DITX_VOTE_PC" [sparse_rows]="-DITX_SPARSE_ROWS" [vote_sparse]="-DITX_SPARSE_ROWS -DITX_VOTE_PC")
for v in "$@"; do bash tools/build_variant.sh $v ${FLAGS[$v]} > $O/build_$v.txt 2>&1; done
libs=""; for v in "$@"; do libs="$libs tools/$v.so"; done
timeout -k 10 600 python tools/stream_measure.py 500000000 10 "" $libs > $O/stream_measure.txt 2>&1
cat $O/stream_measure.txt
for v in "$@"; do
  ITX_LIB=$PWD/tools/$v.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_$v.txt 2>&1
  echo "parity $v rc $?"; tail -2 $O/pytest_$v.txt
done
