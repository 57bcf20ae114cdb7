#!/bin/bash
# round 3, GPU call 15: one copy stream + four compute lanes + eight slots, header by the host reader, buffers pinned beside the streams
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3s
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_bamwin.py tests/test_cli_golden.py tests/test_cli_multi.py -x -q > $O/pytest.txt 2>&1
echo "pytest rc $?"; tail -4 $O/pytest.txt
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 1000 python tools/ab_cli.py 500000000 100 3 \
  s4l4:ITX_PUSHES=4,ITX_LANES=4 \
  q16:GPU_MAX_HW_QUEUES=16 \
  s6l4:ITX_PUSHES=6 \
  > $O/cli_hiseq_500M.json 2> $O/cli_hiseq_500M.err
echo "rc $?"; tail -3 $O/cli_hiseq_500M.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3s/cli_hiseq_500M.json"))
for k in d["walls_s"]:
    print(k, d["walls_s"][k], d["scan_s"][k], [l for l in d["notes"][k] if "device decoder" in l or "record loop" in l or "BAM decode" in l or "HIP runtime" in l or "table build" in l])
PY
WD=/tmp/itx_bench_r500000000_s100_t5500000_c0_hiseq_mixed
timeout -k 10 300 bash tools/trace_cli.sh $WD $O/trace > $O/trace.log 2>&1
python tools/trace_summary.py $O/trace > $O/trace_summary.txt 2>&1
grep "itx timing" $O/trace/stderr.txt >> $O/trace_summary.txt
cat $O/trace_summary.txt
find $O -name "*.csv" -size +2M -delete
