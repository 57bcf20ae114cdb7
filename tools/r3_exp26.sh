#!/bin/bash
# round 3, GPU call 26: chunk buffers as huge-page mappings, touched and registered (25 ms per 384 MB) against hipHostMalloc (62 ms, under the runtime's lock)
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3kk
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_bamwin.py tests/test_cli_golden.py -x -q > $O/pytest.txt 2>&1
echo "pytest rc $?"; tail -3 $O/pytest.txt
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 1000 python tools/ab_cli.py 500000000 100 5 \
  plain:ITX_PIN_PLAIN=1 \
  > $O/cli_hiseq_500M.json 2> $O/cli_hiseq_500M.err
echo "rc $?"; tail -3 $O/cli_hiseq_500M.err
python - <<'PY'
import json, statistics
d = json.load(open("gpurun_out/r3kk/cli_hiseq_500M.json"))
print(d["same_outputs_as_base"])
for k in d["walls_s"]:
    w = d["walls_s"][k]
    print(k, w, "median", statistics.median(w), "mean", round(sum(w) / len(w), 3), [l for l in d["notes"][k] if "HIP runtime" in l or "record loop" in l or "parsed ahead" in l or "load " in l])
PY
