#!/bin/bash
# round 3, GPU call 22: records parsed ahead of the table (the helper thread's backlog) — tests, then against ITX_NO_PREFETCH=1 on 500 M reads
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3dd
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout -k 10 900 python -m pytest tests/test_cli_multi.py tests/test_cli_golden.py -x -q > $O/pytest.txt 2>&1
echo "pytest rc $?"; tail -4 $O/pytest.txt
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 1000 python tools/ab_cli.py 500000000 100 2 \
  noprefetch:ITX_NO_PREFETCH=1 \
  > $O/cli_hiseq_500M.json 2> $O/cli_hiseq_500M.err
echo "rc $?"; tail -3 $O/cli_hiseq_500M.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3dd/cli_hiseq_500M.json"))
print(d["same_outputs_as_base"])
for k in d["walls_s"]:
    print(k, d["walls_s"][k], d["scan_s"][k], [l for l in d["notes"][k] if "record loop" in l or "table build" in l or "load " in l or "parsed ahead" in l or "HIP runtime" in l])
PY
