#!/bin/bash
# round 3, GPU call 37: pass 2 on a second stream of its compute lane (ITX_PASS2_STREAM=1: a lane is busy pass 1 only; scratch per slot), with 12 / 16 hardware
# queues, and five lanes with 16 queues (do the extra streams share queues?)
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3zz
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 900 python tools/ab_cli.py 500000000 100 5 \
  p2s:ITX_PASS2_STREAM=1 \
  p2s_q16:ITX_PASS2_STREAM=1,GPU_MAX_HW_QUEUES=16 \
  lanes5_q16:ITX_LANES=5,GPU_MAX_HW_QUEUES=16 \
  > $O/cli_hiseq_500M.json 2> $O/cli_hiseq_500M.err
echo "rc $?"; tail -3 $O/cli_hiseq_500M.err
python - <<'PY'
import json, statistics
d = json.load(open("gpurun_out/r3zz/cli_hiseq_500M.json"))
print(d["same_outputs_as_base"])
for k in d["walls_s"]:
    w = d["walls_s"][k]
    print(k, w, "median", statistics.median(w), "mean", round(sum(w) / len(w), 3), [l for l in d["notes"][k] if "device decoder" in l or "record loop" in l])
PY
