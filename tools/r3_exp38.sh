#!/bin/bash
# round 3, GPU call 38: the reader's slices. Five lanes, a second stream for pass 2, faster pass 1: the record loop stays at 1.12 s. The reader's eight
# threads copy 56 GB out of the page cache in that time (6.4 GB/s each, about what one thread can do): 12 / 16 slices, alone and with five lanes
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3ab
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 900 python tools/ab_cli.py 500000000 100 5 \
  parts12:ITX_READ_PARTS=12 \
  parts16:ITX_READ_PARTS=16 \
  parts12_lanes5:ITX_READ_PARTS=12,ITX_LANES=5,GPU_MAX_HW_QUEUES=16 \
  > $O/cli_hiseq_500M.json 2> $O/cli_hiseq_500M.err
echo "rc $?"; tail -3 $O/cli_hiseq_500M.err
python - <<'PY'
import json, statistics
d = json.load(open("gpurun_out/r3ab/cli_hiseq_500M.json"))
print(d["same_outputs_as_base"])
for k in d["walls_s"]:
    w = d["walls_s"][k]
    print(k, w, "median", statistics.median(w), "mean", round(sum(w) / len(w), 3), [l for l in d["notes"][k] if "device decoder" in l or "record loop" in l or "BAM decode so far" in l])
PY
