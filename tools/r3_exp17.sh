#!/bin/bash
# round 3, GPU call 17: `stat -w -R` end to end on 100 M reads with pile-ups (exact duplicates by the thousand): the set on the device against the host's
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3w
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
ITX_AB_OPTS="-R" ITX_AB_MKBAM="content=hiseq cigar=mixed pileup=200" timeout -k 10 900 python tools/ab_cli.py 100000000 100 3 \
  host_set:ITX_HOST_DEDUP=1 \
  > $O/cli_R_100M.json 2> $O/cli_R_100M.err
echo "rc $?"; tail -3 $O/cli_R_100M.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3w/cli_R_100M.json"))
print(d["same_outputs_as_base"])
for k in d["walls_s"]:
    print(k, d["walls_s"][k], d["scan_s"][k], [l for l in d["notes"][k] if "-R on" in l or "record loop" in l or "stream of" in l])
PY
