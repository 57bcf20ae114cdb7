#!/bin/bash
# round 3: the driver's bench command (shortened) and a digest of its line
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${1:-r3i}
shift
mkdir -p $O
timeout -k 10 1100 python bench.py "$@" > $O/bench_line.json 2> $O/bench.err
echo rc $?
tail -c 600 $O/bench.err
python - $O <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench_line.json"))
print(d["value"], d["ms_per_step"], d["step_wall_s"], d["scan_only"])
for k in ("phases_last_step", "cpu_baseline", "cpu_baseline_mt", "filter_leg", "roofline", "checks", "vram_settle"):
    print(k, json.dumps(d.get(k))[:3000])
print(json.dumps(d["config"])[:2500])
PY
