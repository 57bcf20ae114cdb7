#!/bin/bash
# round 3, GPU call 39: the process kept on the processors of the memory node the GPU hangs off (ITX_CPUS; the page-locked chunk buffers are first touched there),
# on the other node's, and free to roam (as built)
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3ac
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
node=$(/opt/rocm/bin/rocm-smi --showtoponuma 2>/dev/null | grep "Numa Node:" | head -1 | awk '{print $NF}')
other=$((1 - node))
local_cpus=$(cut -d, -f1 /sys/devices/system/node/node$node/cpulist)
remote_cpus=$(cut -d, -f1 /sys/devices/system/node/node$other/cpulist)
echo "GPU on node $node: local $local_cpus, remote $remote_cpus"
for n in /sys/devices/system/node/node*; do echo "$n filepages $(grep FilePages $n/meminfo | awk '{print $4,$5}')"; done
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 900 python tools/ab_cli.py 500000000 100 5 \
  local:ITX_CPUS=$local_cpus \
  remote:ITX_CPUS=$remote_cpus \
  > $O/cli_hiseq_500M.json 2> $O/cli_hiseq_500M.err
echo "rc $?"; tail -3 $O/cli_hiseq_500M.err
for n in /sys/devices/system/node/node*; do echo "$n filepages $(grep FilePages $n/meminfo | awk '{print $4,$5}')"; done
python - <<'PY'
import json, statistics
d = json.load(open("gpurun_out/r3ac/cli_hiseq_500M.json"))
print(d["same_outputs_as_base"])
for k in d["walls_s"]:
    w = d["walls_s"][k]
    print(k, w, "median", statistics.median(w), "mean", round(sum(w) / len(w), 3), [l for l in d["notes"][k] if "device decoder" in l or "record loop" in l or "load " in l])
PY
