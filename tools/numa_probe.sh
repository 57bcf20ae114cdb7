#!/bin/bash
# where the GPU sits and where this process may run / allocate (development aid)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/numa
{
echo "== nodes"; ls /sys/devices/system/node/ | grep node; cat /sys/devices/system/node/online
for n in /sys/devices/system/node/node*; do echo "$n cpulist $(cat $n/cpulist) mem $(grep MemTotal $n/meminfo | awk '{print $4,$5}') free $(grep MemFree $n/meminfo | awk '{print $4,$5}') filepages $(grep FilePages $n/meminfo | awk '{print $4,$5}')"; done
echo "== gpus"; for d in /sys/class/drm/card*/device; do echo "$d vendor $(cat $d/vendor 2>/dev/null) numa $(cat $d/numa_node 2>/dev/null) local_cpulist $(cat $d/local_cpulist 2>/dev/null) bdf $(basename $(readlink -f $d))"; done
echo "== self"; grep -E "Cpus_allowed_list|Mems_allowed_list" /proc/self/status; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null; cat /sys/fs/cgroup/cpuset.mems.effective 2>/dev/null
echo "== rocm-smi"; /opt/rocm/bin/rocm-smi --showtoponuma 2>/dev/null | head -20
python3 - <<'PY'
import os
print("affinity", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8], "...")
PY
which numactl taskset 2>/dev/null
} > gpurun_out/numa/probe.txt 2>&1
cat gpurun_out/numa/probe.txt
