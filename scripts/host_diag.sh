# what the host side of a GPU box looks like (round 6: why the same code reads 755 it/s on one box and 800 on another)
echo "== nproc $(nproc)  affinity $(taskset -pc $$ 2>/dev/null | sed 's/.*: //')"
lscpu | grep -E "Model name|Socket|NUMA|Thread|Core|MHz|L3" 
echo "== cgroup"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
echo "== gpu pci / numa"
for d in /sys/class/drm/card*/device; do
  [ -f $d/vendor ] || continue
  v=$(cat $d/vendor); [ "$v" = "0x1002" ] || continue
  echo "$d -> $(readlink -f $d | sed 's|.*/||') numa_node=$(cat $d/numa_node 2>/dev/null) local_cpulist=$(cat $d/local_cpulist 2>/dev/null)"
done
rocm-smi --showtopo 2>/dev/null | head -30
cat /sys/devices/system/cpu/cpu0/cpufreq/scaling_governor 2>/dev/null
python - <<'PY'
import os
print("sched_getaffinity", sorted(os.sched_getaffinity(0)))
PY
