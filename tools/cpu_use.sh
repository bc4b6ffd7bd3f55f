#!/bin/bash
# Development aid (GPU box): CPU time the cgroup used and how often it was throttled while a command ran.  bash tools/cpu_use.sh <cmd...>
S=/sys/fs/cgroup/cpu.stat
u0=$(awk '/^usage_usec/{print $2}' $S); t0=$(awk '/^nr_throttled/{print $2}' $S); p0=$(awk '/^nr_periods/{print $2}' $S); w0=$(date +%s.%N)
"$@"
u1=$(awk '/^usage_usec/{print $2}' $S); t1=$(awk '/^nr_throttled/{print $2}' $S); p1=$(awk '/^nr_periods/{print $2}' $S); w1=$(date +%s.%N)
python3 - <<P
print("cpu_use: wall %.1f s, cpu %.1f s = %.1f CPUs on average; %d of %d periods throttled" % ($w1 - $w0, ($u1 - $u0) / 1e6, ($u1 - $u0) / 1e6 / ($w1 - $w0), $t1 - $t0, $p1 - $p0))
P
