#!/bin/bash
# thirty-fifth GPU call of round 6 (host only): tools/out_probe.cc — ways for N threads to put bytes into ONE output file
out=gpurun_out/r06I; mkdir -p $out
g++ -O2 -pthread tools/out_probe.cc -o /tmp/out_probe || exit 1
nproc > $out/out_probe.log; cat /sys/fs/cgroup/cpu.max >> $out/out_probe.log 2>&1; df -h /dev/shm /tmp >> $out/out_probe.log
cat /sys/kernel/mm/transparent_hugepage/shmem_enabled >> $out/out_probe.log
{
echo "== /dev/shm, warm-up pass"; timeout 300 /tmp/out_probe /dev/shm 16 8 pwrite populate
for t in 1 2 4 8 16; do timeout 300 /tmp/out_probe /dev/shm 16 $t pwrite map populate; done
echo "== copy_file_range, /dev/shm"; timeout 120 /tmp/out_probe /dev/shm 2 4 cfr
echo "== /tmp (disk)"; for t in 1 4 16; do timeout 300 /tmp/out_probe /tmp 8 $t pwrite populate; done
timeout 120 /tmp/out_probe /tmp 2 4 cfr
} 2>&1 | tee -a $out/out_probe.log
