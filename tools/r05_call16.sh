#!/bin/bash
# sixteenth GPU call of round 5: (a) plan / scan with eight reads per thread on batches of 3 M reads and more: parity on long
# batches (tools/scale_check.py: configs[2] full size = PER 4, configs[4] = PER 8; every read against the oracle), kernel stats of configs[4];
# (b) configs[4] as the config says through the CLI at FULL size — 5 M-read FASTQ + 500 M-overlap PAF -> report + scrubb, the
# whole output checked (report, totals, windows) — and the same report from two engines on this device (37 GB of text)
out=gpurun_out/r05q; mkdir -p $out
timeout 1500 python tools/scale_check.py 3 5 > $out/scale_check.log 2>&1; tail -4 $out/scale_check.log | cut -c1-300
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/stats -o s -- python /root/repo/bench.py --config 4 --no-extras --no-cpu-baseline > /root/repo/$out/stats.log 2>&1 )
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_configs4_per8.csv \; ; rm -rf $out/stats; head -8 $out/kernel_stats_configs4_per8.csv | cut -c1-160; tail -c 600 $out/stats.log
timeout 2400 python tools/e2e_scrubb_full.py > $out/e2e_scrubb_full.log 2>&1; cat $out/e2e_scrubb_full.log | cut -c1-400
