#!/bin/bash
# tools/pmc_r02.sh <outdir> — kernel stats + PMC passes for bench.py's configs[1] step (run on the GPU
# box: gpurun -- 'bash tools/pmc_r02.sh gpurun_out/r02_final').  Counters in their own runs with
# --kernel-trace only, as the MI355X guide prescribes.
set -u
out=$(realpath -m "$1"); mkdir -p "$out"
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
args=(--no-cpu-baseline --no-extras --steps 100 --warmup 5)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o s -- python "$root/bench.py" "${args[@]}" > "$out/stats.log" 2>&1
run() { local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -o p -- \
      python "$root/bench.py" "${args[@]}" --engines 1 > "$out/$name.log" 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 "$root/tools/pmc_summary.py" "$out" > "$out/pmc_summary.txt" 2>&1
cp "$out/stats/s_kernel_stats.csv" "$out/kernel_stats.csv" 2>/dev/null
# the skewed profile (configs[3]) and configs[2]: kernel stats only
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/skew" -o s -- python "$root/bench.py" --profile skewed --reads 10000 --overlaps 30000000 --steps 20 --warmup 3 --engines 1 --no-extras --no-cpu-baseline > "$out/skew.log" 2>&1
cp "$out/skew/s_kernel_stats.csv" "$out/skew_kernel_stats.csv" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/sequel" -o s -- python "$root/bench.py" --profile sequel --reads 2000000 --overlaps 200000000 --steps 5 --warmup 2 --engines 1 --no-extras --no-cpu-baseline > "$out/sequel.log" 2>&1
cp "$out/sequel/s_kernel_stats.csv" "$out/sequel_kernel_stats.csv" 2>/dev/null
for f in fetch write; do
  timeout 900 rocprofv3 --kernel-trace --pmc $( [ $f = fetch ] && echo FETCH_SIZE || echo WRITE_SIZE ) --output-format csv -d "$out/sequel_$f" -o p -- \
      python "$root/bench.py" --profile sequel --reads 2000000 --overlaps 200000000 --steps 3 --warmup 1 --engines 1 --no-extras --no-cpu-baseline > "$out/sequel_$f.log" 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for tag in ("sequel_fetch", "sequel_write"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for p in glob.glob(os.path.join(out, tag, "*counter_collection.csv")):
        for r in csv.DictReader(open(p)):
            a = acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    with open(os.path.join(out, tag + "_summary.txt"), "w") as f:
        for (k, c), v in sorted(acc.items()):
            f.write("%-60s %-12s mean/dispatch %16.1f (rows %d)\n" % (k[:60], c, v[0] / v[1], v[1]))
PY
ls "$out"
