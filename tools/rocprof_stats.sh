#!/bin/bash
# kernel-trace + stats of one bench run; writes gpurun_out/rocprof_stats/*.csv and a short per-kernel table
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/rocprof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rocprof_stats -- python3 bench.py --steps 5 --warmup 2 --cpu-samples 0 --batched-scenes 0 --no-roofline --no-configs > gpurun_out/rocprof_stats.log 2>&1
f=$(find gpurun_out/rocprof_stats -name "*kernel_stats.csv" | head -1)
echo "stats file: $f"
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print("%-120s calls %6s total_ms %9.3f avg_us %9.2f pct %5s" % (r['Name'][:120], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
