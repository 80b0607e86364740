#!/bin/bash
# kernel stats of the small-call regime (tests/gpu_latency_probe.py: 2 trunk streams per call)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/rocprof_small
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rocprof_small -- python3 tests/gpu_latency_probe.py > gpurun_out/rocprof_small.log 2>&1
tail -4 gpurun_out/rocprof_small.log
f=$(find gpurun_out/rocprof_small -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-100s calls %6s total_ms %9.3f avg_us %8.2f pct %5s" % (r['Name'][:100], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
