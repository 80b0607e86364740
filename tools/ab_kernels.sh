#!/bin/bash
# dev helper: per-kernel average durations (rocprofv3 --kernel-trace --stats) of one bench run per library variant.
# usage: ab_kernels.sh lib1.so lib2.so ...   ("-" = the default build)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for L in "$@"; do
  d=gpurun_out/abk_$(basename $L .so); rm -rf $d
  if [ "$L" != "-" ]; then export SMG_HIP_LIB=$GRAFT_REPO_ROOT/$L; else unset SMG_HIP_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --steps 5 --warmup 2 --cpu-samples 0 --batched-scenes 0 --no-roofline --no-configs > $d.log 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== $L"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel time %.2f ms" % (tot / 1e6))
for r in rows[:22]:
    n = r['Name'].replace('smg::', '').replace('GemmCfg', 'Cfg')
    print("%-100s calls %5s total_ms %8.3f avg_us %8.2f" % (n[:100], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
done
