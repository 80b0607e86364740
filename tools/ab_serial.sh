#!/bin/bash
# dev helper: serialised per-kernel totals (rocprofv3 --stats of bench.py --train-only --serialize) per library variant.
# usage: [LEG=config3] ab_serial.sh "grep pattern" lib1.so lib2.so ...   ("-" = the default build)
pat=$1; shift
LEGARG=${LEG:+--leg $LEG}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for L in "$@"; do
  d=gpurun_out/abs_$(basename $L .so); rm -rf $d
  if [ "$L" != "-" ]; then export SMG_HIP_LIB=$GRAFT_REPO_ROOT/$L; else unset SMG_HIP_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --train-only --steps 4 --warmup 1 --serialize $LEGARG > $d.log 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== $L"
  python3 - "$f" "$pat" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows if 'rocclr' not in r['Name'])
print("total kernel time %.3f ms per step" % (tot / 5e6))
for r in rows:
    if re.search(sys.argv[2], r['Name']):
        n = r['Name'].split('(')[0].replace('smg::', '').replace('GemmCfg', 'Cfg')
        print("%-110s calls %5s ms/step %8.3f avg_us %8.2f" % (n[:110], r['Calls'], float(r['TotalDurationNs'])/5e6, float(r['AverageNs'])/1e3))
PY
done
