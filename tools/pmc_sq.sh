#!/bin/bash
# dev helper: SQ counters per kernel (one rocprofv3 --pmc pass, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_sq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES \
  --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --steps 1 --warmup 1 --cpu-samples 0 --batched-scenes 0 --no-roofline > gpurun_out/pmc_sq.log 2>&1
find gpurun_out/pmc_sq -name "*.csv" | head
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_sq/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'][:110]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES': cnt[k] += 1
out = open('gpurun_out/pmc_sq_summary.txt', 'w')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES', 0)):
    wc = v.get('SQ_WAVE_CYCLES', 1) or 1
    line = "%-110s n=%4d wave_cyc=%.3e busy=%.3e wait_any=%.2f wait_inst=%.2f active=%.2f wait_lds=%.3f bankconf=%.3e mfma_busy=%.3e" % (
        k, cnt[k], wc, v.get('SQ_BUSY_CYCLES', 0), v.get('SQ_WAIT_ANY', 0) / wc, v.get('SQ_WAIT_INST_ANY', 0) / wc, v.get('SQ_ACTIVE_INST_ANY', 0) / wc,
        v.get('SQ_WAIT_INST_LDS', 0) / wc, v.get('SQ_LDS_BANK_CONFLICT', 0), v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0))
    print(line); out.write(line + "\n")
PY
