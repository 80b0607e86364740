"""MFMA utilisation and wave-state split per kernel from one rocprofv3 --pmc pass (SQ_* counters, kernel-trace only).

usage: pmc_sq_summary.py <pmc_dir> <out_prefix>
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x shader clock x 1024 SIMDs): the counter ticks one cycle
per SIMD per busy matrix-pipe cycle (32 per v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md), so its ceiling is every SIMD busy
for the whole dispatch.  Under PMC collection the dispatches are serialised (no overlap between the two backward streams),
which is the per-kernel view wanted here.  wait_any / wait_inst / active are fractions of SQ_WAVE_CYCLES (disjoint states).
"""
import collections
import csv
import glob
import json
import sys

CLOCK_GHZ = 2.4        # shader clock under load (GRBM_GUI_ACTIVE / duration, profiles/README.md)
N_SIMD = 256 * 4


def main():
    d, out = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = {}
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].replace("smg::", "")
            k = k.split("(")[0][:140]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen:
                seen[key] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    dur = collections.defaultdict(float)
    n = collections.Counter()
    for (k, _), ns in seen.items():
        dur[k] += ns
        n[k] += 1
    rows = []
    for k, v in agg.items():
        wc = v.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        util = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(dur[k] * CLOCK_GHZ * N_SIMD, 1.0)
        rows.append({"kernel": k, "dispatches": n[k], "total_ms": dur[k] / 1e6, "mfma_util": util,
                     "wait_any": v.get("SQ_WAIT_ANY", 0) / wc, "wait_inst": v.get("SQ_WAIT_INST_ANY", 0) / wc,
                     "active": v.get("SQ_ACTIVE_INST_ANY", 0) / wc, "wait_lds": v.get("SQ_WAIT_INST_LDS", 0) / wc,
                     "lds_bank_conflict_cycles": v.get("SQ_LDS_BANK_CONFLICT", 0)})
    rows.sort(key=lambda r: -r["total_ms"])
    json.dump(rows, open(out + ".json", "w"), indent=1)
    with open(out + ".md", "w") as md:
        md.write("| kernel | dispatches | total ms | MFMA util | wave parked (wait_any) | issue stall (wait_inst) | issuing (active) | LDS issue stall |\n|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            if r["total_ms"] < 0.05:
                continue
            md.write("| `%s` | %d | %.2f | %.1f %% | %.2f | %.2f | %.2f | %.3f |\n" % (
                r["kernel"], r["dispatches"], r["total_ms"], 100 * r["mfma_util"], r["wait_any"], r["wait_inst"], r["active"], r["wait_lds"]))
    print(open(out + ".md").read())


if __name__ == "__main__":
    main()
