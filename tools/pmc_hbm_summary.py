"""Aggregate rocprofv3 FETCH_SIZE / WRITE_SIZE counter_collection CSVs per kernel class (bench.py's class names).

usage: pmc_hbm_summary.py <fetch_dir> <write_dir> <out_prefix> [commit] [command]
FETCH_SIZE is doubled (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md section HBM); both counters are in KB.
"""
import collections
import csv
import glob
import json
import sys

# training steps / forwards in the profiled bench run: counted from the trace (one loss_kernel per training step, one
# prep_rotate_kernel per forward chain, two chains per forward)
TRAIN_STEPS = 4
FORWARDS = 9
FWD_ONLY = ("prep_rotate", "pack_weights", "pool0_kernel", "feat_kernel", "value_conv", "bn_update", "bn_stat", "FwdConvP<", "conv3x3_halo_fwd", "conv1x1_fwd_ws")


def targs(name, tmpl):
    """Template arguments of `tmpl<GemmCfg<...>, a, b, ...>` behind the tile configuration."""
    tail = name.split(tmpl + "<", 1)[1]
    tail = tail.split(">", 1)[1]              # behind GemmCfg<...>
    depth, out = 0, ""
    for ch in tail:
        if ch == "<": depth += 1
        if ch == ">":
            if depth == 0: break
            depth -= 1
        out += ch
    return [a.strip() for a in out.split(",") if a.strip()]


def kclass(name):
    if "conv1x1_fwd_ws_kernel" in name: return "conv1x1_fwd"
    if "conv1x1_wgrad_ws_kernel" in name: return "conv1x1_wgrad"
    if "conv3x3_halo_fwd" in name: return "conv3x3_fwd"
    if "conv3x3_halo_dgrad" in name: return "conv3x3_dgrad"
    if "conv3x3_halo_wgrad" in name: return "conv3x3_wgrad"
    if "reduce_partials" in name: return "elementwise"       # (one launch reduces a layer's 3x3 AND 1x1 partial tiles: bench.py profiles it as element-wise too)
    if "FwdConvP<" in name:
        a = targs(name, "FwdConvP")           # MODE, PREC, F32IO
        if a[0] == "0" and len(a) > 2 and a[2] == "true": return "head_conv0_fwd"
        return {"0": "conv1x1_fwd", "1": "conv3x3_fwd", "2": "transition_fwd", "3": "stem7x7_fwd", "4": "stem7x7_fwd"}.get(a[0], "conv1x1_fwd")
    if "BwdDataGroupP" in name: return "conv1x1_dgrad"
    if "BwdDataP<" in name:
        a = targs(name, "BwdDataP")           # SHIFT3, EMODE, AFF, PREC, F32IO
        if a[0] == "true": return "conv3x3_dgrad"
        return {"0": "head_conv0_dgrad", "1": "conv1x1_dgrad", "2": "transition_dgrad"}.get(a[1], "conv1x1_dgrad")
    if "BwdWeightP<" in name:
        a = targs(name, "BwdWeightP")         # BMODE, CMAP, PD, AFF, PREC, F32IO
        if a[0] == "0" and len(a) > 5 and a[5] == "true": return "head_conv0_wgrad"
        return {"0": "conv1x1_wgrad", "1": "conv3x3_wgrad", "2": "transition_wgrad", "3": "stem_wgrad", "4": "stem_wgrad"}.get(a[0], "conv1x1_wgrad")
    if "smg::" in name: return "elementwise"
    return None


def count_steps(d):
    global TRAIN_STEPS, FORWARDS
    losses = preps = 0
    for fn in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            losses += "loss_kernel" in r["Kernel_Name"]
            preps += "pack_weights_kernel" in r["Kernel_Name"]        # one per forward
    if losses and preps:
        TRAIN_STEPS, FORWARDS = losses, preps


def load(d, counter):
    tot = collections.defaultdict(float)
    n = collections.Counter()
    step = collections.defaultdict(float)          # bytes per training step (1 forward + 1 backward)
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != counter:
                continue
            k = kclass(r["Kernel_Name"])
            if k is None:
                continue
            b = float(r["Counter_Value"]) * 1024.0
            tot[k] += b
            n[k] += 1
            step[k] += b / (FORWARDS if any(f in r["Kernel_Name"] for f in FWD_ONLY) else TRAIN_STEPS)
    return tot, n, step


def main():
    count_steps(sys.argv[1])
    fetch, nf, fstep = load(sys.argv[1], "FETCH_SIZE")
    write, _, wstep = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0))):
        f = 2.0 * fetch[k] / nf[k]
        w = write.get(k, 0.0) / nf[k]
        out[k] = {"launches": nf[k], "fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "hbm_bytes_per_launch": f + w,
                  "gb_per_train_step": (2.0 * fstep[k] + wstep.get(k, 0.0)) / 1e9}
    total = sum(v["gb_per_train_step"] for v in out.values())
    dump = dict(out)
    dump["_meta"] = {"commit": sys.argv[4] if len(sys.argv) > 4 else "?", "command": sys.argv[5] if len(sys.argv) > 5 else "?",
                     "train_steps": TRAIN_STEPS, "forwards": FORWARDS, "total_gb_per_train_step": total}
    json.dump(dump, open(sys.argv[3] + ".json", "w"), indent=1)
    with open(sys.argv[3] + ".md", "w") as md:
        md.write("commit %s, `%s`\n\n" % (dump["_meta"]["commit"], dump["_meta"]["command"]))
        md.write("| kernel class | launches (%d train steps + %d forward sweeps) | FETCH_SIZE x2 (MB / launch) | WRITE_SIZE (MB / launch) | GB per training step |\n|---|---|---|---|---|\n" % (TRAIN_STEPS, FORWARDS - TRAIN_STEPS))
        for k, v in out.items():
            md.write("| %s | %d | %.1f | %.1f | %.2f |\n" % (k, v["launches"], v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6, v["gb_per_train_step"]))
        md.write("\ntotal %.1f GB per training step (one forward + one backward + Adam)\n" % sum(v["gb_per_train_step"] for v in out.values()))
    print(open(sys.argv[3] + ".md").read())


if __name__ == "__main__":
    main()
