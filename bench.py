"""bench.py - affordance fwd+bwd passes/s on MI355X (BASELINE.json metric).

One STEP = one PASS per GPU = one synthetic 224x224 depth heightmap + one object mask,
R = 16 rotations, every rotation a training sample:
    16 forwards (reinforcement_net branch C, code/models.py:513-539; the masked stream's
    trunk pass de-duplicated: 17 DenseNet-121 passes instead of 32),
    16 Huber losses (code/trainer.py:345-348), backward of their sum,
    gradient all-reduce over RCCL when N > 1, ONE Adam step (code/trainer.py:383).
Inputs (heightmaps, labels, weights) are resident in HBM when the timed region starts.
fp32 in, fp32 out (the reference runs apex O0 = fp32 and parity is gated in fp32, SURVEY.md
section 7); the convolutions run on the fp16 matrix cores as scaled two-piece split products
(three MFMA terms) with fp32-class accuracy (csrc/gemm.cuh, operand kind 3).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "smg-multimodal-grasping_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import synthetic  # noqa: E402

R = 16
# Algorithmic work of one de-duplicated 16-rotation fwd+bwd pass (SURVEY.md 8d / BASELINE.md 4)
PASS_GFLOP = 2331.30
SWEEP_GFLOP = 788.02
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz (the fp32 roof the split scheme breaks)
PEAK_BF16_MFMA_TFLOPS = 2500.0        # dense bf16 MFMA peak (MI355X_MICROARCH.md)
SPLIT_TERMS = 3                       # v_mfma_f32_32x32x16_f16 per fp32 product of the hot classes (two-piece fp16 split, gemm.cuh operand kind 3): MFMA roof 2500 / 3
SPLIT_TERMS_PLAIN = 6                 # ... of the classes that stay on the three-piece bf16 split (operand kind 0)
PLAIN_CLASSES = ("stem7x7_fwd", "transition_wgrad", "transition_dgrad", "stem_wgrad", "head_conv0_wgrad", "head_conv0_dgrad")
PEAK_SPLIT_TFLOPS = PEAK_BF16_MFMA_TFLOPS / SPLIT_TERMS
PEAK_HBM_GBS = 8000.0                 # HBM3E peak (MI355X_MICROARCH.md; ~6300 achievable)
PMC_FILE = "pmc_r06_hbm_traffic.json"                     # tools/profile_round.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes
SERIAL_CSV = "rocprof_r06_kernel_stats_serialized.csv"    # rocprofv3 --kernel-trace --stats of `bench.py --train-only --serialize`
PASS5_GFLOP = 36767.6                 # S=1824, R=32 fwd+bwd pass, masked stream de-duplicated
# rocprofv3 kernel-name fragments that make up each class of roofline.per_kernel (profiles/*kernel_stats*.csv)
# (reduce_partials_kernel serves every weight gradient that goes through partial tiles - one launch per dense layer reduces its 3x3 AND
# 1x1 partial tiles - and carries no policy in its name: it is filed with the element-wise kernels, here, in the engine's live hipEvent
# classes and in tools/pmc_hbm_summary.py alike, so that a class's launches, time and bytes are those of its GEMM kernels in all three)
CLASS_SYMBOLS = {      # fnmatch patterns; template arguments: FwdConvP<Cfg, MODE, PREC, F32IO>, BwdDataP<Cfg, SHIFT3, EMODE, AFF, PREC, F32IO>,
                       # BwdDataGroupP<Cfg, PREC>, BwdWeightP<Cfg, BMODE, CMAP, PD, AFF, PREC, F32IO>, conv3x3_halo_*_kernel<tile, PREC>
    "stem7x7_fwd": ["FwdConvP<*>, 3, ?, false>", "FwdConvP<*>, 4, ?, false>"],
    "conv1x1_fwd": ["FwdConvP<*>, 0, ?, false>", "conv1x1_fwd_ws_kernel"], "head_conv0_fwd": ["FwdConvP<*>, 0, ?, true>"],
    "conv3x3_fwd": ["conv3x3_halo_fwd_kernel"], "transition_fwd": ["FwdConvP<*>, 2, ?, false>"],
    "conv3x3_dgrad": ["conv3x3_halo_dgrad_kernel"], "conv3x3_wgrad": ["conv3x3_halo_wgrad_kernel"],
    "conv1x1_dgrad": ["BwdDataGroupP<", "BwdDataP<*>, false, 1, false, ?, false>"], "conv1x1_wgrad": ["BwdWeightP<*>, 0, 0, 3, false, ?, false>", "conv1x1_wgrad_ws_kernel"],
    "transition_wgrad": ["BwdWeightP<*>, 2, 0, 1, true, ?, false>"], "transition_dgrad": ["BwdDataP<*>, false, 2, true, ?, false>"],
    "stem_wgrad": ["BwdWeightP<*>, 3, 2, 3, true, ?, false>", "BwdWeightP<*>, 4, 3, 3, true, ?, false>"],
    "head_conv0_wgrad": ["BwdWeightP<*>, 0, 0, 3, true, ?, true>"], "head_conv0_dgrad": ["BwdDataP<*>, false, 0, true, ?, true>"],
}


def csv_class_avg_ms(symbols):
    """Call-weighted mean launch duration (ms) of the kernels whose names match the class's fragments in the tracked
    serialised rocprofv3 stats CSV, plus the file's provenance line - so that roofline.frac can be recomputed from
    profiles/ alone.  None if the file is absent."""
    import csv
    import fnmatch
    path = os.path.join(REPO, "profiles", SERIAL_CSV)
    try:
        rows = list(csv.DictReader(open(path)))
    except OSError:
        return None
    calls = tot = 0.0
    for r in rows:
        nm = r["Name"]
        nm = nm.split("(")[0]                    # kernel name without its argument list (which repeats the policy type)
        if any(fnmatch.fnmatchcase(nm, "*" + frag + "*") for frag in symbols):
            calls += float(r["Calls"]); tot += float(r["TotalDurationNs"])
    if calls == 0:
        return None
    meta = {}
    try:
        meta = json.load(open(path.replace(".csv", ".meta.json")))
    except (OSError, ValueError):
        pass
    return {"file": "profiles/" + SERIAL_CSV, "avg_launch_ms": tot / calls / 1e6, "calls": int(calls), **meta}


def layout_names():
    """(name, shape, kind) for synthetic.make_state_dict, from the C library's layout table."""
    import smg_hip
    out = []
    for name, kind, off, shape in smg_hip.layout(1):
        if kind == 3:
            k = "nbt"
        elif kind == 1:
            k = "rm"
        elif kind == 2:
            k = "rv"
        elif "classifier.weight" in name:
            k = "fc_w"
        elif "classifier.bias" in name:
            k = "fc_b"
        elif len(shape) == 4:
            k = "conv"
        elif name.endswith(".weight"):
            k = "bn_w"
        else:
            k = "bn_b"
        out.append((name, shape, k))
    return out


def cpu_baseline(seed, n_samples, labels, thread_settings):
    """The oracle (PyTorch-CPU restatement of the reference) running the REFERENCE's
    schedule: batch-1 samples, masked stream recomputed for every sample, one Adam step per
    sample (code/trainer.py:338-383).  Bounded: n_samples of the 16 samples of a pass per thread
    setting; returns {threads: seconds per 16-sample pass}."""
    from oracle import affordance as orc
    sd = synthetic.make_state_dict(orc.state_layout(1), seed)
    net = orc.OracleNet(1)
    orc.load_numpy_state(net, sd)
    net.gnum_rotations = net.snum_rotations = R
    net.train()
    opt = orc.make_adam(net)
    depth, masks = synthetic.heightmap_scene(seed)
    x = orc.preprocess(depth, [0.01] * 3, [0.03] * 3)
    mx = orc.preprocess(depth * masks[0], [0.01] * 3, [0.03] * 3)
    out = {}
    for nt in thread_settings:              # probe: one warm-up + one timed sample per thread setting
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        orc.train_step(net, opt, x, mx, 0, 0, float(labels[0]))          # warm-up (oneDNN primitives, allocator, thread pool)
        t_first = time.perf_counter() - t0
        if t_first > 8.0:
            # a setting this slow (one thread per physical core of a 2-socket host: oneDNN oversubscribes itself) is measured by
            # its first sample alone - the one-off costs are a second of it - so that the default run stays within minutes
            out[nt] = t_first * R
            continue
        t0 = time.perf_counter()
        orc.train_step(net, opt, x, mx, 0, 1, float(labels[1]))
        out[nt] = (time.perf_counter() - t0) * R                         # seconds per 16-sample pass, from one sample
    # the reported figure: n_samples samples (default: the 16 of ONE FULL pass, no extrapolation) at the best setting found
    best = min(out, key=lambda k: out[k])
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    for r in range(n_samples):
        orc.train_step(net, opt, x, mx, 0, r % R, float(labels[r % R]))
    probe = dict(out)
    out[best] = (time.perf_counter() - t0) / n_samples * R               # seconds per 16-sample pass
    # forward-only sweep (config 2) on the reference's schedule at the best thread setting: 2 of the 16 rotations
    # (each = rotated stream + masked stream, no grad), extrapolated
    torch.set_num_threads(best)
    with torch.no_grad():
        orc.forward(net, x, mx, 0, True, 0)
        t0 = time.perf_counter()
        for r in (3, 9):
            orc.forward(net, x, mx, 0, True, r)
        sweep = (time.perf_counter() - t0) / 2 * R
    return out, sweep, probe


def kernel_roofline(step_fn, eng, dev, terms, ms_per_step, step_rl, n_prof=3, pmc=False):
    """Per-kernel-class roofline of `step_fn` from hipEvents recorded on the launch stream around every launch, with all
    launches serialised on one stream (smg_profile_enable; separate, untimed passes).
    MFMA roof: executed FLOPs x `terms` bf16 / fp16 MFMA terms per product / 2.5 PFLOP/s (6 for the fp32-class split, 1 for the
    16-bit modes).  HBM roof: ALGORITHMIC bytes (every operand read once, every result written once, at the storage width of the
    mode; the engine's BY() figures) / 8 TB/s.  The larger of the two times bounds the class; frac = that time / measured."""
    eng.profile_enable(True)
    for _ in range(n_prof):
        step_fn()
    torch.cuda.synchronize(dev)
    prof = eng.profile_read()
    stages = eng.profile_read_stages()
    eng.profile_enable(False)
    mfma_roof = PEAK_BF16_MFMA_TFLOPS / terms

    def terms_of(k):      # MFMA terms per product of class k: the classes on the three-piece bf16 split in the fp32-class mode pay six
        return SPLIT_TERMS_PLAIN if (terms == SPLIT_TERMS and k in PLAIN_CLASSES) else terms

    def roof(v, k=None):
        ms, n, fl, by = v
        tk = terms_of(k)
        t_mfma, t_hbm = fl * tk / (PEAK_BF16_MFMA_TFLOPS * 1e12), by / (PEAK_HBM_GBS * 1e9)
        bound = "mfma" if t_mfma >= t_hbm else "hbm"
        t = ms * 1e-3
        if bound == "mfma":
            r = {"bound": "mfma", "achieved": fl / t / 1e12, "peak": PEAK_BF16_MFMA_TFLOPS / tk, "unit": "TFLOP/s"}
        else:
            r = {"bound": "hbm", "achieved": by / t / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s"}
        r["frac"] = max(t_mfma, t_hbm) / t
        return r
    conv = {k: v for k, v in prof.items() if k != "elementwise" and v[1] > 0}
    dom = max(conv, key=lambda k: conv[k][0])
    ms, n, fl, by = conv[dom]
    conv_ms = sum(v[0] for v in conv.values())
    conv_fl = sum(v[2] for v in conv.values())
    # HBM bytes per launch of the dominant class from the committed PMC passes of this round (profiles/; rocprofv3
    # cannot run inside this process): bytes per training step there / launches per step HERE.  None if absent.
    traffic, traffic_src = None, None
    if pmc:
        try:
            with open(os.path.join(REPO, "profiles", PMC_FILE)) as f:
                pj = json.load(f)
            if dom in pj:
                traffic = pj[dom]["gb_per_train_step"] * 1e9 / (n / n_prof)
                meta = pj.get("_meta", {})
                traffic_src = ("profiles/%s: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `%s` (%s training steps, commit %s), "
                               "FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md); bytes per training step / launches per step"
                               % (PMC_FILE, meta.get("command", "?"), meta.get("train_steps", "?"), meta.get("commit", "?")))
        except (OSError, ValueError, KeyError):
            pass
    rl = roof(conv[dom], dom)
    rl.update({
        "kernel": dom, "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_src,
        "algorithmic_bytes_per_launch": by / n, "flops_per_launch": fl / n,
        "avg_launch_ms": ms / n, "launches_per_step": n // n_prof,
        "kernel_symbols": CLASS_SYMBOLS.get(dom, []),        # rocprofv3 kernel names that make up `kernel`
        "timing": "hipEvents around every launch of the class, all launches serialised on one stream (smg_profile_enable), %d training steps" % n_prof,
        "rocprof_serialized": csv_class_avg_ms(CLASS_SYMBOLS.get(dom, [])) if pmc else None,   # the same class in the tracked serialised rocprofv3 CSV
        "mfma_terms": terms,
        "arithmetic": ("fp32 in / fp32 out; every product of the dense layers = 3 v_mfma_f32_32x32x16_f16 terms of a scaled 2-piece fp16 split "
                       "(fp32-class accuracy, tools/split16_probe.hip): MFMA roof %.1f TFLOP/s fp32-equivalent; stem / transition and head "
                       "gradients: 6 bf16 terms" % mfma_roof) if terms > 1 else
                      "16-bit storage of activations and gradients, one 16-bit MFMA term per product: MFMA roof %.0f TFLOP/s, bytes counted at 2 B / element" % mfma_roof,
        "all_conv_kernels": {"achieved": conv_fl / (conv_ms * 1e-3) / 1e12, "frac_of_mfma_roof": conv_fl / (conv_ms * 1e-3) / 1e12 / mfma_roof,
                             "frac_of_fp32_mfma_peak": conv_fl / (conv_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                             "ms_per_step": conv_ms / n_prof, "executed_gflop_per_step": conv_fl / n_prof / 1e9},
        "elementwise_ms_per_step": prof["elementwise"][0] / n_prof,
        "launches_per_step_all": sum(v[1] for v in prof.values()) // n_prof,
        "per_kernel": {k: dict(roof(v, k), ms_per_step=v[0] / n_prof, launches_per_step=v[1] // n_prof,
                               tflops=(v[2] / (v[0] * 1e-3) / 1e12) if v[0] > 0 else 0.0,
                               algorithmic_gb_per_step=v[3] / n_prof / 1e9) for k, v in prof.items() if v[1] > 0 and (v[2] > 0 or v[3] > 0)},
        # [ms per step, TFLOP/s] inside dense block 1..4 (160^2, 80^2, 40^2, 20^2 planes at S=640)
        "per_stage": {k: [[round(r[0] / n_prof, 4), round(r[2] / (r[0] * 1e-3) / 1e12, 2) if r[0] > 0 else 0.0] for r in rows]
                      for k, rows in stages.items() if any(r[1] > 0 for r in rows)},
    })
    # whole step: the sum of the per-class roofline times against the measured step
    floor_s = sum(max(v[2] * terms_of(k) / (PEAK_BF16_MFMA_TFLOPS * 1e12), v[3] / (PEAK_HBM_GBS * 1e9)) for k, v in prof.items()) / n_prof
    rl["step"] = dict(step_rl or {}, roofline_ms=floor_s * 1e3, measured_ms=ms_per_step, frac=floor_s * 1e3 / ms_per_step)
    return rl


DETAIL_FILE = "bench_detail.json"
LINE_LIMIT = 1900          # the driver keeps a 2000-character tail of stdout and parses the last line: the whole line must fit in it


def _r(x, nd=4):
    return round(x, nd) if isinstance(x, float) else x


def compact_line(out):
    """The ONE stdout line of the bench contract, < LINE_LIMIT characters: headline fields, the dominant kernel's roofline,
    the CPU baseline and one scalar per side measurement.  Everything else (per-kernel tables, per-stage, config rooflines)
    goes to bench_detail.json (emit)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    c = {k: _r(out[k]) for k in keep if k in out}
    cfgd = out.get("config", {})
    wl = cfgd.get("workload", "")
    c["config"] = {"workload": wl[:wl.index(" | ")] if " | " in wl[:124] else wl[:120], **{k: v for k, v in cfgd.items() if k != "workload"}}
    rl = out.get("roofline") or {}
    if "bound" in rl:
        crl = {k: _r(rl[k]) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                                       "avg_launch_ms", "launches_per_step", "kernel_symbols") if k in rl}
        rs = rl.get("rocprof_serialized")
        if rs:
            crl["rocprof_avg_launch_ms"] = _r(rs.get("avg_launch_ms"))
            crl["rocprof_file"] = rs.get("file")
        if "launches_per_step_all" in rl:
            c["launches_per_step"] = rl["launches_per_step_all"]
        if rl.get("mfma_terms"):
            crl["mfma_terms"] = rl["mfma_terms"]
        ack = rl.get("all_conv_kernels")
        if ack:
            crl["all_conv"] = {"tflops": _r(ack["achieved"], 2), "frac_of_mfma_roof": _r(ack["frac_of_mfma_roof"]), "ms_per_step": _r(ack["ms_per_step"], 3)}
        pk = rl.get("per_kernel")
        if pk:       # class -> [ms per step, frac of its roof]
            crl["classes"] = {k: [_r(v["ms_per_step"], 3), _r(v["frac"], 3)] for k, v in pk.items() if v["ms_per_step"] >= 1.0}
        c["roofline"] = crl
    st = rl.get("step")
    if st:
        c.setdefault("roofline", {})["step"] = {k: _r(v) for k, v in st.items()}
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {k: _r(cb[k]) for k in ("value", "unit", "cores", "kind", "seconds_per_pass", "physical_cores", "sweep_fwd_seconds") if k in cb}
        sm = cb.get("sample", "")
        c["cpu_baseline"]["sample"] = sm[:sm.index(" | ")] if " | " in sm[:170] else sm[:160]
    for k in ("pass_tflops_algorithmic", "host_enqueue_ms_per_step", "ms_per_step_host_inputs", "sweep_fwd_ms", "train_step_ms", "train_step_host_enqueue_ms", "train_step_graph_ms",
              "train_step_kernel_ms", "train_step_launches", "forward_1rot_ms", "launches_per_step", "allreduce_overlapped", "allreduce_exposed_ms_per_step",
              "allreduce_ms", "allreduce_bytes", "allreduce_backend", "rccl_world", "device_count", "devices_seen"):
        if k in out:
            c[k] = _r(out[k])
    if "batched" in out:
        c["batched4_ms"] = _r(out["batched"]["ms_per_step"], 3)
    cf = out.get("configs") or {}
    for name, key in (("config3_three_heads_bf16", "ms"), ("config4_share_8_scenes", "ms_per_step"), ("config5_share_1824_fp16", "ms_per_step")):
        if name in cf:
            c[name.split("_")[0] + "_ms"] = _r(cf[name][key], 3)
    c["detail"] = DETAIL_FILE
    line = json.dumps(c, separators=(",", ":"))
    for k in ("classes", "all_conv", "kernel_symbols", "step"):   # never let the side fields cost the headline its parse
        if len(line) <= LINE_LIMIT:
            break
        c.get("roofline", {}).pop(k, None)
        line = json.dumps(c, separators=(",", ":"))
    return line


def emit(out):
    """Full detail -> bench_detail.json (repo root; also gpurun_out/ when it exists, so it comes back from a GPU box);
    the compact line -> stdout, last."""
    for d in (REPO, os.path.join(REPO, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, DETAIL_FILE), "w") as f:
                    json.dump(out, f, indent=1)
            except OSError:
                pass
    sys.stdout.flush()
    print(compact_line(out), flush=True)


def physical_cores():
    """Physical host cores: distinct (package, core id) pairs of /proc/cpuinfo."""
    try:
        pairs, phys = set(), "0"
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                pairs.add((phys, line.split(":")[1].strip()))
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU, RCCL) as a CHILD process before this
    process has touched the GPU, forward its output and exit with its code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-samples", type=int, default=16, help="reference-schedule samples timed on the host at the best thread setting (16 = one full pass, the default; 0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--batched-scenes", type=int, default=4,
                    help="also time a config-4 style step with this many scenes per engine call (0 = skip)")
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE.json config 3 / 4 / 5 legs")
    ap.add_argument("--leg", choices=("headline", "config3", "config4", "config5"), default="headline",
                    help="what a timed step is (under any --gpus N): the headline 1 scene x 16 rotations per GPU; config3 = E + S + ES "
                         "heads in bf16 per GPU; config4 = 8 scenes x 16 rotations per GPU + all-reduce; config5 = S=1824, 4 of the "
                         "32 rotations per GPU in fp16")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="headline leg under N > 1: weak = one scene x 16 rotations per GPU (default); strong = ONE scene per step, its 16 "
                         "rotations dealt contiguously to the ranks (16 / N each + the masked stream on every rank), one all-reduce")
    ap.add_argument("--scenes-per-rank", type=int, default=8, help="--leg config4: scenes per GPU per step")
    ap.add_argument("--train-only", action="store_true", help="only the timed training steps (profiling runs: every kernel in the trace belongs to a step)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: blocking gradient all-reduce after the backward instead of the one hidden under its second half")
    ap.add_argument("--serialize", action="store_true", help="one HIP stream, launches in issue order (per-kernel durations of a rocprofv3 trace stay per kernel)")
    args = ap.parse_args()
    if args.train_only:
        args.cpu_samples, args.batched_scenes, args.no_configs, args.no_roofline = 0, 0, True, True
    if args.serialize:
        os.environ["SMG_SERIALIZE"] = "1"        # read by the engine at creation: every launch of the run on one stream
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1 or ("RANK" in os.environ and os.environ.get("SMG_FORCE_ALLREDUCE"))
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("SMG_BENCH_BACKEND", "nccl")     # "gloo": ranks sharing one GPU (tests; RCCL needs one GPU per rank)
        local_rank %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import parallel
    from trainer import Trainer
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        tr = Trainer('reinforcement', 0.5, False, None, False)
    lay = layout_names()
    sd = synthetic.make_state_dict(lay, 0)                            # identical replicas on every rank
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    tr.model.gnum_rotations = tr.model.snum_rotations = R

    # weak scaling: rank r works on scene seed r (its own heightmap + mask) every step
    depth, masks = synthetic.heightmap_scene(rank)
    mdepth = depth * masks[0]
    labels = synthetic.uniform(rank, "bench/labels", R, 0.0, 1.5)    # both Huber branches occur
    rots = list(range(R))

    # Gradient all-reduce between backward and Adam.  Default under N > 1: hidden under the second half of the backward
    # (parallel.OverlappedGradSync over smg_backward_phase); --no-overlap: one blocking collective per range after the backward.
    # Events on the launch stream bracket every hook call: with the overlap they measure the EXPOSED part only.
    ar_events = []

    def bracket(fn):
        def wrapped(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(*a)
            e1.record()
            ar_events.append((e0, e1))
        return wrapped

    class BenchSync(parallel.OverlappedGradSync):
        pass
    BenchSync.start = bracket(parallel.OverlappedGradSync.start)
    BenchSync.finish = bracket(parallel.OverlappedGradSync.finish)
    sync = None
    if distributed:
        sync = bracket(parallel.allreduce_grads) if args.no_overlap else BenchSync()

    # inputs resident in HBM before the timed region (the heightmaps as float64, the labels as float32): the numpy form
    # of the same call costs a pageable host-to-device copy per step, which also stalls the host behind the previous step
    def on_dev(a, dtype=np.float64):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(dev)
    depth_d, mdepth_d, labels_d = on_dev(depth), on_dev(mdepth), on_dev(labels, np.float32)

    leg = args.leg
    dtype, dtype_note = "f32", "fp32 storage and results; matrix products on the fp16 MFMA as scaled two-piece splits (3 terms, fp32 accumulate; fp32-class accuracy)"
    if leg == "headline":
        units_per_step, pass_gflop, input_size = 1.0, PASS_GFLOP, 640
        # (the compact line keeps 120 characters: the discriminating facts first)
        workload = ("16 rot x 224^2 (S=640): fwd + Huber + bwd + Adam per GPU per step; reinforcement_net style 0; 17 trunk passes; fp32"
                    " | grasp trunk + graspnet_val head, 1 scene x 1 mask, masked stream de-duplicated")
        assert workload.index(" | ") <= 120          # (the compact line keeps 120 characters: a whole clause)

        if args.scaling == "strong" and world > 1:
            # total work fixed: every rank holds the SAME scene and trains on its contiguous share of the 16 rotations; the masked
            # stream's trunk pass is repeated on every rank (its gradient is the sum over the rank's rotations: exact, linear)
            depth, masks = synthetic.heightmap_scene(0)
            labels = synthetic.uniform(0, "bench/labels", R, 0.0, 1.5)
            rots = parallel.shard(list(range(R)), rank, world)
            lo, hi = rots[0], rots[-1] + 1
            depth_d, mdepth_d, labels_d = on_dev(depth), on_dev(depth * masks[0]), on_dev(labels[lo:hi], np.float32)
            units_per_step = 1.0 / world
            workload = ("strong: ONE scene x 16 rot x 224^2 (S=640) per step over all GPUs, %d rot + masked stream per GPU; "
                        "fwd + Huber + bwd + all-reduce + Adam" % len(rots))

        def step():
            return tr.train_batch(depth_d, mdepth_d, 0, rots, labels_d, grad_sync=sync)
    elif leg == "config4":
        nb = args.scenes_per_rank                                     # 64 scenes over 8 GPUs = 8 per GPU (BASELINE.json configs[3])
        sc = [synthetic.heightmap_scene(200 + nb * rank + k) for k in range(nb)]
        d8_d, m8_d = on_dev(np.stack([c[0] for c in sc])), on_dev(np.stack([c[0] * c[1][0] for c in sc]))
        lab8_d = on_dev(synthetic.uniform(8 + rank, "bench/labels_c4", nb * R, 0.0, 1.5), np.float32)
        units_per_step, pass_gflop, input_size = float(nb), PASS_GFLOP, 640
        workload = ("config 4 share: %d scenes x 16 rotations per GPU per step in ONE engine call (%d trunk streams, %d samples), fwd + Huber "
                    "+ bwd, one gradient all-reduce, one Adam step; a pass = one scene x 16 rotations" % (nb, nb * (R + 1), nb * R))

        def step():
            return tr.train_batch(d8_d, m8_d, 0, [rots] * nb, lab8_d, grad_sync=sync)
    elif leg == "config5":
        dbig, mbig = synthetic.heightmap_scene(4, size=640, n_boxes=8)    # the same scene on every rank: the ranks share its 32 rotations
        dbig_d, mdb_d = on_dev(dbig), on_dev(dbig * mbig[0])
        tr.model.gnum_rotations = tr.model.snum_rotations = 32
        r5 = [(4 * rank + k) % 32 for k in range(4)]                   # 8 GPUs x 4 = the 32 rotations; fewer GPUs: the same per-GPU share
        l5_d = on_dev(synthetic.uniform(50 + rank, "bench/labels_c5", 4, 0.0, 2.0), np.float32)
        tr.model.set_precision("fp16")
        units_per_step, pass_gflop, input_size = 4.0 / 32.0, PASS5_GFLOP, 1824
        dtype, dtype_note = "f16", "fp16 storage of activations + fp16 forward MFMA, bf16 storage of gradients + bf16 backward MFMA; fp32 accumulation / BN statistics / master weights"
        workload = ("config 5 share: 640x640 heightmap -> S=1824, 4 of the 32 rotations per GPU per step as training samples (5 trunk "
                    "streams), dense 38x38 Q maps, Huber on [0,0,0,0], bwd, gradient all-reduce, Adam; a pass = all 32 rotations")

        def step():
            return tr.train_batch(dbig_d, mdb_d, 0, r5, l5_d, grad_sync=sync)
    else:                                                             # config3
        md2_d = on_dev(depth * (masks[1] + masks[2]))
        lab1_d = on_dev(synthetic.uniform(5, "bench/labels_c3", 1, 0.0, 1.5), np.float32)
        tr.model.set_precision("bf16")
        units_per_step, pass_gflop, input_size = 1.0, 2.0 * PASS_GFLOP + 273.99, 640
        dtype, dtype_note = "bf16", "bf16 storage of activations and gradients, one bf16 MFMA term per product; fp32 accumulation / BN statistics / master weights"
        workload = ("config 3: styles 0 and 1 with 16 rotations each + style 2 at rotation 0 per GPU per step (33 samples, 35 trunk streams), "
                    "fwd + Huber + bwd + all-reduce + Adam per head; a pass = the three heads")

        def step():
            tr.train_batch(depth_d, mdepth_d, 0, rots, labels_d, grad_sync=sync)
            tr.train_batch(depth_d, mdepth_d, 1, rots, labels_d, grad_sync=sync)
            return tr.train_batch(depth_d, md2_d, 2, [0], lab1_d, grad_sync=sync)       # ES: rotation 0 only (code/models.py:418)

    def barrier():
        if distributed:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize(dev)

    import models
    for _ in range(args.warmup):
        step()
    barrier()
    del ar_events[:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        import torch.distributed as dist
        on_host = dist.get_backend() == "gloo"
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if on_host else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert bool(torch.isfinite(loss).all()), "non-finite loss in the timed region"
    ms_per_step = elapsed / args.steps * 1e3
    passes_per_s = world * units_per_step * args.steps / elapsed

    out = {
        "metric": "affordance fwd+bwd passes/sec (16-rot 224^2 RGB-D)",
        "value": passes_per_s, "unit": "passes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling if leg == "headline" else "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic", "dtype_note": dtype_note,
        "config": {"workload": workload, "leg": leg, "rotations": 32 if leg == "config5" else R, "input_size": input_size,
                   "passes_per_step_per_gpu": units_per_step, "parallelism": "dp%d" % world},
        "pass_tflops_algorithmic": pass_gflop * passes_per_s / 1e3,
    }
    # whole-job roofline of the step (every N): algorithmic FLOPs of all ranks / wall time against the MFMA roof of the
    # arithmetic in use (bf16 peak / 6 split terms for the fp32-class products, the plain peak for bf16 / fp16 operands)
    mfma_roof = PEAK_SPLIT_TFLOPS if dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
    step_rl = {"algorithmic_tflops": out["pass_tflops_algorithmic"], "mfma_roof_tflops": mfma_roof * world,
               "frac_of_mfma_roof": out["pass_tflops_algorithmic"] / (mfma_roof * world),
               "frac_of_fp32_mfma_peak": out["pass_tflops_algorithmic"] / (PEAK_F32_MFMA_TFLOPS * world)}
    if distributed:
        torch.cuda.synchronize(dev)
        ar = [a.elapsed_time(b) for a, b in ar_events]
        out["allreduce_overlapped"] = not args.no_overlap
        out["allreduce_exposed_ms_per_step"] = float(np.sum(ar)) / args.steps      # launch-stream time inside the sync hooks (hipEvents)
        # the collective itself: blocking all-reduce of the (trunk, head) gradient ranges of the last step, hipEvents around it
        tid, hid = tr.model._saved[2], tr.model._saved[3]
        t_ar = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            parallel.allreduce_grads(tr.model, tid, hid)
            e1.record()
            torch.cuda.synchronize(dev)
            t_ar.append(e0.elapsed_time(e1))
        out["allreduce_ms"] = float(np.min(t_ar))
        out["allreduce_bytes"] = 4 * sum(n for _, n in parallel.grad_segments(tr.model.HEAD_OUT, tid, hid))
        out["allreduce_backend"] = torch.distributed.get_backend()
        out["rccl_world"] = torch.distributed.get_world_size()         # what the process group saw, not what WORLD_SIZE asked for
        out["device_count"] = torch.cuda.device_count()
        devs = [None] * world
        torch.distributed.all_gather_object(devs, "%s:%d" % (os.uname().nodename, torch.cuda.current_device()))
        out["devices_seen"] = len(set(devs))
    out["roofline"] = {"step": step_rl}

    if rank == 0 and world == 1 and leg in ("config3", "config5") and not args.no_roofline:
        eng = models._ENGINES[(local_rank, input_size, 1)]
        out["roofline"] = kernel_roofline(step, eng, dev, 1, ms_per_step, step_rl)
    if rank == 0 and world == 1 and leg == "headline" and not args.train_only:
        # host time to ENQUEUE one step (no synchronisation inside): how far the launch path is from being host-bound
        torch.cuda.synchronize(dev)
        t_h = time.perf_counter()
        step()
        out["host_enqueue_ms_per_step"] = (time.perf_counter() - t_h) * 1e3
        torch.cuda.synchronize(dev)
        # the same step fed from host numpy arrays (what code/main.py hands to Trainer.backprop): PCIe-inclusive rate
        n_h = max(3, args.steps // 4)
        t_h = time.perf_counter()
        for _ in range(n_h):
            tr.train_batch(depth, mdepth, 0, rots, labels)
        torch.cuda.synchronize(dev)
        out["ms_per_step_host_inputs"] = (time.perf_counter() - t_h) / n_h * 1e3
        import models
        eng = models._ENGINES[(local_rank, 640, 1)]
        # the reference's REAL call pattern (code/main.py:338, code/trainer.py:334-384): one (mask, rotation) sample per
        # Trainer.backprop - host numpy inputs, the loss read back every call (a latency, not a throughput), 2 trunk streams
        mk3 = masks.copy()
        with contextlib.redirect_stdout(sys.stderr):
            def one_backprop(i):
                return tr.backprop(depth, 'grasp', (0, i % R), (0, 0), (0, 0), (0, 0), float(labels[i % R]), mk3.reshape(mk3.shape[:3]), None, None, None)
            # as ONE replayed hipGraph per step (smg_train_step_graph; Trainer.use_step_graph): less host time, not less latency
            tr.use_step_graph = True
            for i in range(3):
                one_backprop(i)
            torch.cuda.synchronize(dev)
            n_1 = max(5, args.steps // 2)
            t_h = time.perf_counter()
            enq = 0.0
            for i in range(n_1):
                one_backprop(i)
                enq += tr.last_enqueue_ms
            torch.cuda.synchronize(dev)
            out["train_step_graph_ms"] = (time.perf_counter() - t_h) / n_1 * 1e3
            out["train_step_graph_host_enqueue_ms"] = enq / n_1                     # host time per step up to the loss read-back
            # the default: the step as separate engine calls - and its kernels serialised on one stream with hipEvents around
            # every launch: what the GPU needs for it
            tr.use_step_graph = False
            for i in range(2):
                one_backprop(i)
            torch.cuda.synchronize(dev)
            t_h = time.perf_counter()
            enq = 0.0
            for i in range(n_1):
                one_backprop(i)
                enq += tr.last_enqueue_ms
            torch.cuda.synchronize(dev)
            out["train_step_ms"] = (time.perf_counter() - t_h) / n_1 * 1e3
            out["train_step_host_enqueue_ms"] = enq / n_1
            eng.profile_enable(True)
            for i in range(3):
                one_backprop(i)
            torch.cuda.synchronize(dev)
            p1 = eng.profile_read()
            eng.profile_enable(False)
            out["train_step_kernel_ms"] = sum(v[0] for v in p1.values()) / 3.0
            out["train_step_launches"] = sum(v[1] for v in p1.values()) // 3
            # (detail file only) where the single-sample step's serialised kernel time goes: class -> [ms per step, launches per step]
            out["train_step_per_kernel"] = {k: [round(v[0] / 3.0, 4), v[1] // 3] for k, v in p1.items() if v[1] > 0}
            for i in range(2):
                tr.forward(depth, mdepth, 0, True, False, i)
            t_h = time.perf_counter()
            for i in range(n_1):
                tr.forward(depth, mdepth, 0, True, False, i % R)          # one rotation, Q read back (trainer.py:205-207)
            out["forward_1rot_ms"] = (time.perf_counter() - t_h) / n_1 * 1e3
        # forward-only sweep (BASELINE.json configs[1]), reported beside the headline number
        x_d = tr._heightmaps_to_device(depth, mdepth)
        for _ in range(2):
            tr.model.run(0, rots, R, heightmaps=x_d, mean=tr.image_mean, std=tr.image_std)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        n_sw = max(3, args.steps // 2)
        for _ in range(n_sw):
            tr.model.run(0, rots, R, heightmaps=x_d, mean=tr.image_mean, std=tr.image_std)
        torch.cuda.synchronize(dev)
        sweep_ms = (time.perf_counter() - t1) / n_sw * 1e3
        out["sweep_fwd_ms"] = sweep_ms
        out["sweep_fwd_tflops_algorithmic"] = SWEEP_GFLOP / sweep_ms
        if args.batched_scenes > 1:
            # SURVEY.md config 4 on one GPU: several scenes per optimizer step in ONE engine call
            # (more streams -> the late, small stages fill the chip).  Reported beside the headline.
            nb = args.batched_scenes
            sc = [synthetic.heightmap_scene(100 + k) for k in range(nb)]
            d_b = np.stack([c[0] for c in sc])
            m_b = np.stack([c[0] * c[1][0] for c in sc])
            lab_b = synthetic.uniform(7, "bench/labels_b", nb * R, 0.0, 1.5)
            d_bd, m_bd, lab_bd = on_dev(d_b), on_dev(m_b), on_dev(lab_b, np.float32)
            rots_b = [rots] * nb
            for _ in range(2):
                tr.train_batch(d_bd, m_bd, 0, rots_b, lab_bd)
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            n_b = max(3, args.steps // 3)
            for _ in range(n_b):
                tr.train_batch(d_bd, m_bd, 0, rots_b, lab_bd)
            torch.cuda.synchronize(dev)
            ms_b = (time.perf_counter() - t2) / n_b * 1e3
            out["batched"] = {"scenes_per_step": nb, "samples_per_step": nb * R, "ms_per_step": ms_b,
                              "passes_per_s": nb / (ms_b * 1e-3), "pass_tflops_algorithmic": PASS_GFLOP * nb / ms_b}
        if not args.no_configs:
            def timed(fn, reps):
                fn(); torch.cuda.synchronize(dev)
                t = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize(dev)
                return (time.perf_counter() - t) / reps * 1e3
            cfg = {}
            # ---- config 3: E + S + ES heads, forward + Huber backward, 16 rotations, bf16 MFMA operands ----------------
            md2 = depth * (masks[1] + masks[2])
            lab1 = synthetic.uniform(5, "bench/labels_c3", 1, 0.0, 1.5)
            md2_d, lab1_d = on_dev(md2), on_dev(lab1, np.float32)

            def three_heads():
                tr.train_batch(depth_d, mdepth_d, 0, rots, labels_d)
                tr.train_batch(depth_d, mdepth_d, 1, rots, labels_d)
                tr.train_batch(depth_d, md2_d, 2, [0], lab1_d)              # ES: rotation 0 only (code/models.py:418)

            def q_sweeps():
                return np.concatenate([tr.model.run(st_, rots, R, heightmaps=x_d, mean=tr.image_mean, std=tr.image_std,
                                                    update_bn=False).reshape(-1).cpu().numpy() for st_ in (0, 1)])
            lr0, tr.optimizer.lr = tr.optimizer.lr, 0.0              # weights stay put: the error figures compare like with like
            q_ref = q_sweeps()
            ms32 = timed(three_heads, 3)
            tr.model.set_precision("bf16")
            ms16 = timed(three_heads, 3)
            q16 = q_sweeps()
            rl3 = None if args.no_roofline else kernel_roofline(three_heads, models._ENGINES[(local_rank, 640, 1)], dev, 1, ms16, None)
            tr.model.set_precision("fp32")
            cfg["config3_three_heads_bf16"] = {
                "roofline": rl3,
                "workload": "styles 0 and 1: 16 rotations each, style 2: rotation 0; fwd + Huber + bwd + Adam per head (33 samples, 35 + 35 trunk streams)",
                "dtype": "bf16 storage of activations and gradients, one bf16 MFMA term per product; fp32 BN statistics, accumulation, master weights",
                "ms": ms16, "ms_fp32_class": ms32, "samples_per_s": 33.0 / (ms16 * 1e-3),
                "q_max_abs_err_vs_fp32_class": float(np.abs(q16 - q_ref).max()), "q_max_abs": float(np.abs(q_ref).max()),
                "argmax_agrees": [bool(int(q16[:R].argmax()) == int(q_ref[:R].argmax())), bool(int(q16[R:].argmax()) == int(q_ref[R:].argmax()))]}
            # ---- config 4: per-GPU share of 64 scenes x 16 rotations on 8 GPUs = 8 scenes (136 streams, 128 samples) --------
            nb = 8
            sc = [synthetic.heightmap_scene(200 + k) for k in range(nb)]
            d8 = np.stack([c[0] for c in sc]); m8 = np.stack([c[0] * c[1][0] for c in sc])
            lab8 = synthetic.uniform(8, "bench/labels_c4", nb * R, 0.0, 1.5)
            d8_d, m8_d, lab8_d = on_dev(d8), on_dev(m8), on_dev(lab8, np.float32)
            ms8 = timed(lambda: tr.train_batch(d8_d, m8_d, 0, [rots] * nb, lab8_d), 2)
            cfg["config4_share_8_scenes"] = {"scenes_per_step": nb, "samples_per_step": nb * R, "streams": nb * (R + 1), "ms_per_step": ms8,
                                             "passes_per_s": nb / (ms8 * 1e-3), "pass_tflops_algorithmic": PASS_GFLOP * nb / ms8,
                                             "engine_workspace_gb": models._ENGINES[(local_rank, 640, 1)].workspace_bytes / 1e9}
            # ---- config 5: 640^2 heightmap -> S = 1824, 32 rotations over 8 GPUs = 4 rotations per GPU, fp16 operands ---------
            dbig, mbig = synthetic.heightmap_scene(4, size=640, n_boxes=8)
            mdb = dbig * mbig[0]
            tr.model.gnum_rotations = tr.model.snum_rotations = 32
            r5, l5 = [5, 6, 7, 8], [0.3, 1.9, 0.1, 0.7]
            dbig_d, mdb_d, l5_d = on_dev(dbig), on_dev(mdb), on_dev(l5, np.float32)
            _, qb32 = tr.train_batch(dbig_d, mdb_d, 0, r5, l5_d, return_q=True)
            ms5_32 = timed(lambda: tr.train_batch(dbig_d, mdb_d, 0, r5, l5_d), 4)
            tr.model.set_precision("fp16")
            _, qb16 = tr.train_batch(dbig_d, mdb_d, 0, r5, l5_d, return_q=True)
            ms5_16 = timed(lambda: tr.train_batch(dbig_d, mdb_d, 0, r5, l5_d), 4)
            rl5 = None if args.no_roofline else kernel_roofline(lambda: tr.train_batch(dbig_d, mdb_d, 0, r5, l5_d), models._ENGINES[(local_rank, 1824, 1)], dev, 1, ms5_16, None)
            tr.model.set_precision("fp32")
            tr.model.gnum_rotations = tr.model.snum_rotations = R
            a32, a16 = qb32.reshape(4, -1).cpu().numpy(), qb16.reshape(4, -1).cpu().numpy()
            cfg["config5_share_1824_fp16"] = {
                "workload": "640x640 heightmap -> 1824x1824 input, 4 of 32 rotations as training samples (5 trunk streams), dense 38x38 Q maps, Huber on [0,0,0,0]",
                "dtype": "fp16 storage of activations + fp16 forward MFMA, bf16 storage of gradients + bf16 backward MFMA (one term per product); "
                         "fp32 BN statistics, accumulation, master weights",
                "roofline": rl5,
                "ms_per_step": ms5_16, "ms_fp32_class": ms5_32, "algorithmic_tflops": 4 * 2225.73 / ms5_16 * (5.0 / 8.0),
                "q_max_abs_err_vs_fp32_class": float(np.abs(a16 - a32).max()), "q_max_abs": float(np.abs(a32).max()),
                "argmax_agrees": [bool(int(a16[k].argmax()) == int(a32[k].argmax())) for k in range(4)],
                "engine_workspace_gb": models._ENGINES[(local_rank, 1824, 1)].workspace_bytes / 1e9}
            tr.optimizer.lr = lr0
            out["configs"] = cfg
        if not args.no_roofline:
            eng = models._ENGINES[(local_rank, 640, 1)]        # the batched leg may have regrown the engine
            out["roofline"] = kernel_roofline(step, eng, dev, SPLIT_TERMS, ms_per_step, step_rl, pmc=True)
        if args.cpu_samples > 0:
            # threads = physical cores and two smaller settings; the best one is reported
            pc = physical_cores()
            settings = sorted({pc, max(1, pc // 2), min(pc, 16)}, reverse=True)
            secs, cpu_sweep, probe = cpu_baseline(0, args.cpu_samples, labels, settings)
            best = min(secs, key=lambda k: secs[k])
            out["cpu_baseline"] = {
                "value": 1.0 / secs[best], "unit": "passes/s", "cores": best, "kind": "port",
                # (first 160 characters go into the compact line: keep the facts up front)
                "sample": ("%s (rotation, mask) samples, %d threads, reference schedule (batch 1, masked stream per sample, fwd+Huber+bwd+Adam each), "
                           "PyTorch CPU fp32%s | thread setting = best of %s (1 sample each)"
                           % ("one full pass: 16" if args.cpu_samples >= R else "%d of the 16" % args.cpu_samples, best,
                              "" if args.cpu_samples >= R else ", x%g" % (R / args.cpu_samples), sorted(probe))),
                "seconds_per_pass": secs[best], "physical_cores": pc, "logical_cpus": os.cpu_count(),
                # (detail file only) the one-sample probes of the thread settings; threads = physical cores is among
                # them - on this 2-socket host oneDNN oversubscribes itself there (230-310 s per pass): an artefact, not a baseline
                "seconds_per_pass_by_threads_probe": {str(k): v for k, v in probe.items()},
                "sweep_fwd_seconds": cpu_sweep,       # 16-rotation forward-only sweep (2 rotations timed, x8), same thread setting
            }
    if rank == 0:
        emit(out)
    if distributed:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
