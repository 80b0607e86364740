"""bench.py - affordance fwd+bwd passes/s on MI355X (BASELINE.json metric).

One STEP = one PASS per GPU = one synthetic 224x224 depth heightmap + one object mask,
R = 16 rotations, every rotation a training sample:
    16 forwards (reinforcement_net branch C, code/models.py:513-539; the masked stream's
    trunk pass de-duplicated: 17 DenseNet-121 passes instead of 32),
    16 Huber losses (code/trainer.py:345-348), backward of their sum,
    gradient all-reduce over RCCL when N > 1, ONE Adam step (code/trainer.py:383).
Inputs (heightmaps, labels, weights) are resident in HBM when the timed region starts.
fp32 end to end (v_mfma_f32_32x32x2_f32): the reference runs apex O0 = fp32 and parity
is gated in fp32.

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
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz


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


def cpu_baseline(seed, n_samples, labels):
    """The oracle (PyTorch-CPU restatement of the reference) running the REFERENCE's
    schedule: batch-1 samples, masked stream recomputed for every sample, one Adam step per
    sample (code/trainer.py:338-383).  Bounded: n_samples of the 16 samples of a pass."""
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
    orc.train_step(net, opt, x, mx, 0, 0, float(labels[0]))          # warm-up (oneDNN primitives, allocator)
    t0 = time.perf_counter()
    for r in range(n_samples):
        orc.train_step(net, opt, x, mx, 0, r % R, float(labels[r % R]))
    dt = time.perf_counter() - t0
    return dt / n_samples * R                                         # seconds per 16-sample pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-samples", type=int, default=4, help="reference-schedule samples timed on the host (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--batched-scenes", type=int, default=4,
                    help="also time a config-4 style step with this many scenes per engine call (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1 or ("RANK" in os.environ and os.environ.get("SMG_FORCE_ALLREDUCE"))
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)
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
    sync = parallel.allreduce_grads if distributed else None
    rots = list(range(R))

    def step():
        return tr.train_batch(depth, mdepth, 0, rots, labels, grad_sync=sync)

    def barrier():
        if distributed:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        import torch.distributed as dist
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert bool(torch.isfinite(loss).all()), "non-finite loss in the timed region"
    ms_per_step = elapsed / args.steps * 1e3
    passes_per_s = world * args.steps / elapsed

    out = {
        "metric": "affordance fwd+bwd passes/sec (16-rot 224^2 RGB-D)",
        "value": passes_per_s, "unit": "passes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "reinforcement_net style 0 (grasp trunk + graspnet_val head): 1 scene x 1 mask x 16 rotations per GPU per "
                               "step, fwd + 16 Huber losses + bwd + Adam, S=640 (224^2 heightmap), masked stream de-duplicated (17 trunk passes)",
                   "rotations": R, "input_size": 640, "scenes_per_step": world, "parallelism": "dp%d" % world},
        "pass_tflops_algorithmic": PASS_GFLOP * passes_per_s / 1e3,
    }

    if rank == 0 and world == 1:
        import models
        eng = models._ENGINES[(local_rank, 640, 1)]
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
            rots_b = [rots] * nb
            for _ in range(2):
                tr.train_batch(d_b, m_b, 0, rots_b, lab_b)
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            n_b = max(3, args.steps // 3)
            for _ in range(n_b):
                tr.train_batch(d_b, m_b, 0, rots_b, lab_b)
            torch.cuda.synchronize(dev)
            ms_b = (time.perf_counter() - t2) / n_b * 1e3
            out["batched"] = {"scenes_per_step": nb, "samples_per_step": nb * R, "ms_per_step": ms_b,
                              "passes_per_s": nb / (ms_b * 1e-3), "pass_tflops_algorithmic": PASS_GFLOP * nb / ms_b}
        if not args.no_roofline:
            # per-kernel-class hipEvent timing on the launch stream (separate, untimed passes)
            eng = models._ENGINES[(local_rank, 640, 1)]        # the batched leg may have regrown the engine
            eng.profile_enable(True)
            n_prof = 3
            for _ in range(n_prof):
                step()
            torch.cuda.synchronize(dev)
            prof = eng.profile_read()
            stages = eng.profile_read_stages()
            eng.profile_enable(False)
            conv = {k: v for k, v in prof.items() if k != "elementwise" and v[1] > 0}
            dom = max(conv, key=lambda k: conv[k][0])
            ms, n, fl = conv[dom]
            conv_ms = sum(v[0] for v in conv.values())
            conv_fl = sum(v[2] for v in conv.values())
            achieved = fl / (ms * 1e-3) / 1e12
            # HBM bytes per launch of the dominant class, from the committed PMC passes (profiles/; rocprofv3 cannot run
            # inside this process): FETCH_SIZE x 2 + WRITE_SIZE, see profiles/pmc_r01_hbm_traffic.md.  None if absent.
            traffic = None
            try:
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_r01_hbm_traffic.json")) as f:
                    traffic = json.load(f).get(dom, {}).get("hbm_bytes_per_launch")
            except (OSError, ValueError):
                pass
            out["roofline"] = {
                "bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_unit": "bytes per launch (PMC, profiles/pmc_r01_hbm_traffic.json)",
                "avg_launch_ms": ms / n, "launches_per_step": n // n_prof, "flops_per_launch": fl / n,
                "all_conv_kernels": {"achieved": conv_fl / (conv_ms * 1e-3) / 1e12, "frac": conv_fl / (conv_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                     "ms_per_step": conv_ms / n_prof, "executed_gflop_per_step": conv_fl / n_prof / 1e9},
                "elementwise_ms_per_step": prof["elementwise"][0] / n_prof,
                "per_kernel": {k: {"ms_per_step": v[0] / n_prof, "launches_per_step": v[1] // n_prof,
                                   "tflops": (v[2] / (v[0] * 1e-3) / 1e12) if v[0] > 0 else 0.0} for k, v in prof.items() if v[1] > 0},
                # [ms per step, TFLOP/s] inside dense block 1..4 (160^2, 80^2, 40^2, 20^2 planes at S=640)
                "per_stage": {k: [[round(r[0] / n_prof, 4), round(r[2] / (r[0] * 1e-3) / 1e12, 2) if r[0] > 0 else 0.0] for r in rows]
                              for k, rows in stages.items() if any(r[1] > 0 for r in rows)},
            }
        if args.cpu_samples > 0:
            cores = torch.get_num_threads()
            sec = cpu_baseline(0, args.cpu_samples, labels)
            out["cpu_baseline"] = {
                "value": 1.0 / sec, "unit": "passes/s", "cores": cores, "kind": "port",
                "sample": "%d of the 16 (rotation, mask) training samples of one pass, reference schedule (batch 1, masked stream "
                          "recomputed per sample, fwd + Huber + bwd + Adam per sample, PyTorch CPU fp32), extrapolated x%g" % (args.cpu_samples, R / args.cpu_samples),
                "seconds_per_pass": sec,
            }
    if rank == 0:
        print(json.dumps(out))
    if distributed:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
