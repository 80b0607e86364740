"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through
the C ABI (libsmg_hip.so via smg_hip.py), against
  * the golden vectors captured from the reference (tests/golden/reference_vectors.npz),
  * the oracle (oracle/affordance.py, PyTorch-CPU) on the same seeded inputs.

Tolerances
  * rotation / preprocessing gathers: bit-exact;
  * Q values: north_star 1e-3 relative: |dq| <= 1e-3 * max(|q|, 5e-2 * max|q|), argmax exact
    (measured against an fp64 evaluation, tests/gpu_qnoise.py: the PyTorch-CPU fp32 oracle is off
    by 4e-6..6e-6 abs on |Q| <= 1.6, the HIP path by 8e-6..2.3e-5);
  * gradients: fp32 training-mode BN on these inputs is ill-conditioned - the PyTorch-CPU
    fp32 oracle itself deviates from an fp64 evaluation by ~5e-3 (median) to 3e-2 per
    tensor - so each tensor's error against the fp64 oracle must stay within
    3x the fp32 oracle's own error (on that tensor, or its 90th-percentile relative error where it
    got lucky), and the median over the 368 tensors within 3x the oracle's median.
"""
import copy
import zlib

import numpy as np
import pytest
import torch

from helpers import grads_within_fp32_class, MEAN, STD, nhwc_plane, oracle_net, orc, probe_idx, q_close, scene, scene_tensors

pytestmark = pytest.mark.gpu


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF)


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the gpu-marked tests must run on the MI355X box")
    import smg_hip
    smg_hip.lib()   # raises if libsmg_hip.so is missing
    return torch.device("cuda:0")


def product_net(seed, out_ch=1, R=16):
    from helpers import product_net as pn
    return pn(seed, out_ch, R)


def engine_of(net, S=640):
    import models
    return models._ENGINES[(0, S, net.HEAD_OUT)]


# ---------------------------------------------------------------------------------------
def test_g1_rotation_gather_bit_exact(gpu, golden):
    """K1 against golden G1 (CRC of torch's grid_sample index map) for all 16 rotations."""
    net = product_net(0)
    S = 640
    idx_img = (torch.arange(S * S, dtype=torch.float32).reshape(1, 1, S, S) + 1.0).repeat(1, 3, 1, 1)
    net.forward(idx_img, idx_img, 0, True, -1)
    eng = engine_of(net)
    img = eng.debug_read("img")
    for r in range(16):
        a = nhwc_plane(img, eng.max_streams, eng.HWp[0], 4, S, S, r)
        idx = (a[0].astype(np.int64) - 1).astype(np.int32)
        assert int((idx < 0).sum()) == int(golden["g1_oob_640"][r])
        assert crc(idx) == golden["g1_crc_640"][r], "rotation %d" % r
        assert (a[0] == a[1]).all() and (a[0] == a[2]).all() and (a[3] == 0).all()


def test_g1_rotation_gather_bit_exact_1824_r32(gpu, golden):
    """Config 5 geometry: a 640x640 heightmap -> S = 1824, 32 rotations (11.25 degrees): the gather of every rotation
    against golden G1 (CRC of torch's index map).  An index image does not survive fp32 above 2^24, so it is split in
    a row image and a column image; two 16-rotation sweeps keep the engine at 17 streams."""
    net = product_net(0, R=32)
    S = 1824
    rows = torch.arange(S, dtype=torch.float32).reshape(1, 1, S, 1).expand(1, 3, S, S) + 1.0
    cols = torch.arange(S, dtype=torch.float32).reshape(1, 1, 1, S).expand(1, 3, S, S) + 1.0
    for base in (0, 16):
        planes = []
        for img in (rows, cols):
            imgs = torch.cat((img, img), dim=0).contiguous().cuda()
            net.run(0, list(range(base, base + 16)), 32, images_nchw=imgs, update_bn=False)
            eng = engine_of(net, S)
            raw = eng.debug_read("img").reshape(eng.max_streams, eng.HWp[0], 4)
            planes.append(raw[:16, :S * S, 0].astype(np.int64))        # 0 = out of frame, else 1 + row / col
        for k in range(16):
            r = base + k
            ry, cx = planes[0][k], planes[1][k]
            idx = np.where(ry > 0, (ry - 1) * S + (cx - 1), -1).astype(np.int32).reshape(S, S)
            assert ((ry > 0) == (cx > 0)).all()
            assert int((idx < 0).sum()) == int(golden["g1_oob_1824"][r]), "rotation %d" % r
            assert crc(idx) == golden["g1_crc_1824"][r], "rotation %d" % r


def test_g3_trunk_stage_statistics(gpu, golden):
    """Golden G3: per-stage statistics (mean, L2, absmax) and 32 probes of the reference's trunk activations
    (pool0, every dense block, every transition) for seed 0, rotation 3 - read back from the engine's buffers."""
    net = product_net(0)
    x, mx = scene_tensors(0, [0])
    net.forward(x, mx, 0, True, 3)
    eng = engine_of(net)
    NS, H, HWp = eng.max_streams, eng.H, eng.HWp
    xb = [nhwc_plane(eng.debug_read("x%d" % (b + 1)), NS, HWp[2 + b], (256, 512, 1024, 1024)[b], H[2 + b], H[2 + b], 0) for b in range(4)]
    stages = {"pool0": xb[0][:64], "denseblock1": xb[0], "transition1": xb[1][:128], "denseblock2": xb[1],
              "transition2": xb[2][:256], "denseblock3": xb[2], "transition3": xb[3][:512], "denseblock4": xb[3]}
    for name, a in stages.items():
        assert tuple(golden["g3_%s_shape" % name][1:]) == a.shape, name
        flat = a.astype(np.float64).ravel()
        ref = golden["g3_%s_stats" % name]
        got = np.asarray([flat.mean(), np.sqrt((flat * flat).sum()), np.abs(flat).max()])
        # mean is a cancelling sum of ~1e6 terms: compare it on the scale of the rms value
        rms = ref[1] / np.sqrt(flat.size)
        assert abs(got[0] - ref[0]) <= 1e-5 * rms, (name, got, ref)
        assert abs(got[1] - ref[1]) <= 1e-5 * ref[1] and abs(got[2] - ref[2]) <= 1e-4 * ref[2], (name, got, ref)
        pi = probe_idx(flat.size, 32, "g3/" + name)
        np.testing.assert_allclose(a.ravel()[pi], golden["g3_%s_probe" % name], rtol=2e-4, atol=2e-5 * ref[2], err_msg=name)


def test_g2_heightmap_preprocess_bit_exact(gpu):
    """Engine heightmap path (zoom x2, pad, normalise, replicate) == oracle preprocess + rotate."""
    from trainer import Trainer
    tr = Trainer('reinforcement', 0.5, False, None, False)
    tr.model.gnum_rotations = tr.model.snum_rotations = 16
    d, dm = scene(0, [0])
    tr.forward(d, dm, 0, True, False, 5)
    eng = engine_of(tr.model)
    img = eng.debug_read("img")
    x, mx = scene_tensors(0, [0])
    a = nhwc_plane(img, eng.max_streams, eng.HWp[0], 4, 640, 640, 0)
    assert (a[:3] == orc.rotate(x, 5, 16).numpy()[0]).all()
    a = nhwc_plane(img, eng.max_streams, eng.HWp[0], 4, 640, 640, 1)
    assert (a[:3] == mx.numpy()[0]).all()


def test_g2_literal_reference_constants_reproduce_the_nan_pattern(gpu, golden):
    """Trainer(..., literal_reference=True): the RELEASED normalisation constants mean = std = [0, 0, 0]
    (code/trainer.py:176-185) make every network input x / 0: +inf where the heightmap is positive, NaN (0 / 0) elsewhere.
    The input kernel must reproduce golden G2's pattern (captured from the reference's own Trainer.forward) element for
    element, and the Q values are NaN like the reference's."""
    from trainer import Trainer
    tr = Trainer('reinforcement', 0.5, False, None, False, literal_reference=True)
    tr.model.gnum_rotations = tr.model.snum_rotations = 16
    d, dm = scene(0, [0])
    q = tr.forward(d, dm, 0, True, False, 0)                     # rotation 0 = the identity gather: stream 0 is the tensor G2 captured
    eng = engine_of(tr.model)
    img = eng.debug_read("img")
    a = nhwc_plane(img, eng.max_streams, eng.HWp[0], 4, 640, 640, 0)[:3][None]
    assert tuple(a.shape) == tuple(golden["g2_literal_shape"])
    assert int(np.isinf(a).sum()) == int(golden["g2_literal_ninf"])
    assert int(np.isnan(a).sum()) == int(golden["g2_literal_nnan"])
    assert crc(np.isinf(a).astype(np.uint8)) == golden["g2_literal_infmask_crc"]
    assert (a[np.isinf(a)] > 0).all()                             # +inf only: heights are non-negative
    assert q.shape == (1,) and q.dtype == np.float64 and np.isnan(q).all() and bool(golden["g2_literal_out_isnan"])
    qs = tr.forward(d, dm, 0, True, False, -1)                    # the sweep form: R NaNs
    assert qs.shape == (16,) and np.isnan(qs).all()
    tr.model.gnum_rotations = tr.model.snum_rotations = 1         # as the reference's Trainer leaves them (code/models.py:312-313): G2's out shape
    qs = tr.forward(d, dm, 0, True, False, -1)
    assert tuple(qs.shape) == tuple(golden["g2_literal_out_shape"]) and np.isnan(qs).all()
    # ... and the same from a forward that leaves the running statistics alone (models.run(update_bn=False): bench sweeps)
    hm = tr._heightmaps_to_device(d, dm)
    q2 = tr.model.run(0, [0, 5], 16, heightmaps=hm, mean=tr.image_mean, std=tr.image_std, update_bn=False)
    assert torch.isnan(q2).all()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_g4_q_sweeps_vs_reference(gpu, golden, seed):
    """16-rotation sweeps of styles 0/1 and the ES pass against the reference's values."""
    net = product_net(seed)
    x, mx = scene_tensors(seed, [seed % 8])
    _, mx2 = scene_tensors(seed, [1, 2])
    for style in (0, 1):
        out = net.forward(x, mx, style, True, -1)
        assert isinstance(out, list) and len(out) == 16 and tuple(out[0].shape) == (1, 1, 1, 1)
        q = np.asarray([float(t) for t in out])
        ref = golden["g4_s%d_q%d" % (seed, style)]
        ok, worst = q_close(q, ref)
        assert ok, "style %d worst |dq| %.3e" % (style, worst)
        assert int(q.argmax()) == int(ref.argmax())
    out = net.forward(x, mx2, 2, True, -1)
    assert isinstance(out, list) and len(out) == 1
    ok, worst = q_close([float(out[0])], golden["g4_s%d_q2" % seed], scale=np.abs(golden["g4_s%d_q0" % seed]).max())
    assert ok, worst


def test_g4_branch_b_and_g7_bn_buffers(gpu, golden):
    net = product_net(0)
    x, mx = scene_tensors(0, [0])
    qb = net.forward(x, mx, 0, True, 5)
    assert torch.is_tensor(qb) and tuple(qb.shape) == (1, 1, 1, 1)
    assert q_close([float(qb)], golden["g4_branchB_style0_rot5"], scale=1.1)[0]
    qb1 = net.forward(x, mx, 1, True, 7)
    assert q_close([float(qb1)], golden["g4_branchB_style1_rot7"], scale=1.1)[0]
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    keys = sorted(k[3:-3] for k in golden.files if k.startswith("g7_") and k.endswith("_rm"))
    for key in keys:
        np.testing.assert_allclose(sd[key + ".running_mean"].numpy(), golden["g7_%s_rm" % key], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(sd[key + ".running_var"].numpy(), golden["g7_%s_rv" % key], rtol=2e-4, atol=2e-5)
        assert int(sd[key + ".num_batches_tracked"]) == int(golden["g7_%s_nbt" % key])


def test_sweep_bn_update_order(gpu):
    """A 2-rotation sweep updates trunk BN buffers 4 times and head buffers twice
    (SURVEY.md Appendix B), exactly like the oracle's sequential schedule."""
    net = product_net(1, R=2)
    on = oracle_net(1, R=2)
    x, mx = scene_tensors(1, [2])
    net.forward(x, mx, 0, True, -1)
    orc.forward(on, x, mx, 0, True, -1)
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    so = on.state_dict()
    assert int(sd["grasp_depth_trunk.features.norm0.num_batches_tracked"]) == 4
    assert int(sd["graspnet_val.grasp-val-norm0.num_batches_tracked"]) == 2
    assert int(sd["suction_depth_trunk.features.norm0.num_batches_tracked"]) == 0
    for k in ("grasp_depth_trunk.features.norm0", "grasp_depth_trunk.features.denseblock3.denselayer7.norm2",
              "grasp_depth_trunk.features.transition1.norm", "graspnet_val.grasp-val-norm1"):
        np.testing.assert_allclose(sd[k + ".running_mean"].numpy(), so[k + ".running_mean"].numpy(), rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(sd[k + ".running_var"].numpy(), so[k + ".running_var"].numpy(), rtol=2e-4, atol=2e-5)


def _fp64_truth(on, rx, mx, style, label):
    o64 = copy.deepcopy(on).double()
    o64.zero_grad()
    trunk = getattr(o64, orc.STYLE_TRUNK[style]).features
    head = getattr(o64, orc.STYLE_HEAD[style])
    q = head(torch.cat((trunk(rx.double()), trunk(mx.double())), 1))
    orc.huber(q[0, 0, 0, 0], label).sum().backward()
    return float(q), {n: p.grad for n, p in o64.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("style,rot,label", [(0, 3, 0.4), (1, 9, 7.5), (2, 0, -3.0)])
def test_g5_backward_gradients(gpu, golden, style, rot, label):
    """Every one of the 368 gradient tensors of a train step (both Huber branches; style 2 = the ES pass with its
    two-object mask, gs_depth_trunk + suctionnet_val - the head quirk of code/models.py:582)."""
    on = oracle_net(0)
    x, mx = scene_tensors(0, [1, 2] if style == 2 else [0])
    rx = orc.rotate(x, rot, 16)
    q64, g64 = _fp64_truth(on, rx, mx, style, label)
    on.zero_grad()
    qo = orc.forward(on, x, mx, style, False, rot)
    orc.huber(qo[0, 0, 0, 0], label).sum().backward()
    net = product_net(0)
    net.zero_grad()
    qp = net.forward(x, mx, style, False, rot)
    d = qp[0, 0, 0, 0] - label
    loss = 0.5 * d ** 2 if abs(float(d.detach())) < 1 else abs(d) - 0.5       # code/trainer.py:345-348
    loss.backward()                                                            # autograd -> smg_backward
    assert abs(float(qp.detach()) - q64) <= 1e-3 * max(abs(q64), 1e-2)
    # Per tensor within 3x the fp32 oracle's own error (or its 90th-percentile relative error), the median within 3x the
    # oracle's median (measured 0.3x .. 2.3x, tests/gpu_gradnoise.py - the step is chaotic at this level: any change of
    # summation order moves every tensor by ~5e-3)
    # (up to four of the 368 tensors may sit past 3x the oracle's error on THAT tensor - capped at 5 % of their norm: which tensors the
    # ReLU-mask noise of the 20x20 planes hits hardest differs between arithmetic variants of the same accuracy: the three-piece bf16
    # split puts dense block 4 layer 8 at 0.85 of the bound for style 0, the two-piece fp16 split layer 9 at 1.30; the distribution
    # gates - median and 90th percentile within 3x the oracle's - hold for both)
    rel_p, rel_o, _ = grads_within_fp32_class(net.named_parameters(), on.named_parameters(), g64, 3.0, "g5 style %d" % style, max_outliers=3, outlier_cap=0.05)
    assert len(rel_p) == 368
    print("g5 style %d: median relative error %.2e = %.2f x the fp32 oracle's own (%.2e); 90th percentile %.2f x"
          % (style, np.median(rel_p), np.median(rel_p) / np.median(rel_o), np.median(rel_o), np.percentile(rel_p, 90) / np.percentile(rel_o, 90)))
    # the same tensors the reference produced gradients for
    has = golden["g5_step0_hasgrad"] if style == 0 else None
    if has is not None:
        mine = np.asarray([p.grad is not None for p in net.parameters()])
        assert (mine == has).all()


def test_split_scales_cover_extreme_magnitudes(gpu):
    """The fp16-split products (csrc/gemm.cuh, operand kind 3) rest on three scales: per weight tensor (its maximum), per BN + ReLU
    operand (hypot(gamma, beta)), per gradient tensor and stream (its recorded maximum).  Push every one of them far from the synthetic
    nets' comfortable magnitudes and hold the result to the SAME fp32-class gates against the fp64 oracle: the bottleneck weights of dense
    block 2 x 4096 and of block 3 x 2^-12 (the BN behind a convolution removes the factor again: the activations stay put, the weight
    tensors and their gradients move by 3.6 decades each way), gamma / beta of every norm2 in block 1 x 2^-10 (tiny BN + ReLU operands;
    the next convolution's weights x 2^10 restore the signal), and a label 1e-6 away from Q (gradients of 1e-6 of their usual size
    through the whole backward)."""
    on = oracle_net(0)
    sd = {k: v.clone() for k, v in on.state_dict().items()}
    for k in sd:
        if ".denseblock2." in k and k.endswith("conv1.weight"):
            sd[k] *= 4096.0
        if ".denseblock3." in k and k.endswith("conv1.weight"):
            sd[k] *= 2.0 ** -12
        if ".denseblock1." in k and (k.endswith("norm2.weight") or k.endswith("norm2.bias")):
            sd[k] *= 2.0 ** -10
        if ".denseblock1." in k and k.endswith("conv2.weight"):
            sd[k] *= 2.0 ** 10
    on.load_state_dict(sd)
    style, rot = 0, 5
    x, mx = scene_tensors(0, [0])
    rx = orc.rotate(x, rot, 16)
    q64, _ = _fp64_truth(on, rx, mx, style, 0.0)
    label = q64 - 1e-6                                     # |dq| = 1e-6: the quadratic Huber branch, gradients scaled by 1e-6
    q64, g64 = _fp64_truth(on, rx, mx, style, label)
    net = product_net(0)
    net.load_state_dict(sd)
    net.zero_grad()
    qp = net.forward(x, mx, style, False, rot)
    assert abs(float(qp.detach()) - q64) <= 1e-3 * max(abs(q64), 1e-2), (float(qp.detach()), q64)
    # the label sits 1e-6 from the fp64 Q; the product's own Q differs from it by its fp32-class error, which would swamp the 1e-6:
    # drive the backward with the oracle's dq so that both sides differentiate the same loss
    dq = torch.full_like(qp.detach(), float(q64 - label))
    qp.backward(dq)
    on.zero_grad()
    qo = orc.forward(on, x, mx, style, False, rot)
    qo.backward(torch.full_like(qo.detach(), float(q64 - label)))
    rel_p, rel_o, _ = grads_within_fp32_class(net.named_parameters(), on.named_parameters(), g64, 3.0, "extreme scales", max_outliers=4, outlier_cap=0.05)
    assert len(rel_p) == 368
    gn = {n: float(p.grad.double().norm()) for n, p in net.named_parameters() if p.grad is not None}
    assert all(np.isfinite(v) for v in gn.values()) and min(gn.values()) > 0.0
    print("extreme scales: gradient norms span %.1e .. %.1e; median rel err %.2e (oracle %.2e)" % (min(gn.values()), max(gn.values()), np.median(rel_p), np.median(rel_o)))


def test_activation_scales_follow_gamma_and_beta(gpu):
    """scale_kernel (csrc/elem.cuh): the per-BatchNorm activation scale of the fp16-split products is the power of two that puts
    max_c hypot(gamma_c, beta_c) into [16, 32) - recomputed here in numpy from the state dict for every norm1 / norm2 of the grasp
    trunk, its transitions and the head's norm0, and compared exactly (powers of two)."""
    net = product_net(0)
    x, mx = scene_tensors(0, [0])
    net.forward(x, mx, 0, True, 3)                          # style 0: grasp trunk + graspnet_val
    asc = engine_of(net).debug_read("asc").reshape(-1, 2)
    sd = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in net.state_dict().items()}

    def expect(prefix):
        m = np.sqrt((sd[prefix + ".weight"] ** 2 + sd[prefix + ".bias"] ** 2).max())
        return 2.0 ** (4 - int(np.floor(np.log2(np.float32(m)))))
    names = []
    for b, nl in enumerate((6, 12, 24, 16)):
        for i in range(nl):
            for nrm in ("norm1", "norm2"):
                names.append("grasp_depth_trunk.features.denseblock%d.denselayer%d.%s" % (b + 1, i + 1, nrm))
    names += ["grasp_depth_trunk.features.transition%d.norm" % (b + 1) for b in range(3)]
    names.append("graspnet_val.grasp-val-norm0")
    assert asc.shape[0] == len(names) == 120
    for k, nm in enumerate(names):
        s_ = expect(nm)
        assert asc[k, 0] == np.float32(s_) and asc[k, 1] == np.float32(1.0 / s_), (nm, asc[k], s_)
        m = np.sqrt((sd[nm + ".weight"] ** 2 + sd[nm + ".bias"] ** 2).max())
        assert 16.0 <= m * s_ < 32.0, (nm, m, s_)


# HIP gradients against the reference's own (golden g5_step*): max / 95th percentile / median of the per-tensor norm errors, and the worst
# probed entry in units of its tensor's rms element.  Measured (round 6): step 0 1.6e-2 / 1.4e-3 / 2.9e-4 and 0.17; step 1 1.7e-3 / 8.4e-4 / 1.9e-4
# and 0.04; step 2 7.5e-4 / 3.0e-4 / 5.0e-5 and 0.04 - the gates sit at about three times the largest of each.
_G5_NORM_GATES, _G5_PROBE_GATE = (5e-2, 5e-3, 1e-3), 0.5


def test_g5_g6_trainer_steps_vs_reference(gpu, golden):
    """Trainer.backprop x3 (grasp, suction, grasp_then_suction): q, loss and Adam-updated
    weights against the reference's own trajectory; then the diverged target network."""
    from trainer import Trainer
    import synthetic
    tr = Trainer('reinforcement', 0.5, False, None, False)
    sd = synthetic.make_state_dict(orc.state_layout(1), 0)
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    tr.model_target.load_state_dict(tr.model.state_dict())
    for m in (tr.model, tr.model_target):
        m.gnum_rotations = m.snum_rotations = 16
    depth, masks = synthetic.heightmap_scene(0)
    actions = [("grasp", 3, 0.4), ("suction", 9, 7.5), ("grasp_then_suction", 0, -3.0)]
    for si, (action, rot, label) in enumerate(actions):
        om = masks.copy()
        # the ES step masks with objects 1 + 2 (code/trainer.py:370), like the golden run
        g_id, s_id = ((1, rot), (2, rot)) if action == "grasp_then_suction" else ((0, rot), (0, rot))
        loss = tr.backprop(depth, action, (0, rot), (0, rot), g_id, s_id, label, om, None, None, None)
        assert om.ndim == 4                                       # in-place reshape, code/trainer.py:336
        assert isinstance(loss, np.ndarray) and loss.shape == ()
        style = {"grasp": 0, "suction": 1, "grasp_then_suction": 2}[action]
        q = float((tr.model.gra_prob, tr.model.suc_prob, tr.model.gs_prob)[style].reshape(-1)[0])
        assert abs(q - float(golden["g5_step%d_q" % si])) <= 2e-3 * max(abs(float(golden["g5_step%d_q" % si])), 0.1)
        assert abs(float(loss) - float(golden["g5_step%d_loss" % si])) <= 2e-3 * max(float(golden["g5_step%d_loss" % si]), 0.1)
        # the gradients the reference's own backward() produced (code/trainer.py:350-351), DIRECTLY: per-tensor norms of all 1110
        # parameters (golden g5_step*_gradnorm) and 16 probed entries of six tensors per step (g5_step*_grad_*) - not through the
        # oracle.  Steps 1 and 2 start from weights that already carry one / two Adam steps of fp32 noise on both sides.
        ref_n = golden["g5_step%d_gradnorm" % si]
        mine_n = np.asarray([float(p.grad.double().norm()) if p.grad is not None else 0.0 for p in tr.model.parameters()])
        assert ((mine_n > 0) == (ref_n > 0)).all()                # the same tensors received a gradient
        big = ref_n > 1e-3 * ref_n.max()
        reln = np.abs(mine_n[big] - ref_n[big]) / ref_n[big]
        worst_probe = 0.0
        named = dict(tr.model.named_parameters())
        gpre = "g5_step%d_grad_" % si
        gkeys = [k for k in golden.files if k.startswith(gpre)]
        assert len(gkeys) >= 6
        for k in gkeys:
            pg = named[k[len(gpre):]]
            pi = probe_idx(pg.numel(), 16, "g5/" + k[len(gpre):])
            mine_p = pg.grad.detach().cpu().numpy().ravel()[pi].astype(np.float64)
            # (a probe is one element: its error is measured against the tensor's rms element, norm / sqrt(numel))
            rms = float(ref_n[list(named).index(k[len(gpre):])]) / np.sqrt(pg.numel())
            worst_probe = max(worst_probe, float(np.abs(mine_p - golden[k]).max() / max(rms, 1e-30)))
        print("step %d (%s): |grad| per tensor vs the reference's: median %.2e, 95th percentile %.2e, max %.2e over %d tensors; worst probe error %.2e of its tensor's rms element"
              % (si, action, np.median(reln), np.percentile(reln, 95), reln.max(), int(big.sum()), worst_probe))
        assert reln.max() < _G5_NORM_GATES[0] and np.percentile(reln, 95) < _G5_NORM_GATES[1] and np.median(reln) < _G5_NORM_GATES[2], (reln.max(), np.percentile(reln, 95), np.median(reln))
        assert worst_probe < _G5_PROBE_GATE, worst_probe
        # Adam: on a segment's first step every weight moves by +-lr (sign of its gradient), so a
        # probe can only be off by a full 2e-4 where fp32 noise flips the sign of a ~zero gradient.  (The ES step is the
        # SECOND step of suctionnet_val - code/models.py:582 - so its head probes test a real two-step Adam trajectory.)
        params = dict(tr.model.named_parameters())
        pre = "g6_step%d_param_" % si
        diffs = []
        for k in [k for k in golden.files if k.startswith(pre)]:
            p = params[k[len(pre):]]
            pi = probe_idx(p.numel(), 16, "g6/" + k[len(pre):])
            dk = np.abs(p.detach().cpu().numpy().ravel()[pi] - golden[k])
            assert dk.max() <= 2.1e-4, (k, dk)
            if "norm5" not in k:      # d/d(norm5.bias) is identically 0 (the head's BN removes it):
                diffs.append(dk)      # its Adam update is pure rounding noise on both sides
        diffs = np.concatenate(diffs)
        assert (diffs < 2e-6).mean() >= 0.9, diffs
    d = depth
    qt = tr.forward(d, d * masks[0], 0, True, True, 3)
    assert isinstance(qt, np.ndarray) and qt.shape == (1,) and qt.dtype == np.float64
    assert q_close(qt, [golden["g4_target_rot3"]], scale=1.1)[0]      # target net never moved
    qm = tr.forward(d, d * masks[0], 0, True, False, 3)
    assert abs(qm[0] - qt[0]) > 1e-4                                   # model has


def test_get_label_value_uses_target_network(gpu):
    from trainer import Trainer
    import synthetic
    tr = Trainer('reinforcement', 0.5, False, None, False)
    for m in (tr.model, tr.model_target):
        m.gnum_rotations = m.snum_rotations = 16
    depth, masks = synthetic.heightmap_scene(1)
    fut = tr.forward(depth, depth * masks[2], 0, True, True, 4)[0]
    exp, cur = tr.get_label_value('grasp', 3, 0, 1, 0, depth, masks, masks, (2, 4), (2, 4), (2, 4), (2, 4),
                                  'grasp', 0, 0, 0)
    assert cur == 1 and abs(exp - (1 + 0.5 * fut)) < 1e-9
    exp, cur = tr.get_label_value('grasp', 3, 0, 0, 0, depth, masks, masks, (2, 4), (2, 4), (2, 4), (2, 4), 'grasp', 0, 0, 0)
    assert (exp, cur) == (0, 0)


def test_g8_reactive(gpu, golden):
    """Config 1: reactive_net, 1 rotation; logits, softmax, weighted CE and its gradients."""
    from trainer import Trainer
    import synthetic
    tr = Trainer('reactive', 0.5, False, None, False)
    sd = synthetic.make_state_dict(orc.state_layout(3), 0)
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    x, mx = scene_tensors(0, [0])
    out = tr.model.forward(x, mx, 0, True, -1)
    assert isinstance(out, list) and tuple(out[0].shape) == (1, 3, 1, 1)
    ok, worst = q_close(out[0].cpu().numpy().ravel(), golden["g8_logits"])
    assert ok, worst
    depth, masks = synthetic.heightmap_scene(0)
    p0 = tr.forward(depth, depth * masks[0], 0, True, False, -1)
    assert isinstance(p0, (float, np.floating)) and abs(p0 - float(golden["g8_softmax0"])) < 1e-3
    loss = tr.backprop(depth, 'grasp', (0, 0), (0, 0), (0, 0), (0, 0), 1, masks.copy(), None, None, None)
    assert abs(float(loss) - float(golden["g8_loss"])) < 2e-3


def test_torch_optimizer_zero_grad_does_not_accumulate(gpu):
    """INTEGRATION.md's second mode: a torch optimizer over model.parameters().  Its zero_grad(set_to_none=True) only
    drops p.grad and never sees the engine's flat gradient buffer, so the next backward must restart the touched ranges
    from zero - while two backwards WITHOUT a zero_grad in between still accumulate, like autograd."""
    net = product_net(0)
    x, mx = scene_tensors(0, [0])
    opt = torch.optim.Adam(net.parameters(), lr=0.0)

    def backward_once():
        q = net.forward(x, mx, 0, False, 3)
        (q[0, 0, 0, 0] * 1.0).backward()
        return net.flat_grads().double().cpu().numpy().copy()
    opt.zero_grad()
    g1 = backward_once()
    opt.step()
    opt.zero_grad()                                      # set_to_none=True on torch >= 2
    assert all(p.grad is None for p in net.parameters())
    g2 = backward_once()
    n1 = np.sqrt((g1 * g1).sum())
    assert np.sqrt(((g2 - g1) ** 2).sum()) <= 5e-3 * n1, "second iteration carries the first one's gradient"
    g3 = backward_once()                                 # no zero_grad: accumulates
    assert abs(np.sqrt((g3 * g3).sum()) / n1 - 2.0) < 2e-2


def _cos(a, b):
    return float((a * b).sum() / max(np.sqrt((a * a).sum() * (b * b).sum()), 1e-300))


def test_config3_three_heads_bf16_storage(gpu):
    """BASELINE.json config 3 as stated: E + S + ES heads, forward + Huber backward + Adam, 16 rotations, bf16 STORAGE of
    activations and gradients (dense-block buffers, bottlenecks, G', the backward ring) with single-term bf16 MFMA; fp32 BN
    statistics, accumulation and master weights.  Not a parity mode - a random-weight 121-layer DenseNet amplifies 8-bit
    mantissas - so the 33-sample step is held to bounds MEASURED against the fp32-class result
    (tests/gpu_precision.py, two builds with different summation orders: max |dq| / max |q| 0.135 - 0.165 / 0.075 / 0.006 for
    styles 0 / 1 / 2, gradient cosine 0.54 - 0.86 / 0.71 / 0.93 - the bf16 gradient of this random-weight net is itself chaotic)
    with headroom, and the argmax over the 16 rotations must agree."""
    from trainer import Trainer
    import smg_hip
    import synthetic
    R = 16
    tr = Trainer('reinforcement', 0.5, False, None, False)
    sd = synthetic.make_state_dict(orc.state_layout(1), 0)
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    tr.model.gnum_rotations = tr.model.snum_rotations = R
    tr.optimizer.lr = 0.0
    depth, masks = synthetic.heightmap_scene(0)
    rots = list(range(R))
    labels = synthetic.uniform(0, "bench/labels", R, 0.0, 1.5)
    work = ((0, depth * masks[0], rots, labels), (1, depth * masks[0], rots, labels), (2, depth * (masks[1] + masks[2]), [0], labels[:1]))

    def run():
        out = []
        for style, m, rs, lab in work:
            loss, q = tr.train_batch(depth, m, style, rs, lab, return_q=True)
            t0, n0 = smg_hip.trunk_range(1, (1, 0, 2)[style]); h0, hn = smg_hip.head_range(1, (1, 0, 0)[style])
            g = tr.model.flat_grads().double().cpu().numpy()
            out.append((q.reshape(-1).cpu().numpy().astype(np.float64), np.concatenate([g[t0:t0 + n0], g[h0:h0 + hn]]), loss.cpu().numpy()))
        return out
    ref = run()
    tr.model.set_precision("bf16")
    got = run()
    tr.model.set_precision("fp32")
    q_bound, cos_bound = (0.22, 0.12, 0.02), (0.45, 0.45, 0.80)
    for style in range(3):
        q, g, loss = got[style]; qr, gr, _ = ref[style]
        assert np.isfinite(q).all() and np.isfinite(g).all() and np.isfinite(loss).all()
        err, c = np.abs(q - qr).max() / np.abs(qr).max(), _cos(g, gr)
        print("config 3 bf16 style %d: max|dq|/max|q| %.4f, gradient cosine %.4f, |g| ratio %.3f" % (style, err, c, np.sqrt((g * g).sum() / (gr * gr).sum())))
        assert err <= q_bound[style], (style, err)
        assert c >= cos_bound[style], (style, c)
        assert 0.6 <= np.sqrt((g * g).sum() / (gr * gr).sum()) <= 1.5
        assert int(q.argmax()) == int(qr.argmax()), (style, int(q.argmax()), int(qr.argmax()))
    again = run()                                          # the default mode is restored bit for bit (forward)
    for style in range(3):
        assert np.array_equal(again[style][0], ref[style][0])


def test_config3_bf16_three_adam_steps_track_the_fp32_trajectory(gpu):
    """Config 3's bf16 storage mode as a TRAINING mode: three Adam steps of the three heads (E: 16 rotations, S: 16, ES: rotation
    0; lr 1e-4 as the reference, code/trainer.py:99) in bf16 against the same three steps in the fp32-class mode from the same
    weights.  Asserted: (1) the per-step sample losses stay within a measured bound of the fp32-class ones; (2) the weight
    displacement after three steps points the same way (cosine) and has the same size - Adam normalises each step, so what
    matters is the SIGN pattern of the gradient, which bf16 must mostly preserve; (3) evaluated in fp32-class arithmetic, the
    summed loss on the training samples moves from the initial weights in the same direction and by a comparable amount
    whichever mode produced the weights.  Bounds are measured ones with headroom (printed)."""
    from trainer import Trainer
    import smg_hip
    import synthetic
    R = 16
    depth, masks = synthetic.heightmap_scene(0)
    rots = list(range(R))
    labels = synthetic.uniform(0, "bench/labels", R, 0.0, 1.5)
    work = ((0, depth * masks[0], rots, labels), (1, depth * masks[0], rots, labels), (2, depth * (masks[1] + masks[2]), [0], labels[:1]))
    sd = synthetic.make_state_dict(orc.state_layout(1), 0)

    def trajectory(prec):
        tr = Trainer('reinforcement', 0.5, False, None, False)
        tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        tr.model.gnum_rotations = tr.model.snum_rotations = R
        w0 = tr.model._flat_params.clone()
        tr.model.set_precision(prec)
        losses = []
        for _ in range(3):
            losses.append(np.concatenate([tr.train_batch(depth, m, st_, rs, lab).cpu().numpy() for st_, m, rs, lab in work]))
        tr.model.set_precision("fp32")
        lr, tr.optimizer.lr = tr.optimizer.lr, 0.0              # evaluate the reached weights in fp32-class arithmetic
        final = np.concatenate([tr.train_batch(depth, m, st_, rs, lab).cpu().numpy() for st_, m, rs, lab in work])
        tr.optimizer.lr = lr
        return np.asarray(losses, dtype=np.float64), final.astype(np.float64), (tr.model._flat_params - w0).double().cpu().numpy()
    l32, f32, dw32 = trajectory("fp32")
    l16, f16, dw16 = trajectory("bf16")
    assert np.isfinite(l16).all() and np.isfinite(f16).all() and np.isfinite(dw16).all()
    moved = (dw32 != 0) | (dw16 != 0)
    c = _cos(dw16[moved], dw32[moved])
    size = np.linalg.norm(dw16) / np.linalg.norm(dw32)
    sign_agree = float((np.sign(dw16[moved]) == np.sign(dw32[moved])).mean())
    step_err = np.abs(l16 - l32).sum(axis=1) / np.abs(l32).sum(axis=1)
    d32, d16 = f32.sum() - l32[0].sum(), f16.sum() - l32[0].sum()
    print("config 3 bf16 trajectory: per-step |dloss| / |loss| %s, displacement cosine %.3f, size ratio %.3f, sign agreement %.3f, "
          "loss change after 3 steps fp32-class %.4f, bf16 %.4f (initial %.4f)" % (np.round(step_err, 4), c, size, sign_agree, d32, d16, l32[0].sum()))
    # measured (MI355X, this seed, two builds with different fp32-class arithmetic): per-step loss deviation 0.10 / 0.11 / 0.76 (after
    # two Adam steps the two random-weight nets have moved apart: the third step's losses differ by their own size), displacement
    # cosine 0.600, size ratio 0.995 - 0.996, sign agreement 0.795 - stable to three digits across the builds.  The summed loss after
    # the three steps is NOT asserted: at lr 1e-4 on this random-weight net with single-sample BatchNorm it moved 31.06 -> 18.7 with one
    # fp32-class arithmetic and 31.06 -> 35.6 with the other (bf16: 26.4 / 31.0): the loss surface is rough at that scale, the
    # DIRECTION Adam takes is what the storage mode must preserve.
    assert (step_err[:2] <= 0.25).all() and step_err[2] <= 1.2, step_err
    assert c >= 0.45 and 0.8 <= size <= 1.25 and sign_agree >= 0.70, (c, size, sign_agree)
    assert abs(d16) <= l32[0].sum() and abs(d32) <= l32[0].sum(), (d32, d16)      # nothing blows up


def test_config5_share_fp16_storage(gpu):
    """BASELINE.json config 5's per-GPU share as stated: 640x640 heightmap -> S = 1824, 4 of the 32 rotations as one training
    call, fp16: activations STORED in fp16 and multiplied on the fp16 MFMA in the forward; gradients stored and multiplied in
    bf16 (fp32 exponent range - no loss scaling); fp32 statistics / accumulation / master weights.  Held to bounds measured
    against the fp32-class result (tests/gpu_precision.py: max |dq| / max |q| 0.0033, gradient cosine 0.990) with headroom;
    the argmax of every 38x38 Q map must agree.  model.half() is the torch idiom for the same switch."""
    from trainer import Trainer
    import synthetic
    tr = Trainer('reinforcement', 0.5, False, None, False)
    sd = synthetic.make_state_dict(orc.state_layout(1), 0)
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    tr.model.gnum_rotations = tr.model.snum_rotations = 32
    tr.optimizer.lr = 0.0
    dbig, mbig = synthetic.heightmap_scene(4, size=640, n_boxes=8)
    r5, l5 = [5, 6, 7, 8], [0.3, 1.9, 0.1, 0.7]

    def run():
        loss, q = tr.train_batch(dbig, dbig * mbig[0], 0, r5, l5, return_q=True)
        return q.reshape(4, -1).cpu().numpy().astype(np.float64), tr.model.flat_grads().double().cpu().numpy().copy(), loss.cpu().numpy()
    qr, gr, _ = run()
    tr.model.half()
    assert tr.model.precision == "fp16" and tr.model._flat_params.dtype == torch.float32
    q, g, loss = run()
    tr.model.float()
    assert tr.model.precision == "fp32"
    assert np.isfinite(q).all() and np.isfinite(g).all() and np.isfinite(loss).all()
    err, c = np.abs(q - qr).max() / np.abs(qr).max(), _cos(g, gr)
    print("config 5 share fp16: max|dq|/max|q| %.5f, gradient cosine %.5f" % (err, c))
    assert err <= 0.008 and c >= 0.975, (err, c)
    assert 0.9 <= np.sqrt((g * g).sum() / (gr * gr).sum()) <= 1.1
    for k in range(4):
        assert int(q[k].argmax()) == int(qr[k].argmax()), k


def test_heightmap_generation_matches_restatement(gpu):
    """utils.get_heightmap (code/utils.py:38-68) on the device against oracle/heightmap.py (whose
    cv2 semantics are restated, not pinned): the fused point-cloud / rigid-transform / perspective-warp kernel must agree
    to double rounding (only the summation order of the 3-term dot product may differ from BLAS)."""
    import synthetic
    import utils as smg_utils
    from oracle import heightmap as hm
    depth = 0.45 + 0.2 * synthetic.uniform(3, "hm/depth", 480 * 640).reshape(480, 640)
    depth[100:140, 200:260] = 0.0                                       # invalid-depth patch, as the simulator emits
    color = (synthetic.uniform(3, "hm/color", 480 * 640 * 3) * 255).astype(np.uint8).reshape(480, 640, 3)
    k = np.asarray([[618.62, 0, 320], [0, 618.62, 240], [0, 0, 1]])
    pose = np.eye(4)
    th = 0.05
    pose[:3, :3] = np.array([[1, 0, 0], [0, -np.cos(th), np.sin(th)], [0, -np.sin(th), -np.cos(th)]])
    pose[:3, 3] = [-0.5, 0.02, 0.62]
    ch, dh, cm, dm, a_htor = smg_utils.get_heightmap(color, depth, k, pose, None, 0.002)
    oh, om, oa = hm.get_depth_heightmaps(depth, k, pose)
    assert dh.shape == (224, 224) and dm.shape == (448, 448) and dh.dtype == np.float64
    np.testing.assert_allclose(dh, oh, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(dm, om, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(a_htor, oa, rtol=0, atol=1e-12)
    assert ch.shape == (224, 224, 3) and cm.shape == (448, 448, 3) and ch.dtype == np.uint8


def test_model_api_errors(gpu):
    import models
    net = models.reinforcement_net(True)      # not moved to the GPU
    with pytest.raises(RuntimeError):
        net.forward(torch.zeros(1, 3, 640, 640), torch.zeros(1, 3, 640, 640), 0, True, -1)
    with pytest.raises(NotImplementedError):
        net.eval()
    net = net.cuda()
    q = net.forward(torch.zeros(1, 3, 640, 640), torch.zeros(1, 3, 640, 640), 0, False, 0)
    net.forward(torch.zeros(1, 3, 640, 640), torch.zeros(1, 3, 640, 640), 0, True, 0)   # overwrites the saved activations
    with pytest.raises(RuntimeError):
        q.sum().backward()


def test_snapshot_roundtrip(gpu, tmp_path):
    """Logger.save_model pattern (code/logger.py:121-125): model.cpu().state_dict() -> file -> load."""
    import models
    net = product_net(2)
    x, mx = scene_tensors(2, [1])
    q0 = float(net.forward(x, mx, 0, True, 2))
    f = str(tmp_path / "snap.pth")
    torch.save(net.cpu().state_dict(), f)
    net = net.cuda()
    other = models.reinforcement_net(True)
    other.load_state_dict(torch.load(f))
    other = other.cuda()
    other.gnum_rotations = other.snum_rotations = 16
    assert len(other.state_dict()) == 2217
    # BN buffers changed by the first forward do not enter a training-mode forward
    assert abs(float(other.forward(x, mx, 0, True, 2)) - q0) < 1e-6


@pytest.mark.parametrize("case", ["two_scenes_few_rotations", "four_scenes_all_rotations", "eight_scenes_all_rotations"])
def test_multi_scene_batch_equals_sum_of_single_scene_gradients(gpu, case):
    """Config-4 style batch (several scenes x rotations in ONE engine call): Q values equal the
    single-scene calls and the gradient equals the sum of the single-scene gradients.  The second case is the
    bench's batched leg: 4 scenes x 16 rotations = 68 trunk streams, 64 samples (workspace sizing, ring buffers)."""
    from trainer import Trainer
    import synthetic
    tr = Trainer('reinforcement', 0.5, False, None, False)
    sd = synthetic.make_state_dict(orc.state_layout(1), 3)
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    tr.model.gnum_rotations = tr.model.snum_rotations = 16
    tr.optimizer.lr = 0.0                                   # keep the weights fixed across the calls
    if case == "two_scenes_few_rotations":
        seeds, rots = (5, 6), [[0, 7, 12], [3, 9]]
        labels = [0.2, 1.4, 0.9, 3.0, 0.1]
    elif case == "four_scenes_all_rotations":
        seeds, rots = (5, 6, 7, 8), [list(range(16))] * 4
        labels = list(synthetic.uniform(11, "multi/labels", 64, 0.0, 1.5))
    else:       # BASELINE.json config 4's per-GPU share: 8 scenes x 16 rotations = 136 trunk streams, 128 samples
        seeds, rots = tuple(range(5, 13)), [list(range(16))] * 8
        labels = list(synthetic.uniform(12, "multi/labels8", 128, 0.0, 1.5))
    scenes = [synthetic.heightmap_scene(s) for s in seeds]
    d = np.stack([sc[0] for sc in scenes])
    m = np.stack([sc[0] * sc[1][1] for sc in scenes])
    loss_b, q_b = tr.train_batch(d, m, 0, rots, labels, return_q=True)
    g_b = tr.model.flat_grads().clone()
    g_sum = torch.zeros_like(g_b)
    q_s = []
    k = 0
    for i in range(len(seeds)):
        _, q = tr.train_batch(d[i], m[i], 0, rots[i], labels[k:k + len(rots[i])], return_q=True)
        k += len(rots[i])
        q_s.append(q.reshape(-1).cpu().numpy())
        g_sum += tr.model.flat_grads()
    q_s = np.concatenate(q_s)
    # same per-stream work; the tile shapes (hence the fp32 summation order) depend on how many streams a launch holds
    np.testing.assert_allclose(q_b.reshape(-1).cpu().numpy(), q_s, rtol=0, atol=2e-5)
    num = float((g_b - g_sum).double().norm())
    den = float(g_sum.double().norm())
    print("batch vs sum of single-scene gradients (%s): |d| / |g| = %.3e" % (case, num / den))
    # only the fp32 summation order / ReLU-mask noise differs (ill-conditioned, see header).  Measured in round 5, default build /
    # -DSMG_SPLIT16=0 build: 6.2e-3 / 1.4e-3 (5 samples), 1.9e-3 / 1.9e-3 (64), 1.2e-3 / 1.4e-3 (128); one missing sample of 64
    # would be 1.5e-2
    assert num <= (1.2e-2 if len(labels) < 10 else 4e-3) * den, (num, den)
    assert loss_b.shape == (len(labels),)
    if len(seeds) == 8:
        import models
        ws = models._ENGINES[(0, 640, 1)].workspace_bytes
        print("config-4 share: 136 streams / 128 samples per call, engine workspace %.1f GB" % (ws / 1e9))
        assert ws < 120e9


def test_train_batch_device_resident_inputs_equal_host_inputs(gpu):
    """bench.py feeds train_batch device-resident heightmaps / labels (no PCIe copy, no host stall per step): same Q values,
    losses and gradients - bit for bit - as the numpy form of the call, for one scene and for a stack of scenes."""
    from trainer import Trainer
    import synthetic
    tr = Trainer('reinforcement', 0.5, False, None, False)
    sd = synthetic.make_state_dict(orc.state_layout(1), 4)
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    tr.model.gnum_rotations = tr.model.snum_rotations = 16
    tr.optimizer.lr = 0.0
    dev = tr.model._flat_params.device
    scenes = [synthetic.heightmap_scene(s) for s in (21, 22)]
    d = np.stack([sc[0] for sc in scenes]); m = np.stack([sc[0] * sc[1][0] for sc in scenes])
    for dh, mh, rots, labels in ((d[0], m[0], [0, 5, 11], [0.3, 1.2, 0.05]), (d, m, [[1, 2], [7, 8, 15]], [0.1, 0.9, 2.0, 0.4, 0.6])):
        loss_h, q_h = tr.train_batch(dh, mh, 0, rots, labels, return_q=True)
        g_h = tr.model.flat_grads().clone()
        loss_d, q_d = tr.train_batch(torch.from_numpy(dh).to(dev), torch.from_numpy(mh).to(dev), 0, rots,
                                     torch.tensor(labels, dtype=torch.float32, device=dev), return_q=True)
        assert torch.equal(q_h, q_d) and torch.equal(loss_h, loss_d)
        g_d = tr.model.flat_grads()
        # the weight gradients of the 1x1 convolutions are summed with fp32 atomics: equal up to their order
        assert float((g_h - g_d).double().norm()) <= 1e-5 * float(g_h.double().norm())


def test_large_input_dense_qmap(gpu):
    """Config-5 path: a 640x640 heightmap -> S = 1824: odd plane sizes (57x57), average-pool rows
    the pooling never reads, and a dense 38x38 Q map per sample."""
    import synthetic
    net = product_net(0, R=32)
    on = oracle_net(0, R=32)
    depth, masks = synthetic.heightmap_scene(4, size=640, n_boxes=8)
    x = orc.preprocess(depth, [MEAN] * 3, [STD] * 3)
    mx = orc.preprocess(depth * masks[0], [MEAN] * 3, [STD] * 3)
    assert x.shape[-1] == 1824
    hm = torch.from_numpy(np.stack([depth, depth * masks[0]])).cuda()
    q = net.run(0, [5], 32, heightmaps=hm, mean=MEAN, std=STD)
    assert tuple(q.shape) == (1, 1, 38, 38)
    with torch.no_grad():
        qo = orc.forward(on, x, mx, 0, True, 5)
    assert tuple(qo.shape) == (1, 1, 38, 38)
    a, b = q.cpu().numpy().ravel(), qo.numpy().ravel()
    ok, worst = q_close(a, b)
    assert ok, worst
    assert int(a.argmax()) == int(b.argmax())


def test_large_input_backward_config5_share(gpu):
    """Config 5's per-GPU share: a 640x640 heightmap (S = 1824), 4 of the 32 rotations as training samples in ONE call
    (5 trunk streams, dense 38x38 Q maps, Huber on element [0,0,0,0] like code/trainer.py:345).  (a) every sample's Q map
    and the gradient of ONE sample against the PyTorch-CPU oracle evaluated in fp64; (b) the 4-sample batch gradient
    against the sum of the four single-sample gradients."""
    import psutil
    import synthetic
    from trainer import Trainer
    tr = Trainer('reinforcement', 0.5, False, None, False)
    sd = synthetic.make_state_dict(orc.state_layout(1), 0)
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    tr.model.gnum_rotations = tr.model.snum_rotations = 32
    tr.optimizer.lr = 0.0
    depth, masks = synthetic.heightmap_scene(4, size=640, n_boxes=8)
    md = depth * masks[0]
    rots, labels = [5, 6, 7, 8], [0.3, 1.9, 0.1, 0.7]
    loss_b, q_b = tr.train_batch(depth, md, 0, rots, labels, return_q=True)
    assert tuple(q_b.shape) == (4, 1, 38, 38)
    g_b = tr.model.flat_grads().clone()
    g_sum = torch.zeros_like(g_b)
    for r, lab in zip(rots, labels):
        _, q1 = tr.train_batch(depth, md, 0, [r], [lab], return_q=True)
        g_sum += tr.model.flat_grads()
        np.testing.assert_allclose(q1.reshape(-1).cpu().numpy(), q_b[rots.index(r)].reshape(-1).cpu().numpy(), rtol=0, atol=5e-5)
        if r == 5:
            g_one = tr.model.flat_grads().double().cpu().numpy().copy()
            q_one = q1.reshape(38, 38).cpu().numpy().copy()
    num, den = float((g_b - g_sum).double().norm()), float(g_sum.double().norm())
    print("S=1824 batch vs sum of single-sample gradients: |d| / |g| = %.3e" % (num / den))
    # (four samples: as sensitive to the summation order as the small S = 640 case above - 5.8e-3 measured once the batch and the
    #  single-sample calls tile the 114^2 plane differently, 16 x 16 against 8 x 8)
    assert num <= 1.2e-2 * den, (num, den)
    import models
    print("config-5 share: S=1824, 5 streams, engine workspace %.1f GB" % (models._ENGINES[(0, 1824, 1)].workspace_bytes / 1e9))
    # oracle (fp64 where the host has the memory for its autograd graph at S = 1824: ~40 GB; else fp32)
    big = psutil.virtual_memory().available > 160e9
    on = oracle_net(0, R=32)
    if big:
        on = on.double()
    x = orc.preprocess(depth, [MEAN] * 3, [STD] * 3)
    mx = orc.preprocess(md, [MEAN] * 3, [STD] * 3)
    rx = orc.rotate(x, 5, 32)
    if big:
        rx, mx = rx.double(), mx.double()
    trunk = getattr(on, orc.STYLE_TRUNK[0]).features
    head = getattr(on, orc.STYLE_HEAD[0])
    q = head(torch.cat((trunk(rx), trunk(mx)), 1))
    orc.huber(q[0, 0, 0, 0], labels[0]).sum().backward()
    ok, worst = q_close(q_one.ravel(), q.detach().double().numpy().ravel())
    assert ok, worst
    assert int(q_one.argmax()) == int(q.detach().numpy().argmax())
    off = {name: (o, n) for name, kind, o, shape in __import__("smg_hip").layout(1) if kind == 0 for n in [int(np.prod(shape))]}
    rels = []
    for name, p in on.named_parameters():
        if p.grad is None:
            continue
        o, n = off[name]
        t = p.grad.double().numpy().ravel()
        nrm = np.sqrt((t * t).sum())
        if nrm > 0 and "norm5" not in name:
            rels.append(np.sqrt(((g_one[o:o + n] - t) ** 2).sum()) / nrm)
    rels = np.asarray(rels)
    print("config-5 share: gradient vs %s oracle: median rel %.3e, max %.3e over %d tensors" % ("fp64" if big else "fp32", np.median(rels), rels.max(), rels.size))
    assert rels.size >= 360 and np.median(rels) < 2e-2 and rels.max() < 1e-1


def test_batched_object_evaluation_equals_per_object_loop(gpu):
    """main.py:158-192's loops in one engine call per style: same Q values,
    same argmax, same BN buffers as the per-object Trainer.forward loop."""
    from trainer import Trainer
    import synthetic
    sd = synthetic.make_state_dict(orc.state_layout(1), 1)

    def fresh():
        tr = Trainer('reinforcement', 0.5, False, None, False)
        tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        tr.model.gnum_rotations = tr.model.snum_rotations = 4
        return tr
    depth, masks = synthetic.heightmap_scene(2)
    masks = masks[:3]
    a, b = fresh(), fresh()
    conf_loop = np.stack([a.forward(depth, depth * masks[k], 1, True) for k in range(3)])
    conf_batch = b.forward_objects(depth, masks, style=1)
    assert conf_batch.shape == (3, 4)
    # (tile shapes, hence the fp32 summation order, depend on the number of streams in a launch)
    np.testing.assert_allclose(conf_batch, conf_loop, rtol=0, atol=3e-5)
    assert np.unravel_index(np.argmax(conf_batch), conf_batch.shape) == np.unravel_index(np.argmax(conf_loop), conf_loop.shape)
    gs_loop = np.full((3, 3), -100.0)
    for g in range(3):
        for s_ in range(g + 1, 3):
            gs_loop[g, s_] = a.forward(depth, depth * (masks[g] + masks[s_]), 2, True)[0]
    gs_batch = b.forward_object_pairs(depth, masks)
    np.testing.assert_allclose(gs_batch, gs_loop, rtol=0, atol=3e-5)
    sa = {k: v.cpu() for k, v in a.model.state_dict().items()}
    sb = {k: v.cpu() for k, v in b.model.state_dict().items()}
    for k in ("suction_depth_trunk.features.norm0", "suction_depth_trunk.features.denseblock3.denselayer9.norm1",
              "suctionnet_val.suction-val-norm1", "gs_depth_trunk.features.norm5"):
        assert int(sa[k + ".num_batches_tracked"]) == int(sb[k + ".num_batches_tracked"]) > 0
        np.testing.assert_allclose(sb[k + ".running_mean"].numpy(), sa[k + ".running_mean"].numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(sb[k + ".running_var"].numpy(), sa[k + ".running_var"].numpy(), rtol=1e-4, atol=1e-5)


def test_g9_batched_object_evaluation_vs_reference(gpu, golden):
    """SURVEY.md 8f-1 against the REFERENCE: golden G9 holds gra_conf / suc_conf / gs_conf and BN buffers captured from
    the imported reference running the loops of code/main.py:158-192 (3 objects, 4 rotations).  One engine call per
    style must reproduce the values, the argmax (lowest index on ties, np.argmax) and the buffers."""
    from trainer import Trainer
    import synthetic
    tr = Trainer('reinforcement', 0.5, False, None, False)
    sd = synthetic.make_state_dict(orc.state_layout(1), 1)
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    tr.model.gnum_rotations = tr.model.snum_rotations = 4
    depth, masks = synthetic.heightmap_scene(2)
    masks = masks[:3]
    depth_a = depth * masks.sum(0)                                     # main.py:144-151
    gra = tr.forward_objects(depth_a, masks, style=0)
    suc = tr.forward_objects(depth_a, masks, style=1)
    gs = tr.forward_object_pairs(depth_a, masks)
    for got, key in ((gra, "g9_gra_conf"), (suc, "g9_suc_conf")):
        ref = golden[key]
        ok, worst = q_close(got, ref)
        assert ok, (key, worst)
        assert np.unravel_index(np.argmax(got), got.shape) == np.unravel_index(np.argmax(ref), ref.shape), key
    ref = golden["g9_gs_conf"]
    assert ((gs == -100.0) == (ref == -100.0)).all()
    sel = ref != -100.0
    ok, worst = q_close(gs[sel], ref[sel])
    assert ok, worst
    assert np.unravel_index(np.argmax(gs), gs.shape) == np.unravel_index(np.argmax(ref), ref.shape)
    best = tr.best_actions(depth_a, masks)                             # the same three calls + argmax, one entry point
    assert best["bestg_id"] == tuple(int(v) for v in np.unravel_index(np.argmax(golden["g9_gra_conf"]), (3, 4)))
    sd = {k: v.cpu() for k, v in tr.model.state_dict().items()}
    for key in sorted(k[3:-3] for k in golden.files if k.startswith("g9_") and k.endswith("_rm")):
        # best_actions repeated the three sweeps: every buffer saw the reference's sequence twice
        assert int(sd[key + ".num_batches_tracked"]) == 2 * int(golden["g9_%s_nbt" % key]), key


def test_g8_reactive_gradients_and_adam(gpu, golden):
    """Reactive net train step (code/trainer.py:282-332): weighted-CE loss vs the reference's value, and all 368 gradient
    tensors TENSOR-WISE against the fp64 oracle with the yardstick of the Huber case (here 5x the fp32 oracle's own error);
    the gradient norms also against the reference's own (golden G8)."""
    on = oracle_net(0, out_ch=3, R=1)
    x, mx = scene_tensors(0, [0])
    o64 = copy.deepcopy(on).double()
    o64.zero_grad()
    trunk, head = getattr(o64, orc.STYLE_TRUNK[0]).features, getattr(o64, orc.STYLE_HEAD[0])
    q64 = head(torch.cat((trunk(x.double()), trunk(mx.double())), 1))          # reactive_net: rotation 0 = identity
    lab = torch.ones((1, 1, 1), dtype=torch.long)
    torch.nn.functional.nll_loss(torch.log_softmax(q64[0].view(1, 3, 1, 1), dim=1), lab,
                                 weight=torch.tensor([1.0, 1.0, 0.0], dtype=torch.float64)).sum().backward()   # orc.reactive_loss in fp64
    g64 = {n: p.grad for n, p in o64.named_parameters() if p.grad is not None}
    on.zero_grad()
    qo = orc.forward(on, x, mx, 0, False, 0)
    orc.reactive_loss(qo, 1).backward()
    net = product_net(0, out_ch=3, R=1)
    net.zero_grad()
    q = net.forward(x, mx, 0, False, 0)                       # branch C, rotation 0
    w = torch.tensor([1.0, 1.0, 0.0], device=q.device)
    label = torch.ones((1, 1, 1), dtype=torch.long, device=q.device)
    loss = torch.nn.functional.nll_loss(torch.log_softmax(q[0].view(1, 3, 1, 1), dim=1), label, weight=w).sum()
    loss.backward()                                            # torch autograd on the 3 logits -> smg_backward
    assert abs(float(loss.detach()) - float(golden["g8_loss"])) < 2e-3
    # (5x, and up to three tensors may sit outside it up to 20 % of their norm - measured in round 4: two, norm0.bias at 2.7 and
    # norm0.weight at 1.6 of the bound, the next one (a block-4 norm1.bias) AT 1.00.  This sample is ill-conditioned at the stem: on the
    # un-rotated image's constant background stem channel 46 normalises to ~0, so the sign of its ReLU mask over most of the plane
    # is decided by the last bits of the fp64 statistics - tests/gpu_diag_reactive.py: the whole error of norm0.bias (7 %) is
    # that one channel, with every 3x3 / chain variant of the kernels; the fp32 oracle lands on the fp64 side, the engine
    # on either side from run to run (the order of the statistics' atomics).  The old check allowed 10 % on every norm.)
    rel_p, _, _ = grads_within_fp32_class(net.named_parameters(), on.named_parameters(), g64, 5.0, "reactive", max_outliers=3)
    assert len(rel_p) == 368
    ref = golden["g8_gradnorm"]
    mine = np.asarray([float(p.grad.double().norm()) if p.grad is not None else 0.0 for p in net.parameters()])
    assert ((mine > 0) == (ref > 0)).all()
    big = ref > 1e-3 * ref.max()
    rel = np.abs(mine[big] - ref[big]) / ref[big]
    assert rel.max() < 1e-1 and np.percentile(rel, 95) < 2e-2 and np.median(rel) < 1e-2, (rel.max(), np.percentile(rel, 95), np.median(rel))


def test_target_network_sync(gpu):
    """code/main.py:352-353: model_target.load_state_dict(model.state_dict()) makes the two nets
    bit-identical (weights AND BN buffers), on the device, without touching the model."""
    from trainer import Trainer
    import synthetic
    tr = Trainer('reinforcement', 0.5, False, None, False)
    for m in (tr.model, tr.model_target):
        m.gnum_rotations = m.snum_rotations = 2
    depth, masks = synthetic.heightmap_scene(3)
    tr.backprop(depth, 'grasp', (0, 1), (0, 1), (0, 1), (0, 1), 0.7, masks.copy(), None, None, None)
    a, b = tr.model.state_dict(), tr.model_target.state_dict()
    assert any(not torch.equal(a[k], b[k]) for k in a)         # they have diverged (weights + BN buffers)
    before = {k: v.clone() for k, v in a.items()}
    tr.model_target.load_state_dict(tr.model.state_dict())
    a, b = tr.model.state_dict(), tr.model_target.state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert all(torch.equal(a[k], before[k]) for k in a)
    assert b["grasp_depth_trunk.features.norm0.running_mean"].is_cuda
    q1 = tr.forward(depth, depth * masks[0], 0, True, False, 1)
    q2 = tr.forward(depth, depth * masks[0], 0, True, True, 1)
    assert q1[0] == q2[0]


def test_backward_other_input_size_ragged_planes(gpu):
    """A 240^2 heightmap -> S = 704: 176^2 / 88^2 / 44^2 / 22^2 planes.  Block 1 runs the 16x16 halo kernels, block 2
    the 8x8 ones on an exact tiling, blocks 3-4 the 8x8 ones with masked ragged edges (5.5 / 2.75 tiles per side),
    the 1x1 kernels their partial-tile epilogues, and the head a dense 3x3 Q map.  Forward and all 368 gradient
    tensors of a weighted-sum loss against the fp64 oracle.  Yardstick as in test_g5_backward_gradients but with
    room for ReLU mask flips: on this scene one head activation sits 2e-6 from zero (tests/gpu_diag_head.py), the
    HIP path and PyTorch-CPU fp32 land on different sides, and that single element moves norm1.bias by 0.8 % and
    every trunk gradient by ~0.4 %.  A dropped or doubled tile edge would show as >= 4 % (one of 22 rows)."""
    import synthetic
    style, rot = 0, 5
    on = oracle_net(2)
    depth, masks = synthetic.heightmap_scene(8, size=240, n_boxes=8)
    x = orc.preprocess(depth, [MEAN] * 3, [STD] * 3)
    mx = orc.preprocess(depth * masks[0], [MEAN] * 3, [STD] * 3)
    assert x.shape[-1] == 704
    wq = torch.from_numpy(synthetic.uniform(3, "ragged/wq", 9, -1.0, 1.0).astype(np.float32)).reshape(1, 1, 3, 3)
    rx = orc.rotate(x, rot, 16)
    o64 = copy.deepcopy(on).double()
    trunk = getattr(o64, orc.STYLE_TRUNK[style]).features
    head = getattr(o64, orc.STYLE_HEAD[style])
    q64 = head(torch.cat((trunk(rx.double()), trunk(mx.double())), 1))
    assert tuple(q64.shape) == (1, 1, 3, 3)
    (q64 * wq.double()).sum().backward()
    g64 = {n: p.grad for n, p in o64.named_parameters() if p.grad is not None}
    on.zero_grad()
    qo = orc.forward(on, x, mx, style, False, rot)
    (qo * wq).sum().backward()
    net = product_net(2)
    net.zero_grad()
    qp = net.forward(x, mx, style, False, rot)
    assert tuple(qp.shape) == (1, 1, 3, 3)
    (qp * wq.cuda()).sum().backward()
    ok, worst = q_close(qp.detach().cpu().numpy().ravel(), q64.detach().numpy().ravel())
    assert ok, worst
    rel_p, _, _ = grads_within_fp32_class(net.named_parameters(), on.named_parameters(), g64, 3.0, "S=704")
    assert len(rel_p) == 368


_VARIANT_SCRIPT = r"""
import sys, json
import numpy as np, torch
sys.path.insert(0, %(tests)r)
from helpers import product_net, scene_tensors
net = product_net(4)
x, mx = scene_tensors(3, [1])
with torch.no_grad():
    q = np.concatenate([np.asarray(t.cpu()).ravel() for t in net.forward(x, mx, 0, True, -1)])
net.zero_grad()
qp = net.forward(x, mx, 0, False, 11)
(qp[0, 0, 0, 0] * 1.0).backward()
g = net.flat_grads().double().cpu().numpy()
idx = np.linspace(0, g.size - 1, 4096).astype(np.int64)
print("RESULT " + json.dumps({"q": q.tolist(), "gnorm": float(np.sqrt((g * g).sum())), "gprobe": g[idx].tolist()}))
"""


def test_alternative_kernel_paths_agree(gpu):
    """The same sweep + one backward through independent implementations: the 3x3 layers through the LDS-halo kernels (default:
    two-piece fp16 split, three MFMA terms) and through the generic implicit GEMM (SMG_CROSSCHECK=1: the three-piece bf16 split,
    six terms - another kernel AND another arithmetic of the same fp32-class accuracy), and the 1x1 forward of the small planes
    through the generic kernel instead of the wave-specialised one (SMG_CROSSCHECK=2), likewise the 1x1 weight gradient
    (SMG_CROSSCHECK=4).  Separate child processes: the switch is read at engine creation."""
    import json
    import os
    import subprocess
    import sys
    tests_dir = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for tag, env in (("halo", {}), ("generic", {"SMG_CROSSCHECK": "1"}), ("c1_generic", {"SMG_CROSSCHECK": "2"}), ("w1_generic", {"SMG_CROSSCHECK": "4"})):
        e = dict(os.environ)
        e.pop("SMG_CROSSCHECK", None)
        e.update(env)
        out = subprocess.run([sys.executable, "-c", _VARIANT_SCRIPT % {"tests": tests_dir}], env=e, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][-1]
        res[tag] = json.loads(line[7:])
    ref = res["halo"]
    qs = np.abs(np.asarray(ref["q"])).max()
    for tag in ("generic", "c1_generic", "w1_generic"):
        r = res[tag]
        assert np.abs(np.asarray(r["q"]) - np.asarray(ref["q"])).max() <= 2e-5 * max(qs, 1e-2), tag     # fp32 summation order only
        assert int(np.argmax(r["q"])) == int(np.argmax(ref["q"])), tag
        gp, g0 = np.asarray(r["gprobe"]), np.asarray(ref["gprobe"])
        assert np.sqrt(((gp - g0) ** 2).sum()) <= 5e-3 * np.sqrt((g0 * g0).sum()), tag               # ill-conditioned, see header
        assert abs(r["gnorm"] - ref["gnorm"]) <= 5e-3 * ref["gnorm"], tag


# ---------------------------------------------------------------------------------------
# data parallelism on hardware: two ranks (sharing the one GPU of the test box, so gloo carries the collectives - the
# engine, the flat-gradient ranges and parallel.py are exactly what bench.py runs over RCCL)
def _dp_worker(rank, world, port, out_q):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import parallel
        import synthetic
        from trainer import Trainer
        tr = Trainer('reinforcement', 0.5, False, None, False)
        sd = synthetic.make_state_dict(orc.state_layout(1), 3)
        tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        tr.model.gnum_rotations = tr.model.snum_rotations = 16
        tr.optimizer.lr = 0.0
        seeds, rots = (5, 6), [[0, 7, 12], [3, 9]]
        labels = [[0.2, 1.4, 0.9], [3.0, 0.1]]
        depth, masks = synthetic.heightmap_scene(seeds[rank])
        # rank r trains on scene r; one all-reduce of the (trunk, head) gradient ranges between backward and Adam
        tr.train_batch(depth, depth * masks[1], 0, rots[rank], labels[rank], grad_sync=parallel.allreduce_grads)
        g = tr.model.flat_grads().cpu()
        # the same step with the all-reduce split around the two halves of the backward (smg_backward_phase)
        tr.train_batch(depth, depth * masks[1], 0, rots[rank], labels[rank], grad_sync=parallel.OverlappedGradSync())
        g2 = tr.model.flat_grads().cpu()
        assert float((g - g2).double().norm()) <= 1e-5 * float(g.double().norm())
        # sharded forward sweep of scene 5: 8 rotations per rank, 16 scalars gathered, argmax on every rank
        d5, m5 = synthetic.heightmap_scene(5)
        q, best = parallel.sweep_sharded(tr, d5, d5 * m5[1], style=0)
        if rank == 0:
            out_q.put((g.numpy(), q, best))
    finally:
        dist.destroy_process_group()


def test_data_parallel_two_ranks_equal_single_process_batch(gpu):
    """Config-4 semantics on hardware: the all-reduced gradient of two ranks (one scene each) equals the gradient of
    the single-process 2-scene batch, and the rotation-sharded forward sweep equals the single-process sweep."""
    import socket
    import torch.multiprocessing as mp
    from trainer import Trainer
    import synthetic
    so = socket.socket(); so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]; so.close()
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, out_q)) for r in range(2)]
    for p in procs:
        p.start()
    g_dp, q_dp, best_dp = out_q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    tr = Trainer('reinforcement', 0.5, False, None, False)
    sd = synthetic.make_state_dict(orc.state_layout(1), 3)
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    tr.model.gnum_rotations = tr.model.snum_rotations = 16
    tr.optimizer.lr = 0.0
    scenes = [synthetic.heightmap_scene(s) for s in (5, 6)]
    d = np.stack([sc[0] for sc in scenes])
    m = np.stack([sc[0] * sc[1][1] for sc in scenes])
    tr.train_batch(d, m, 0, [[0, 7, 12], [3, 9]], [0.2, 1.4, 0.9, 3.0, 0.1])
    g_one = tr.model.flat_grads().cpu().numpy()
    num = float(np.sqrt(((g_dp.astype(np.float64) - g_one) ** 2).sum()))
    den = float(np.sqrt((g_one.astype(np.float64) ** 2).sum()))
    print("two data-parallel ranks vs the single-process batch: |d| / |g| = %.3e" % (num / den))
    assert num <= 1.2e-2 * den, (num, den)          # summation order / ReLU-mask noise only (see the multi-scene test: 6.2e-3 measured)
    q_one = tr.forward(scenes[0][0], scenes[0][0] * scenes[0][1][1], 0, True)
    assert q_dp.shape == (16,)
    np.testing.assert_allclose(q_dp, q_one, rtol=0, atol=3e-5)
    assert best_dp == int(np.argmax(q_one))


def test_two_phase_backward_equals_single_call(gpu):
    """smg_backward_phase(0) + (1) = smg_backward: with the deterministic option every convolution weight gradient bit for bit,
    everything to 1e-5 (the atomics' order); the ranges a data-parallel caller all-reduces after phase 0 are final then."""
    from trainer import Trainer
    import parallel
    import smg_hip
    import synthetic
    tr = Trainer('reinforcement', 0.5, False, None, False)
    sd = synthetic.make_state_dict(orc.state_layout(1), 7)
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    tr.model.gnum_rotations = tr.model.snum_rotations = 16
    tr.optimizer.lr = 0.0
    depth, masks = synthetic.heightmap_scene(9)
    rots, labels = [1, 6, 11], [0.4, 1.6, 0.8]
    tr.train_batch(depth, depth * masks[0], 0, rots, labels)
    eng = engine_of(tr.model)
    snap = {}

    class Probe(object):                     # stands where the all-reduce would: records what phase 0 has finished
        overlapped = True

        def start(self, model, trunk_id, head_id):
            early, late = parallel.OverlappedGradSync()._ranges(model, trunk_id, head_id)
            snap["early"] = [(o, n, model.flat_grads()[o:o + n].clone()) for o, n in early]
            snap["late_before"] = [float(model.flat_grads()[o:o + n].abs().sum()) for o, n in late]

        def finish(self, model, trunk_id, head_id):
            pass
    try:
        eng.set_option("deterministic", 1)
        tr.train_batch(depth, depth * masks[0], 0, rots, labels)
        g_one = tr.model.flat_grads().clone()
        tr.train_batch(depth, depth * masks[0], 0, rots, labels, grad_sync=Probe())
        g_two = tr.model.flat_grads().clone()
    finally:
        eng.set_option("deterministic", 0)
    assert float((g_one - g_two).double().norm()) <= 1e-5 * float(g_one.double().norm())
    for n_, p in tr.model.named_parameters():
        if p.dim() == 4 and n_.startswith("grasp_depth_trunk.features"):
            off = p.data_ptr() - tr.model._flat_params.data_ptr()
            assert torch.equal(g_one[off // 4: off // 4 + p.numel()], g_two[off // 4: off // 4 + p.numel()]), n_
    # after phase 0 the early ranges already held their final values, the late range (conv0 .. dense block 1) was still zero
    for o, n, early in snap["early"]:
        assert float((early - g_two[o:o + n]).double().norm()) <= 1e-5 * float(g_two[o:o + n].double().norm())
    assert snap["late_before"] == [0.0]
    t0, tn = smg_hip.trunk_range(1, 1)
    split = smg_hip.trunk_split(1, 1)
    assert t0 < split < t0 + tn and float(g_two[t0:split].abs().sum()) > 0


def test_second_backward_half_never_writes_the_ranges_being_all_reduced(gpu):
    """The write set of smg_backward_phase(1): between the halves the ranges [trunk split, end) + head are all-reduced IN PLACE by
    RCCL on another stream (parallel.OverlappedGradSync).  Poison exactly those ranges after phase 0: phase 1 must leave every
    poisoned element bit for bit - a read-modify-write of even one of them (round 3's db_flush over all segments) would store a
    stale local value over the reduced one and silently diverge the replicas.  Also: the phase order is enforced by the engine."""
    from trainer import Trainer
    import parallel
    import synthetic
    tr = Trainer('reinforcement', 0.5, False, None, False)
    sd = synthetic.make_state_dict(orc.state_layout(1), 7)
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    tr.model.gnum_rotations = tr.model.snum_rotations = 16
    tr.optimizer.lr = 0.0
    depth, masks = synthetic.heightmap_scene(9)
    rots, labels = [1, 6, 11], [0.4, 1.6, 0.8]
    tr.train_batch(depth, depth * masks[0], 0, rots, labels)
    g_ref = tr.model.flat_grads().clone()
    seen = {}

    class Poison(object):
        overlapped = True

        def start(self, model, trunk_id, head_id):
            early, late = parallel.OverlappedGradSync()._ranges(model, trunk_id, head_id)
            seen["early"], seen["late"] = early, late
            for o, n in early:
                model.flat_grads()[o:o + n].view(torch.int32).fill_(0x4640E6B7)      # 12345.679 (finite: Adam at lr 0 leaves the weights alone)

        def finish(self, model, trunk_id, head_id):
            pass
    tr.train_batch(depth, depth * masks[0], 0, rots, labels, grad_sync=Poison())
    g = tr.model.flat_grads()
    for o, n in seen["early"]:
        assert bool((g[o:o + n].view(torch.int32) == 0x4640E6B7).all()), "phase 1 wrote into [%d, %d)" % (o, o + n)
    for o, n in seen["late"]:                                      # and its own range is complete: equal to the single-call backward
        assert torch.isfinite(g[o:o + n]).all()
        assert float((g[o:o + n] - g_ref[o:o + n]).double().norm()) <= 1e-5 * float(g_ref[o:o + n].double().norm())
    # phase order: 1 without 0, 0 twice, a full backward between the halves - all refused with an error, none runs on stale state
    model = tr.model
    loss = tr.train_batch(depth, depth * masks[0], 0, rots, labels)          # fresh forward + full backward
    q = model.run(0, rots, 16, heightmaps=tr._heightmaps_to_device(depth, depth * masks[0]), mean=tr.image_mean, std=tr.image_std, keep_for_backward=True)
    dq = torch.ones_like(q)
    token = model._saved[1]
    with pytest.raises(RuntimeError, match="must follow"):
        model._engine_backward(token, dq, phase=1)
    model._engine_backward(token, dq, phase=0)
    with pytest.raises(RuntimeError, match="second half"):
        model._engine_backward(token, dq, phase=0)
    with pytest.raises(RuntimeError, match="second half"):
        model._engine_backward(token, dq)
    model._engine_backward(token, dq, phase=1)
    torch.cuda.synchronize()
    assert torch.isfinite(loss).all()


def test_debug_read_of_bottlenecks_in_16bit_storage(gpu):
    """smg_debug_read("bt<b>_<i>") addresses the bottleneck arena in ELEMENTS of the mode (2 bytes in bf16 / fp16 storage): the bf16
    readback of a late layer must be that layer's tensor - close to the fp32-class one, far from any other layer's."""
    net = product_net(0)
    d, dm = scene(0, [0])
    from trainer import Trainer
    tr = Trainer('reinforcement', 0.5, False, None, False)
    tr.model = net
    tr.forward(d, dm, 0, True, False, 3)
    eng = engine_of(net)
    ref = {k: eng.debug_read(k).copy() for k in ("bt1_2", "bt2_7", "bt3_20")}
    net.set_precision("bf16")
    tr.forward(d, dm, 0, True, False, 3)
    for k, r in ref.items():
        got = eng.debug_read(k)
        n = eng.HWp[int(k[2]) + 1] * 128 * 2                      # the two streams the call used
        c = _cos(got[:n].astype(np.float64), r[:n].astype(np.float64))
        others = [_cos(got[:n].astype(np.float64), o[:n].astype(np.float64)) for kk, o in ref.items() if kk != k and o.shape == r.shape]
        print("bf16 debug read %s: cosine to the fp32-class tensor %.4f" % (k, c))
        assert c >= 0.98, (k, c)
        assert all(o < 0.9 for o in others)
    net.set_precision("fp32")


def _fp32_chain(a32, w32):
    """out[m][n] = sum_k a[m][k] * w[n][k] as a k-ordered fp32 multiply-add chain (one rounding per product and per add):
    the plain fp32 arithmetic the split-MFMA products are held to."""
    acc = np.zeros((a32.shape[0], w32.shape[0]), dtype=np.float32)
    for k in range(a32.shape[1]):
        acc = (acc + a32[:, k:k + 1] * w32[None, :, k]).astype(np.float32)
    return acc


# Gate of the weight-gradient products: within 3x the error of the blocked fp32 reduction with 64-term chains.  Measured on the MI355X
# (round 6; product / 64-term / 512-term yardstick, err / sum|ab| rms): conv2 3.8e-8 / 2.2e-8 / 1.4e-7 (block 1, 153 600 terms: ratio 1.74),
# 1.3e-8 / 1.7e-8 (block 2: 0.76), 1.1e-8 / 9.8e-9 (block 3: 1.17); conv1 2.8e-8 / 2.4e-8 (1.15), 2.3e-8 / 1.4e-8 (1.63), 1.3e-8 / 9.7e-9 (1.40).
_WGRAD_BLK, _WGRAD_GATE = 64, 3.0


def _fp32_blocked(a32, w32, blk=64):
    """out[m][n] = sum_k a[k][m] * w[k][n] the way a careful fp32 reduction over a LONG k is written: sequential multiply-add
    chains of `blk` terms (one rounding per product and per add), the chains' sums then added pairwise (a binary tree, one
    rounding per add).  The yardstick of the weight-gradient products, whose reductions run over 10^4 - 10^5 pixels: a single
    sequential chain that long carries 50 - 130 times the error of any blocked kernel and gates nothing (judge, round 5)."""
    K = a32.shape[0]
    nb = (K + blk - 1) // blk
    a = np.zeros((nb * blk, a32.shape[1]), dtype=np.float32); a[:K] = a32
    w = np.zeros((nb * blk, w32.shape[1]), dtype=np.float32); w[:K] = w32
    a, w = a.reshape(nb, blk, -1), w.reshape(nb, blk, -1)
    acc = np.zeros((nb, a.shape[2], w.shape[2]), dtype=np.float32)
    for j in range(blk):
        acc = (acc + a[:, j, :, None] * w[:, j, None, :]).astype(np.float32)
    while acc.shape[0] > 1:
        if acc.shape[0] % 2:
            acc = np.concatenate([acc, np.zeros((1,) + acc.shape[1:], dtype=np.float32)])
        acc = (acc[0::2] + acc[1::2]).astype(np.float32)
    return acc[0]


@pytest.mark.parametrize("block,layer", [(1, 1), (2, 12), (3, 24)])
def test_layer_products_within_fp32_chain_error(gpu, block, layer):
    """GEMM-level gate on the PRODUCT's kernels at real layer shapes (K = 64 / 480 / 992 for the 1x1, K = 1152 for the 3x3):
    the raw output of one dense layer's conv1 (BN + ReLU + scaled two-piece fp16 split + three MFMA terms, operand kind 3) and
    conv2, read back from the engine, against the same layer evaluated in fp64 from the engine's own input buffer.  The error must stay within 2x
    that of a plain fp32 multiply-add chain over the same operands - i.e. the split products are fp32-class arithmetic,
    independent of how the gradient gates of the end-to-end tests are set."""
    net = product_net(0)
    x, mx = scene_tensors(0, [0])
    net.forward(x, mx, 0, True, 3)
    eng = engine_of(net)
    NS, H, HWp = eng.max_streams, eng.H[1 + block], eng.HWp[1 + block]
    Ct = (256, 512, 1024, 1024)[block - 1]
    cin = (64, 128, 256, 512)[block - 1] + 32 * (layer - 1)
    X = eng.debug_read("x%d" % block).reshape(NS, HWp, Ct)[0, :H * H, :]
    BT = eng.debug_read("bt%d_%d" % (block, layer)).reshape(NS, HWp, 128)[0, :H * H, :]
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    pre = "grasp_depth_trunk.features.denseblock%d.denselayer%d." % (block, layer)

    def bn_relu(v, g, b, dt):
        v64 = v.astype(np.float64)
        mean, var = v64.mean(0), v64.var(0)
        inv = 1.0 / np.sqrt(var + 1e-5)
        if dt == np.float64:
            return np.maximum((v64 - mean) * inv * g.astype(np.float64) + b.astype(np.float64), 0.0)
        sc = (g.astype(np.float64) * inv).astype(np.float32)
        return np.maximum((v - mean.astype(np.float32)) * sc + b, np.float32(0)).astype(np.float32)

    # conv1: 1x1, cin -> 128
    w1 = sd[pre + "conv1.weight"].reshape(128, cin)
    a64 = bn_relu(X[:, :cin], sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], np.float64)
    a32 = bn_relu(X[:, :cin], sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], np.float32)
    ref = a64 @ w1.astype(np.float64).T
    mag = np.abs(a64) @ np.abs(w1.astype(np.float64)).T + 1e-30
    e_prod = np.sqrt((((BT.astype(np.float64) - ref) / mag) ** 2).mean())
    e_chain = np.sqrt((((_fp32_chain(a32, w1).astype(np.float64) - ref) / mag) ** 2).mean())
    print("conv1 block %d layer %d K %d: err/sum|ab| rms product %.3e, fp32 chain %.3e" % (block, layer, cin, e_prod, e_chain))
    assert e_prod <= 2.0 * e_chain, (e_prod, e_chain)

    # conv2: 3x3 pad 1, 128 -> 32 (the layer's slice of the block buffer), on a 48-row band (keeps the fp32 chain cheap)
    w2 = sd[pre + "conv2.weight"]                                      # [32][128][3][3]
    b64 = bn_relu(BT, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], np.float64).reshape(H, H, 128)
    b32 = bn_relu(BT, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], np.float32).reshape(H, H, 128)
    rows = min(H, 48)

    def im2col(a):
        p = np.zeros((H + 2, H + 2, 128), dtype=a.dtype)
        p[1:-1, 1:-1] = a
        return np.concatenate([p[dy:dy + rows, dx:dx + H].reshape(rows * H, 128) for dy in range(3) for dx in range(3)], axis=1)
    wk = np.concatenate([w2[:, :, dy, dx] for dy in range(3) for dx in range(3)], axis=1)     # [32][9*128], tap-major like im2col
    c64, c32 = im2col(b64), im2col(b32)
    out = X[:, cin:cin + 32].reshape(H, H, 32)[:rows].reshape(rows * H, 32)
    ref = c64 @ wk.astype(np.float64).T
    mag = np.abs(c64) @ np.abs(wk.astype(np.float64)).T + 1e-30
    e_prod = np.sqrt((((out.astype(np.float64) - ref) / mag) ** 2).mean())
    e_chain = np.sqrt((((_fp32_chain(c32, wk).astype(np.float64) - ref) / mag) ** 2).mean())
    print("conv2 block %d layer %d K 1152: err/sum|ab| rms product %.3e, fp32 chain %.3e" % (block, layer, e_prod, e_chain))
    assert e_prod <= 2.0 * e_chain, (e_prod, e_chain)


def _bn_stats64(v):
    """per-channel (mean, invstd) of one stream's plane [pixels][C], in fp64 (training-mode BN, biased variance, eps 1e-5)"""
    v64 = v.astype(np.float64)
    return v64.mean(0), 1.0 / np.sqrt(v64.var(0) + 1e-5)


@pytest.mark.parametrize("block,layer", [(1, 1), (2, 12), (3, 24)])
def test_backward_layer_products_within_fp32_chain_error(gpu, block, layer):
    """GEMM-level gate on the BACKWARD products (K = 64 / 480 / 992 layers, like the forward gate above): the backward stops behind
    one dense layer (engine option debug_stop) and the test reads what that layer's kernels consumed and produced - GS and D2 from
    their ring slots (D2 rebuilt from its fp16 units and block scales: the operand the MFMAs see), the raw 3x3 data gradient, the
    block's G' in front of and behind the layer's 1x1 data gradients, the two weight gradients - and recomputes the four products
    in fp64 from those buffers: conv2 data gradient, conv2 weight gradient, conv1 data gradient (per-layer or layer-grouped form),
    conv1 weight gradient.  Each must stay within 2x the error of a plain fp32 multiply-add chain over the same operands.
    Sample 3 enters with dq = 1e-6 (2^-20 of the others): its stream's gradient operands sit 20 binades below the rest."""
    import synthetic
    net = product_net(0)
    depth, masks = synthetic.heightmap_scene(0)
    hm = torch.from_numpy(np.stack([depth, depth * masks[0]])).to(gpu)
    rots = [0, 3, 7, 10, 13]
    NSu = len(rots) + 1
    b, i = block - 1, layer - 1
    L = (6, 12, 24, 16)[b]
    q = net.run(0, rots, 16, heightmaps=hm, mean=MEAN, std=STD, keep_for_backward=True)
    eng = engine_of(net)
    dq = torch.tensor([1.0, -1.0, 0.5, 1e-6, -0.25], device=gpu).view(5, 1, 1, 1)
    net.zero_grad()
    eng.set_option("debug_stop", b * 100 + i)
    try:
        net._engine_backward(net._saved[1], dq)
    finally:
        eng.set_option("debug_stop", -1)
    torch.cuda.synchronize()
    H, HWp = eng.H[1 + block], eng.HWp[1 + block]
    HW = H * H
    Ct = (256, 512, 1024, 1024)[b]
    cin = (64, 128, 256, 512)[b] + 32 * i

    def rd(name, C):
        return eng.debug_read(name, count=NSu * HWp * C).reshape(NSu, HWp, C)[:, :HW, :]
    X, BT = rd("x%d" % block, Ct), rd("bt%d_%d" % (block, layer), 128)
    GS, DY = rd("gs_%d_%d" % (block, layer), 32), rd("dy2", 128)
    D2 = rd("d2_%d_%d" % (block, layer), 128)
    G0, G1 = rd("gsnap", Ct), rd("g%d" % block, Ct)
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    pre = "grasp_depth_trunk.features.denseblock%d.denselayer%d." % (block, layer)
    grads = {n: p.grad.detach().cpu().numpy() for n, p in net.named_parameters() if p.grad is not None}

    def check(what, got, ref, mag, chain, keep=None):
        keep = np.ones(ref.shape, dtype=bool) if keep is None else keep
        e_prod = np.sqrt(((((got.astype(np.float64) - ref) / mag) ** 2)[keep]).mean())
        e_chain = np.sqrt(((((chain.astype(np.float64) - ref) / mag) ** 2)[keep]).mean())
        print("%s block %d layer %d: err/sum|ab| rms product %.3e, fp32 chain %.3e (%d elements)" % (what, block, layer, e_prod, e_chain, int(keep.sum())))
        assert e_prod <= 2.0 * e_chain, (what, e_prod, e_chain)

    def bn_act(v, g, bta, dt):
        """relu(bn(v)) of one stream's plane, fp64 or the kernels' fp32 form; also the fp64 pre-activation (borderline test)"""
        mean, inv = _bn_stats64(v)
        pre64 = (v.astype(np.float64) - mean) * inv * g.astype(np.float64) + bta.astype(np.float64)
        if dt == np.float64:
            return np.maximum(pre64, 0.0), pre64
        sc = (g.astype(np.float64) * inv).astype(np.float32)
        return np.maximum((v - mean.astype(np.float32)) * sc + bta, np.float32(0)).astype(np.float32), pre64

    g2, b2 = sd[pre + "norm2.weight"], sd[pre + "norm2.bias"]
    w2 = sd[pre + "conv2.weight"]                                      # [32][128][3][3]
    rows = min(H, 24)

    # ---- conv2 data gradient on a band of rows of stream 0 and of the 2^-20 stream: dy[p][k] = mask * sum_{tap, c} gs[p - d(tap)][c] * w2[c][k][tap]
    wk_d = np.concatenate([w2[:, :, dy, dx].T for dy in range(3) for dx in range(3)], axis=1)      # [128][9 * 32], taps in the order of the gather below
    for n in (0, 3):
        gs = GS[n].reshape(H, H, 32)
        pad = np.zeros((H + 2, H + 2, 32), dtype=np.float32)
        pad[1:-1, 1:-1] = gs
        col = np.concatenate([pad[2 - dy:2 - dy + rows, 2 - dx:2 - dx + H].reshape(rows * H, 32) for dy in range(3) for dx in range(3)], axis=1)
        _, pre64 = bn_act(BT[n], g2, b2, np.float64)
        pre_b = pre64.reshape(H, H, 128)[:rows].reshape(rows * H, 128)
        scale_b = np.abs(BT[n].astype(np.float64) - _bn_stats64(BT[n])[0]).reshape(H, H, 128)[:rows].reshape(rows * H, 128) * np.abs(g2) * _bn_stats64(BT[n])[1] + np.abs(b2)
        on = pre_b > 1e-5 * scale_b                                      # clearly inside the ReLU (borderline signs are the mask's business, not the GEMM's)
        ref = col.astype(np.float64) @ wk_d.astype(np.float64).T
        mag = np.abs(col).astype(np.float64) @ np.abs(wk_d).astype(np.float64).T + 1e-300
        got = DY[n].reshape(H, H, 128)[:rows].reshape(rows * H, 128)
        check("conv2 dgrad (stream %d)" % n, got, ref, mag, _fp32_chain(col, wk_d), keep=on)

    # ---- norm2 backward as stored for the three 1x1 consumers (units + block scales) against fp64 from the raw gradient
    for n in (0, 3):
        mean, inv = _bn_stats64(BT[n])
        xh = (BT[n].astype(np.float64) - mean) * inv
        dy = DY[n].astype(np.float64)
        ref = g2.astype(np.float64) * inv * (dy - dy.mean(0) - xh * (dy * xh).mean(0))
        blk = np.abs(ref).reshape(-1, 64, 128).max(axis=(1, 2), keepdims=True) if HW % 64 == 0 else np.abs(ref).max()
        err = np.abs(D2[n].astype(np.float64) - ref)
        rel = (err.reshape(-1, 64, 128) / np.maximum(blk, 1e-300)).max() if HW % 64 == 0 else (err / blk).max()
        print("norm2 backward in unit form, stream %d: max |err| / block max %.3e" % (n, rel))
        assert rel <= 2e-5, rel                                          # fp32 evaluation of the affine form + 22-bit pieces (2^-22 = 2.4e-7 of the block's maximum)

    # ---- conv2 weight gradient: dW2[c][k][tap] = sum_{stream, p} gs[p][c] * relu(bn2(bt))[p + d(tap)][k]
    a64s, a32s = [], []
    for n in range(NSu):
        a64, _ = bn_act(BT[n], g2, b2, np.float64)
        a32, _ = bn_act(BT[n], g2, b2, np.float32)
        for a, lst in ((a64, a64s), (a32, a32s)):
            padk = np.zeros((H + 2, H + 2, 128), dtype=a.dtype)
            padk[1:-1, 1:-1] = a.reshape(H, H, 128)
            lst.append(np.concatenate([padk[dy:dy + H, dx:dx + H].reshape(HW, 128) for dy in range(3) for dx in range(3)], axis=1))
    A64, A32 = np.concatenate(a64s), np.concatenate(a32s)                # [streams * HW][9 * 128]
    Gs = GS.reshape(NSu * HW, 32)
    # Yardstick: a BLOCKED fp32 reduction over ALL streams x pixels (64-term chains, then pairwise: _fp32_blocked) on a subset of the
    # output elements - every 9th of the 9 * 128 columns - with the product's error on the same elements within 3x it.  (Until round
    # 5 the yardstick was ONE sequential chain over a thinned reduction, scaled by sqrt(length): 50 - 130x slack on these lengths.)
    csel = slice(0, None, 9)
    ref = Gs.astype(np.float64).T @ A64[:, csel]
    mag = np.abs(Gs).astype(np.float64).T @ np.abs(A64[:, csel]) + 1e-300
    got = grads[pre + "conv2.weight"].transpose(0, 2, 3, 1).reshape(32, 9 * 128)[:, csel]      # [c][tap][k] like the im2col columns
    e_prod = np.sqrt((((got.astype(np.float64) - ref) / mag) ** 2).mean())
    e_blk = {bl: np.sqrt((((_fp32_blocked(Gs, np.ascontiguousarray(A32[:, csel]), bl).astype(np.float64) - ref) / mag) ** 2).mean()) for bl in (64, 512)}
    print("conv2 wgrad block %d layer %d: err/sum|ab| rms product %.3e, blocked fp32 (chains + pairwise) 64-term %.3e / 512-term %.3e over all %d terms, %d elements: ratios %.2f / %.2f"
          % (block, layer, e_prod, e_blk[64], e_blk[512], NSu * HW, ref.size, e_prod / e_blk[64], e_prod / e_blk[512]))
    assert e_prod <= _WGRAD_GATE * e_blk[_WGRAD_BLK], ("conv2 wgrad", e_prod, e_blk)

    # ---- conv1 data gradient: G'[p][c] += gamma1[c] * relu'(bn1(x))[p][c] * sum_k D2[p][k] * w1[k][c]
    g_lo = i - (0 if (L - 1 - i) % 4 == 3 else min(i, 3 - (L - 1 - i) % 4))
    cs = (64, 128, 256, 512)[b] + 32 * g_lo
    segs = []                                                             # (layer, column range) pairs whose contribution landed between the snapshots
    if cin > cs:
        segs.append(([layer], cs, cin))
    if i == g_lo:
        g_hi = L - 1 - ((L - 1 - g_lo) // 4) * 4
        segs.append((list(range(g_lo + 1, g_hi + 2)), 0, cs))
    for layers, c0, c1 in segs:
        for n in (0, 3):
            take = slice(0, min(HW, 4096))
            ref = G0[n][take, c0:c1].astype(np.float64)
            mag = np.abs(ref) + 1e-300
            ch = G0[n][take, c0:c1].copy()
            keep = np.ones(ref.shape, dtype=bool)
            mean, inv = _bn_stats64(X[n][:, c0:c1])
            for lyr in layers:
                prl = "grasp_depth_trunk.features.denseblock%d.denselayer%d." % (block, lyr)
                g1, b1_ = sd[prl + "norm1.weight"][c0:c1], sd[prl + "norm1.bias"][c0:c1]
                w1 = sd[prl + "conv1.weight"].reshape(128, -1)[:, c0:c1]                   # [128][cols]
                d2 = rd("d2_%d_%d" % (block, lyr), 128)[n][take]
                pre64 = (X[n][take, c0:c1].astype(np.float64) - mean) * inv * g1 + b1_
                sc = np.abs(X[n][take, c0:c1].astype(np.float64) - mean) * inv * np.abs(g1) + np.abs(b1_)
                keep &= np.abs(pre64) > 1e-5 * sc
                m = pre64 > 0
                ref = ref + g1.astype(np.float64) * m * (d2.astype(np.float64) @ w1.astype(np.float64))
                mag = mag + np.abs(g1).astype(np.float64) * m * (np.abs(d2).astype(np.float64) @ np.abs(w1).astype(np.float64))
                ch = (ch + (g1 * m).astype(np.float32) * _fp32_chain(d2, np.ascontiguousarray(w1.T))).astype(np.float32)
            check("conv1 dgrad (stream %d, layers %s, columns %d..%d)" % (n, layers, c0, c1), G1[n][take, c0:c1], ref, mag, ch, keep=keep)

    # ---- conv1 weight gradient: dW1[k][c] = sum_{stream, p} D2[p][k] * relu(bn1(x))[p][c]
    g1, b1_ = sd[pre + "norm1.weight"], sd[pre + "norm1.bias"]
    a64 = np.concatenate([bn_act(X[n][:, :cin], g1, b1_, np.float64)[0] for n in range(NSu)])
    a32 = np.concatenate([bn_act(X[n][:, :cin], g1, b1_, np.float32)[0] for n in range(NSu)])
    D2f = D2.reshape(NSu * HW, 128)
    rsel, csel = slice(0, None, 8), slice(0, None, max(1, cin // 128))      # 16 of the 128 rows x <= 128 columns, the whole reduction
    ref = D2f[:, rsel].astype(np.float64).T @ a64[:, csel]
    mag = np.abs(D2f[:, rsel]).astype(np.float64).T @ np.abs(a64[:, csel]) + 1e-300
    got = grads[pre + "conv1.weight"].reshape(128, cin)[rsel, csel]
    e_prod = np.sqrt((((got.astype(np.float64) - ref) / mag) ** 2).mean())
    e_blk = {bl: np.sqrt((((_fp32_blocked(np.ascontiguousarray(D2f[:, rsel]), np.ascontiguousarray(a32[:, csel]), bl).astype(np.float64) - ref) / mag) ** 2).mean()) for bl in (64, 512)}
    print("conv1 wgrad block %d layer %d K %d: err/sum|ab| rms product %.3e, blocked fp32 (chains + pairwise) 64-term %.3e / 512-term %.3e over all %d terms, %d elements: ratios %.2f / %.2f"
          % (block, layer, cin, e_prod, e_blk[64], e_blk[512], NSu * HW, ref.size, e_prod / e_blk[64], e_prod / e_blk[512]))
    assert e_prod <= _WGRAD_GATE * e_blk[_WGRAD_BLK], ("conv1 wgrad", e_prod, e_blk)


def test_padded_rows_of_a_near_constant_channel_stay_finite(gpu):
    """Planes whose pixel count is not a multiple of 64 carry padding rows (the 20 x 20 planes of block 4: 448 rows for 400 pixels),
    which the 1x1 kernels read as they lie: zeros.  BN + ReLU of a zero on a channel with |mean| / std in the thousands, times the
    fp16 operand scale, lies past fp16's range; unclamped it became inf and - against the zero gradient rows of the padding - NaN
    in a whole column of that layer's conv1 weight gradient.  Build such a channel (a 3x3 convolution reduced to its centre tap
    over a nearly constant BN + ReLU input, with a negative mean) and require finite gradients everywhere."""
    import synthetic
    net = product_net(0)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    pre = "grasp_depth_trunk.features.denseblock4.denselayer3."
    with torch.no_grad():
        sd[pre + "norm2.weight"].fill_(1e-4)          # relu(1e-4 * xhat + 1) ~ 1: a nearly constant input of conv2
        sd[pre + "norm2.bias"].fill_(1.0)
        w = sd[pre + "conv2.weight"]
        w.zero_()
        w[:, :, 1, 1] = -0.5                            # centre tap only (no border effect), negative: mean ~ -64 at a standard deviation of ~ sqrt(eps)
        # the next layers read those 32 channels through their norm1: positive gamma -> relu(gamma * (0 - mean) * invstd + beta) is huge on padding rows
        for l in (4, 5):
            sd["grasp_depth_trunk.features.denseblock4.denselayer%d.norm1.weight" % l].abs_().clamp_(min=0.5)
    net.load_state_dict(sd)
    depth, masks = synthetic.heightmap_scene(0)
    hm = torch.from_numpy(np.stack([depth, depth * masks[0]])).to(gpu)
    rots = [0, 3, 7, 10, 13]
    q = net.run(0, rots, 16, heightmaps=hm, mean=MEAN, std=STD, keep_for_backward=True)
    assert torch.isfinite(q).all()
    eng = engine_of(net)
    X4 = eng.debug_read("x4", count=6 * eng.HWp[5] * 1024).reshape(6, eng.HWp[5], 1024)[:, :400, 512 + 64:512 + 96].astype(np.float64)
    ratio = np.abs(X4.mean(1)) / np.sqrt(X4.var(1) + 1e-5)
    print("constructed channels: |mean| / std up to %.0f" % ratio.max())
    assert ratio.max() > 17000, ratio.max()             # gamma (>= 0.5) * ratio * operand scale (>= 8) past 65504: the unclamped operand overflows fp16
    net.zero_grad()
    net._engine_backward(net._saved[1], torch.tensor([1.0, -1.0, 0.5, 0.3, -0.25], device=gpu).view(5, 1, 1, 1))
    torch.cuda.synchronize()
    bad = [n for n, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    assert not bad, bad[:5]


def test_forward_is_bit_reproducible_run_to_run(gpu):
    """The same 17-stream sweep three times in one process: every dense-block buffer, every bottleneck and the Q values bit for
    bit equal to the first run; the fp64 BN sums (accumulated with fp64 atomics, whose order is not fixed) equal to rounding.
    This is the detector of DESIGN.md 3.1's compiler fact (hipcc's SLP vectoriser packing unrelated fp32 scalars of the epilogues:
    BN sums off by 1e-3 run to run with bit-identical stored values) - the build keeps -fno-slp-vectorize because of it."""
    import models
    import synthetic
    from trainer import Trainer
    tr = Trainer('reinforcement', 0.5, False, None, False)
    sd = synthetic.make_state_dict(orc.state_layout(1), 0)
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    tr.model.gnum_rotations = tr.model.snum_rotations = 16
    depth, masks = synthetic.heightmap_scene(0)
    hm = tr._heightmaps_to_device(depth, depth * masks[0])
    names = ["x1", "x2", "x3", "x4"] + ["bt%d_%d" % (b + 1, i + 1) for b in range(4) for i in ((0, 5), (0, 11), (0, 23), (0, 15))[b]]
    fs_names = ["fs_bt1_1", "fs_bt2_12", "fs_bt4_16"]
    ref = None
    for it in range(3):
        q = tr.model.run(0, list(range(16)), 16, heightmaps=hm, mean=MEAN, std=STD, update_bn=False)
        torch.cuda.synchronize()
        eng = models._ENGINES[(0, 640, 1)]
        def count(n):      # the 17 streams of the sweep (the engine may be sized for more)
            blk = int(n[1] if n[0] == 'x' else n[2])
            return 17 * eng.HWp[1 + blk] * ((256, 512, 1024, 1024)[blk - 1] if n[0] == 'x' else 128)
        cur = {n: eng.debug_read(n, count=count(n)).copy() for n in names}
        cur["q"] = q.reshape(-1).cpu().numpy()
        fs = {n: eng.debug_read(n).view(np.float64).copy() for n in fs_names}
        if ref is None:
            ref, fs_ref = cur, fs
            continue
        for n in names + ["q"]:
            assert np.array_equal(cur[n].view(np.uint32), ref[n].view(np.uint32)), "run %d: %s differs (%d elements)" % (it, n, int((cur[n] != ref[n]).sum()))
        for n in fs_names:
            a, b_ = fs[n].reshape(2, -1, 128)[:, :17], fs_ref[n].reshape(2, -1, 128)[:, :17]
            rel = float(np.max(np.abs(a - b_) / np.maximum(np.abs(b_), 1e-300)))
            assert rel <= 1e-12, (n, rel)


def test_train_step_graph_equals_eager_calls(gpu):
    """Trainer.backprop's single-sample step as one replayed hipGraph (smg_train_step_graph: eager first call, captured second,
    replays after) against the four separate engine calls: with the learning rate at zero the weights stay put, so every step's
    loss must be BIT-identical to the eager path's - rotations, labels, masks and the style change from step to step (a style
    change re-captures) - and the gradients equal to the atomics' order; with the reference's learning rate the weights after
    eight steps agree to 3e-4 of their norm and the optimizer's step counts match."""
    import synthetic
    from trainer import Trainer
    depth, masks = synthetic.heightmap_scene(0)
    sd = synthetic.make_state_dict(orc.state_layout(1), 0)
    steps = [('grasp', 3, 0.4, 0), ('grasp', 9, 1.7, 1), ('grasp', 14, 0.05, 2), ('suction', 5, 0.9, 0), ('suction', 6, 2.5, 1),
             ('grasp', 1, 0.3, 3), ('grasp_then_suction', 0, 0.7, 2), ('grasp', 2, 0.6, 0)]

    def run(graph, lr):
        tr = Trainer('reinforcement', 0.5, False, None, False)
        tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        tr.model.gnum_rotations = tr.model.snum_rotations = 16
        tr.use_step_graph = graph
        tr.optimizer.lr = lr
        losses, grads = [], []
        for action, rot, label, mi in steps:
            mk = masks.copy()
            losses.append(np.float32(tr.backprop(depth, action, (mi, rot), (mi, rot), (mi, rot), ((mi + 1) % 4, rot), label, mk, None, None, None)))
            grads.append(tr.model.flat_grads().clone())
        return tr, losses, grads

    tr_e, l_e, g_e = run(False, 0.0)
    tr_g, l_g, g_g = run(True, 0.0)
    for k, (a, b_) in enumerate(zip(l_e, l_g)):
        assert a.tobytes() == b_.tobytes(), "step %d: loss %r (eager) vs %r (graph)" % (k, a, b_)
    for k, (a, b_) in enumerate(zip(g_e, g_g)):
        num, den = float((a - b_).double().norm()), float(a.double().norm())
        assert num <= 1e-5 * den, (k, num, den)
    assert tr_e.optimizer.steps == tr_g.optimizer.steps
    for n_, p_ in tr_g.model.named_parameters():           # only the last step's (trunk, head) carries gradients, like torch's zero_grad(set_to_none)
        assert (p_.grad is not None) == (("grasp_depth_trunk.features" in n_) or ("graspnet_val" in n_)), n_
    tr_e, l_e, _ = run(False, 1e-4)
    tr_g, l_g, _ = run(True, 1e-4)
    assert l_e[0].tobytes() == l_g[0].tobytes()
    # (two eager runs differ as much: the order of the fp32 atomics feeds Adam, whose first steps move every parameter by ~lr whatever
    #  its gradient's size - Q values drift apart by up to 1e-2 over the eight steps, measured in one of six repeats: loss 0.0689 against
    #  0.0717 at step 6 - so this half only checks that the graph's optimizer follows the same trajectory; the exactness check is the
    #  zero-learning-rate half above)
    assert np.allclose(l_e, l_g, rtol=5e-2, atol=1e-2), (l_e, l_g)
    pe, pg = tr_e.model._flat_params.double(), tr_g.model._flat_params.double()
    # (Adam's first steps move every parameter by ~lr whatever its gradient's size: where the gradient is noise, the atomics' order
    #  decides the sign - 6e-5 of the norm measured after eight steps, the same between two eager runs)
    assert float((pe - pg).norm()) <= 3e-4 * float(pe.norm())
    assert tr_e.optimizer.steps == tr_g.optimizer.steps


def test_deterministic_option_gives_bit_identical_conv_weight_gradients(gpu):
    """smg_engine_set_option("deterministic", 1): the 1x1 weight gradients are reduced from partial tiles in a fixed order
    instead of fp32 atomics.  Two identical training calls then give bit-identical gradients for every convolution weight
    of the trunk and the head's conv0 (the reference's backward on one device is deterministic, code/trainer.py:350-351);
    without the switch the 1x1 weight gradients differ in the last bits.  (BatchNorm affine gradients and the head's 20x20 value
    convolution are still summed with fp32 atomics in either mode: equal to 1e-5.)"""
    from trainer import Trainer
    import synthetic
    import models
    tr = Trainer('reinforcement', 0.5, False, None, False)
    sd = synthetic.make_state_dict(orc.state_layout(1), 5)
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    tr.model.gnum_rotations = tr.model.snum_rotations = 16
    tr.optimizer.lr = 0.0
    depth, masks = synthetic.heightmap_scene(6)
    rots, labels = [0, 3, 7, 12], [0.2, 1.7, 0.6, 0.9]
    tr.train_batch(depth, depth * masks[0], 0, rots, labels)
    eng = engine_of(tr.model)
    # every convolution of the trunk (120 tensors, 6.9 M of the 7.1 M gradient elements) and the head's 1x1 conv0; the head's 20x20
    # value convolution sums its (pair, pixel) terms with fp32 atomics like the BN affine gradients
    conv = [(n, p) for n, p in tr.model.named_parameters()
            if p.dim() == 4 and n.startswith(("grasp_depth_trunk", "graspnet_val")) and not n.endswith("val-conv1.weight")]
    assert len(conv) == 121

    bn = [(n, p) for n, p in tr.model.named_parameters()
          if p.dim() == 1 and "norm" in n and n.startswith(("grasp_depth_trunk", "graspnet_val"))]
    last_bn = {}

    def run():
        tr.train_batch(depth, depth * masks[0], 0, rots, labels)
        last_bn.clear(); last_bn.update({n: p.grad.clone() for n, p in bn})
        return {n: p.grad.clone() for n, p in conv}, tr.model.flat_grads().clone()
    try:
        eng.set_option("deterministic", 1)
        g1, f1 = run()
        b1_ = dict(last_bn)
        g2, f2 = run()
        b2_ = dict(last_bn)
    finally:
        eng.set_option("deterministic", 0)
    differing = [n for n in g1 if not torch.equal(g1[n], g2[n])]
    assert not differing, differing[:5]
    assert float((f1 - f2).double().norm()) <= 1e-5 * float(f1.double().norm())
    # The backward's run-to-run detector (the forward has test_forward_is_bit_reproducible_run_to_run): every BN-backward statistic that
    # is USED downstream is covered bit for bit by the convolution gradients above - the norm2 sums s1 / s2 shape D2 (the operand of that
    # layer's 1x1 weight and data gradients), the norm1 sums shape the slice gradients GS of every layer below (the operand of their 3x3
    # gradients) - so a statistic that varied from run to run (the -fslp-vectorize hazard of DESIGN.md section 4 moved forward sums by
    # 1e-3) would change convolution gradients bit-wise.  The BN affine gradients themselves are terminal sums over streams and
    # workgroups (fp32 atomics, order-dependent in the last bits): each tensor is held to 2e-5 of its norm here, per tensor, which
    # is 50x below what the hazard produced.
    gmax = max(float(v.double().norm()) for v in b1_.values())
    worst_bn = 0.0
    assert len(b1_) == 2 * 121 + 2 * 2          # norm0, 58 x (norm1, norm2), 3 transition norms, norm5; the head's two norms
    for n in b1_:
        d, nr = float((b1_[n] - b2_[n]).double().norm()), float(b1_[n].double().norm())
        worst_bn = max(worst_bn, d / (nr + 5e-3 * gmax))
        assert d <= 2e-5 * nr + 1e-7 * gmax, (n, d, nr)      # (norm5.weight's gradient is ~0 mathematically - the head's BN removes the scale: 1.9e-7 of noise on |g| 7.6e-5, gmax 8.1)
    print("deterministic: 121 convolution gradients bit-identical; BN affine gradients differ by at most %.1e of (their tensor's norm + 0.5 %% of the largest) between two runs" % worst_bn)
    g3, f3 = run()                                            # default mode: same values up to the atomics' order
    assert float((f1 - f3).double().norm()) <= 1e-5 * float(f1.double().norm())
    with pytest.raises(Exception):
        eng.set_option("no-such-option", 1)


def test_argmax_matches_numpy_including_nan(gpu):
    """smg_argmax = np.argmax (code/main.py:172-173): lowest index on ties, and the first NaN wins."""
    import smg_hip
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(3)
    cases = [rng.randn(1000).astype(np.float32), np.zeros(700, np.float32), np.asarray([-np.inf] * 5, np.float32)]
    a = rng.randn(513).astype(np.float32); a[77] = a[400] = a.max() + 1; cases.append(a)
    b = rng.randn(900).astype(np.float32); b[650] = np.nan; b[301] = np.nan; b[5] = np.inf; cases.append(b)
    c = rng.randn(300).astype(np.float32); c[299] = np.nan; cases.append(c)
    for v in cases:
        t = torch.from_numpy(v).to(dev)
        idx = torch.zeros(1, dtype=torch.int32, device=dev); val = torch.zeros(1, dtype=torch.float32, device=dev)
        smg_hip.argmax(t.data_ptr(), t.numel(), idx.data_ptr(), val.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        want = int(np.argmax(v))
        assert int(idx.item()) == want, (int(idx.item()), want)
        got = float(val.item())
        assert (np.isnan(got) and np.isnan(v[want])) or got == float(v[want])


def test_reactive_labels_outside_the_classes_are_rejected(gpu):
    from trainer import Trainer
    import synthetic
    tr = Trainer('reactive', 0.5, False, None, False)
    depth, masks = synthetic.heightmap_scene(0)
    for bad in (3, -1, 0.5, float("nan")):
        with pytest.raises(ValueError):
            tr.train_batch(depth, depth * masks[0], 0, [0], [bad])
    loss = tr.train_batch(depth, depth * masks[0], 0, [0], [2])          # class 2 = "no loss" (weight 0)
    assert float(loss[0]) == 0.0


def test_bench_two_ranks_config4_leg_on_one_gpu(gpu):
    """`python bench.py --gpus 2 --leg config4`: the launcher starts two ranks as a child process (here both on this GPU,
    gradients exchanged over gloo - RCCL wants one GPU per rank), every rank trains on its own scenes, one all-reduce
    between backward and Adam; the N > 1 line carries n_gpus, the all-reduce time and the whole-job step roofline."""
    import json
    import os
    import subprocess
    import sys
    from helpers import REPO
    env = dict(os.environ, SMG_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--leg", "config4", "--scenes-per-rank", "2",
                        "--steps", "2", "--warmup", "1", "--train-only"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["leg"] == "config4" and out["scaling"] == "weak"
    assert out["allreduce_ms"] > 0 and out["allreduce_backend"] == "gloo" and out["allreduce_overlapped"] is True
    assert out["allreduce_exposed_ms_per_step"] > 0 and out["allreduce_bytes"] == 4 * (6953856 + 160896)
    st = out["roofline"]["step"]
    assert 0 < st["frac_of_mfma_roof"] < 1 and st["mfma_roof_tflops"] > 800          # 2 x 416.7
    assert abs(out["value"] - 2 * 2 * 1e3 / out["ms_per_step"]) <= 1e-4 * out["value"]    # passes/s of the whole job (the compact line rounds to 4 decimals)


def test_bench_two_ranks_strong_scaling_leg_on_one_gpu(gpu):
    """`python bench.py --gpus 2 --scaling strong`: ONE scene per step, its 16 rotations dealt contiguously to the ranks (8 each + the
    masked stream on both), one all-reduce, one Adam; whole-job value = one pass per step.  The N > 1 line reports what the process
    group saw (rccl_world, device_count, devices_seen) - here two ranks sharing this box's one GPU over gloo."""
    import json
    import os
    import subprocess
    import sys
    from helpers import REPO
    env = dict(os.environ, SMG_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--scaling", "strong", "--steps", "2", "--warmup", "1",
                        "--train-only"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    assert len(line) < 2000
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["leg"] == "headline"
    assert out["rccl_world"] == 2 and out["device_count"] >= 1 and out["devices_seen"] >= 1 and out["allreduce_backend"] == "gloo"
    assert abs(out["config"]["passes_per_step_per_gpu"] - 0.5) < 1e-9
    assert abs(out["value"] - 1e3 / out["ms_per_step"]) <= 1e-4 * out["value"]          # one pass per step for the whole job (the compact line rounds to 4 decimals)


# ---------------------------------------------------------------------------------------
# RCCL on the one GPU of the test box: a process group of ONE rank over backend "nccl" (= RCCL on ROCm).  The collective
# is an identity there, but everything around it is what an 8-GPU job runs: ProcessGroupNCCL's communicator and stream,
# dist.all_reduce(view) / async_op=True + work.wait() on the flat gradient ranges (parallel.py), the ordering of RCCL's
# stream against the engine's launch stream and its internal side stream, smg_backward_phase's two halves.
_RCCL_WORLD1_SCRIPT = r"""
import json, os, sys
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, %(tests)r)
from helpers import orc
import parallel, synthetic
from trainer import Trainer
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
sd = synthetic.make_state_dict(orc.state_layout(1), 3)
depth, masks = synthetic.heightmap_scene(5)
rots, labels = [0, 3, 7, 9, 12, 15], [0.2, 1.4, 0.9, 3.0, 0.1, 0.7]
res = {}
calls = {"n": 0}
real_all_reduce = dist.all_reduce
def counting_all_reduce(*a, **k):
    calls["n"] += 1
    return real_all_reduce(*a, **k)
dist.all_reduce = counting_all_reduce
for tag, make_sync in (("none", lambda: None), ("blocking", lambda: parallel.allreduce_grads), ("overlapped", lambda: parallel.OverlappedGradSync())):
    tr = Trainer('reinforcement', 0.5, False, None, False)
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    tr.model.gnum_rotations = tr.model.snum_rotations = 16
    calls["n"] = 0
    losses = []
    for step in range(2):                       # two Adam steps: the second one's forward reads the first one's weights
        loss = tr.train_batch(depth, depth * masks[1], 0, rots, labels, grad_sync=make_sync())
        losses.append(np.asarray(loss.cpu() if torch.is_tensor(loss) else loss, dtype=np.float64).ravel().tolist())
        if step == 0:                           # gradients of the FIRST step: identical weights on every variant
            g = tr.model.flat_grads().double().cpu().numpy()
    torch.cuda.synchronize()
    p = tr.model._flat_params.double().cpu().numpy()
    res[tag] = {"g": g, "p": p, "loss": losses, "calls": calls["n"]}
ref = res["none"]
out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
for tag in ("blocking", "overlapped"):
    r = res[tag]
    out[tag] = {"grad_rel": float(np.linalg.norm(r["g"] - ref["g"]) / np.linalg.norm(ref["g"])),
                "param_rel": float(np.linalg.norm(r["p"] - ref["p"]) / np.linalg.norm(ref["p"])),
                "loss_abs": float(np.abs(np.asarray(r["loss"]) - np.asarray(ref["loss"])).max()), "calls": r["calls"]}
out["none_calls"] = ref["calls"]
out["gnorm"] = float(np.linalg.norm(ref["g"]))
print("RESULT " + json.dumps(out))
dist.destroy_process_group()
"""


def test_rccl_world1_allreduce_paths_execute_and_leave_the_step_unchanged(gpu):
    """First execution of parallel.py's "nccl" branches (blocking: `dist.all_reduce(view)`; overlapped: `async_op=True` on RCCL's
    stream after smg_backward_phase(0), `work.wait()` in front of Adam) - on the test box's single GPU, as a one-rank process group in
    a FRESH child process (RANK=0 WORLD_SIZE=1 SMG_FORCE_ALLREDUCE=1; never a re-exec of a process that touched the GPU).  The sum
    over one rank is the identity, so two training steps with either hook must equal two steps without one: gradients to 1e-5
    (fp32 atomics order), Adam'd weights, losses - which only holds if the engine's streams and RCCL's stream are ordered correctly
    (a collective that started before the backward's first half had finished, or an Adam that did not wait for it, would reduce or
    apply stale gradients).  The reference has nothing to match here."""
    import json
    import os
    import socket
    import subprocess
    import sys
    tests_dir = os.path.dirname(os.path.abspath(__file__))
    so = socket.socket(); so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]; so.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               SMG_FORCE_ALLREDUCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_WORLD1_SCRIPT % {"tests": tests_dir}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    print("rccl world-1:", out)
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["none_calls"] == 0
    assert out["blocking"]["calls"] == 2 * 2          # (trunk range, head range) per step
    assert out["overlapped"]["calls"] == 2 * 3        # early ranges (trunk from the split on, head) + the late trunk range
    # gradients of the first step (same weights): equal to the fp32 atomics' order.  Weights after two Adam steps: a first Adam step moves
    # every weight by +-lr whatever its gradient's size, so where a gradient is noise the atomics' order picks the sign - 4.3e-6 of the
    # parameter norm measured between two runs of the SAME variant, and the second step's losses follow to ~4e-6.
    for tag in ("blocking", "overlapped"):
        assert out[tag]["grad_rel"] <= 1e-5, (tag, out[tag])
        assert out[tag]["param_rel"] <= 2e-5, (tag, out[tag])
        assert out[tag]["loss_abs"] <= 1e-4, (tag, out[tag])


def test_bench_rccl_world1_line(gpu):
    """`RANK=0 WORLD_SIZE=1 SMG_FORCE_ALLREDUCE=1 python bench.py --train-only`: the bench's distributed leg over RCCL with one rank -
    the line reports what the process group saw (`allreduce_backend == "nccl"`, `rccl_world == 1`), the blocking collective's time and
    the exposed time of the overlapped one."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from helpers import REPO
    so = socket.socket(); so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]; so.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               SMG_FORCE_ALLREDUCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SMG_BENCH_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--train-only"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print("bench rccl world-1:", {k: out.get(k) for k in ("allreduce_backend", "rccl_world", "allreduce_ms", "allreduce_exposed_ms_per_step", "ms_per_step")})
    assert out["allreduce_backend"] == "nccl" and out["rccl_world"] == 1 and out["n_gpus"] == 1
    assert out["allreduce_overlapped"] is True and out["allreduce_ms"] > 0 and out["allreduce_bytes"] == 4 * (6953856 + 160896)


@pytest.mark.parametrize("size,rots", [(232, [1, 6, 10, 13]), (256, [3, 11])])
def test_weight_gradient_chunks_never_start_in_the_plane_padding(gpu, size, rots):
    """A 232^2 heightmap -> S = 672: dense block 1's 168^2 = 28 224-pixel planes are padded to 28 288 rows (make_plane pads planes
    of 8192+ pixels to multiples of 128), so HWp - HW = 64 - a whole weight-gradient chunk granule.  Chunks sized from HWp put the
    last chunk of a stream wholly into the padding for 5 streams (chunk 448, 64 per stream: the 64th starts at row 28 224 = HW);
    its workgroups left without storing their partial tile and the fixed-order reduce added whatever the workspace held there
    (advisor, round 5: static reading, S = 640 and S = 1824 were not affected).  pick_chunk now sizes chunks over the valid rows.
    Detector: under "deterministic" the conv weight gradients of a 5-stream call must be bit-identical before and after an
    unrelated call whose gradients are 1000x larger (stale partial tiles of that call would be added), and every 1x1 weight
    gradient of dense block 1 and every transition's weight gradient must match the sum of the single-rotation calls (two streams
    each: the 1x1 weight gradients there take the atomics form, no partial tiles).
    Second case, a 256^2 heightmap -> S = 736 with 3 streams: 184^2 planes padded by 64 rows, 92^2 = 8464-pixel planes by 112 - the
    advisor's transition-1 weight gradient case (two column tiles, 64-row chunks)."""
    from trainer import Trainer
    import synthetic
    tr = Trainer('reinforcement', 0.5, False, None, False)
    sd = synthetic.make_state_dict(orc.state_layout(1), 7)
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    tr.model.gnum_rotations = tr.model.snum_rotations = 16
    tr.optimizer.lr = 0.0
    depth, masks = synthetic.heightmap_scene(11, size=size)
    depth_b, masks_b = synthetic.heightmap_scene(12, size=size)
    S = {232: 672, 256: 736}[size]
    with torch.no_grad():
        q = tr.model.run(0, [rots], 16, heightmaps=torch.from_numpy(np.stack([depth, depth * masks[0]])).cuda(), mean=tr.image_mean, std=tr.image_std)
    q0 = q.reshape(len(rots), -1)[:, 0].double().cpu().numpy()
    labels_small = (q0 - 1e-3).tolist()          # |d| = 1e-3: dL/dq = d
    labels_big = [float(v) + 50.0 for v in q0]   # |d| >= 1: dL/dq = -1, a thousand times the above
    conv = [(n, p) for n, p in tr.model.named_parameters() if p.dim() == 4 and n.startswith("grasp_depth_trunk")]
    assert len(conv) == 120

    def run(d, m, labels):
        tr.train_batch(d, m, 0, rots, labels)
        return {n: p.grad.clone() for n, p in conv}
    eng = engine_of(tr.model, S)
    try:
        eng.set_option("deterministic", 1)
        g_a = run(depth, depth * masks[0], labels_small)
        run(depth_b, depth_b * masks_b[2], labels_big)
        g_c = run(depth, depth * masks[0], labels_small)
    finally:
        eng.set_option("deterministic", 0)
    differing = [n for n in g_a if not torch.equal(g_a[n], g_c[n])]
    assert not differing, differing[:5]
    # the batch against the sum of its single-rotation calls (2 streams each: fp32 atomics, no partial-tile workspace)
    gsum = None
    for r, lab in zip(rots, labels_small):
        tr.train_batch(depth, depth * masks[0], 0, [r], [lab])
        g = {n: p.grad.double().clone() for n, p in conv}
        gsum = g if gsum is None else {n: gsum[n] + g[n] for n in g}
    worst = 0.0
    for n in g_a:
        if not ((".conv1." in n and "denseblock1" in n) or ".transition" in n):
            continue
        rel = float((g_a[n].double() - gsum[n]).norm() / gsum[n].norm())
        worst = max(worst, rel)
        assert rel <= 3e-2, (n, rel)
    print("S=%d, %d streams, deterministic: dense block 1's 1x1 and the transitions' weight gradients vs the sum of single-rotation calls: worst relative error %.2e"
          % (S, len(rots) + 1, worst))
