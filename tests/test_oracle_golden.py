"""Pins the oracle (oracle/affordance.py, oracle/densenet121.py) against vectors
captured from the reference's own Python (tests/golden/reference_vectors.npz, made by
oracle/make_golden.py).  CPU only.  Same torch CPU operators in the same order, so the
expectation is bit-identical; a 1e-6 guard band is allowed for thread-count effects."""
import zlib

import numpy as np
import pytest
import torch

import synthetic
from oracle import affordance as orc

MEAN, STD = 0.01, 0.03
TOL = dict(rtol=2e-5, atol=2e-6)


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF)


def probe_idx(n, k, tag):
    return (synthetic.uniform(1234, "probe/" + tag, k) * n).astype(np.int64)


def make_net(seed, out_ch=1, R=16):
    net = orc.OracleNet(out_ch)
    orc.load_numpy_state(net, synthetic.make_state_dict(orc.state_layout(out_ch), seed))
    net.gnum_rotations = net.snum_rotations = R
    net.train()
    return net


def scene_inputs(seed, mask_ids):
    depth, masks = synthetic.heightmap_scene(seed)
    m = sum(masks[i] for i in mask_ids)
    return (orc.preprocess(depth, [MEAN] * 3, [STD] * 3),
            orc.preprocess(depth * m, [MEAN] * 3, [STD] * 3))


def test_state_layout_counts():
    lay = orc.state_layout(1)
    assert len(lay) == 2217                      # SURVEY.md section 5 (measured on the reference)
    n_params = sum(int(np.prod(s)) for _, s, k in lay if k not in ("rm", "rv", "nbt"))
    assert n_params == 24419256


@pytest.mark.parametrize("size,R", [(640, 16), (1824, 32)])
def test_g1_rotation_index_tables(golden, size, R):
    for r in range(R):
        idx = orc.rotation_index_map(r, R, size)
        assert int((idx < 0).sum()) == int(golden["g1_oob_%d" % size][r])
        assert crc(idx) == golden["g1_crc_%d" % size][r], "rotation %d" % r


def test_g2_preprocess(golden):
    depth, masks = synthetic.heightmap_scene(0)
    rep = np.repeat(np.repeat(depth, 2, axis=0), 2, axis=1)
    assert crc(rep) == golden["g2_zoom_crc"]     # ndimage.zoom(order=0, x2) == pixel replication
    with np.errstate(all="ignore"):
        lit = orc.preprocess(depth, [0.0] * 3, [0.0] * 3).numpy()
    assert tuple(lit.shape) == tuple(golden["g2_literal_shape"])
    assert int(np.isinf(lit).sum()) == int(golden["g2_literal_ninf"])
    assert int(np.isnan(lit).sum()) == int(golden["g2_literal_nnan"])
    assert crc(np.isinf(lit).astype(np.uint8)) == golden["g2_literal_infmask_crc"]


def test_g3_trunk_stages(golden):
    net = make_net(0)
    x, _ = scene_inputs(0, [0])
    feats = net.grasp_depth_trunk.features
    with torch.no_grad():
        t = orc.rotate(x, 3, 16)
        for name, mod in feats.named_children():
            t = mod(t)
            if "g3_%s_stats" % name in golden.files:
                a = t.numpy().astype(np.float64).ravel()
                st = np.asarray([a.mean(), np.sqrt((a * a).sum()), np.abs(a).max()])
                np.testing.assert_allclose(st, golden["g3_%s_stats" % name], rtol=1e-5, atol=1e-7)
                pi = probe_idx(a.size, 32, "g3/" + name)
                np.testing.assert_allclose(t.numpy().ravel()[pi], golden["g3_%s_probe" % name], **TOL)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_g4_q_values(golden, seed):
    net = make_net(seed)
    x, mx = scene_inputs(seed, [seed % 8])
    _, mx2 = scene_inputs(seed, [1, 2])
    for style, rots in ((0, (0, 3, 9)), (1, (5,))):
        for r in rots:
            q = orc.forward(net, x, mx, style, True, r)
            assert torch.is_tensor(q) and tuple(q.shape) == (1, 1, 1, 1)
            np.testing.assert_allclose(float(q), golden["g4_s%d_q%d" % (seed, style)][r], **TOL)
    q2 = orc.forward(net, x, mx2, 2, True, -1)
    assert isinstance(q2, list) and len(q2) == 1
    np.testing.assert_allclose(float(q2[0]), golden["g4_s%d_q2" % seed][0], **TOL)


def test_g4_branches_and_g7_bn_buffers(golden):
    net = make_net(0)
    x, mx = scene_inputs(0, [0])
    qb = orc.forward(net, x, mx, 0, True, 5)
    np.testing.assert_allclose(float(qb), golden["g4_branchB_style0_rot5"], **TOL)
    np.testing.assert_allclose(float(qb), golden["g4_s0_q0"][5], **TOL)   # sweep element == branch B
    qb1 = orc.forward(net, x, mx, 1, True, 7)
    np.testing.assert_allclose(float(qb1), golden["g4_branchB_style1_rot7"], **TOL)
    sd = net.state_dict()
    keys = sorted(k[3:-3] for k in golden.files if k.startswith("g7_") and k.endswith("_rm"))
    assert keys
    for key in keys:
        np.testing.assert_allclose(sd[key + ".running_mean"].numpy(), golden["g7_%s_rm" % key], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(sd[key + ".running_var"].numpy(), golden["g7_%s_rv" % key], rtol=1e-5, atol=1e-7)
        assert int(sd[key + ".num_batches_tracked"]) == int(golden["g7_%s_nbt" % key])


def test_g5_g6_training_steps(golden):
    net = make_net(0)
    target = orc.clone_target(net)
    opt = orc.make_adam(net)
    x, mx = scene_inputs(0, [0])
    _, mx_es = scene_inputs(0, [1, 2])            # the ES step's two-object mask (code/trainer.py:370)
    names = [n for n, _ in net.named_parameters()]
    assert names == [str(s) for s in golden["g5_param_names"]]
    for si, (style, rot, label) in enumerate([(0, 3, 0.4), (1, 9, 7.5), (2, 0, -3.0)]):
        opt.zero_grad()
        q = orc.forward(net, x, mx_es if style == 2 else mx, style, False, rot)
        loss = orc.huber(q[0, 0, 0, 0], label).sum()
        loss.backward()
        np.testing.assert_allclose(float(q), golden["g5_step%d_q" % si], **TOL)
        np.testing.assert_allclose(float(loss), golden["g5_step%d_loss" % si], **TOL)
        has = np.asarray([p.grad is not None for p in net.parameters()])
        assert (has == golden["g5_step%d_hasgrad" % si]).all()
        gn = np.asarray([float(p.grad.double().norm()) if p.grad is not None else 0.0 for p in net.parameters()])
        np.testing.assert_allclose(gn, golden["g5_step%d_gradnorm" % si], rtol=1e-4, atol=1e-9)
        params = dict(net.named_parameters())
        pre = "g5_step%d_grad_" % si
        for k in [k for k in golden.files if k.startswith(pre)]:
            p = params[k[len(pre):]]
            pi = probe_idx(p.numel(), 16, "g5/" + k[len(pre):])
            np.testing.assert_allclose(p.grad.numpy().ravel()[pi], golden[k], rtol=1e-4, atol=1e-9)
        opt.step()
        pre = "g6_step%d_param_" % si
        for k in [k for k in golden.files if k.startswith(pre)]:
            p = params[k[len(pre):]]
            pi = probe_idx(p.numel(), 16, "g6/" + k[len(pre):])
            np.testing.assert_allclose(p.detach().numpy().ravel()[pi], golden[k], rtol=1e-6, atol=1e-9)
    qt = orc.forward(target, x, mx, 0, True, 3)
    qm = orc.forward(net, x, mx, 0, True, 3)
    np.testing.assert_allclose(float(qt), golden["g4_target_rot3"], **TOL)
    np.testing.assert_allclose(float(qm), golden["g4_model_after3_rot3"], rtol=1e-4, atol=1e-5)
    assert abs(float(qt) - float(qm)) > 1e-4     # the two nets really have diverged


def test_g8_reactive(golden):
    net = make_net(0, out_ch=3, R=1)
    x, mx = scene_inputs(0, [0])
    out = orc.forward(net, x, mx, 0, True, -1)
    np.testing.assert_allclose(out[0].numpy().ravel(), golden["g8_logits"], **TOL)
    sm = torch.softmax(out[0].view(1, 3, 1, 1), 1).numpy()[0, 0, 0, 0]
    np.testing.assert_allclose(sm, golden["g8_softmax0"], **TOL)
    opt = orc.make_adam(net)
    opt.zero_grad()
    q = orc.forward(net, x, mx, 0, False, 0)
    loss = orc.reactive_loss(q, 1)
    loss.backward()
    np.testing.assert_allclose(float(loss), golden["g8_loss"], **TOL)
    gn = np.asarray([float(p.grad.double().norm()) if p.grad is not None else 0.0 for p in net.parameters()])
    np.testing.assert_allclose(gn, golden["g8_gradnorm"], rtol=1e-4, atol=1e-9)


def test_label_value_known_answers():
    # code/trainer.py:238-274 reward arithmetic
    assert orc.label_value("reinforcement", "grasp", 3, 0, 1, 0, 0.8, 0.5) == (1 + 0.5 * 0.8, 1)
    assert orc.label_value("reinforcement", "grasp", 3, 0, 0, 0, 0.8, 0.5) == (0, 0)
    assert orc.label_value("reinforcement", "suction", 1, 1, 0, 0, 0.8, 0.5) == (1, 1)
    assert orc.label_value("reinforcement", "grasp_then_suction", 2, 0, 0, 2.5, 0.8, 0.5) == (2.5, 2.5)
    assert orc.label_value("reinforcement", "grasp_then_suction", 3, 0, 0, 0.5, 0.8, 0.5) == (0.5 + 0.4, 0.5)
    assert orc.label_value("reactive", "grasp", 3, 0, 0, 0, 0, 0.5) == (1, 0)
    assert orc.label_value("reactive", "grasp_then_suction", 3, 0, 0, 2.5, 0, 0.5) == (0, 2.5)


def test_heightmap_restatement_known_answers():
    """oracle/heightmap.py (utils.get_heightmap, code/utils.py:38-68; parity unpinned - OpenCV is not installed): the
    homography maps the four corners exactly, an axis-aligned crop + scale warp of a linear ramp is the ramp sampled on
    the 1/32-pixel grid, taps outside the image count as zero."""
    from oracle import heightmap as hm
    dst = np.array([[0, 0], [0, 224], [224, 224], [224, 0]], np.float32)
    m = hm.perspective_transform(hm.SRC_SIM, dst)
    for (x, y), (u, v) in zip(hm.SRC_SIM, dst):
        p = m @ np.array([x, y, 1.0])
        assert np.allclose(p[:2] / p[2], [u, v], atol=1e-9)
    assert np.allclose(hm.perspective_transform(dst, hm.SRC_SIM) @ m, np.eye(3) * (hm.perspective_transform(dst, hm.SRC_SIM) @ m)[2, 2], atol=1e-9)
    yy, xx = np.meshgrid(np.arange(480.0), np.arange(640.0), indexing="ij")
    ramp = 0.25 * xx + 2.0 * yy + 1.0
    out = hm.warp_perspective(ramp, m, (224, 224))
    # destination (x, y) samples the source at (110 + x*400/224, y*400/224), rounded to 1/32 pixel
    sx = np.rint((110 + np.arange(224) * 400.0 / 224) * 32) / 32
    sy = np.rint((np.arange(224) * 400.0 / 224) * 32) / 32
    want = 0.25 * sx[None, :] + 2.0 * sy[:, None] + 1.0
    assert np.allclose(out, want, rtol=0, atol=1e-4)          # float32 table weights: ~1e-7 relative
    shifted = hm.perspective_transform(np.array([[-10, 0], [-10, 400], [390, 400], [390, 0]], np.float32), dst)
    edge = hm.warp_perspective(np.ones((480, 640)), shifted, (224, 224))
    assert edge[100, 0] == 0.0 and edge[100, 5] == 0.0 and abs(edge[100, 6] - 1.0) < 1e-6      # x_src = -10 + x*400/224 < 0 for x <= 5
    k = np.asarray([[618.62, 0, 320], [0, 618.62, 240], [0, 0, 1]])
    pose = np.eye(4)
    pose[:3, :3] = [[1, 0, 0], [0, -1, 0], [0, 0, -1]]
    pose[:3, 3] = [-0.5, 0.0, 0.6]
    z = hm.world_z(np.full((480, 640), 0.55), k, pose)       # a plane 0.55 m in front of a downward camera 0.6 m up
    assert np.allclose(z, 0.05)


# ---------------------------------------------------------------------------------------
# The arithmetic of the HIP path's dense-layer products (csrc/gemm.cuh, operand kind 3), restated in numpy: a scaled two-piece fp16
# split with three fp32-accumulated terms h*l + l*h + h*h.  No GPU: this pins the CLAIMS the kernels rest on - the
# GPU-side measurement of the same scheme is tools/split16_probe.hip, the product's own gate test_layer_products_within_fp32_chain_error.
def _split16(x, s):
    y = (x * np.float32(s)).astype(np.float32)
    h = y.astype(np.float16)                                  # round to nearest even, subnormals kept (v_cvt_pk_f16_f32)
    l = (y - h.astype(np.float32)).astype(np.float16)         # the residual is exact in fp32
    return h.astype(np.float64), l.astype(np.float64)


def _pow2_scale(m, target):
    return 2.0 ** (target - int(np.floor(np.log2(m))))


def _chain_fp32(a, b):
    acc = np.zeros((a.shape[0], b.shape[1]), dtype=np.float32)
    for k in range(a.shape[1]):                               # one fp32 FMA per k, like v_mfma_f32_32x32x2_f32 (rounded once per step)
        acc = (acc.astype(np.float64) + a[:, k:k + 1].astype(np.float64) * b[k:k + 1, :].astype(np.float64)).astype(np.float32)
    return acc.astype(np.float64)


@pytest.mark.parametrize("K,mode", [(64, "act"), (128, "act"), (288, "grad"), (128, "small")])
def test_two_piece_fp16_split_is_fp32_class_when_scaled(K, mode):
    rng = np.random.RandomState(K + len(mode))
    if mode == "grad":                                        # gradient-like: tiny, heavy-tailed; dynamic scale from the recorded maximum
        a = (rng.randn(32, K) * 1e-6 * np.exp(2.0 * rng.randn(32, K))).astype(np.float32)
        sa = _pow2_scale(np.abs(a).max(), 13)
    else:                                                     # BN + ReLU activations; scale from hypot(gamma, beta) ~ 1.4 -> 2^4
        a = np.maximum(rng.randn(32, K) * 1.3 + 0.1, 0.0).astype(np.float32) * (1e-3 if mode == "small" else 1.0)
        sa = 16.0
    w = (rng.randn(K, 32) * np.sqrt(2.0 / K)).astype(np.float32)
    sw = _pow2_scale(np.abs(w).max(), 13)
    assert np.abs(w).max() * sw < 65504 and np.abs(a).max() * sa < 65504          # nothing overflows fp16
    ref = a.astype(np.float64) @ w.astype(np.float64)
    mag = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64)
    ah, al = _split16(a, sa)
    wh, wl = _split16(w, sw)
    got = (ah @ wl + al @ wh + ah @ wh) / (sa * sw)           # three terms; the inverse scales are exact powers of two
    e_split = np.sqrt((((got - ref) / mag) ** 2).mean())
    e_chain = np.sqrt((((_chain_fp32(a, w) - ref) / mag) ** 2).mean())
    # pieces alone (no accumulation rounding here): the representation error of the scheme against the fp32 chain's total error
    if mode == "small":
        # unscaled-small activations sit in fp16's subnormal floor: the scheme is NOT fp32-class there - which is what the per-BN
        # activation scale (from gamma / beta) exists to prevent; with the scale the kernels would use (values near 2^5) it is again
        assert e_split > 1.5 * e_chain
        ah, al = _split16(a, 16.0 * 1024.0)
        got = (ah @ wl + al @ wh + ah @ wh) / (16.0 * 1024.0 * sw)
        e_split = np.sqrt((((got - ref) / mag) ** 2).mean())
    assert e_split <= 1.0 * e_chain, (e_split, e_chain)
    # a single fp16 piece is three orders of magnitude off: the low piece carries the accuracy
    e_one = np.sqrt(((((ah @ wh) / (sa * (1024.0 if mode == "small" else 1.0) * sw) - ref) / mag) ** 2).mean())
    assert e_one > 100 * e_chain
