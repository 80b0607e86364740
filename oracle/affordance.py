"""ORACLE (test infrastructure, never on the product path).

CPU restatement, in plain PyTorch-CPU fp32 ops, of the reference's affordance hot
path.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import this file; the product (smg-multimodal-grasping_amd/) never does.

Each function cites the reference lines it follows (paths relative to
/root/reference/).  The arithmetic itself lives in third-party code the reference
does not vendor (torch conv / batch_norm / affine_grid / grid_sample / Adam;
torchvision densenet121 - restated in oracle/densenet121.py), so the restatement
calls the same torch CPU operators in the same order, and is pinned against the
reference's own Python executed in the build container (oracle/make_golden.py ->
tests/golden/*.npz; tests/test_oracle_golden.py).
"""
import copy
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .densenet121 import DenseNet121

STYLE_TRUNK = ("grasp_depth_trunk", "suction_depth_trunk", "gs_depth_trunk")
# code/models.py:387,410,434 - style 2 goes through suctionnet_val (sic);
# gsnet_val is constructed (models.py:336-343) but never used.
STYLE_HEAD = ("graspnet_val", "suctionnet_val", "suctionnet_val")


def _head(prefix, out_ch):
    """Head layout, code/models.py:316-343 (reinforcement, 1 ch) / :28-55 (reactive, 3 ch)."""
    return nn.Sequential(OrderedDict([
        (prefix + "-val-norm0", nn.BatchNorm2d(2048)),
        (prefix + "-val-relu0", nn.ReLU(inplace=True)),
        (prefix + "-val-conv0", nn.Conv2d(2048, 64, kernel_size=1, stride=1, bias=False)),
        (prefix + "-val-norm1", nn.BatchNorm2d(64)),
        (prefix + "-val-relu1", nn.ReLU(inplace=True)),
        (prefix + "-val-conv1", nn.Conv2d(64, out_ch, kernel_size=20, stride=1, bias=False)),
    ]))


class OracleNet(nn.Module):
    """reinforcement_net (out_ch=1, code/models.py:301-358) or reactive_net
    (out_ch=3, code/models.py:15-69): three trunks, three heads, same key names."""

    def __init__(self, out_ch=1):
        super().__init__()
        self.suction_depth_trunk = DenseNet121()
        self.grasp_depth_trunk = DenseNet121()
        self.gs_depth_trunk = DenseNet121()
        self.gnum_rotations = 1
        self.snum_rotations = 1
        self.suctionnet_val = _head("suction", out_ch)
        self.graspnet_val = _head("grasp", out_ch)
        self.gsnet_val = _head("grasp", out_ch)  # models.py:336-343 reuses the 'grasp-' names
        for name, m in self.named_modules():  # models.py:347-353
            if "suction-" in name or "grasp-" in name or "gs-" in name:
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight.data)
                elif isinstance(m, nn.BatchNorm2d):
                    m.weight.data.fill_(1)
                    m.bias.data.zero_()


def rotation_matrix(rotate_idx, num_rotations):
    """code/models.py:372-376: float64 2x3 matrix -> float32 [1,2,3]."""
    theta = np.radians(rotate_idx * (360 / num_rotations))
    a = np.asarray([[np.cos(-theta), np.sin(-theta), 0], [-np.sin(-theta), np.cos(-theta), 0]])
    a.shape = (2, 3, 1)
    return torch.from_numpy(a).permute(2, 0, 1).float()


def rotate(x, rotate_idx, num_rotations):
    """code/models.py:378-382: affine_grid + grid_sample(nearest), align_corners=True."""
    grid = F.affine_grid(rotation_matrix(rotate_idx, num_rotations), x.size(), align_corners=True)
    return F.grid_sample(x, grid, mode="nearest", align_corners=True)


def rotation_index_map(rotate_idx, num_rotations, size):
    """Source index (y*size+x, or -1 when out of frame) each output pixel of
    `rotate` copies from.  Second stage restated from torch's grid_sample nearest /
    align_corners=True definition; the grid itself comes from
    torch's own affine_grid so the rounding of its bmm is inherited, not guessed."""
    grid = F.affine_grid(rotation_matrix(rotate_idx, num_rotations), [1, 1, size, size], align_corners=True)
    gx = grid[0, :, :, 0].numpy()
    gy = grid[0, :, :, 1].numpy()
    ix = np.rint(((gx + np.float32(1)) / np.float32(2)) * np.float32(size - 1))
    iy = np.rint(((gy + np.float32(1)) / np.float32(2)) * np.float32(size - 1))
    ok = (ix >= 0) & (ix <= size - 1) & (iy >= 0) & (iy <= size - 1)
    idx = (iy.astype(np.int64) * size + ix.astype(np.int64)).astype(np.int32)
    idx[~ok] = -1
    return idx


def _sample(net, x, mx, style, rotate_idx, num_rot):
    """One (rotation, mask) sample, code/models.py:372-387 (and the 17 sibling blocks)."""
    trunk = getattr(net, STYLE_TRUNK[style]).features
    head = getattr(net, STYLE_HEAD[style])
    rot = rotate(x, rotate_idx, num_rot)
    feat = torch.cat((trunk(rot), trunk(mx)), dim=1)  # models.py:384-386
    return head(feat)


def forward(net, x, mx, style=0, is_volatile=False, specific_rotation=-1):
    """reinforcement_net.forward / reactive_net.forward, code/models.py:361-586, :72-296.

    Branch A (is_volatile, specific_rotation == -1): no-grad sweep, returns a python list.
    Branch B (is_volatile, specific_rotation != -1): no-grad single sample, returns a tensor.
    Branch C (else): grad-enabled single sample, stored on net.{gra,suc,gs}_prob."""
    if is_volatile and specific_rotation == -1:
        with torch.no_grad():
            if style == 0:
                return [_sample(net, x, mx, 0, r, net.gnum_rotations) for r in range(net.gnum_rotations)]
            if style == 1:
                return [_sample(net, x, mx, 1, r, net.snum_rotations) for r in range(net.snum_rotations)]
            return [_sample(net, x, mx, 2, 0, net.gnum_rotations)]
    if is_volatile:
        with torch.no_grad():
            r = 0 if style == 2 else specific_rotation
            return _sample(net, x, mx, style, r, net.gnum_rotations)
    r = 0 if style == 2 else specific_rotation
    q = _sample(net, x, mx, style, r, net.gnum_rotations)
    net.gra_prob, net.suc_prob, net.gs_prob = [], [], []
    setattr(net, ("gra_prob", "suc_prob", "gs_prob")[style], q)
    return q


def preprocess(depth_heightmap, image_mean, image_std):
    """Trainer.forward preprocessing, code/trainer.py:165-191.

    ndimage.zoom(order=0, x2) is restated as 2x pixel replication (verified identical
    in tests/test_oracle_golden.py through golden vector G2); pad to the next
    multiple of 32 above the diagonal; replicate to 3 channels; normalise in float64;
    cast to float32; NCHW.  mean/std are parameters because the released constants
    are [0,0,0]/[0,0,0] (trainer.py:176-177) which yields inf/NaN."""
    h2 = np.repeat(np.repeat(np.asarray(depth_heightmap, dtype=np.float64), 2, axis=0), 2, axis=1)
    diag = float(h2.shape[0]) * np.sqrt(2)
    diag = np.ceil(diag / 32) * 32
    pad = int((diag - h2.shape[0]) / 2)
    h2 = np.pad(h2, pad, "constant", constant_values=0)
    img = np.stack([h2, h2, h2], axis=2)
    with np.errstate(divide="ignore", invalid="ignore"):
        for c in range(3):
            img[:, :, c] = (img[:, :, c] - image_mean[c]) / image_std[c]
    return torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None].contiguous()


def huber(q, label):
    """code/trainer.py:345-348."""
    d = q - label
    if abs(d) < 1:
        return 0.5 * d ** 2
    return abs(d) - 0.5


def reactive_loss(logits, label_value):
    """code/trainer.py:296-299 + code/utils.py:306-313: NLLLoss2d(log_softmax) with
    class weights [1,1,0], size_average=True, then .sum()."""
    w = torch.ones(3)
    w[2] = 0
    label = torch.from_numpy(np.full((1, 1, 1), label_value)).long()
    return F.nll_loss(F.log_softmax(logits[0].view(1, 3, 1, 1), dim=1), label, weight=w).sum()


def train_step(net, optimizer, x, mx, style, rotate_idx, label_value, method="reinforcement"):
    """Trainer.backprop, code/trainer.py:334-384 (reinforcement) / :282-332 (reactive).
    Returns (loss float, q tensor detached)."""
    optimizer.zero_grad()
    q = forward(net, x, mx, style, False, rotate_idx)
    if method == "reinforcement":
        loss = huber(q[0, 0, 0, 0], label_value)
    else:
        loss = reactive_loss(q, label_value)
    loss = loss.sum()
    loss.backward()
    optimizer.step()
    return float(loss.detach()), q.detach().clone()


def make_adam(net):
    """code/trainer.py:99."""
    return torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0)


def label_value(method, primitive_action, objects_number, suction_success, grasp_success, gs_success,
                future_reward, discount):
    """Reward arithmetic of Trainer.get_label_value, code/trainer.py:218-274, with the
    network evaluation (`future_reward`, :259-270) passed in by the caller."""
    if method == "reactive":
        lab = 0
        if primitive_action == "suction":
            ok = suction_success
            lab = 0 if suction_success else 1
        elif primitive_action == "grasp":
            ok = grasp_success
            lab = 0 if grasp_success else 1
        else:
            ok = gs_success
            lab = 0 if gs_success == 2.5 else 1
        return lab, ok
    cur = {"suction": suction_success, "grasp": grasp_success, "grasp_then_suction": gs_success}.get(primitive_action, 0)
    if suction_success == 0 and grasp_success == 0 and gs_success == 0:
        fut = 0
    elif (objects_number == 1 and suction_success == 1) or (objects_number == 1 and grasp_success == 1) or \
            (objects_number == 2 and gs_success == 2.5):
        fut = 0
    else:
        fut = future_reward
    return cur + discount * fut, cur


def state_layout(out_ch=1):
    """(name, shape, kind) for every state_dict entry, in torch's order."""
    net = OracleNet(out_ch)
    lay = []
    for name, t in net.state_dict().items():
        shape = tuple(t.shape)
        if name.endswith("num_batches_tracked"):
            kind = "nbt"
        elif name.endswith("running_mean"):
            kind = "rm"
        elif name.endswith("running_var"):
            kind = "rv"
        elif "classifier.weight" in name:
            kind = "fc_w"
        elif "classifier.bias" in name:
            kind = "fc_b"
        elif len(shape) == 4:
            kind = "conv"
        elif name.endswith(".weight"):
            kind = "bn_w"
        else:
            kind = "bn_b"
        lay.append((name, shape, kind))
    return lay


def load_numpy_state(net, sd):
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return net


def clone_target(net):
    """code/trainer.py:74-75."""
    t = copy.deepcopy(net)
    t.load_state_dict(net.state_dict())
    return t
