"""MI355X-native stand-in for the one function of the reference's `utils` module that sits directly in front
of the affordance path: `get_heightmap` (/root/reference/code/utils.py:38-68, called at code/main.py:111).
Same signature and return tuple; the two DEPTH outputs (the 224x224 heightmap Trainer.forward consumes and the
448x448 one) are produced by one HIP kernel each (point cloud -> robot frame -> perspective warp, fused); the two COLOUR
outputs only feed the reference's Mask R-CNN (code/masks.py, out of scope) and are warped with the
same inverse map in numpy, nearest to cv2's 8-bit bilinear path (not bit-pinned: OpenCV is not available to pin it).

The rest of the reference's utils.py (grasp-angle heuristics, rotation helpers, CrossEntropyLoss2d) is robot-side / host code
and is NOT restated here - but this module sits in front of the reference's `utils` on the module path (INTEGRATION.md
section 1), and the reference's callers need those names from `import utils` (code/robot.py:4,97 `utils.euler2rotm`,
code/main.py:253,264 `utils.get_best_grasp_angle / get_best_suction_angle`, code/trainer.py:9 `from utils import
CrossEntropyLoss2d`).  Every attribute this module does not define is therefore FORWARDED to the next `utils` module on
sys.path (PEP 562 module __getattr__, resolved lazily on first use), so the drop-in shadows `get_heightmap` only.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

import smg_hip

_NEXT_UTILS = None


def _next_utils():
    """The `utils` module that `import utils` would have found had this directory not been in front of it on sys.path (the
    reference's code/utils.py in the INTEGRATION.md set-up), loaded under a private name; None if there is none."""
    global _NEXT_UTILS
    if _NEXT_UTILS is None:
        here = os.path.dirname(os.path.abspath(__file__))
        for d in sys.path:
            d = os.path.abspath(d or os.getcwd())
            if d == here:
                continue
            for cand in (os.path.join(d, "utils.py"), os.path.join(d, "utils", "__init__.py")):
                if os.path.isfile(cand):
                    spec = importlib.util.spec_from_file_location("_smg_forwarded_utils", cand)
                    mod = importlib.util.module_from_spec(spec)
                    spec.loader.exec_module(mod)
                    _NEXT_UTILS = mod
                    return mod
        _NEXT_UTILS = False
    return _NEXT_UTILS or None


def __getattr__(name):
    if name.startswith("__") and name.endswith("__"):
        raise AttributeError(name)
    nxt = _next_utils()
    if nxt is None or not hasattr(nxt, name):
        raise AttributeError("module 'utils' (MI355X drop-in: get_heightmap) has no attribute %r, and no other `utils` module on "
                             "sys.path provides it" % name)
    return getattr(nxt, name)

HEIGHTMAP_SIZE = (224, 224)      # code/utils.py:41-42
COLORMASK_SIZE = (448, 448)
SRC_SIM = np.array([[110, 0], [110, 400], [510, 400], [510, 0]], np.float32)      # code/utils.py:49-50 (simulation)


def perspective_transform(src, dst):
    """cv2.getPerspectiveTransform(src, dst) (code/utils.py:56-59): 3x3 M with M[2,2] = 1."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    a, b = np.zeros((8, 8)), np.zeros(8)
    for i in range(4):
        (x, y), (u, v) = src[i], dst[i]
        a[i] = [x, y, 1, 0, 0, 0, -x * u, -y * u]
        a[i + 4] = [0, 0, 0, x, y, 1, -x * v, -y * v]
        b[i], b[i + 4] = u, v
    return np.append(np.linalg.solve(a, b), 1.0).reshape(3, 3)


def _warp_color(img, m, size):
    """Colour image through the same inverse map (1/32-pixel bilinear, zero border), rounded back to the input dtype."""
    img = np.asarray(img)
    w_dst, h_dst = size
    mi = np.linalg.inv(m)
    xs, ys = np.meshgrid(np.arange(w_dst, dtype=np.float64), np.arange(h_dst, dtype=np.float64))
    den = mi[2, 0] * xs + mi[2, 1] * ys + mi[2, 2]
    scale = np.where(den != 0, 32.0 / np.where(den != 0, den, 1.0), 0.0)
    ix = np.rint((mi[0, 0] * xs + mi[0, 1] * ys + mi[0, 2]) * scale).astype(np.int64)
    iy = np.rint((mi[1, 0] * xs + mi[1, 1] * ys + mi[1, 2]) * scale).astype(np.int64)
    x0, y0, ax, ay = ix >> 5, iy >> 5, (ix & 31) / 32.0, (iy & 31) / 32.0
    h, w = img.shape[:2]

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        v = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.float64)
        return v * (ok[..., None] if v.ndim == 3 else ok)
    e = (lambda a_: a_[..., None]) if img.ndim == 3 else (lambda a_: a_)
    out = tap(y0, x0) * e((1 - ay) * (1 - ax)) + tap(y0, x0 + 1) * e((1 - ay) * ax) + tap(y0 + 1, x0) * e(ay * (1 - ax)) + tap(y0 + 1, x0 + 1) * e(ay * ax)
    return np.rint(out).astype(img.dtype) if np.issubdtype(img.dtype, np.integer) else out.astype(img.dtype)


def get_heightmap(color_img, depth_img, cam_intrinsics, cam_pose, workspace_limits, heightmap_resolution, device=None, src=SRC_SIM):
    """code/utils.py:38-68.  Returns (color_heightmap, depth_heightmap, color_mask, depth_mask, A_htor); workspace_limits
    and heightmap_resolution are accepted and unused, as in the reference."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    depth = np.ascontiguousarray(np.asarray(depth_img, dtype=np.float64))
    h, w = depth.shape
    dst_h = np.array([[0, 0], [0, HEIGHTMAP_SIZE[0]], [HEIGHTMAP_SIZE[1], HEIGHTMAP_SIZE[0]], [HEIGHTMAP_SIZE[1], 0]], np.float32)
    dst_m = np.array([[0, 0], [0, COLORMASK_SIZE[0]], [COLORMASK_SIZE[1], COLORMASK_SIZE[0]], [COLORMASK_SIZE[1], 0]], np.float32)
    a_h, a_m = perspective_transform(src, dst_h), perspective_transform(src, dst_m)
    d_dev = torch.from_numpy(depth).to(dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    outs = []
    for m, (ow, oh) in ((a_h, HEIGHTMAP_SIZE), (a_m, COLORMASK_SIZE)):
        o = torch.empty((oh, ow), dtype=torch.float64, device=dev)
        smg_hip.heightmap(d_dev.data_ptr(), h, w, cam_intrinsics, np.asarray(cam_pose, np.float64)[:4, :4], np.linalg.inv(m), ow, oh,
                          o.data_ptr(), stream)
        outs.append(o)
    depth_heightmap, depth_mask = outs[0].cpu().numpy(), outs[1].cpu().numpy()
    color_heightmap = _warp_color(color_img, a_h, HEIGHTMAP_SIZE)
    color_mask = _warp_color(color_img, a_m, COLORMASK_SIZE)
    return color_heightmap, depth_heightmap, color_mask, depth_mask, perspective_transform(dst_h, src)
