"""MI355X-native stand-in for the one function of the reference's `utils` module that sits directly in front
of the affordance path: `get_heightmap` (/root/reference/code/utils.py:38-68, called at code/main.py:111).
Same signature and return tuple; the two DEPTH outputs (the 224x224 heightmap Trainer.forward consumes and the
448x448 one) are produced by one HIP kernel each (point cloud -> robot frame -> perspective warp, fused); the two COLOUR
outputs only feed the reference's Mask R-CNN (code/masks.py, out of scope) and are warped with the
same inverse map in numpy, nearest to cv2's 8-bit bilinear path (not bit-pinned: OpenCV is not available to pin it).

The rest of the reference's utils.py (grasp-angle heuristics, rotation helpers) is robot-side code and out of scope.
"""
import numpy as np
import torch

import smg_hip

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
