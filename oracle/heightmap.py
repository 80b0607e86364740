"""ORACLE / test infrastructure - never imported by the product path.

CPU (numpy) restatement of the heightmap generation that feeds the affordance path,
`utils.get_heightmap` of the reference (/root/reference/code/utils.py:12-68):

    get_pointcloud        (:12-35)  camera pixels + depth -> camera-frame points
    rigid transform       (:47)     cam_pose[0:3,0:3] . p + cam_pose[0:3,3]  -> robot frame; z = height above the table
    cv2.getPerspectiveTransform (:56-59)  the 3x3 homography of four point pairs
    cv2.warpPerspective   (:62-66)  480x640 world-z image -> 224x224 / 448x448 heightmaps

PARITY UNPINNED: the two cv2 calls live in OpenCV, which is not installed in the build
image (the reference's utils.py cannot even be imported: `import cv2`, utils.py:4), and the
reference holds no fixture for them.  What is restated is OpenCV's published algorithm for
the default flags the reference uses (INTER_LINEAR, BORDER_CONSTANT 0, no WARP_INVERSE_MAP):
the matrix is inverted, each destination pixel is mapped to source coordinates in double,
the coordinates are rounded to 1/32 pixel (INTER_BITS = 5, saturate_cast<int>(x * 32) =
round-half-even), and the four taps are blended with float32 table weights
((1-fy)(1-fx), (1-fy)fx, fy(1-fx), fy fx), taps outside the image counting as 0.
The HIP kernel (csrc/elem.cuh heightmap_warp_kernel) is tested against THIS file.
"""
import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS

# the four source corners of the table in the camera image, simulation setup (utils.py:49-50)
SRC_SIM = np.array([[110, 0], [110, 400], [510, 400], [510, 0]], np.float32)


def perspective_transform(src, dst):
    """cv2.getPerspectiveTransform: the 3x3 M (M[2,2] = 1) with dst_i ~ M . src_i, from the 8x8 linear system
    (utils.py:56-59)."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    a = np.zeros((8, 8))
    b = np.zeros(8)
    for i in range(4):
        x, y = src[i]
        u, v = dst[i]
        a[i] = [x, y, 1, 0, 0, 0, -x * u, -y * u]
        a[i + 4] = [0, 0, 0, x, y, 1, -x * v, -y * v]
        b[i], b[i + 4] = u, v
    m = np.linalg.solve(a, b)
    return np.append(m, 1.0).reshape(3, 3)


def world_z(depth_img, cam_intrinsics, cam_pose):
    """Rows 2 of utils.py:12-47: the robot-frame z coordinate of every camera pixel (float64, [H, W])."""
    depth_img = np.asarray(depth_img, np.float64)
    h, w = depth_img.shape
    k, t = np.asarray(cam_intrinsics, np.float64), np.asarray(cam_pose, np.float64)
    px, py = np.meshgrid(np.linspace(0, w - 1, w), np.linspace(0, h - 1, h))
    cx = np.multiply(px - k[0][2], depth_img / k[0][0])
    cy = np.multiply(py - k[1][2], depth_img / k[1][1])
    pts = np.stack([cx.reshape(-1), cy.reshape(-1), depth_img.reshape(-1)], axis=1)
    surf = np.transpose(np.dot(t[0:3, 0:3], np.transpose(pts)) + np.tile(t[0:3, 3:], (1, pts.shape[0])))
    return surf[:, 2].reshape(h, w)


def warp_perspective(img, m, size):
    """cv2.warpPerspective(img, M, (w, h)) with the default flags, float64 single-channel image."""
    img = np.asarray(img, np.float64)
    h_src, w_src = img.shape
    w_dst, h_dst = size
    mi = np.linalg.inv(np.asarray(m, np.float64))
    xs, ys = np.meshgrid(np.arange(w_dst, dtype=np.float64), np.arange(h_dst, dtype=np.float64))
    den = mi[2, 0] * xs + mi[2, 1] * ys + mi[2, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        scale = np.where(den != 0, INTER_TAB_SIZE / den, 0.0)
    fx = np.clip((mi[0, 0] * xs + mi[0, 1] * ys + mi[0, 2]) * scale, -2147483648.0, 2147483647.0)
    fy = np.clip((mi[1, 0] * xs + mi[1, 1] * ys + mi[1, 2]) * scale, -2147483648.0, 2147483647.0)
    ix, iy = np.rint(fx).astype(np.int64), np.rint(fy).astype(np.int64)          # saturate_cast<int>: round half to even
    x0, y0 = ix >> INTER_BITS, iy >> INTER_BITS
    ax = ((ix & (INTER_TAB_SIZE - 1)).astype(np.float32) * np.float32(1.0 / INTER_TAB_SIZE))
    ay = ((iy & (INTER_TAB_SIZE - 1)).astype(np.float32) * np.float32(1.0 / INTER_TAB_SIZE))
    one = np.float32(1.0)
    w00, w01, w10, w11 = (one - ay) * (one - ax), (one - ay) * ax, ay * (one - ax), ay * ax     # float32 table weights

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h_src) & (xx >= 0) & (xx < w_src)
        return np.where(ok, img[np.clip(yy, 0, h_src - 1), np.clip(xx, 0, w_src - 1)], 0.0)
    return (tap(y0, x0) * w00.astype(np.float64) + tap(y0, x0 + 1) * w01.astype(np.float64)
            + tap(y0 + 1, x0) * w10.astype(np.float64) + tap(y0 + 1, x0 + 1) * w11.astype(np.float64))


def get_depth_heightmaps(depth_img, cam_intrinsics, cam_pose, src=SRC_SIM):
    """The depth outputs of utils.get_heightmap (:38-68): (depth_heightmap [224,224], depth_mask [448,448], A_htor)."""
    z = world_z(depth_img, cam_intrinsics, cam_pose)
    dst_h = np.array([[0, 0], [0, 224], [224, 224], [224, 0]], np.float32)
    dst_m = np.array([[0, 0], [0, 448], [448, 448], [448, 0]], np.float32)
    a_h, a_m = perspective_transform(src, dst_h), perspective_transform(src, dst_m)
    return warp_perspective(z, a_h, (224, 224)), warp_perspective(z, a_m, (448, 448)), perspective_transform(dst_h, src)
