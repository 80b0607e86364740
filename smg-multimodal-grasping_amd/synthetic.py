"""Counter-based synthetic data generator (seeded weights, heightmaps, masks).

Used by bench.py, __graft_entry__.smoke(), the tests and oracle/make_golden.py to
feed BOTH sides of a parity check the same inputs.  It computes nothing on the
hot path: Trainer / models never import it.

Everything here is a pure function of (seed, name): splitmix64 -> uniform ->
Box-Muller, in numpy, so the reference side (oracle/make_golden.py, run in the
build container) and the GPU side regenerate identical weights / heightmaps
without depending on torch or numpy RNG streams.
"""
import zlib

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    """Vectorised splitmix64 finaliser on a uint64 array."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        z = z ^ (z >> np.uint64(31))
    return z


def _key(seed, name):
    h = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    return np.uint64((int(seed) * 0x100000001B3 + h * 0x9E3779B1 + 0x1234567) % (1 << 64))


def uniform(seed, name, n, lo=0.0, hi=1.0):
    """n float64 uniforms in [lo, hi) determined by (seed, name)."""
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        bits = _splitmix64(_splitmix64(idx + _key(seed, name)))
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return lo + (hi - lo) * u


def normal(seed, name, n, mean=0.0, std=1.0):
    """n float64 normals (Box-Muller on two uniform streams)."""
    u1 = uniform(seed, name + "/u1", n)
    u2 = uniform(seed, name + "/u2", n)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return mean + std * r * np.cos(2.0 * np.pi * u2)


def heightmap_scene(seed, size=224, n_boxes=8):
    """Synthetic depth heightmap + per-object masks.

    Returns (depth [size,size] float64, masks [n_boxes,size,size] float64).
    0 background plus axis-aligned boxes: height U(0.02,0.10) m, sides U(10,40) px.
    Later boxes overwrite earlier ones, and the masks are made disjoint the same way.
    """
    depth = np.zeros((size, size), dtype=np.float64)
    owner = -np.ones((size, size), dtype=np.int64)
    h = uniform(seed, "scene/h", n_boxes, 0.02, 0.10)
    sx = uniform(seed, "scene/sx", n_boxes, 10, 40).astype(np.int64)
    sy = uniform(seed, "scene/sy", n_boxes, 10, 40).astype(np.int64)
    cx = uniform(seed, "scene/cx", n_boxes, 0, 1)
    cy = uniform(seed, "scene/cy", n_boxes, 0, 1)
    for b in range(n_boxes):
        x0 = int(cx[b] * (size - sx[b]))
        y0 = int(cy[b] * (size - sy[b]))
        depth[y0:y0 + sy[b], x0:x0 + sx[b]] = h[b]
        owner[y0:y0 + sy[b], x0:x0 + sx[b]] = b
    masks = np.stack([(owner == b).astype(np.float64) for b in range(n_boxes)])
    return depth, masks


def make_state_dict(layout, seed):
    """Seeded weights for a list of (name, shape, kind) entries.

    kind: 'conv' -> kaiming normal (fan_in, gain sqrt 2); 'bn_w' -> U(0.5,1.5);
    'bn_b' -> N(0,0.1); 'rm' -> zeros; 'rv' -> ones; 'nbt' -> int64 zero;
    'fc_w' / 'fc_b' -> N(0, 0.01) / zeros (the never-executed classifier).
    Returns {name: np.ndarray} (float32 except nbt).
    """
    out = {}
    for name, shape, kind in layout:
        n = int(np.prod(shape)) if len(shape) else 1
        if kind == "conv":
            fan_in = int(np.prod(shape[1:]))
            v = normal(seed, name, n, 0.0, np.sqrt(2.0 / fan_in))
        elif kind == "bn_w":
            v = uniform(seed, name, n, 0.5, 1.5)
        elif kind == "bn_b":
            v = normal(seed, name, n, 0.0, 0.1)
        elif kind == "rm":
            v = np.zeros(n)
        elif kind == "rv":
            v = np.ones(n)
        elif kind == "nbt":
            out[name] = np.zeros((), dtype=np.int64)
            continue
        elif kind == "fc_w":
            v = normal(seed, name, n, 0.0, 0.01)
        elif kind == "fc_b":
            v = np.zeros(n)
        else:
            raise ValueError(kind)
        out[name] = v.astype(np.float32).reshape(shape)
    return out
