"""ctypes binding of libsmg_hip.so (include/smg_hip.h) - the only way the Python
layer reaches the GPU kernels.  There is no CPU fallback: if the library is missing
or no MI355X is visible every entry point raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMG_HIP_LIB") or os.path.join(_HERE, "libsmg_hip.so")     # SMG_HIP_LIB: dev A/B of two builds


class SmgError(RuntimeError):
    pass


class SmgNet(C.Structure):
    _fields_ = [("params", C.c_void_p), ("grads", C.c_void_p), ("bufs", C.c_void_p), ("nbt", C.c_void_p)]


class SmgBatch(C.Structure):
    _fields_ = [
        ("n_images", C.c_int), ("images_nchw_dev", C.c_void_p), ("heightmaps_dev", C.c_void_p),
        ("hm_size", C.c_int), ("image_mean", C.c_double), ("image_std", C.c_double),
        ("n_streams", C.c_int), ("stream_image", C.POINTER(C.c_int)), ("stream_affine", C.POINTER(C.c_float)),
        ("stream_rotated", C.POINTER(C.c_int)),
        ("n_pairs", C.c_int), ("pair_a", C.POINTER(C.c_int)), ("pair_b", C.POINTER(C.c_int)),
        ("n_bn_seq_trunk", C.c_int), ("bn_seq_trunk", C.POINTER(C.c_int)),
        ("n_bn_seq_head", C.c_int), ("bn_seq_head", C.POINTER(C.c_int)),
        ("masks_dev", C.c_void_p), ("n_masks", C.c_int), ("stream_mask_a", C.POINTER(C.c_int)), ("stream_mask_b", C.POINTER(C.c_int)),
    ]


class SmgAdam(C.Structure):
    _fields_ = [("m", C.c_void_p), ("v", C.c_void_p), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("step_trunk", C.c_int), ("step_head", C.c_int)]


ABI_VERSION = 4     # SMG_ABI_VERSION of include/smg_hip.h this binding was written against

_lib = None


def lib():
    """Load the shared library once.  torch must already be imported so that the HIP
    runtime the library binds to (soname libamdhip64.so.7) is the one torch uses."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads torch's libamdhip64 first)
    if not os.path.exists(LIB_PATH):
        raise SmgError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback for the affordance path." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.smg_last_error.restype = C.c_char_p
    L.smg_version.restype = C.c_int
    if not hasattr(L, "smg_abi_struct_bytes") or L.smg_version() != ABI_VERSION:
        raise SmgError("%s is ABI version %d, this binding needs %d: rebuild it (make -C csrc)" % (LIB_PATH, L.smg_version(), ABI_VERSION))
    L.smg_abi_struct_bytes.argtypes = [C.c_int]
    for which, ty in ((0, SmgBatch), (1, SmgNet), (2, SmgAdam)):
        if L.smg_abi_struct_bytes(which) != C.sizeof(ty):
            raise SmgError("%s: struct %s is %d bytes in the library, %d in this binding" % (LIB_PATH, ty.__name__, L.smg_abi_struct_bytes(which), C.sizeof(ty)))
    L.smg_layout_count.argtypes = [C.c_int]
    for f in (L.smg_layout_param_floats, L.smg_layout_buffer_floats, L.smg_layout_nbt_count):
        f.argtypes = [C.c_int]
        f.restype = C.c_int64
    L.smg_layout_entry.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    L.smg_layout_trunk_range.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.smg_layout_head_range.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.smg_engine_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.smg_engine_destroy.argtypes = [C.c_void_p]
    L.smg_engine_destroy.restype = None
    L.smg_engine_workspace_bytes.argtypes = [C.c_void_p]
    L.smg_engine_workspace_bytes.restype = C.c_int64
    L.smg_engine_geometry.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.smg_forward.argtypes = [C.c_void_p, C.POINTER(SmgNet), C.c_int, C.c_int, C.POINTER(SmgBatch), C.c_void_p, C.c_void_p]
    L.smg_loss.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.smg_backward.argtypes = [C.c_void_p, C.POINTER(SmgNet), C.c_void_p, C.c_void_p]
    if hasattr(L, "smg_backward_phase"):          # (absent from the dev builds tools/ab_kernels.sh compares against)
        L.smg_backward_phase.argtypes = [C.c_void_p, C.POINTER(SmgNet), C.c_void_p, C.c_void_p, C.c_int]
        L.smg_layout_trunk_split.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int64)]
    L.smg_engine_set_precision.argtypes = [C.c_void_p, C.c_int]
    L.smg_engine_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.smg_heightmap.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.smg_argmax.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.smg_adam_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int,
                                C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
    L.smg_train_step_graph.argtypes = [C.c_void_p, C.POINTER(SmgNet), C.c_int, C.c_int, C.POINTER(SmgBatch), C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.POINTER(SmgAdam), C.c_void_p]
    L.smg_debug_read.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.smg_debug_read.restype = C.c_int64
    L.smg_profile_enable.argtypes = [C.c_void_p, C.c_int]
    L.smg_profile_kinds.restype = C.c_int
    L.smg_profile_kind_name.argtypes = [C.c_int]
    L.smg_profile_kind_name.restype = C.c_char_p
    L.smg_profile_read.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    L.smg_profile_read_bytes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    _lib = L
    return L


EXPORTS = (
    "smg_last_error", "smg_version", "smg_abi_struct_bytes", "smg_engine_set_option", "smg_layout_count", "smg_layout_param_floats", "smg_layout_buffer_floats",
    "smg_layout_nbt_count", "smg_layout_entry", "smg_layout_trunk_range", "smg_layout_head_range",
    "smg_engine_create", "smg_engine_destroy", "smg_engine_workspace_bytes", "smg_engine_geometry",
    "smg_forward", "smg_loss", "smg_backward", "smg_backward_phase", "smg_train_step_graph", "smg_layout_trunk_split", "smg_adam_step", "smg_argmax", "smg_heightmap", "smg_engine_set_precision", "smg_debug_read",
    "smg_profile_enable", "smg_profile_kinds", "smg_profile_kind_name", "smg_profile_read", "smg_profile_read_bytes",
)


def check(rc):
    if rc != 0:
        raise SmgError("libsmg_hip: %s (code %d)" % (lib().smg_last_error().decode(), rc))


def layout(head_out):
    """[(name, kind, offset, shape)] in the reference's state_dict order."""
    L = lib()
    out = []
    name = C.create_string_buffer(256)
    kind, ndim, off = C.c_int(), C.c_int(), C.c_int64()
    shape = (C.c_int64 * 4)()
    for i in range(L.smg_layout_count(head_out)):
        check(L.smg_layout_entry(head_out, i, name, 256, C.byref(kind), C.byref(off), C.byref(ndim), shape))
        out.append((name.value.decode(), kind.value, off.value, tuple(shape[k] for k in range(ndim.value))))
    return out


def trunk_range(head_out, trunk_id):
    o, n = C.c_int64(), C.c_int64()
    check(lib().smg_layout_trunk_range(head_out, trunk_id, C.byref(o), C.byref(n)))
    return o.value, n.value


def trunk_split(head_out, trunk_id):
    """Element offset of the first parameter behind dense block 1 (smg_layout_trunk_split)."""
    o = C.c_int64()
    check(lib().smg_layout_trunk_split(head_out, trunk_id, C.byref(o)))
    return o.value


def head_range(head_out, head_id):
    o, n = C.c_int64(), C.c_int64()
    check(lib().smg_layout_head_range(head_out, head_id, C.byref(o), C.byref(n)))
    return o.value, n.value


def _iarr(v):
    a = np.ascontiguousarray(v, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int))


class Engine(object):
    """One smg_engine (workspaces for up to max_streams trunk passes / max_pairs heads)."""

    def __init__(self, device, input_size, max_streams, max_pairs, head_out):
        self.h = C.c_void_p()
        self.device, self.S, self.max_streams, self.max_pairs, self.head_out = device, input_size, max_streams, max_pairs, head_out
        check(lib().smg_engine_create(device, input_size, max_streams, max_pairs, head_out, C.byref(self.h)))
        H = (C.c_int * 6)()
        HWp = (C.c_int * 6)()
        check(lib().smg_engine_geometry(self.h, H, HWp))
        self.H, self.HWp = list(H), list(HWp)
        self.OH = self.OW = self.H[5] - 20 + 1
        self.forward_id = 0
        self.precision = "fp32"

    def close(self):
        if self.h:
            lib().smg_engine_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_precision(self, name):
        """Precision mode: 'fp32' (fp32 storage, split products on the 16-bit matrix cores: fp32-class; default), 'bf16' (bf16 storage of
        activations and gradients) or 'fp16' (fp16 activations, bf16 gradients); include/smg_hip.h."""
        code = PRECISIONS[str(name).replace("torch.", "")]
        check(lib().smg_engine_set_precision(self.h, code))
        self.precision = PRECISION_NAMES[code]          # canonical name: callers compare against 'fp32' / 'bf16' / 'fp16'

    def set_option(self, name, value):
        """Engine switches by name (smg_engine_set_option), e.g. ('deterministic', 1)."""
        check(lib().smg_engine_set_option(self.h, name.encode(), int(value)))

    @property
    def workspace_bytes(self):
        return lib().smg_engine_workspace_bytes(self.h)

    @staticmethod
    def _batch(images_nchw=None, heightmaps=None, hm_size=0, mean=0.0, std=1.0, n_images=0, stream_image=(), stream_affine=(),
               stream_rotated=(), pair_a=(), pair_b=(), bn_seq_trunk=None, bn_seq_head=None, masks=None, n_masks=0,
               stream_mask_a=None, stream_mask_b=None):
        """(smg_batch, arrays to keep alive while it is in use)"""
        b = SmgBatch()
        b.n_images = n_images
        b.images_nchw_dev = images_nchw
        b.heightmaps_dev = heightmaps
        b.hm_size = hm_size
        b.image_mean, b.image_std = mean, std
        keep = []
        a, p = _iarr(stream_image); keep.append(a); b.stream_image = p; b.n_streams = len(a)
        aff = np.ascontiguousarray(stream_affine, dtype=np.float32).reshape(-1); keep.append(aff)
        assert aff.size == 6 * b.n_streams
        b.stream_affine = aff.ctypes.data_as(C.POINTER(C.c_float))
        a, p = _iarr(stream_rotated); keep.append(a); b.stream_rotated = p
        a, p = _iarr(pair_a); keep.append(a); b.pair_a = p; b.n_pairs = len(a)
        a, p = _iarr(pair_b); keep.append(a); b.pair_b = p
        if bn_seq_trunk is not None and len(bn_seq_trunk):
            a, p = _iarr(bn_seq_trunk); keep.append(a); b.bn_seq_trunk = p; b.n_bn_seq_trunk = len(a)
        if bn_seq_head is not None and len(bn_seq_head):
            a, p = _iarr(bn_seq_head); keep.append(a); b.bn_seq_head = p; b.n_bn_seq_head = len(a)
        if masks is not None:
            b.masks_dev, b.n_masks = masks, n_masks
            a, p = _iarr(stream_mask_a); keep.append(a); b.stream_mask_a = p
            a, p = _iarr(stream_mask_b); keep.append(a); b.stream_mask_b = p
            assert len(stream_mask_a) == b.n_streams == len(stream_mask_b)
        return b, keep

    def forward(self, net, trunk_id, head_id, q_out, stream, **batch):
        b, keep = self._batch(**batch)
        check(lib().smg_forward(self.h, C.byref(net), trunk_id, head_id, C.byref(b), q_out, stream))
        self.forward_id += 1
        return self.forward_id

    def train_step_graph(self, net, trunk_id, head_id, loss_mode, labels, q_out, loss_out, dq, adam, stream, **batch):
        """smg_train_step_graph: zero grads + forward + loss + backward + Adam as one replayable hipGraph.  Returns the
        forward token (the engine holds that forward's activations afterwards)."""
        b, keep = self._batch(**batch)
        check(lib().smg_train_step_graph(self.h, C.byref(net), trunk_id, head_id, C.byref(b), loss_mode, labels, q_out, loss_out, dq,
                                         C.byref(adam), stream))
        self.forward_id += 1
        return self.forward_id

    def loss(self, mode, q, labels, n_pairs, loss_out, dq_out, stream):
        check(lib().smg_loss(self.h, mode, q, labels, n_pairs, loss_out, dq_out, stream))

    def backward(self, net, dq, stream, phase=None):
        """phase None: the whole backward; 0 / 1: its two halves (smg_backward_phase)."""
        if phase is None:
            check(lib().smg_backward(self.h, C.byref(net), dq, stream))
        else:
            check(lib().smg_backward_phase(self.h, C.byref(net), dq, stream, int(phase)))

    def debug_read(self, name, stream=None, count=None):
        """count: read only the first `count` floats (a prefix of whole streams for the ring-slot names)."""
        n = lib().smg_debug_read(self.h, name.encode(), None, 0, stream)
        if n < 0:
            check(int(n))
        if count is not None:
            n = min(n, int(count))
        out = np.empty(n, dtype=np.float32)
        got = lib().smg_debug_read(self.h, name.encode(), out.ctypes.data_as(C.c_void_p), n, stream)
        if got < 0:
            check(int(got))
        return out[:got]

    def profile_enable(self, on):
        check(lib().smg_profile_enable(self.h, 1 if on else 0))

    def profile_read(self):
        """{kind_name: (ms, launches, flops, algorithmic HBM bytes)}"""
        L = lib()
        out = {}
        ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        for k in range(L.smg_profile_kinds()):
            check(L.smg_profile_read(self.h, k, C.byref(ms), C.byref(n), C.byref(fl)))
            check(L.smg_profile_read_bytes(self.h, k, C.byref(by)))
            out[L.smg_profile_kind_name(k).decode()] = (ms.value, n.value, fl.value, by.value)
        return out

    def profile_read_stages(self):
        """{kind_name: [(ms, launches, flops) for dense block 1..4]} -- the share of each dense block
        (kind ids kinds*(1+block) + kind of smg_profile_read)."""
        L = lib()
        out = {}
        ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
        K = L.smg_profile_kinds()
        for k in range(K):
            rows = []
            for b in range(4):
                check(L.smg_profile_read(self.h, K * (1 + b) + k, C.byref(ms), C.byref(n), C.byref(fl)))
                rows.append((ms.value, n.value, fl.value))
            out[L.smg_profile_kind_name(k).decode()] = rows
        return out


PRECISIONS = {"fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "fp16": 2, "float16": 2, "half": 2}
PRECISION_NAMES = {0: "fp32", 1: "bf16", 2: "fp16"}


def heightmap(depth_img, h, w, intrinsics, cam_pose, inv_homography, out_w, out_h, out, stream):
    def d(a, n):
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
        assert a.size == n
        return a, a.ctypes.data_as(C.POINTER(C.c_double))
    k, kp = d(intrinsics, 9)
    t, tp = d(cam_pose, 16)
    m, mp = d(inv_homography, 9)
    check(lib().smg_heightmap(depth_img, h, w, kp, tp, mp, out_w, out_h, out, stream))


def argmax(values, n, idx_out, val_out, stream):
    check(lib().smg_argmax(values, n, idx_out, val_out, stream))


def adam_step(params, grads, m, v, offset, count, step, lr, beta1, beta2, eps, stream):
    check(lib().smg_adam_step(params, grads, m, v, offset, count, step, lr, beta1, beta2, eps, stream))
