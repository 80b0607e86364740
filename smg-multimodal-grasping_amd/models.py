"""MI355X-native stand-ins for the reference's `models.reinforcement_net` and
`models.reactive_net` (/root/reference/code/models.py:301-586 and :15-296).

Same constructor, attributes, forward() signature, return types per branch and
state_dict keys (2217 entries, torchvision densenet121 naming) - so
`from models import reactive_net, reinforcement_net` (code/trainer.py:10) and
snapshot load/save (code/logger.py:121-125, code/main.py:104,353) keep working - but
forward() is one call into libsmg_hip.so (hand-written HIP for gfx950), not a chain
of torch ops.  All parameters are views into ONE flat fp32 buffer whose layout the
C library defines (smg_hip.layout), so Adam and the gradient all-reduce are single
passes over contiguous memory.

What forward() does NOT do: run on CPU.  The reference itself only builds its
sampling grids under `if self.use_cuda` (models.py:377-382); here a missing GPU or
missing libsmg_hip.so raises instead of silently computing something else.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

import smg_hip

# style -> (trunk_id, head_id) in layout order (suction, grasp, gs).
# Style 2 deliberately uses suctionnet_val: code/models.py:434,507,582.
STYLE_TRUNK = (1, 0, 2)
STYLE_HEAD = (1, 0, 0)
_PROB_ATTR = ("gra_prob", "suc_prob", "gs_prob")

_ENGINES = {}


def get_engine(device, input_size, head_out, n_streams, n_pairs):
    """Engines (activation / gradient workspaces) are cached per (device, S, head_out)
    and regrown when a call needs more streams than the cached one holds."""
    key = (device, input_size, head_out)
    eng = _ENGINES.get(key)
    if eng is None or eng.max_streams < n_streams or eng.max_pairs < n_pairs:
        cap_s = max(n_streams, eng.max_streams if eng else 0, 17)
        cap_p = max(n_pairs, eng.max_pairs if eng else 0, 16)
        torch.cuda.synchronize(device)      # nothing of the old engine is in flight any more
        if eng is not None:
            eng.close()                     # models that saved activations on it get "activations are gone" from backward
        eng = smg_hip.Engine(device, input_size, cap_s, cap_p, head_out)
        _ENGINES[key] = eng
    return eng


def release_engines():
    for e in _ENGINES.values():
        e.close()
    _ENGINES.clear()


def rotation_theta(rotate_idx, num_rotations):
    """2x3 float32 affine matrix of code/models.py:372-376 (float64 trig, then .float())."""
    t = np.radians(rotate_idx * (360 / num_rotations))
    a = np.asarray([[np.cos(-t), np.sin(-t), 0], [-np.sin(-t), np.cos(-t), 0]])
    return a.astype(np.float32).reshape(6)


class _Node(nn.Module):
    """Parameter container; only exists to reproduce the reference's key hierarchy."""


class _EngineFn(torch.autograd.Function):
    """Lets the reference's own Trainer.backprop (`loss.backward()`,
    code/trainer.py:350-351) drive smg_backward through autograd."""

    @staticmethod
    def forward(ctx, hook, net, q, token):
        ctx.net, ctx.token = net, token
        return q.clone()

    @staticmethod
    def backward(ctx, dq):
        ctx.net._engine_backward(ctx.token, dq.contiguous())
        return None, None, None, None


class _AffordanceNet(nn.Module):
    HEAD_OUT = 1

    def __init__(self, use_cuda):
        super(_AffordanceNet, self).__init__()
        self.use_cuda = use_cuda
        self.gnum_rotations = 1     # code/models.py:312-313, :25-26
        self.snum_rotations = 1
        self._layout = smg_hip.layout(self.HEAD_OUT)
        L = smg_hip.lib()
        self._flat_params = torch.zeros(L.smg_layout_param_floats(self.HEAD_OUT), dtype=torch.float32)
        self._flat_bufs = torch.zeros(L.smg_layout_buffer_floats(self.HEAD_OUT), dtype=torch.float32)
        self._flat_nbt = torch.zeros(L.smg_layout_nbt_count(self.HEAD_OUT), dtype=torch.int64)
        self._flat_grads = None
        self._entries = []          # (tensor-or-param, kind, offset, shape)
        for name, kind, off, shape in self._layout:
            parts = name.split(".")
            node = self
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, _Node())
                node = node._modules[p]
            n = int(np.prod(shape)) if shape else 1
            if kind == 0:
                t = nn.Parameter(self._flat_params[off:off + n].view(shape))
                node.register_parameter(parts[-1], t)
            elif kind in (1, 2):
                t = self._flat_bufs[off:off + n].view(shape)
                node.register_buffer(parts[-1], t)
            else:
                t = self._flat_nbt[off:off + 1].view(())
                node.register_buffer(parts[-1], t)
            self._entries.append((node, parts[-1], kind, off, n, shape))
        # first parameter of every trunk / head gradient range (tells whether the range's .grad views are alive)
        self._range_probe = {}
        for kind_, fn in (("t", smg_hip.trunk_range), ("h", smg_hip.head_range)):
            for i in range(3):
                off0 = fn(self.HEAD_OUT, i)[0]
                node, leaf = next((nd, lf) for nd, lf, k, off, n, shp in self._entries if k == 0 and off == off0)
                self._range_probe[(kind_, i)] = node._parameters[leaf]
        self._init_weights()
        self.gra_prob = []          # code/models.py:356-358
        self.suc_prob = []
        self.gs_prob = []
        self._saved = None
        self._autograd_hook = None
        self._grads_clean = False
        self._graph_exposed = None      # (trunk, head) whose p.grad views a graph-replayed training step left in place
        self.precision = "fp32"     # operand precision of the matrix products (set_precision)
        self._prec_from_cast = False

    # ---- initialisation ------------------------------------------------------------------
    def _init_weights(self):
        """Heads: kaiming-normal convs, BN gamma=1 beta=0 (code/models.py:347-353).
        Trunks: the reference loads ImageNet weights (densenet121(pretrained=True),
        :308-310) which cannot be fetched offline; load a snapshot with load_state_dict.
        Until then the trunks get torch's default Conv2d/BatchNorm2d/Linear init."""
        with torch.no_grad():
            for node, leaf, kind, off, n, shape in self._entries:
                t = getattr(node, leaf)
                if kind == 0 and len(shape) == 4:
                    nn.init.kaiming_normal_(t)
                elif kind == 0 and len(shape) == 2:
                    nn.init.normal_(t, 0.0, 0.01)
                elif kind == 0 and leaf == "weight":
                    t.fill_(1.0)
                elif kind == 0:
                    t.zero_()
                elif kind == 2:
                    t.fill_(1.0)

    # ---- storage management ----------------------------------------------------------------
    def _rebind(self):
        # every p.grad is dropped below (and _apply drops the flat gradient buffer): forget the graph-replayed step's "views already in
        # place" key and the cached views / exposed list with it, or the next graph step would skip zero_grad + expose_grads and leave p.grad None
        self._graph_exposed = None
        self._grad_views = None
        self._exposed_params = None
        for node, leaf, kind, off, n, shape in self._entries:
            if kind == 0:
                p = node._parameters[leaf]
                p.data = self._flat_params[off:off + n].view(shape)
                p.grad = None
            elif kind in (1, 2):
                node._buffers[leaf] = self._flat_bufs[off:off + n].view(shape)
            else:
                node._buffers[leaf] = self._flat_nbt[off:off + 1].view(())

    def set_precision(self, name):
        """Precision mode of the engine calls of this model: 'fp32' (default - fp32 storage, fp32-class products on the 16-bit
        matrix cores - scaled two-piece fp16 splits for the dense layers, three-piece bf16 splits elsewhere: the accuracy of
        the reference's apex O0 arithmetic, code/trainer.py:101), 'bf16'
        (activations and gradients STORED in bf16, one bf16 MFMA term per product; BASELINE.json config 3) or 'fp16' (activations
        stored in fp16 with fp16 forward products, gradients stored and multiplied in bf16; config 5).  Parameters, their
        gradients, BN statistics, every accumulation and Adam stay fp32 in every mode."""
        name = str(name).replace("torch.", "")
        if name not in smg_hip.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(smg_hip.PRECISIONS))
        self.precision = {0: "fp32", 1: "bf16", 2: "fp16"}[smg_hip.PRECISIONS[name]]
        self._prec_from_cast = False       # an explicit choice: a later .float() does not undo it (only .half() / .bfloat16() casts are undone)
        self._saved = None
        return self

    def _apply(self, fn, recurse=True):
        probe = fn(torch.zeros(1, dtype=torch.float32, device=self._flat_params.device))
        if probe.dtype in (torch.bfloat16, torch.float16):
            # model.half() / .bfloat16(): the master copy stays fp32 (like apex O1/O2 keep fp32 weights); only the
            # matrix products change their operand precision
            self.set_precision("bf16" if probe.dtype == torch.bfloat16 else "fp16")
            self._prec_from_cast = True
            dev = probe.device
            fn = lambda t: t.to(dev)                                     # noqa: E731
        elif probe.dtype != torch.float32:
            raise TypeError("the affordance engine stores fp32 and computes in fp32 / bf16 / fp16 operands; got %s" % probe.dtype)
        elif (self.precision != "fp32" and self._prec_from_cast
              and fn(torch.zeros(1, dtype=torch.float16, device=self._flat_params.device)).dtype == torch.float32):
            # model.float() / .to(torch.float32) after a .half(): a dtype cast (it turns a half probe into fp32; a device
            # move leaves it half) - back to the fp32-class products.  A precision chosen with set_precision() stays: the
            # common `model.float().cuda()` idiom must not silently change the numerics and speed of configs 3 / 5.
            self.set_precision("fp32")
        self._flat_params = fn(self._flat_params)
        self._flat_bufs = fn(self._flat_bufs)
        nbt = fn(self._flat_nbt)
        self._flat_nbt = nbt if nbt.dtype == torch.int64 else nbt.to(torch.int64)
        self._flat_grads = None
        self._rebind()
        return self

    def __deepcopy__(self, memo):
        new = type(self)(self.use_cuda)
        new._flat_params = self._flat_params.clone()
        new._flat_bufs = self._flat_bufs.clone()
        new._flat_nbt = self._flat_nbt.clone()
        new._rebind()
        new.gnum_rotations, new.snum_rotations = self.gnum_rotations, self.snum_rotations
        new.precision, new._prec_from_cast = self.precision, self._prec_from_cast
        return new

    def train(self, mode=True):
        if not mode:
            raise NotImplementedError("the reference never leaves training-mode BatchNorm (code/trainer.py:95); "
                                      "eval-mode statistics are not part of this path")
        return self

    def flat_grads(self):
        if self._flat_grads is None or self._flat_grads.device != self._flat_params.device:
            self._flat_grads = torch.zeros_like(self._flat_params)
            self._grads_clean = True
            self._dirty_ranges = set()
        return self._flat_grads

    def zero_grad(self, set_to_none=True):
        if self._flat_grads is not None:
            dirty = getattr(self, "_dirty_ranges", None)
            if dirty is None:
                self._flat_grads.zero_()
            else:
                # only the (trunk, head) ranges written since the last zero_grad: 7.1 M of the 24.4 M floats for a reinforcement step
                # (the whole-buffer fill was 70 us at the head of every step)
                for off, n in dirty:
                    self._flat_grads[off:off + n].zero_()
        self._dirty_ranges = set()
        self._grads_clean = True
        self._graph_exposed = None
        # p.grad = None for the parameters that can have one: those expose_grads() handed a view since the last call (a walk over all
        # 2217 parameters cost ~1 ms of host time at the head of every step, in front of the forward's first launch)
        exposed = getattr(self, "_exposed_params", None)
        if exposed is None:
            for p in self.parameters():
                p.grad = None
        else:
            for p in exposed:
                p.grad = None
        self._exposed_params = []

    def expose_grads(self, trunk_id, head_id):
        """Make p.grad views of the flat gradient buffer for the parameters the last
        backward touched (everything else stays None, like torch>=2 zero_grad)."""
        g = self.flat_grads()
        cache = getattr(self, "_grad_views", None)
        if cache is None or cache[0] is not g:
            cache = self._grad_views = (g, {})
        key = (trunk_id, head_id)
        views = cache[1].get(key)
        if views is None:      # (the views are fixed for a (trunk, head) once made: ~370 slice + view calls, 1 ms, not per step)
            t0, tn = smg_hip.trunk_range(self.HEAD_OUT, trunk_id)
            h0, hn = smg_hip.head_range(self.HEAD_OUT, head_id)
            views = [(node._parameters[leaf], g[off:off + n].view(shape)) for node, leaf, kind, off, n, shape in self._entries
                     if kind == 0 and (t0 <= off < t0 + tn or h0 <= off < h0 + hn)]
            cache[1][key] = views
        for p, v in views:
            p.grad = v
        if getattr(self, "_exposed_params", None) is not None:
            self._exposed_params.extend(p for p, _ in views)

    def _net_struct(self, with_grads):
        net = smg_hip.SmgNet()
        net.params = self._flat_params.data_ptr()
        net.grads = self.flat_grads().data_ptr() if with_grads else None
        net.bufs = self._flat_bufs.data_ptr()
        net.nbt = self._flat_nbt.data_ptr()
        return net

    # ---- engine calls ----------------------------------------------------------------------
    def _require_gpu(self):
        if not self.use_cuda or not self._flat_params.is_cuda:
            raise RuntimeError("the affordance network runs only on an MI355X through libsmg_hip.so; construct it with "
                               "use_cuda=True and call .cuda() (the reference has no CPU path either: "
                               "code/models.py:377-385)")

    def run(self, style, rotations, num_rot, images_nchw=None, heightmaps=None, mean=0.0, std=1.0,
            keep_for_backward=False, update_bn=True):
        """Evaluate (rotation, mask) samples.  Inputs hold 2*n_scenes images: image 2k is scene k's
        depth image (rotated per sample), image 2k+1 its masked depth image (never rotated, so its
        trunk pass is computed once per scene - the reference recomputes it per rotation,
        code/models.py:385).  `rotations` is a list of rotation indices (one scene) or a list of such
        lists (one per scene).  Returns a cuda tensor [n_samples, out, OH, OW], scene-major."""
        self._require_gpu()
        dev = self._flat_params.device
        if images_nchw is not None:
            S = int(images_nchw.shape[-1])
            n_images = int(images_nchw.shape[0])
            src = dict(images_nchw=images_nchw.data_ptr(), n_images=n_images)
        else:
            hm = int(heightmaps.shape[-1])
            n_images = int(heightmaps.shape[0])
            diag = np.ceil(float(2 * hm) * np.sqrt(2) / 32) * 32          # code/trainer.py:169-171
            S = 2 * hm + 2 * int((diag - 2 * hm) / 2)
            src = dict(heightmaps=heightmaps.data_ptr(), hm_size=hm, mean=float(mean), std=float(std), n_images=n_images)
        per_scene = [list(rotations)] if (len(rotations) == 0 or np.isscalar(rotations[0])) else [list(r) for r in rotations]
        if 2 * len(per_scene) != n_images:
            raise ValueError("need one (depth, masked depth) image pair per scene")
        stream_image, stream_rot, thetas, pair_a, pair_b, seq_t, seq_h = [], [], [], [], [], [], []
        for k, rots in enumerate(per_scene):
            base = len(stream_image)
            mask_stream = base + len(rots)
            for j, r in enumerate(rots):
                stream_image.append(2 * k); stream_rot.append(1); thetas.append(rotation_theta(r, num_rot))
                pair_a.append(base + j); pair_b.append(mask_stream)
                seq_t += [base + j, mask_stream]                         # reference order: trunk(rot r), trunk(mask), head(r)
                seq_h.append(len(pair_a) - 1)
            stream_image.append(2 * k + 1); stream_rot.append(0); thetas.append(rotation_theta(0, 1))
        n_pairs = len(pair_a)
        eng = get_engine(dev.index or 0, S, self.HEAD_OUT, len(stream_image), n_pairs)
        if eng.precision != self.precision:
            eng.set_precision(self.precision)
        q = torch.empty((n_pairs, self.HEAD_OUT, eng.OH, eng.OW), dtype=torch.float32, device=dev)
        trunk_id, head_id = STYLE_TRUNK[style], STYLE_HEAD[style]
        net = self._net_struct(keep_for_backward)
        stream = torch.cuda.current_stream(dev).cuda_stream
        token = eng.forward(net, trunk_id, head_id, q.data_ptr(), stream,
                            stream_image=stream_image, stream_affine=np.concatenate(thetas), stream_rotated=stream_rot,
                            pair_a=pair_a, pair_b=pair_b, bn_seq_trunk=seq_t if update_bn else None,
                            bn_seq_head=seq_h if update_bn else None, **src)
        self._saved = (eng, token, trunk_id, head_id) if keep_for_backward else None
        return q

    def train_step_graph(self, style, rot, num_rot, heightmaps, labels, loss, q, dq, optimizer, loss_mode, mean=0.0, std=1.0):
        """One (mask, rotation) training sample - zero the (trunk, head) gradient ranges, forward (branch C), loss, backward, Adam -
        as ONE replayed hipGraph (smg_train_step_graph; the reference's real call pattern, code/main.py:338 -> code/trainer.py:334-384).
        `heightmaps` [2, H, H] float64 (depth, masked depth), `labels` [1], `loss` [1], `q` / `dq` [1, out, OH, OW] are PERSISTENT
        device tensors: the graph is keyed on their addresses; their contents and the rotation may change from step to step."""
        self._require_gpu()
        dev = self._flat_params.device
        hm = int(heightmaps.shape[-1])
        diag = np.ceil(float(2 * hm) * np.sqrt(2) / 32) * 32                  # code/trainer.py:169-171
        S = 2 * hm + 2 * int((diag - 2 * hm) / 2)
        eng = get_engine(dev.index or 0, S, self.HEAD_OUT, 2, 1)
        if eng.precision != self.precision:
            eng.set_precision(self.precision)
        trunk_id, head_id = STYLE_TRUNK[style], STYLE_HEAD[style]
        if optimizer.m is None or optimizer.m.device != dev:
            optimizer.m = torch.zeros_like(self._flat_params)
            optimizer.v = torch.zeros_like(self._flat_params)
        names = ("trunk%d" % trunk_id, "head%d" % head_id)
        for nm in names:
            optimizer.steps[nm] = optimizer.steps.get(nm, 0) + 1
        adam = smg_hip.SmgAdam()
        adam.m, adam.v = optimizer.m.data_ptr(), optimizer.v.data_ptr()
        adam.lr, adam.beta1, adam.beta2, adam.eps = optimizer.lr, optimizer.betas[0], optimizer.betas[1], optimizer.eps
        adam.step_trunk, adam.step_head = optimizer.steps[names[0]], optimizer.steps[names[1]]
        stream = torch.cuda.current_stream(dev).cuda_stream
        try:
            token = eng.train_step_graph(self._net_struct(True), trunk_id, head_id, loss_mode, labels.data_ptr(), q.data_ptr(), loss.data_ptr(),
                                         dq.data_ptr(), adam, stream,
                                         heightmaps=heightmaps.data_ptr(), hm_size=hm, mean=float(mean), std=float(std), n_images=2,
                                         stream_image=[0, 1], stream_affine=np.concatenate([rotation_theta(rot, num_rot), rotation_theta(0, 1)]),
                                         stream_rotated=[1, 0], pair_a=[0], pair_b=[1], bn_seq_trunk=[0, 1], bn_seq_head=[0])
        except Exception:
            for nm in names:
                optimizer.steps[nm] -= 1
            raise
        self._saved = (eng, token, trunk_id, head_id)
        self._grads_clean = False
        if getattr(self, "_dirty_ranges", None) is not None:
            self._dirty_ranges.add(smg_hip.trunk_range(self.HEAD_OUT, trunk_id))
            self._dirty_ranges.add(smg_hip.head_range(self.HEAD_OUT, head_id))
        return trunk_id, head_id

    def run_pairs(self, style, num_rot, heightmaps, rot_streams, mask_images, pairs, mean=0.0, std=1.0,
                  bn_seq_trunk=None, bn_seq_head=None, masks=None, mask_a=None, mask_b=None):
        """General form: image 0 is the scene's depth heightmap; `rot_streams` lists the rotation
        indices evaluated on it (ONE trunk pass each, shared by every pair that uses it);
        `mask_images` lists image indices (>= 1) fed un-rotated (one trunk pass each);
        `pairs` = [(i, j)]: head evaluation on rot_streams[i] x mask_images[j].
        `masks` (device tensor [n_masks, H, H] float64) with `mask_a` / `mask_b` (one index per entry of
        mask_images, -1 = none): those streams read image * (masks[a] + masks[b]), formed on the device.
        Returns q [len(pairs), out, OH, OW].  Inference only (no saved activations)."""
        self._require_gpu()
        dev = self._flat_params.device
        hm = int(heightmaps.shape[-1])
        diag = np.ceil(float(2 * hm) * np.sqrt(2) / 32) * 32
        S = 2 * hm + 2 * int((diag - 2 * hm) / 2)
        n_rot, n_mask = len(rot_streams), len(mask_images)
        stream_image = [0] * n_rot + list(mask_images)
        stream_rot = [1] * n_rot + [0] * n_mask
        thetas = [rotation_theta(r, num_rot) for r in rot_streams] + [rotation_theta(0, 1)] * n_mask
        pair_a = [i for i, _ in pairs]
        pair_b = [n_rot + j for _, j in pairs]
        eng = get_engine(dev.index or 0, S, self.HEAD_OUT, len(stream_image), len(pairs))
        if eng.precision != self.precision:
            eng.set_precision(self.precision)
        q = torch.empty((len(pairs), self.HEAD_OUT, eng.OH, eng.OW), dtype=torch.float32, device=dev)
        trunk_id, head_id = STYLE_TRUNK[style], STYLE_HEAD[style]
        stream = torch.cuda.current_stream(dev).cuda_stream
        mk = {}
        if masks is not None:
            mk = dict(masks=masks.data_ptr(), n_masks=int(masks.shape[0]),
                      stream_mask_a=[-1] * n_rot + list(mask_a), stream_mask_b=[-1] * n_rot + list(mask_b))
        eng.forward(self._net_struct(False), trunk_id, head_id, q.data_ptr(), stream,
                    heightmaps=heightmaps.data_ptr(), hm_size=hm, mean=float(mean), std=float(std),
                    n_images=int(heightmaps.shape[0]), stream_image=stream_image, stream_affine=np.concatenate(thetas),
                    stream_rotated=stream_rot, pair_a=pair_a, pair_b=pair_b,
                    bn_seq_trunk=bn_seq_trunk, bn_seq_head=bn_seq_head, **mk)
        self._saved = None
        return q

    def _engine_backward(self, token, dq, phase=None):
        """phase None: the whole backward; 0 / 1: its two halves (a data-parallel caller all-reduces the gradients the first half
        finished while the second runs, parallel.OverlappedGradSync)."""
        if self._saved is None or self._saved[1] != token or not self._saved[0].h or self._saved[0].forward_id != token:
            raise RuntimeError("backward: the activations of that forward are gone (another forward ran on the engine)")
        eng, _, trunk_id, head_id = self._saved
        if eng.precision != self.precision:
            raise RuntimeError("backward: the engine's precision changed since the forward")
        stream = torch.cuda.current_stream(dq.device).cuda_stream
        # smg_backward ACCUMULATES into the flat gradient buffer (like autograd into p.grad).  A torch optimizer's
        # zero_grad(set_to_none=True) only drops p.grad and never sees that buffer, so a range whose parameters
        # have no .grad is started from zero here; a range that still has its .grad keeps accumulating.
        g = self.flat_grads()
        if phase in (None, 0):
            for (off, n), probe in ((smg_hip.trunk_range(self.HEAD_OUT, trunk_id), self._range_probe[("t", trunk_id)]),
                                    (smg_hip.head_range(self.HEAD_OUT, head_id), self._range_probe[("h", head_id)])):
                if probe.grad is None and not self._grads_clean:
                    g[off:off + n].zero_()
                if getattr(self, "_dirty_ranges", None) is not None:
                    self._dirty_ranges.add((off, n))
            self._grads_clean = False
        eng.backward(self._net_struct(True), dq.data_ptr(), stream, phase)
        if phase in (None, 1):
            self.expose_grads(trunk_id, head_id)

    # ---- the reference interface -------------------------------------------------------------
    def forward(self, input_depth_data, m_input_depth_data, style=0, is_volatile=False, specific_rotation=-1):
        """code/models.py:361 (reinforcement_net) / :72 (reactive_net)."""
        self._require_gpu()
        dev = self._flat_params.device
        imgs = torch.cat((input_depth_data, m_input_depth_data), dim=0).to(device=dev, dtype=torch.float32).contiguous()
        if is_volatile and specific_rotation == -1:                     # branch A, models.py:363-437
            if style == 0:
                rots, num = list(range(self.gnum_rotations)), self.gnum_rotations
            elif style == 1:
                rots, num = list(range(self.snum_rotations)), self.snum_rotations
            else:
                rots, num = [0], self.gnum_rotations
            q = self.run(style, rots, num, images_nchw=imgs)
            return [q[i:i + 1] for i in range(len(rots))]
        rot = 0 if style == 2 else specific_rotation                    # models.py:418,446,469 (gnum for style 1 too)
        if is_volatile:                                                 # branch B, models.py:439-510
            return self.run(style, [rot], self.gnum_rotations, images_nchw=imgs)
        q = self.run(style, [rot], self.gnum_rotations, images_nchw=imgs, keep_for_backward=True)   # branch C, :513-586
        if self._autograd_hook is None or self._autograd_hook.device != dev:
            self._autograd_hook = torch.zeros(1, device=dev, requires_grad=True)
        out = _EngineFn.apply(self._autograd_hook, self, q, self._saved[1])
        self.gra_prob, self.suc_prob, self.gs_prob = [], [], []
        setattr(self, _PROB_ATTR[style], out)
        return out


class reinforcement_net(_AffordanceNet):
    """code/models.py:301-586: three DenseNet-121 trunks + three 1-channel Q heads."""
    HEAD_OUT = 1

    def __init__(self, use_cuda):
        super(reinforcement_net, self).__init__(use_cuda)


class reactive_net(_AffordanceNet):
    """code/models.py:15-296: the same with 3-class heads."""
    HEAD_OUT = 3

    def __init__(self, use_cuda):
        super(reactive_net, self).__init__(use_cuda)
