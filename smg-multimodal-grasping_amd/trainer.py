"""MI355X-native stand-in for the reference's `trainer.Trainer`
(/root/reference/code/trainer.py:17-384): same constructor, attributes and
forward / get_label_value / backprop signatures and return types, so
`from trainer import Trainer` (code/main.py:17) keeps working - but every network
evaluation, the Huber / cross-entropy loss, the backward pass and the Adam step run
in libsmg_hip.so (hand-written HIP for gfx950).

Differences that are deliberate and documented:
  * image_mean / image_std are constructor arguments (default 0.01 / 0.03); the
    released constants are [0,0,0] / [0,0,0] (code/trainer.py:176-177) which makes
    every network input inf/NaN.  `literal_reference=True` reproduces that.
  * the masked stream's trunk pass is computed once per sweep instead of once per
    rotation (code/models.py:385 sits inside the rotation loop); results are identical.
  * no CPU mode: without a GPU (or with force_cpu=True) the first forward raises.
"""
import copy
import os
import time

import numpy as np
import torch

import smg_hip
from models import STYLE_HEAD, STYLE_TRUNK, reactive_net, reinforcement_net

_ACTION_STYLE = {"grasp": 0, "suction": 1, "grasp_then_suction": 2}


class FusedAdam(object):
    """torch.optim.Adam(lr=1e-4, betas=(0.9,0.999), eps=1e-8, weight_decay=0)
    (code/trainer.py:99) over the model's flat parameter buffer.  Like torch >= 2
    (zero_grad(set_to_none=True)) only parameters that received a gradient in this
    step are updated: the engine reports which (trunk, head) segments those are."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.model, self.lr, self.betas, self.eps = model, lr, betas, eps
        self.m = None
        self.v = None
        self.steps = {}

    def zero_grad(self, set_to_none=True):
        self.model.zero_grad()

    def _segments(self):
        if self.model._saved is None:
            return []
        _, _, trunk_id, head_id = self.model._saved
        return [("trunk%d" % trunk_id, smg_hip.trunk_range(self.model.HEAD_OUT, trunk_id)),
                ("head%d" % head_id, smg_hip.head_range(self.model.HEAD_OUT, head_id))]

    def step(self, segments=None):
        model = self.model
        p = model._flat_params
        if self.m is None or self.m.device != p.device:
            self.m = torch.zeros_like(p)
            self.v = torch.zeros_like(p)
        stream = torch.cuda.current_stream(p.device).cuda_stream
        for name, (off, n) in (segments if segments is not None else self._segments()):
            self.steps[name] = self.steps.get(name, 0) + 1
            smg_hip.adam_step(p.data_ptr(), model.flat_grads().data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                              off, n, self.steps[name], self.lr, self.betas[0], self.betas[1], self.eps, stream)


class Trainer(object):
    def __init__(self, method, future_reward_discount, load_snapshot, snapshot_file, force_cpu,
                 image_mean=0.01, image_std=0.03, literal_reference=False):
        self.method = method
        # code/trainer.py:22-31
        if torch.cuda.is_available() and not force_cpu:
            print("CUDA detected. Running with GPU acceleration.")
            self.use_cuda = True
        elif force_cpu:
            print("CUDA detected, but overriding with option '--cpu'. Running with only CPU.")
            self.use_cuda = False
        else:
            print("CUDA is *NOT* detected. Running with only CPU.")
            self.use_cuda = False
        self.image_mean = 0.0 if literal_reference else float(image_mean)
        self.image_std = 0.0 if literal_reference else float(image_std)

        if self.method == 'reactive':                                   # code/trainer.py:34-69
            self.model = reactive_net(self.use_cuda)
            if load_snapshot:
                self.model.load_state_dict(torch.load(snapshot_file))
                print('Pre-trained model snapshot loaded from: %s' % (snapshot_file))
            if self.use_cuda:
                self.model = self.model.cuda()
        elif self.method == 'reinforcement':                            # code/trainer.py:72-92
            self.model = reinforcement_net(self.use_cuda)
            self.model_target = copy.deepcopy(self.model)
            self.model_target.load_state_dict(self.model.state_dict())
            self.future_reward_discount = future_reward_discount
            if load_snapshot:
                self.model.load_state_dict(torch.load(snapshot_file))
                print('Pre-trained model snapshot loaded from: %s' % (snapshot_file))
            if self.use_cuda:
                self.model = self.model.cuda()
                self.model_target = self.model_target.cuda()
        else:
            raise ValueError("method must be 'reactive' or 'reinforcement'")

        self.model.train()                                              # code/trainer.py:95
        self.optimizer = FusedAdam(self.model)                          # code/trainer.py:99
        self.iteration = 0
        # code/trainer.py:105-114
        self.executed_action_log = []
        self.label_value_log = []
        self.reward_value_log = []
        self.predicted_value_log = []
        self.use_heuristic_log = []
        self.is_exploit_log = []
        self.clearance_log = []
        self.grasping_type_log = []
        self.episode_success_log = []
        self.training_loss_log = []

    # ---- session resume ---------------------------------------------------------------------
    # (file stem, attribute, layout): how code/trainer.py:118-160 re-reads each text log of a
    # previous session.  "rows" keeps the first `iteration` rows of a 2-D log, "col" the first
    # `iteration` entries of a 1-D log as an [iteration, 1] column, "col_all" every entry.
    _LOG_FILES = (
        ("executed-action", "executed_action_log", "rows"),
        ("label-value", "label_value_log", "col"),
        ("predicted-value", "predicted_value_log", "col"),
        ("reward-value", "reward_value_log", "col"),
        ("use-heuristic", "use_heuristic_log", "col"),
        ("is-exploit", "is_exploit_log", "col"),
        ("clearance", "clearance_log", "col_all"),
        ("grasping_type", "grasping_type_log", "col"),
        ("episode_success", "episode_success_log", "rows"),
        ("training_loss", "training_loss_log", "rows"),
    )

    def preload(self, transitions_directory):
        """code/trainer.py:118-160 (`--continue_logging`, code/main.py:74): reload the ten
        `<name>.log.txt` files of a logging session as python lists and resume the
        iteration counter at (rows of executed-action.log.txt) - 2."""
        def read(stem):
            return np.loadtxt(os.path.join(transitions_directory, stem + ".log.txt"), delimiter=" ")
        self.iteration = read("executed-action").shape[0] - 2
        n = self.iteration
        for stem, attr, how in self._LOG_FILES:
            a = read(stem)
            if how == "rows":
                a = a[0:n, :]
            elif how == "col":
                a = a[0:n].reshape(n, 1)
            else:
                a = a.reshape(a.shape[0], 1)
            setattr(self, attr, a.tolist())

    # ---- network evaluation -----------------------------------------------------------------
    def _heightmaps_to_device(self, depth_heightmap, m_depth_heightmap):
        hm = np.stack([np.asarray(depth_heightmap, dtype=np.float64), np.asarray(m_depth_heightmap, dtype=np.float64)])
        if hm.ndim != 3 or hm.shape[1] != hm.shape[2]:
            raise ValueError("heightmaps must be square 2-D arrays")
        dev = self.model._flat_params.device
        return torch.from_numpy(np.ascontiguousarray(hm)).to(dev)

    def _evaluate(self, model, depth_heightmap, m_depth_heightmap, style, is_volatile, specific_rotation):
        """Rotation selection of reinforcement_net.forward / reactive_net.forward
        (code/models.py:363-586) on the heightmap fast path: the x2 zoom, padding,
        3-channel replication and normalisation of code/trainer.py:165-191 happen inside
        the engine's input kernel."""
        model._require_gpu()
        hm = self._heightmaps_to_device(depth_heightmap, m_depth_heightmap)
        if is_volatile and specific_rotation == -1:
            if style == 0:
                rots, num = list(range(model.gnum_rotations)), model.gnum_rotations
            elif style == 1:
                rots, num = list(range(model.snum_rotations)), model.snum_rotations
            else:
                rots, num = [0], model.gnum_rotations
        else:
            rots, num = [0 if style == 2 else specific_rotation], model.gnum_rotations
        return model.run(style, rots, num, heightmaps=hm, mean=self.image_mean, std=self.image_std,
                         keep_for_backward=not is_volatile)

    def forward(self, depth_heightmap, m_depth_heightmap, style=0, is_volatile=False, is_target=False, specific_rotation=-1):
        """code/trainer.py:162-209.  Returns np.ndarray float64 of length R (reinforcement)
        or a python float P(success) (reactive, :195-199)."""
        with np.errstate(divide="ignore", invalid="ignore"):
            if self.method == 'reactive':
                q = self._evaluate(self.model, depth_heightmap, m_depth_heightmap, style, is_volatile, specific_rotation)
                self._last_q = q
                logits = q[0].reshape(1, 3, 1, 1)
                return torch.softmax(logits, dim=1).cpu().numpy()[0, 0, 0][0]
            model = self.model_target if is_target else self.model
            q = self._evaluate(model, depth_heightmap, m_depth_heightmap, style, is_volatile, specific_rotation)
            self._last_q = q
            if q.shape[2] * q.shape[3] != 1:
                raise NotImplementedError("dense Q maps (input larger than 640) cannot be returned through the "
                                          "reference's scalar-per-rotation array (code/trainer.py:205-207)")
            return q.reshape(-1).cpu().numpy().astype(np.float64)

    def _objects_on_device(self, model, depth_heightmap, mask_depth):
        dev = model._flat_params.device
        d = torch.from_numpy(np.ascontiguousarray(np.asarray(depth_heightmap, dtype=np.float64))[None]).to(dev)
        m = torch.from_numpy(np.ascontiguousarray(np.asarray(mask_depth, dtype=np.float64))).to(dev)
        if m.ndim != 3 or m.shape[1:] != d.shape[1:]:
            raise ValueError("mask_depth must be [n_objects, H, H] like the heightmap")
        return d, m

    def forward_objects(self, depth_heightmap, mask_depth, style=0, is_target=False, return_device=False):
        """All objects of one scene in ONE engine call - the loop of code/main.py:158-166:

            for num in range(objects_number):
                gra_conf[num] = trainer.forward(depth, depth * mask_depth[num], style, is_volatile=True)

        The rotated full-depth streams are the same for every object, so n objects x R rotations
        cost R + n trunk passes (the reference runs 2*n*R).  The products depth * mask[k] are formed
        on the device by the input kernel (the host ships the heightmap and the masks once).  BN
        running statistics are updated in the reference's order and count.
        Returns conf[n_objects, R] float64 (styles 0 / 1), or the device tensor [n*R] if asked."""
        if self.method != 'reinforcement' or style not in (0, 1):
            raise ValueError("forward_objects: reinforcement styles 0 (grasp) and 1 (suction)")
        model = self.model_target if is_target else self.model
        d, m = self._objects_on_device(model, depth_heightmap, mask_depth)
        n = int(m.shape[0])
        R = model.gnum_rotations if style == 0 else model.snum_rotations
        pairs = [(r, k) for k in range(n) for r in range(R)]
        seq_t = [v for k in range(n) for r in range(R) for v in (r, R + k)]       # trunk(rot r), trunk(mask k) per sample
        q = model.run_pairs(style, R, d, list(range(R)), [0] * n, pairs, self.image_mean, self.image_std,
                            bn_seq_trunk=seq_t, bn_seq_head=list(range(len(pairs))),
                            masks=m, mask_a=list(range(n)), mask_b=[-1] * n)
        if q.shape[2] * q.shape[3] != 1:
            raise NotImplementedError("dense Q maps")
        if return_device:
            return q.reshape(-1)
        return q.reshape(n, R).cpu().numpy().astype(np.float64)

    def forward_object_pairs(self, depth_heightmap, mask_depth, is_target=False, return_device=False):
        """The enveloping-then-sucking loop of code/main.py:183-192 in one engine call: for every
        unordered object pair (g < s) the mask is mask[g] + mask[s] (summed and applied on the device),
        style 2, rotation 0.  Returns gs_conf[n, n] with -100 where the reference leaves its fill value
        (main.py:184), or (device tensor of the pair values, [(g, s)]) if asked."""
        model = self.model_target if is_target else self.model
        n = int(np.asarray(mask_depth).shape[0])
        gs = np.full((n, n), -100.0)
        idx = [(g, s) for g in range(n) for s in range(g + 1, n)]
        if not idx:
            return (None, idx) if return_device else gs
        d, m = self._objects_on_device(model, depth_heightmap, mask_depth)
        pairs = [(0, k) for k in range(len(idx))]
        seq_t = [v for k in range(len(idx)) for v in (0, 1 + k)]
        q = model.run_pairs(2, model.gnum_rotations, d, [0], [0] * len(idx), pairs, self.image_mean, self.image_std,
                            bn_seq_trunk=seq_t, bn_seq_head=list(range(len(idx))),
                            masks=m, mask_a=[g for g, _ in idx], mask_b=[s_ for _, s_ in idx])
        if return_device:
            return q.reshape(-1), idx
        vals = q.reshape(-1).cpu().numpy().astype(np.float64)
        for (g, s), v in zip(idx, vals):
            gs[g, s] = v
        return gs

    def best_actions(self, depth_heightmap, mask_depth, is_ets=True):
        """The whole per-step evaluation of code/main.py:158-195 - grasp and suction sweeps over every object, the ES
        pass over every object pair, and the three np.argmax selections - in three engine calls whose maxima are
        found on the device (smg_argmax: lowest index on ties, like np.argmax); the host reads back three
        (index, value) pairs instead of 2*n*R + n(n-1)/2 scalars.
        Returns bestg_id / bests_id = (object, rotation), bestg_conf / bests_conf, and for ES bestgs_num = (g, s),
        bestgs_conf (None / 0 with fewer than two objects, main.py:180-183)."""
        model = self.model
        dev = model._flat_params.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        idx = torch.empty(3, dtype=torch.int32, device=dev)
        val = torch.zeros(3, dtype=torch.float32, device=dev)
        R = (model.gnum_rotations, model.snum_rotations)
        for style in (0, 1):
            q = self.forward_objects(depth_heightmap, mask_depth, style, return_device=True)
            smg_hip.argmax(q.data_ptr(), q.numel(), idx[style:].data_ptr(), val[style:].data_ptr(), stream)
        pair_list = []
        if is_ets and np.asarray(mask_depth).shape[0] > 1:
            q, pair_list = self.forward_object_pairs(depth_heightmap, mask_depth, return_device=True)
            smg_hip.argmax(q.data_ptr(), q.numel(), idx[2:].data_ptr(), val[2:].data_ptr(), stream)
        i, v = idx.cpu().numpy(), val.cpu().numpy().astype(np.float64)
        out = {"bestg_id": (int(i[0]) // R[0], int(i[0]) % R[0]), "bestg_conf": v[0],
               "bests_id": (int(i[1]) // R[1], int(i[1]) % R[1]), "bests_conf": v[1],
               "bestgs_num": None, "bestgs_conf": 0}
        if pair_list:
            out["bestgs_num"], out["bestgs_conf"] = pair_list[int(i[2])], v[2]
        return out

    def get_label_value(self, primitive_action, objects_number,
                        suction_success, grasp_success, gs_success,
                        depth_heightmap, mask_depth, objects_mask,
                        bestg_id, bests_id, bestgs_g_id, bestgs_s_id,
                        exploit_action, bestg_conf, bests_conf, bestgs_conf):
        """code/trainer.py:212-274."""
        if self.method == 'reactive':
            label_value = 0
            if primitive_action == 'suction':
                success_value = suction_success
                if not suction_success:
                    label_value = 1
            elif primitive_action == 'grasp':
                success_value = grasp_success
                if not grasp_success:
                    label_value = 1
            elif primitive_action == 'grasp_then_suction':
                success_value = gs_success
                label_value = 0 if gs_success == 2.5 else 1
            print('Label value: %d' % (label_value))
            return label_value, success_value

        current_reward = 0
        if primitive_action == 'suction':
            current_reward = suction_success
        elif primitive_action == 'grasp':
            current_reward = grasp_success
        elif primitive_action == 'grasp_then_suction':
            current_reward = gs_success
        if suction_success == 0 and grasp_success == 0 and gs_success == 0:
            future_reward = 0
        elif (objects_number == 1 and suction_success == 1) or (objects_number == 1 and grasp_success == 1) or \
                (objects_number == 2 and gs_success == 2.5):
            future_reward = 0
        else:
            if exploit_action == 'grasp':
                m = depth_heightmap * mask_depth[bestg_id[0]]
                future_reward = self.forward(depth_heightmap, m, style=0, is_volatile=True, is_target=True, specific_rotation=bestg_id[1])[0]
            elif exploit_action == 'suction':
                m = depth_heightmap * mask_depth[bests_id[0]]
                future_reward = self.forward(depth_heightmap, m, style=1, is_volatile=True, is_target=True, specific_rotation=bests_id[1])[0]
            elif exploit_action == 'grasp_then_suction':
                m = depth_heightmap * (mask_depth[bestgs_g_id[0]] + mask_depth[bestgs_s_id[0]])
                future_reward = self.forward(depth_heightmap, m, style=2, is_volatile=True, is_target=True, specific_rotation=bestgs_g_id[1])[0]
        expected_reward = current_reward + self.future_reward_discount * future_reward
        print('Expected reward: %f + %f x %f = %f' % (current_reward, self.future_reward_discount, future_reward, expected_reward))
        return expected_reward, current_reward

    def backprop(self, depth_heightmap, primitive_action,
                 bestg_id, bests_id, bestgs_g_id, bestgs_s_id,
                 label_value, objects_mask, sro_best, gro_best, bestgs_num):
        """code/trainer.py:278-384: one sample, one optimizer step; returns a 0-d array."""
        mask_depth = objects_mask.copy()
        objects_mask.shape = (objects_mask.shape[0], objects_mask.shape[1], objects_mask.shape[2], 1)   # trainer.py:288,336
        style = _ACTION_STYLE[primitive_action]
        if style == 0:
            m = depth_heightmap * mask_depth[bestg_id[0]]
            rot = bestg_id[1]
        elif style == 1:
            m = depth_heightmap * mask_depth[bests_id[0]]
            rot = bests_id[1]
        else:
            m = depth_heightmap * (mask_depth[bestgs_g_id[0]] + mask_depth[bestgs_s_id[0]])
            rot = bestgs_g_id[1]
        loss_value = self.train_step(depth_heightmap, m, style, rot, label_value)
        print('Training loss: %f' % (loss_value))
        return loss_value

    def train_batch(self, depth_heightmap, m_depth_heightmap, style, rotations, labels, grad_sync=None, return_q=False):
        """Batched form of backprop: every (scene, rotation) is a training sample (forward as
        branch C, Huber / CE against its label); the gradient of the SUM of the losses is
        accumulated in one backward pass (each scene's masked stream is walked once with the summed
        gradient - exact, the trunk is linear in its output gradient), then ONE Adam step.  Equals
        that many reference backprop calls with the optimizer step deferred to the end.

        One scene: 2-D heightmaps, `rotations` a list of rotation indices.  Several scenes
        (SURVEY.md config 4): heightmaps [n_scenes, H, H], `rotations` a list of lists; labels are
        flat, scene-major.  `grad_sync(model, trunk_id, head_id)` is the data-parallel hook
        (parallel.allreduce_grads) called between backward and Adam; a hook with `.overlapped` set (parallel.OverlappedGradSync)
        gets `.start()` after the first half of the backward and `.finish()` after the second.  Returns the loss vector."""
        model = self.model
        self.optimizer.zero_grad()
        model._require_gpu()
        if torch.is_tensor(depth_heightmap) and depth_heightmap.is_cuda:
            # device-resident inputs (float64 heightmaps, float32 labels): nothing crosses PCIe and - unlike a pageable
            # host-to-device copy - nothing makes the host wait for the previous step's kernels
            d = depth_heightmap.to(dtype=torch.float64)
            m = m_depth_heightmap.to(device=d.device, dtype=torch.float64)
            if d.dim() == 2:
                d, m, rotations = d[None], m[None], [list(rotations)]
            hm = torch.stack((d, m), dim=1).reshape((2 * d.shape[0],) + tuple(d.shape[1:])).contiguous()
        else:
            d = np.asarray(depth_heightmap, dtype=np.float64)
            m = np.asarray(m_depth_heightmap, dtype=np.float64)
            if d.ndim == 2:
                d, m, rotations = d[None], m[None], [list(rotations)]
            hm = np.empty((2 * d.shape[0],) + d.shape[1:], dtype=np.float64)
            hm[0::2], hm[1::2] = d, m
            hm = torch.from_numpy(hm).to(model._flat_params.device)
        num = model.gnum_rotations                     # code/models.py:522,545,568 (gnum for every style)
        rots = [[0 if style == 2 else int(r) for r in rs] for rs in rotations]
        dev = model._flat_params.device
        if torch.is_tensor(labels):
            lab = labels.to(device=dev, dtype=torch.float32).reshape(-1)
        else:      # (uploaded BEFORE the forward is enqueued: behind it, the pageable copy would hold the host until the forward has run)
            lab_h = np.asarray(labels, dtype=np.float32).reshape(-1)
            if self.method == 'reactive' and not np.isin(lab_h, (0.0, 1.0, 2.0)).all():
                # torch's nll_loss (code/utils.py:311) raises on a class index outside [0, 3); the loss kernel would
                # silently treat it as the weight-0 class.  (Device-resident labels are not read back: same contract.)
                raise ValueError("reactive labels must be class indices 0, 1 or 2")
            lab = torch.as_tensor(lab_h, device=dev)
        q = model.run(style, rots, num, heightmaps=hm, mean=self.image_mean, std=self.image_std, keep_for_backward=True)
        n = q.shape[0]
        eng, token, trunk_id, head_id = model._saved
        stream = torch.cuda.current_stream(dev).cuda_stream
        if lab.numel() != n:
            raise ValueError("one label per (scene, rotation) sample")
        loss = torch.empty(n, dtype=torch.float32, device=dev)
        dq = torch.empty_like(q)
        eng.loss(0 if self.method == 'reinforcement' else 1, q.data_ptr(), lab.data_ptr(), n, loss.data_ptr(), dq.data_ptr(), stream)
        if grad_sync is not None and getattr(grad_sync, "overlapped", False):
            # the all-reduce of everything behind dense block 1 (most of the parameters) runs under the second half of the backward
            model._engine_backward(token, dq, phase=0)
            grad_sync.start(model, trunk_id, head_id)
            model._engine_backward(token, dq, phase=1)
            grad_sync.finish(model, trunk_id, head_id)
        else:
            model._engine_backward(token, dq)
            if grad_sync is not None:
                grad_sync(model, trunk_id, head_id)
        self.optimizer.step()
        return (loss, q) if return_q else loss

    # The single-sample step of Trainer.backprop as ONE replayed hipGraph (smg_train_step_graph): ~560 launches of 2-20 us each are
    # enqueued by one hipGraphLaunch instead of one by one (same results, bit for bit at zero learning rate).  It saves a little host time
    # (1.8 ms per step instead of 2.3) and costs latency: the graph's ~560 dependent nodes execute no faster than the same launches from
    # two streams whose host stays ahead of the GPU - 6.6-6.8 ms per step as a graph against 5.5 ms as separate calls (bench.py;
    # splitting the graph so that its launch cost hides changed nothing).  The reference's loop reads the loss of every step before it
    # continues (latency, not throughput), so the separate calls are the default.
    use_step_graph = False

    def _train_step_graph(self, depth_heightmap, m_depth_heightmap, style, rotation, label_value):
        t_host = time.perf_counter()
        model = self.model
        model._require_gpu()
        dev = model._flat_params.device
        hm = np.stack([np.asarray(depth_heightmap, dtype=np.float64), np.asarray(m_depth_heightmap, dtype=np.float64)])
        if hm.ndim != 3 or hm.shape[1] != hm.shape[2]:
            raise ValueError("heightmaps must be square 2-D arrays")
        st = getattr(self, "_step_state", None)
        if st is None or st["model"] is not model or st["hm"].shape != hm.shape or st["hm"].device != dev:
            # persistent device buffers: the captured graph is keyed on their addresses
            st = self._step_state = dict(model=model, hm=torch.empty(hm.shape, dtype=torch.float64, device=dev),
                                         label=torch.empty(1, dtype=torch.float32, device=dev), loss=torch.empty(1, dtype=torch.float32, device=dev),
                                         q=None, dq=None)
        if st["q"] is None or st["q"].shape[1] != model.HEAD_OUT:
            st["q"] = torch.empty((1, model.HEAD_OUT, 1, 1), dtype=torch.float32, device=dev)
            st["dq"] = torch.empty_like(st["q"])
        st["hm"].copy_(torch.from_numpy(np.ascontiguousarray(hm)))
        st["label"].fill_(float(label_value))
        key = (STYLE_TRUNK[style], STYLE_HEAD[style])
        if model._graph_exposed != key:
            self.optimizer.zero_grad()          # the reference's zero_grad: every p.grad dropped (the graph zeroes its own two ranges)
        rot = 0 if style == 2 else rotation
        model.train_step_graph(style, rot, model.gnum_rotations, st["hm"], st["label"], st["loss"], st["q"], st["dq"], self.optimizer,
                               0 if self.method == 'reinforcement' else 1, mean=self.image_mean, std=self.image_std)
        if model._graph_exposed != key:
            model.expose_grads(*key)
            model._graph_exposed = key
        setattr(model, ("gra_prob", "suc_prob", "gs_prob")[style], st["q"])
        self.last_enqueue_ms = (time.perf_counter() - t_host) * 1e3        # host time of the step up to the loss read-back (bench.py)
        return np.asarray(st["loss"].cpu().numpy()[0])

    def train_step(self, depth_heightmap, m_depth_heightmap, style, rotation, label_value):
        """zero_grad -> forward (branch C) -> loss -> backward -> Adam, all on the device;
        the only host synchronisation is reading the loss back (as code/trainer.py:352 does)."""
        if self.use_step_graph and np.shape(depth_heightmap)[-1] == 224:      # (S = 640: one Q value per sample; larger inputs keep the eager calls)
            return self._train_step_graph(depth_heightmap, m_depth_heightmap, style, rotation, label_value)
        t_host = time.perf_counter()
        model = self.model
        self.optimizer.zero_grad()
        # (the label goes up BEFORE the forward is enqueued: a pageable host-to-device copy returns only when the stream has reached it -
        #  behind the forward it held the host for the forward's ~1.5 ms, with the backward not yet enqueued)
        labels = torch.tensor([float(label_value)], dtype=torch.float32, device=model._flat_params.device)
        q = self._evaluate(model, depth_heightmap, m_depth_heightmap, style, False, rotation)
        dev = q.device
        eng, token, trunk_id, head_id = model._saved
        stream = torch.cuda.current_stream(dev).cuda_stream
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dq = torch.empty_like(q)
        eng.loss(0 if self.method == 'reinforcement' else 1, q.data_ptr(), labels.data_ptr(), 1, loss.data_ptr(), dq.data_ptr(), stream)
        model._engine_backward(token, dq)
        self.optimizer.step()
        setattr(model, ("gra_prob", "suc_prob", "gs_prob")[style], q)
        self.last_enqueue_ms = (time.perf_counter() - t_host) * 1e3
        return np.asarray(loss.cpu().numpy()[0])
