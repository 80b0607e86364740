/*
 * smg_hip.h - C ABI of the MI355X (gfx950) affordance engine.
 *
 * Drop-in boundary for ONE path of fukangl/SMG-multimodal-grasping: the
 * Trainer.forward / Trainer.backprop -> reinforcement_net / reactive_net .forward
 * loop.  The reference is pure Python on torch (no FFI of its own), so these entry
 * points are what a ctypes binding for that path binds (INTEGRATION.md shows the
 * stub); every function cites the reference interface it stands in for, paths
 * relative to the reference repository root.
 *
 * Conventions
 *   - plain pointers and sizes only; every `dev` pointer is device memory of the
 *     engine's GPU, every `host` pointer is ordinary host memory;
 *   - all work is enqueued on the hipStream_t passed as `stream` (void*, 0 = the
 *     null stream); functions with host outputs synchronise that stream;
 *   - return 0 on success, a negative errno-style code on failure;
 *     smg_last_error() gives the message (thread-local);
 *   - one engine per GPU per process; an engine is not re-entrant.
 *
 * Network state lives in three caller-owned flat device arrays whose layout is
 * defined by smg_layout_*():
 *     params  float32[smg_layout_param_floats]   (weights, BN gamma/beta, classifier)
 *     grads   float32[same]                      (same offsets)
 *     bufs    float32[smg_layout_buffer_floats]  (BN running_mean / running_var)
 *     nbt     int64  [smg_layout_nbt_count]      (BN num_batches_tracked)
 * Entry i of the layout carries the torchvision/torch state_dict key of the
 * reference model (code/models.py:308-343; torchvision densenet121 names), so a
 * reference snapshot (code/logger.py:121-125) maps 1:1 onto the arrays.
 */
#ifndef SMG_HIP_H
#define SMG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smg_engine smg_engine;

/* ---- library ------------------------------------------------------------------ */
const char* smg_last_error(void);
/* ABI revision of this header: a binding must refuse a library whose smg_version() differs (stale .so) and should
 * compare its own struct sizes with smg_abi_struct_bytes(0 = smg_batch, 1 = smg_net, 2 = smg_adam) before the first call. */
#define SMG_ABI_VERSION 4
int smg_version(void);
int smg_abi_struct_bytes(int which);

/* ---- state layout (replaces nn.Module.state_dict() ordering of
 *      code/models.py:301-358 reinforcement_net / :15-69 reactive_net) ------------- */
enum { SMG_KIND_PARAM = 0, SMG_KIND_RUNNING_MEAN = 1, SMG_KIND_RUNNING_VAR = 2, SMG_KIND_NBT = 3 };
int     smg_layout_count(int head_out);               /* 2217 entries                  */
int64_t smg_layout_param_floats(int head_out);        /* 24 419 256 (head_out=1)       */
int64_t smg_layout_buffer_floats(int head_out);
int64_t smg_layout_nbt_count(int head_out);
/* name: caller buffer of name_cap bytes; offset: element offset into the array the
 * kind selects; shape: up to 4 dims (ndim 0 for nbt scalars). */
int smg_layout_entry(int head_out, int index, char* name, int name_cap, int* kind,
                     int64_t* offset, int* ndim, int64_t shape[4]);

/* ---- engine ------------------------------------------------------------------- */
/* input_size S = side of the padded network input (640 for a 224x224 heightmap,
 * code/trainer.py:165-173); max_streams = most trunk passes per call; max_pairs =
 * most (rotated, masked) head evaluations per call; head_out = 1 (reinforcement_net)
 * or 3 (reactive_net). Allocates all activation / gradient workspaces once. */
int smg_engine_create(int device, int input_size, int max_streams, int max_pairs, int head_out,
                      smg_engine** out);
void smg_engine_destroy(smg_engine* e);
int64_t smg_engine_workspace_bytes(const smg_engine* e);

/* Network binding: the three flat arrays of ONE model instance (model or
 * model_target, code/trainer.py:73-75). */
typedef struct {
    float*   params;
    float*   grads;     /* may be NULL for inference-only nets */
    float*   bufs;
    int64_t* nbt;
} smg_net;

/* One batch of work.  A "stream" is one DenseNet-121 `.features` pass over one image
 * (code/models.py:384-385); a "pair" is one head evaluation on the concatenation of
 * two streams' features (code/models.py:386-387).
 *
 * Images come in one of two forms:
 *   images_nchw != NULL : n_images tensors [3,S,S] float32, what
 *       reinforcement_net.forward receives (code/models.py:361);
 *   heightmaps  != NULL : n_images arrays [hm_size,hm_size] float64, what
 *       Trainer.forward receives (code/trainer.py:162); the engine applies the x2
 *       nearest zoom, zero padding to S, 3-channel replication and (x-mean)/std of
 *       code/trainer.py:165-191 on the fly.
 * stream_image[s]  : which image stream s reads;
 * stream_affine[s] : 6 float32 (row-major 2x3 theta of F.affine_grid,
 *       code/models.py:374-378); the stream samples its image through
 *       affine_grid + grid_sample(nearest, align_corners=True) (code/models.py:378-382);
 * stream_rotated[s]: 0 = feed the image as is (the masked stream, models.py:385).
 * pair_a / pair_b  : stream indices whose norm5 features form channels 0..1023 /
 *       1024..2047 of the head input.
 * bn_seq_trunk     : stream indices in the order the reference would have run them
 *       (BN running statistics are updated once per entry);
 *       bn_seq_head the same for pairs.  NULL/0 = do not touch running statistics.
 */
typedef struct {
    int n_images;
    const float*  images_nchw_dev;
    const double* heightmaps_dev;
    int hm_size;
    double image_mean, image_std;
    int n_streams;
    const int*   stream_image;     /* host */
    const float* stream_affine;    /* host, 6 per stream */
    const int*   stream_rotated;   /* host */
    int n_pairs;
    const int* pair_a;             /* host */
    const int* pair_b;             /* host */
    int n_bn_seq_trunk;
    const int* bn_seq_trunk;       /* host */
    int n_bn_seq_head;
    const int* bn_seq_head;        /* host */
    /* Object masks applied on the device (heightmap form only), replacing the host products
     * `depth * mask[k]` / `depth * (mask[g] + mask[s])` of code/main.py:160,187: masks_dev holds n_masks arrays
     * [hm_size,hm_size] float64; stream s reads heightmap * (mask[stream_mask_a[s]] + mask[stream_mask_b[s]]), an
     * index of -1 meaning "no such term" (both -1: the plain heightmap).  NULL = no masking. */
    const double* masks_dev;
    int n_masks;
    const int* stream_mask_a;      /* host */
    const int* stream_mask_b;      /* host */
} smg_batch;

/* Forward: trunk `trunk_id` (0 suction_depth_trunk, 1 grasp_depth_trunk,
 * 2 gs_depth_trunk - layout order) and head `head_id` (0 suctionnet_val,
 * 1 graspnet_val, 2 gsnet_val).  q_out_dev: float32 [n_pairs][head_out][OH][OW]
 * (OH=OW=1 for S=640).  Keeps every activation needed by smg_backward until the next
 * forward on this engine.  Replaces reinforcement_net.forward / reactive_net.forward
 * (code/models.py:361-586, :72-296) for any of their branches.
 * NaN / inf: a non-finite BatchNorm batch statistic of a stream (pair) makes every Q value of the samples that use it NaN,
 * as it does in the reference (checked in every forward, with or without bn_seq_trunk / bn_seq_head). */
int smg_forward(smg_engine* e, const smg_net* net, int trunk_id, int head_id,
                const smg_batch* batch, float* q_out_dev, void* stream);

/* Loss on the device, replacing the python Huber of code/trainer.py:345-348
 * (mode 0: element [0,0,0] of each pair's output vs labels[pair]) and the weighted
 * cross entropy of code/trainer.py:296-299 + code/utils.py:306-313 (mode 1: 3 logits
 * vs integer class labels[pair], class weights {1,1,0}).  Writes loss_dev[n_pairs]
 * and dq_dev (same shape as q). */
int smg_loss(smg_engine* e, int mode, const float* q_dev, const float* labels_dev, int n_pairs,
             float* loss_dev, float* dq_dev, void* stream);

/* Backward of the last smg_forward: accumulates (+=) d(sum of losses)/d(param) into
 * net->grads for the trunk and head that forward used.  Replaces loss.backward() at
 * code/trainer.py:350-351. */
int smg_backward(smg_engine* e, const smg_net* net, const float* dq_dev, void* stream);

/* The same backward in two halves, for a data-parallel caller that hides its gradient all-reduce under compute:
 * phase 0 runs the head and dense blocks 4, 3, 2; when it returns (in stream order) every gradient of the parameters from
 * smg_layout_trunk_split() to the end of the trunk range, and of the head range, is final - their all-reduce can start while
 * phase 1 (dense block 1, pool0, the stem: about a third of the backward) computes the rest, [trunk begin, split).
 * smg_backward == phase 0 followed by phase 1.
 * Phase 1 never reads or writes a gradient element of [split, trunk end) or of the head range (they may be in an in-place
 * all-reduce on another stream).  The order is enforced: phase 1 without phase 0 of the SAME forward, phase 0 twice, or
 * smg_backward between the two halves return -22 and launch nothing. */
int smg_backward_phase(smg_engine* e, const smg_net* net, const float* dq_dev, void* stream, int phase);
/* Element offset (params / grads) of the first parameter behind dense block 1 of trunk `trunk_id` (transition1.norm.weight). */
int smg_layout_trunk_split(int head_out, int trunk_id, int64_t* offset);

/* Precision mode of the engine (the reference runs apex O0 = fp32, code/trainer.py:101; modes 1 and 2 are BASELINE.json configs 3
 * and 5):
 *   0  (default) fp32 storage; fp32-class accuracy, what the parity suite gates: the dense layers' products as a scaled
 *      two-piece fp16 split per operand, three MFMA terms (scales per weight tensor, per BatchNorm operand and per gradient
 *      tensor and stream, maintained by the engine); the stem's, the transitions' and the head's gradient products as three
 *      bf16 pieces, six terms;
 *   1  bf16 STORAGE of activations and gradients (dense-block buffers, bottlenecks, G', the backward ring), one bf16 MFMA term
 *      per product;
 *   2  fp16 storage of activations with fp16 forward products; gradients stored and multiplied in bf16 (fp32 exponent range:
 *      no loss scaling).
 * In every mode parameters, their gradients, Adam, BN statistics, every accumulation, the input image, the stem plane and the
 * head's feature buffers stay fp32.  Takes effect with the next smg_forward (a saved forward of another mode is dropped). */
int smg_engine_set_precision(smg_engine* e, int precision);

/* Engine switches by name.  "deterministic" (0 / 1): the 1x1-convolution weight gradients (conv1 of every dense layer,
 * the largest gradient tensors) are reduced from partial tiles in a fixed order instead of fp32 atomics, so the trunk's
 * convolution weight gradients of two identical calls are bit-identical like the reference's (code/trainer.py:350-351 on one
 * device); BatchNorm affine gradients and the head's value convolution keep their fp32 atomics.  (Since round 3 the fixed-order
 * path is also the default for batches of more than four streams whenever the partial tiles fit the workspace - it is the
 * faster one there; the option guarantees it for every batch: a launch whose partial tiles do not fit the workspace fails
 * with -12 instead of falling back to atomics.)
 * "serialize" (0 / 1): every kernel on the caller's stream in issue order instead of two concurrent chains (profiling).
 * "debug_stop" (tests only; -1 = off): the next smg_backward returns behind the launches of dense layer (block, layer) =
 * (value / 100, value % 100), 0-based - or, with value % 100 == 50, in front of that block's first layer - with both streams
 * joined, so that smg_debug_read sees the ring slots, "dy2" and G' as that layer left them.  The forward's saved state is
 * consumed (run a new smg_forward before the next backward). */
int smg_engine_set_option(smg_engine* e, const char* name, int value);

/* Heightmap generation in front of the path (utils.get_heightmap, code/utils.py:38-68): the robot-frame height of every
 * camera pixel (get_pointcloud + cam_pose, :12-47) warped onto the table plane (cv2.warpPerspective, INTER_LINEAR,
 * constant 0 border, :62-66).  depth_img_dev float64 [h][w]; intrinsics row-major 3x3, cam_pose row-major 4x4 and the
 * INVERSE of cv2.getPerspectiveTransform(src, dst) row-major 3x3 in host memory; out_dev float64 [out_h][out_w]. */
int smg_heightmap(const double* depth_img_dev, int h, int w, const double* intrinsics3x3, const double* cam_pose4x4,
                  const double* inv_homography3x3, int out_w, int out_h, double* out_dev, void* stream);

/* Index and value of the largest of n float32 values (lowest index on ties, like np.argmax at
 * code/main.py:172-173,195), on the device: idx_out_dev int32[1], val_out_dev float32[1]. */
int smg_argmax(const float* values_dev, int n, int* idx_out_dev, float* val_out_dev, void* stream);

/* Adam over [offset, offset+count) of params/grads with moments m, v (same layout),
 * replacing torch.optim.Adam.step (code/trainer.py:99,383): lr 1e-4, betas
 * (0.9,0.999), eps 1e-8, no weight decay; `step` is the 1-based step count of this
 * segment. */
int smg_adam_step(float* params, const float* grads, float* m, float* v, int64_t offset, int64_t count,
                  int step, float lr, float beta1, float beta2, float eps, void* stream);

/* One training step - zero the (trunk, head) gradient ranges, smg_forward, smg_loss, smg_backward, Adam on both ranges - as ONE
 * replayable hipGraph: the reference's real call pattern is one (mask, rotation) sample per Trainer.backprop (code/main.py:338,
 * code/trainer.py:334-384), ~560 launches of 2-20 us, which an eager host enqueues no faster than the GPU runs them.  The first
 * call with a given set of pointers / shapes runs eagerly, the second is captured (stream capture across the engine's two
 * streams), later ones replay; what changes between steps without re-capture: the CONTENTS of the input images, masks and labels,
 * the batch description arrays of `batch` (rotations, pairings: same counts), and the Adam step counts.  Anything else
 * (pointers, counts, precision mode, options) re-captures.  Results are bit-identical to the four separate calls with
 * smg_adam_step(step) on the two ranges.  Asynchronous on `stream` like the calls it replaces. */
typedef struct {
    float* m; float* v;               /* Adam moments, flat, laid out like params */
    float lr, beta1, beta2, eps;
    int step_trunk, step_head;        /* 1-based step count of the trunk range / the head range for THIS step */
} smg_adam;
int smg_train_step_graph(smg_engine* e, const smg_net* net, int trunk_id, int head_id, const smg_batch* batch, int loss_mode,
                         const float* labels_dev, float* q_out_dev, float* loss_out_dev, float* dq_dev, const smg_adam* adam, void* stream);

/* Element range of params/grads used by (trunk_id) features or (head_id) head. */
int smg_layout_trunk_range(int head_out, int trunk_id, int64_t* offset, int64_t* count);
int smg_layout_head_range(int head_out, int head_id, int64_t* offset, int64_t* count);

/* ---- debug / test access (used by tests/ only) ------------------------------------ */
/* Copies an internal buffer to host as float32 (16-bit storage is widened).  name: "img", "stem", "x1".."x4",
 * "feat", "g1".."g4", "bt<block>_<layer>", "fs_bt<block>_<layer>", "asc" ...; precision mode 0 only: "dy2" (raw 3x3 data
 * gradient of the dense layer the backward processed last), "gs_<block>_<layer>" / "d2_<block>_<layer>" (that layer's ring
 * slot: its finished output-slice gradient [streams][HWp][32] / bottleneck gradient [streams][HWp][128], the latter rebuilt from
 * its fp16 units and block scales - exactly the operand the consumers' MFMAs see).  Block / layer are 1-based.
 * Returns the element count or a negative error. */
int64_t smg_debug_read(smg_engine* e, const char* name, float* host_out, int64_t cap, void* stream);
/* Geometry of the engine: fills H (per block spatial size), HWp (padded rows). */
int smg_engine_geometry(const smg_engine* e, int H[6], int HWp[6]);
/* Per-kernel-class timing, measured with hipEvents recorded on the launch stream
 * around every launch while profiling is enabled (bench.py's roofline leg).
 * kind in [0, smg_profile_kinds()): the MFMA convolution classes, last = everything
 * else.  smg_profile_read drains pending events (synchronises) and returns the
 * accumulated milliseconds, launch count and executed FLOPs (2*M*N*K over valid
 * pixels) of one class since smg_profile_enable.  kind + smg_profile_kinds()*(1+b),
 * b in 0..3, reads the share of that class issued inside dense block b's layer loops. */
int smg_profile_enable(smg_engine* e, int on);
int smg_profile_kinds(void);
const char* smg_profile_kind_name(int kind);
int smg_profile_read(smg_engine* e, int kind, double* ms, int64_t* launches, double* flops);
/* Algorithmic HBM bytes of the same class (every operand read once, every result written once, fp32) - the
 * numerator of bench.py's HBM roofline.  Valid after smg_profile_read of that class. */
int smg_profile_read_bytes(smg_engine* e, int kind, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* SMG_HIP_H */
