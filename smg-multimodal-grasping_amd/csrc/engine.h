// engine.h - shared by the translation units of libsmg_hip.so (engine.hip: workspace + C ABI, forward.hip, backward.hip):
// the engine's state, tile configurations and launch helpers.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/smg_hip.h"
#include "elem.cuh"
#include "gemm.cuh"
#include "halo.cuh"
#include "ws.cuh"
#include "wsw.cuh"
#include "plan.h"

using namespace smg;

// ------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------
int smg_fail(int code, const std::string& msg);      // engine.hip: sets the thread's smg_last_error() text, returns code
static inline int fail(int code, const std::string& msg) { return smg_fail(code, msg); }
#define HIP_OK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return fail(-5, std::string(#expr) + ": " + hipGetErrorString(_e));                   \
    } while (0)

static const Layout& layout_for(int head_out) {
    static Layout L1 = build_layout(1);
    static Layout L3 = build_layout(3);
    return head_out == 3 ? L3 : L1;
}

// ------------------------------------------------------------------------------------
// GEMM tile configurations (BM, BN, BK, waves M x N, A pixel-major?)
// ------------------------------------------------------------------------------------
// big stages (pixel planes that are multiples of 128 rows)
using CfgP128x128 = GemmCfg<128, 128, 16, 2, 2, 1, true>;    // 1x1 fwd, transitions, 3x3 dgrad
using CfgP128x32 = GemmCfg<128, 32, 32, 4, 1, 1, true>;      // 3x3 fwd (N = growth 32)
using CfgP128x64 = GemmCfg<128, 64, 16, 2, 2, 1, true>;      // stem, head conv0, 1x1 dgrad
// small stages (late blocks: few pixels per stream -> 64-row tiles, 4x the workgroups)
using CfgP64x64 = GemmCfg<64, 64, 32, 2, 2, 1, true>;        // 1x1 fwd / dgrad, transitions
using CfgP32x64 = GemmCfg<32, 64, 32, 1, 2, 2, true>;        // 1x1 fwd of a launch too small to fill the chip: 2x the workgroups, half the K chain per wave
using CfgP64x128 = GemmCfg<64, 128, 16, 2, 2, 1, true>;      // 3x3 dgrad, late 1x1 fwd / transitions (A operand read once)
using CfgP64x32 = GemmCfg<64, 32, 64, 2, 1, 2, true>;        // 3x3 fwd, k-tile split over 2 waves
// weight gradients (reduction over pixels)
using CfgW32x128 = GemmCfg<32, 128, 32, 1, 4, 1, false>;     // 3x3 wgrad (32 x 128 per tap)
using CfgW128x64 = GemmCfg<128, 64, 16, 2, 2, 1, false>;     // 1x1 wgrad (128 x cin)
using CfgW128x128 = GemmCfg<128, 128, 16, 2, 2, 1, false>;   // transition wgrad
using CfgW64x64 = GemmCfg<64, 64, 16, 2, 2, 1, false>;       // head conv0 wgrad
using CfgW64x256 = GemmCfg<64, 256, 16, 2, 2, 1, false>;     // stem wgrad: all 196 (tap, channel) columns in one tile

// 16-bit storage modes move half the bytes per k-tile with the same latency per k-tile (global load -> LDS -> barrier): their
// k-tiles are MUL times deeper (the single-piece LDS images are a third of the fp32-class ones, so the tiles still fit).
template <class C, int PREC, int MUL = 2>
using MC = typename std::conditional<PREC == 0, C, GemmCfg<C::BM, C::BN, MUL * C::BK, C::WM, C::WN, C::WK, C::AT>>::type;
enum Kind {
    K_STEM = 0, K_C1, K_C3, K_TRANS, K_HEAD0, K_D3, K_W3, K_D1, K_W1, K_TW, K_TD, K_SW, K_HW0, K_HD0, K_OTHER, K_COUNT
};
static const char* kKindNames[K_COUNT] = {"stem7x7_fwd", "conv1x1_fwd", "conv3x3_fwd", "transition_fwd", "head_conv0_fwd",
                                          "conv3x3_dgrad", "conv3x3_wgrad", "conv1x1_dgrad", "conv1x1_wgrad",
                                          "transition_wgrad", "transition_dgrad", "stem_wgrad", "head_conv0_wgrad",
                                          "head_conv0_dgrad", "elementwise"};

// Backward ring: the finished bottleneck gradients (D2) of one layer group stay alive until the group's joint
// 1x1 data-gradient kernel has read them, while the side stream may still be two layers behind.
constexpr int kGroup = GROUP_MAX;          // dense layers per 1x1-dgrad group
constexpr int kRing = kGroup + 2;

constexpr int kDbRep = 8;
struct DbSeg { int64_t grad_off; int scr_off; int n; };      // n floats at scr_off (per replica) -> grads[grad_off ..]

struct ProfRec { hipEvent_t a, b; int kind; double flops; int stage; double bytes; };

struct StatArr { int64_t off; int stride; };   // into a double arena: sum at off, sumsq at off + span

// ------------------------------------------------------------------------------------
// engine
// ------------------------------------------------------------------------------------
struct smg_engine {
    int device = 0, S = 0, max_streams = 0, max_pairs = 0, head_out = 1;
    Plane p_img, p_stem, p_blk[4];
    int OH = 1, OW = 1;
    const Layout* L = nullptr;

    // activations
    // (X, Bt, G, GS, D2 hold fp32 elements in mode 0 and 16-bit elements in modes 1 / 2: they are allocated for fp32 and
    // addressed through el(); everything else is fp32 in every mode)
    float* img4 = nullptr; float* stem = nullptr; float* X[4] = {}; float* Bt = nullptr;
    std::vector<int64_t> bt_off[4];          // ELEMENT offset of each layer's bottleneck buffer
    unsigned char* argmax = nullptr;
    float* F = nullptr; float* H1 = nullptr;
    // gradients
    float* G[4] = {}; float* GS[kRing] = {}; float* D2[kRing] = {}; float* part = nullptr; int64_t part_floats = 0;
    // precision mode 0: the ring's D2 slots hold the bottleneck gradient in UNIT form (gemm.cuh, kD2K8: two fp16 pieces per element,
    // the same bytes) with one inverse scale per 64-pixel block in D2S; the raw 3x3 data gradient of the layer in flight goes to DY2
    float* DY2 = nullptr; float* D2S[kRing] = {};
    // second stream for the weight-gradient kernels (independent of the data-gradient chain)
    hipStream_t side = nullptr; hipEvent_t ev_gs[kRing] = {}, ev_d2[kRing] = {}, ev_side[kRing] = {}, ev_misc = nullptr, ev_end = nullptr; float* DY0 = nullptr; float* DH1 = nullptr; float* DF = nullptr;
    // statistics arenas (doubles). fwd: [sum | sumsq] halves; bwd: [s1 | s2] halves
    double* fstat = nullptr; int64_t fstat_span = 0;
    double* bstat = nullptr; int64_t bstat_span = 0;
    // dbeta / dgamma of the trunk's norm1 / transition norms: the data-gradient GEMMs add one value per (workgroup, column) to ONE
    // address per channel - thousands of same-address atomics per launch, which the L2 serialises (about 50-90 ns each: the
    // kernels' tails).  They go to kDbRep replicas of a compact scratch instead (replica = workgroup index mod kDbRep);
    // db_flush_kernel sums the replicas into the gradient array at the end of the backward (half) and re-zeroes them.
    float* dbscr = nullptr; int db_total = 0; int db_off[4][24] = {}; int db_toff[3] = {}; struct DbSeg* d_dbseg = nullptr; int n_dbseg = 0;
    StatArr st_stem, st_X[4], st_F, st_H1; std::vector<StatArr> st_Bt[4];
    StatArr bs_stem, bs_X[4], bs_F, bs_H1; std::vector<StatArr> bs_Bt[4];
    // packed weights: split bf16 units for the MFMA GEMMs (packed_u) and fp32 K-major layouts for the halo 3x3
    // kernels / the value convolution (packed_f)
    u32x4* packed_u = nullptr; int64_t packed_units = 0;
    float* packed_f = nullptr; int64_t packed_floats = 0;
    PackDesc* d_pack = nullptr; std::vector<PackDesc> h_pack[3]; std::vector<PackDesc> h_pack_head[3];
    int pack_stride = 0, bnupd_stride = 0, n_bnupd = 0;   // d_pack / d_bnupd hold one table per (trunk, head)
    int64_t pk_conv0 = 0, pk_conv0_1 = 0, pk_head0 = 0, pk_hd0 = 0, pk_head1 = 0;
    std::vector<int64_t> pk_c1[4], pk_d1[4], pk_g3f[4], pk_g3d[4], pk_hf[4], pk_hd[4]; int64_t pk_t[3] = {}, pk_td[3] = {};
    int max_pack = 0;
    // operand kind 3 (gemm.cuh): activation scales {s, 1 / s} per BatchNorm + ReLU operand (norm1 / norm2 of every dense layer, the
    // transition norms, the head's norm0), written by scale_kernel at the start of every forward from the descriptor table of
    // the (trunk, head) in use; recorded gradient maxima per (dense layer, GS | D2, stream), zeroed at the start of a backward
    float* asc = nullptr; ActScaleDesc* d_asc = nullptr; int n_asc = 0;
    unsigned* gamax = nullptr; int64_t gamax_words = 0;
    // BN statistics as fp32 tables (mean | invstd, [rows][C] each): one per dense-block buffer, one per bottleneck, one
    // for the head's features; written by the first consumer of a channel (BnTab, gemm.cuh), kept until the backward
    float* stab = nullptr; int64_t stab_floats = 0;
    int64_t sx_tab[4] = {}, sb_tab[4][24] = {}, sf_tab = 0;
    // bn update descriptors
    BnUpdDesc* d_bnupd = nullptr;
    // last forward
    bool have_fwd = false; int f_trunk = 0, f_head = 0, f_streams = 0, f_pairs = 0;
    bool bw_phase0_done = false;   // smg_backward_phase(0) ran on the last forward and its second half is still due (reset by every forward / precision change)
    bool f_stem1 = false;      // the last forward ran the one-channel stem (heightmap input form): img4 holds [streams][HWp] single floats
    int* d_stream_image = nullptr; int* d_stream_rot = nullptr; int* d_pair_a = nullptr; int* d_pair_b = nullptr;
    int* d_seq_t = nullptr; int* d_seq_h = nullptr; int* d_user_ptr = nullptr; int* d_user_pair = nullptr; int* d_user_slot = nullptr;
    float* d_affine = nullptr;
    // batch description staging: one pinned ping-pong host block -> one device block per forward
    int* d_stage = nullptr; int* h_stage[2] = {}; hipEvent_t ev_stage[2] = {}; int stage_ints = 0, stage_turn = 0;
    int so_image = 0, so_rot = 0, so_pa = 0, so_pb = 0, so_seq_t = 0, so_seq_h = 0, so_uptr = 0, so_upair = 0, so_uslot = 0, so_aff = 0, so_ma = 0, so_mb = 0, so_adam = 0;
    // One training step as a replayable hipGraph (smg_train_step_graph, engine.hip): while `capturing`, do_forward takes the batch
    // description from h_stage_g - filled by the caller, together with the Adam scalars at so_adam - and issues no host
    // synchronisation; ev_graph orders the host's next refill of h_stage_g behind the replay that reads it.
    // The step runs on the engine's own stream gstream (the caller's may be the NULL stream, which cannot be captured), ordered
    // behind the caller's stream by ev_gin and in front of it by ev_graph.
    bool capturing = false; int* h_stage_g = nullptr; hipEvent_t ev_graph = nullptr, ev_gin = nullptr; hipStream_t gstream = nullptr; struct StepGraph* step_graph = nullptr;
    int64_t workspace_bytes = 0;
    int n_cu = 256;            // compute units of the device (persistent-launch sizing)
    int prec = 0;              // precision mode: 0 fp32 storage + fp32-class split products, 1 bf16 storage, 2 fp16 activations + bf16 gradients (smg_engine_set_precision)
    bool serialize = false;       // smg_engine_set_option("serialize"): every launch on the caller's stream in issue order (profiling: a trace's
                                  // per-kernel durations are not inflated by a kernel of the other chain sharing the chip)
    bool deterministic = false;   // smg_engine_set_option("deterministic"): 1x1 weight gradients as partial tiles + fixed-order reduce instead of fp32 atomics
    // SMG_CROSSCHECK (read at engine creation; tests/test_gpu_parity.py::test_alternative_kernel_paths_agree): independent implementations of
    // two kernel families for cross-checks - bit 0: dense-layer 3x3 convolutions through the generic implicit GEMM instead of the LDS-halo
    // kernels; bit 1: the 1x1 forward of the small planes through the generic kernel instead of the wave-specialised one
    // bit 2: the 1x1 weight gradient through the generic kernel instead of the wave-specialised one (wsw.cuh)
    bool generic3x3 = false, generic_c1 = false, generic_w1 = false;
    int dbg_stop = -1;         // smg_engine_set_option("debug_stop", block * 100 + layer) (0-based): the backward returns behind that dense layer's
                               // launches (block * 100 + 50: in front of the block's first layer) - the GEMM-level tests read the ring
                               // slots, DY2 and G' at that point (smg_debug_read); -1 = off
    float* dbg_gsnap = nullptr; int64_t dbg_gsnap_floats = 0;      // ... and G' of the block as it was in front of that layer's 1x1 data gradients ("gsnap")
    // profiling
    bool prof = false; std::vector<ProfRec> recs; std::vector<hipEvent_t> ev_pool;
    // totals per kind in slot 0, and the share of dense block b (kernels issued inside its layer loops) in slot 1 + b
    double prof_ms[5][K_COUNT] = {}; int64_t prof_n[5][K_COUNT] = {}; double prof_flops[5][K_COUNT] = {}; double prof_bytes[5][K_COUNT] = {}; int prof_stage = -1;
    double next_bytes = 0;     // algorithmic HBM bytes of the next profiled launch (set with BY() right before it)
};

// Tile side of the LDS-halo 3x3 kernels for a plane: 16 where it tiles exactly, else 8 (ragged edges masked) - and 8
// as well when the launch would have fewer than 320 16x16 tiles (few streams per call; 80x80 planes of a 9-stream
// forward chain): four times the workgroups fill the chip (forward sweep 9.05 -> 8.77 ms, single-rotation forward
// 4.3 -> 3.6 ms).
static inline int halo_tile(const Plane& p, int n_streams = 1 << 20) {
    // (tiles that hang over the edge are masked: S = 1824's 456^2 / 228^2 / 114^2 planes take 16 x 16 tiles too - round 5)
    return (int64_t)((p.H + 15) / 16) * ((p.W + 15) / 16) * n_streams >= 200 ? 16 : 8;      // (round 5, 320 -> 200: a single-sample step's 160^2 planes and the 80^2 planes of
                                                                                            //  an 8-stream forward chain take 16 x 16 tiles - step 5.71 -> 5.63 ms, headline 16.0 both ways; 100: the 17-stream
                                                                                            //  40^2 planes would too, 16.05)
}

// 3x3 weight-gradient halo kernel: tiles per workgroup.  The launch runs in rounds of 512 resident workgroups (2 per
// CU), each lasting tiles_per_wg tile-times plus a fixed prologue + 9-tap flush (~0.6 of a 16x16 tile-time, measured);
// take the run length with the shortest total (e.g. 100 tiles x 17 streams -> 7, 25 tiles -> 4), then lengthen it
// until the partial tiles fit the workspace.
static int w3_tiles_per_wg(int n_tiles, int ts, int n_streams, int64_t part_floats, double fix_scale = 1.0) {
    const double fix = (ts == 16 ? 0.6 : 2.4) * fix_scale;
    double best = 1e30;
    int tpw_best = 1;
    for (int tpw = 1; tpw <= n_tiles; ++tpw) {
        const int g = (n_tiles + tpw - 1) / tpw;
        const int rounds = (g * (kBottleneck / 32) * n_streams + 511) / 512;
        const double cost = rounds * (tpw + fix);
        if (cost < best - 1e-9) { best = cost; tpw_best = tpw; }
    }
    // (taking the LONGEST run length within 4-40 % of the shortest total - fewer partial tiles for the reduce to read back - measured
    //  17.3-17.4 against 17.27 ms per step: the partial-tile traffic is not what the side stream waits for)
    while (tpw_best < n_tiles && (int64_t)((n_tiles + tpw_best - 1) / tpw_best) * n_streams * 9 * 32 * kBottleneck > part_floats) ++tpw_best;
    return tpw_best;
}

static Plane make_plane(int H, int W) {
    // rows per stream: a multiple of 64 (tiles and scale blocks never straddle two streams) - of 128 on the big planes, so that the
    // 128-row tile configurations serve them whatever the input size (S = 1824: 456^2 = 207 936 = 64 x 3249 pixels took 64-row tiles)
    Plane p; p.H = H; p.W = W; p.HW = H * W;
    const int g = p.HW >= 8192 ? 128 : 64;
    p.HWp = (p.HW + g - 1) / g * g;
    return p;
}

template <class T>
static int dev_alloc(smg_engine* e, T** out, int64_t count) {
    void* p = nullptr;
    hipError_t err = hipMalloc(&p, (size_t)count * sizeof(T));
    if (err != hipSuccess) return fail(-12, std::string("hipMalloc ") + std::to_string(count * sizeof(T)) + " B: " + hipGetErrorString(err));
    e->workspace_bytes += count * (int64_t)sizeof(T);
    *out = (T*)p;
    return 0;
}
#define ALLOC(ptr, count) do { int _r = dev_alloc(e, &(ptr), (count)); if (_r) return _r; } while (0)

// ------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------
static hipEvent_t prof_event(smg_engine* e) {
    if (!e->ev_pool.empty()) { hipEvent_t ev = e->ev_pool.back(); e->ev_pool.pop_back(); return ev; }
    hipEvent_t ev; (void)hipEventCreate(&ev); return ev;
}
// Algorithmic HBM bytes of the launch that follows: what the kernel must move once (inputs read once, outputs written
// once, fp32), the yardstick of bench.py's HBM roofline.
#define BY(e, x) ((e)->next_bytes = (double)(x))
#define ESZ(e) ((e)->prec ? 2.0 : 4.0)      // bytes per element of the mode-typed buffers (X, Bt, G', GS, D2)
struct ProfScope {
    smg_engine* e; hipStream_t st; int kind; double flops, bytes; hipEvent_t a{}, b{};
    ProfScope(smg_engine* e_, hipStream_t s, int k, double f) : e(e_), st(s), kind(k), flops(f), bytes(e_->next_bytes) {
        e->next_bytes = 0;
        if (e->prof) { a = prof_event(e); b = prof_event(e); (void)hipEventRecord(a, st); }
    }
    ~ProfScope() {
        if (e->prof) { (void)hipEventRecord(b, st); e->recs.push_back({a, b, kind, flops, e->prof_stage, bytes}); }
    }
};

// Runs CALL with a compile-time PREC equal to the engine's run-time precision mode (0 fp32-class, 1 bf16 storage, 2 fp16
// activations + bf16 gradients; gemm.cuh).  PTAG is PREC as a type, for generic lambdas that build a policy.
#define PTAG std::integral_constant<int, PREC>{}
#define PREC_DISPATCH(e, CALL)                                            \
    switch ((e)->prec) {                                                  \
        case 1: { constexpr int PREC = 1; CALL; } break;                  \
        case 2: { constexpr int PREC = 2; CALL; } break;                  \
        default: { constexpr int PREC = 0; CALL; } break;                 \
    }

// Dev instrumentation: per-workgroup phase stamps of ONE launch (SMG_TRACE_KIND = kernel class, SMG_TRACE_SKIP = how many
// launches of that class to skip).  The kernels store s_memtime at up to five points (slots 0..4) and the device-wide
// 100 MHz counter at start / end (slots 5, 6) through g_smg_trace; the scope prints the mean phase lengths.
namespace {   // per translation unit: the scope writes THIS unit's g_smg_trace (a shared inline copy would write another unit's)
struct TraceScope {
    hipStream_t st; int kind; dim3 grid; const char* what; unsigned long long* tbuf = nullptr; size_t n_wg = 0;
    TraceScope(hipStream_t s, int k, dim3 g, const char* w = "") : st(s), kind(k), grid(g), what(w) {
        static const int tr_kind = getenv("SMG_TRACE_KIND") ? atoi(getenv("SMG_TRACE_KIND")) : -1;
        static const int tr_skip = getenv("SMG_TRACE_SKIP") ? atoi(getenv("SMG_TRACE_SKIP")) : 0;
        static int tr_seen = 0;
        if (!(kind == tr_kind && tr_seen++ == tr_skip)) return;
        n_wg = (size_t)grid.x * grid.y * grid.z;
        (void)hipMalloc((void**)&tbuf, n_wg * 64);
        (void)hipMemsetAsync(tbuf, 0, n_wg * 64, st);
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_smg_trace), &tbuf, sizeof(tbuf), 0, hipMemcpyHostToDevice, st);
    }
    ~TraceScope() {
        if (!tbuf) return;
        unsigned long long* nul = nullptr;
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_smg_trace), &nul, sizeof(nul), 0, hipMemcpyHostToDevice, st);
        std::vector<unsigned long long> h(n_wg * 8);
        (void)hipMemcpyAsync(h.data(), tbuf, n_wg * 64, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        (void)hipFree(tbuf);
        // phases in shader cycles (s_memtime, per-XCD base); span / residency from the device-wide 100 MHz counter
        double sum[4] = {0, 0, 0, 0}, life = 0, extra = 0; size_t live = 0; unsigned long long t_min = ~0ull, t_max = 0;
        for (size_t w = 0; w < n_wg; ++w) {
            const unsigned long long* r = &h[w * 8];
            if (!r[4]) continue;
            ++live;
            if (r[7]) extra += (double)(r[7] - r[1]);      // (kernels with a seventh stamp: slot 1 -> slot 7)
            for (int k = 0; k < 4; ++k) sum[k] += (double)(r[k + 1] - r[k]);
            t_min = std::min(t_min, r[5]); t_max = std::max(t_max, r[6]);
            life += (double)(r[6] - r[5]);
        }
        const double span = (double)(t_max - t_min) * 0.01, resid = life / (double)(t_max - t_min) / 256.0;
        {   // distribution: when workgroups start (relative to the first) and how long they live
            std::vector<double> st0, lf;
            for (size_t w = 0; w < n_wg; ++w) { const unsigned long long* r = &h[w * 8]; if (!r[4]) continue; st0.push_back((double)(r[5] - t_min) * 0.01); lf.push_back((double)(r[6] - r[5]) * 0.01); }
            std::sort(st0.begin(), st0.end()); std::sort(lf.begin(), lf.end());
            auto q = [](const std::vector<double>& v, double f) { return v.empty() ? 0.0 : v[std::min(v.size() - 1, (size_t)(f * v.size()))]; };
            fprintf(stderr, "[smg trace]   start us p50 %.1f p90 %.1f max %.1f | life us p10 %.1f p50 %.1f p90 %.1f max %.1f\n",
                    q(st0, 0.5), q(st0, 0.9), q(st0, 1.0), q(lf, 0.1), q(lf, 0.5), q(lf, 0.9), q(lf, 1.0));
        }
        if (extra > 0) fprintf(stderr, "[smg trace]   slot 1 -> slot 7: %.0f cycles/WG (mean)\n", extra / live);
        fprintf(stderr, "[smg trace] kind %d grid %ux%ux%u live %zu: init %.0f | first tile %.0f | k-loop %.0f | epilogue %.0f cycles/WG (mean); span %.1f us, %.2f workgroups resident per CU, mean life %.1f us  %.160s\n",
                kind, grid.x, grid.y, grid.z, live, sum[0] / live, sum[1] / live, sum[2] / live, sum[3] / live, span, resid, life / live * 0.01, strstr(what, "[P = ") ? strstr(what, "[P = ") : what);
    }
};
}  // namespace

template <class P>
static void launch_gemm(smg_engine* e, hipStream_t st, P p, dim3 grid, int kind, double flops) {
    const size_t smem = (size_t)(GeoOf<P>::TILE_FLOATS + p.param_floats()) * sizeof(float);
    if (smem > 64 * 1024) {      // more dynamic LDS than the default limit: raise it once per (instantiation, device)
        static bool raised[64] = {};
        if (!raised[e->device & 63]) {
            (void)hipFuncSetAttribute((const void*)gemm_kernel<P>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            raised[e->device & 63] = true;
        }
    }
    p.tm = TileMap{0, 0, 0};
    if constexpr (P::kSwizzle == 1) {            // x = M tiles, y = N tiles sharing one A operand
        if (grid.y > 1 && grid.z == 1) {
            p.tm = TileMap{(int)grid.x, (int)grid.y, 0};
            grid = dim3(tile_grid(p.tm), 1, 1);
        }
    } else {                                     // weight gradient: z = pixel chunks, x*y = tiles sharing them
        p.gx = grid.x; p.gy = grid.y;
        const int tiles = grid.x * grid.y;
        if (tiles > 1) {
            p.tm = TileMap{(int)grid.z, tiles, (int)grid.x};
            grid = dim3(tile_grid(p.tm), 1, 1);
        }
    }
    TraceScope ts(st, kind, grid, __PRETTY_FUNCTION__);
    {
        ProfScope ps(e, st, kind, flops);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_kernel<P>), dim3((unsigned)(grid.x * grid.y * grid.z)), dim3(256), smem, st, p, (int)grid.x, (int)grid.y);
    }
}

// Weight-gradient launch: partial tiles to the workspace + one reduce kernel (falls back to
// atomics if the workspace is too small for this launch).
// part_off: first float of the partial-tile workspace this launch may use (a dense layer's two weight gradients keep their partial
// tiles side by side and share ONE reduce launch: `defer` receives this launch's reduction instead of it being launched here).
template <class P>
static int launch_wgrad(smg_engine* e, hipStream_t st, P& p, dim3 grid, int kind, double flops, int taps, int cmap, bool use_part = true,
                        int64_t part_off = 0, ReduceArgs* defer = nullptr) {
    using C = typename P::Cfg;
    const int64_t ldp = (int64_t)grid.y * C::BN, rowsp = (int64_t)grid.x * C::BM;
    const int64_t need = (int64_t)grid.z * rowsp * ldp;
    p.part = (use_part && part_off + need <= e->part_floats) ? e->part + part_off : nullptr;
    if (use_part && !p.part && e->deterministic)      // never a silent loss of the bit-reproducibility the option promises
        return fail(-12, "deterministic: a weight-gradient launch needs " + std::to_string(need) + " partial-tile floats, the workspace holds " +
                             std::to_string(e->part_floats - part_off) + " (smaller batch per call, or a larger engine)");
    launch_gemm(e, st, p, grid, kind, flops);
    if (defer) defer->Z = 0;
    if (p.part) {
        ReduceArgs r;
        r.part = p.part; r.Z = p.n_chunks; r.taps = taps; r.rows = p.MA; r.cols = p.NB; r.ldp = (int)ldp;
        r.z_stride = rowsp * ldp; r.tap_stride = (int64_t)p.n_chunks * rowsp * ldp;
        r.dw = p.dw; r.ldw_out = p.ldw_out; r.cmap = cmap;
        if (defer) { *defer = r; return 0; }
        const int total = taps * p.MA * p.NB;
        ProfScope ps(e, st, K_OTHER, 0);       // (the reduce is profiled with the element-wise kernels: a class's launches are its GEMM kernels only)
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((total + 63) / 64), dim3(256), 0, st, r, ReduceArgs{}, (total + 63) / 64);
    }
    return 0;
}
// ONE launch for two pending reductions (either may be empty: Z == 0)
static void launch_reduce2(smg_engine* e, hipStream_t st, int kind, const ReduceArgs& ra, const ReduceArgs& rb) {
    const int ba = ra.Z ? (ra.taps * ra.rows * ra.cols + 63) / 64 : 0, bb = rb.Z ? (rb.taps * rb.rows * rb.cols + 63) / 64 : 0;
    if (ba + bb == 0) return;
    (void)kind;
    ProfScope ps(e, st, K_OTHER, 0);           // (one launch serves the layer's 3x3 AND 1x1 partial tiles: profiled with the element-wise kernels)
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(ba + bb), dim3(256), 0, st, ra, rb, ba);
}

// element `elems` of a mode-typed buffer (X, Bt, G, GS, D2): 4-byte elements in mode 0, 2-byte elements in modes 1 / 2
static inline float* el(const smg_engine* e, float* base, int64_t elems) {
    return reinterpret_cast<float*>(reinterpret_cast<char*>(base) + (e->prec ? 2 : 4) * elems);
}
static inline double* fsum(smg_engine* e, const StatArr& s) { return e->fstat + s.off; }
static inline double* fsq(smg_engine* e, const StatArr& s) { return e->fstat + e->fstat_span + s.off; }
static inline double* b1(smg_engine* e, const StatArr& s) { return e->bstat + s.off; }
static inline double* b2(smg_engine* e, const StatArr& s) { return e->bstat + e->bstat_span + s.off; }

static const float kEps = 1e-5f;

// position of dense layer (block b, layer i; 0-based) in the backward's ring sequence of GS / D2 / D2S slots (slot = position % kRing)
static inline int ring_pos(int b, int i) {
    int r = kBlockLayers[b] - 1 - i;
    for (int bb = 3; bb > b; --bb) r += kBlockLayers[bb];
    return r;
}

// indices into the activation-scale table / the gradient-maximum buffer
static inline int layer_seq(int b, int i) { static const int first[4] = {0, 6, 18, 42}; return first[b] + i; }
constexpr int kDenseLayers = 58;
static inline const float* asc_n1(const smg_engine* e, int b, int i) { return e->asc + 2 * (2 * layer_seq(b, i)); }
static inline const float* asc_n2(const smg_engine* e, int b, int i) { return e->asc + 2 * (2 * layer_seq(b, i) + 1); }
static inline const float* asc_trans(const smg_engine* e, int b) { return e->asc + 2 * (2 * kDenseLayers + b); }
static inline const float* asc_head0(const smg_engine* e) { return e->asc + 2 * (2 * kDenseLayers + 3); }
// kind 0: GS (finished gradient of the layer's 32 output channels), 1: D2 (finished bottleneck gradient)
static inline unsigned* gamax_of(const smg_engine* e, int b, int i, int kind) {
    return e->gamax + ((int64_t)(2 * layer_seq(b, i) + kind) * e->max_streams) * kAmaxRep;
}

// BN statistics table at float offset `at` of the table arena ([rows_max][C] mean, then invstd), from row r0 on, with the
// affine parameters of the consuming BatchNorm
static StatTab stat_table(smg_engine* e, int64_t at, int rows_max, int C) {      // the same table, read-only view for the backward
    StatTab t; t.mean = e->stab + at; t.invstd = t.mean + (int64_t)rows_max * C; t.ld = C; return t;
}
static BnTab bn_table(smg_engine* e, int64_t at, int rows_max, int r0, int C, const float* gamma, const float* beta) {
    BnTab t;
    t.mean = e->stab + at + (int64_t)r0 * C; t.invstd = t.mean + (int64_t)rows_max * C; t.ld = C; t.gamma = gamma; t.beta = beta;
    return t;
}


int validate_batch(const smg_engine* e, const smg_batch* B);
void fill_stage(const smg_engine* e, const smg_batch* B, int* h);
int do_forward(smg_engine* e, const smg_net* net, int trunk_id, int head_id, const smg_batch* B, float* q_out, hipStream_t st);
int do_backward(smg_engine* e, const smg_net* net, const float* dq, hipStream_t st, int phases = 3);
