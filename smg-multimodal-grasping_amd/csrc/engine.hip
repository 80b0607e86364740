// engine.hip - host side of the MI355X affordance engine: workspace ownership, the
// walk over DenseNet-121 (forward and backward) and the C ABI of include/smg_hip.h.
//
// Replaces, for one (trunk, head) of the reference model:
//   reinforcement_net.forward / reactive_net.forward   code/models.py:361-586, :72-296
//   loss.backward() of Trainer.backprop                code/trainer.py:350-351
//   torch.optim.Adam.step                              code/trainer.py:383
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
#include "plan.h"

using namespace smg;

// ------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
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
using CfgP64x64k16 = GemmCfg<64, 64, 16, 2, 2, 1, true>;     // 1x1 fwd with many input channels: 29 KB LDS incl. BN parameters
using CfgP32x64 = GemmCfg<32, 64, 32, 1, 2, 2, true>;        // 1x1 fwd of a launch too small to fill the chip: 2x the workgroups, half the K chain per wave
using CfgP128x128d = GemmCfg<128, 128, 32, 2, 2, 1, true>;   // 1x1 fwd of launches with fewer 128-row tiles than CUs: one workgroup per CU, deep register prefetch
using CfgP64x128d = GemmCfg<64, 128, 32, 2, 2, 1, true>;     // the same for planes that tile by 64 rows only (40^2)
using CfgP64x64w = GemmCfg<64, 64, 32, 1, 2, 2, true>;       // 64x64 with 64x32 wave tiles over half the k-steps each: 25% fewer fragment reads per MFMA
using CfgP64x128 = GemmCfg<64, 128, 16, 2, 2, 1, true>;      // 3x3 dgrad, late 1x1 fwd / transitions (A operand read once)
using CfgP64x32 = GemmCfg<64, 32, 64, 2, 1, 2, true>;        // 3x3 fwd, k-tile split over 2 waves
// weight gradients (reduction over pixels)
using CfgW32x128 = GemmCfg<32, 128, 32, 1, 4, 1, false>;     // 3x3 wgrad (32 x 128 per tap)
using CfgW128x64 = GemmCfg<128, 64, 16, 2, 2, 1, false>;     // 1x1 wgrad (128 x cin)
using CfgW128x128 = GemmCfg<128, 128, 16, 2, 2, 1, false>;   // transition wgrad
using CfgW64x64 = GemmCfg<64, 64, 16, 2, 2, 1, false>;       // head conv0 wgrad
using CfgW64x256 = GemmCfg<64, 256, 16, 2, 2, 1, false>;     // stem wgrad: all 196 (tap, channel) columns in one tile

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
    float* img4 = nullptr; float* stem = nullptr; float* X[4] = {}; float* Bt = nullptr;
    std::vector<int64_t> bt_off[4];          // float offset of each layer's bottleneck buffer
    unsigned char* argmax = nullptr;
    float* F = nullptr; float* H1 = nullptr;
    // gradients
    float* G[4] = {}; float* GS[kRing] = {}; float* D2[kRing] = {}; float* part = nullptr; int64_t part_floats = 0;
    // second stream for the weight-gradient kernels (independent of the data-gradient chain)
    hipStream_t side = nullptr; hipEvent_t ev_gs[kRing] = {}, ev_d2[kRing] = {}, ev_side[kRing] = {}, ev_misc = nullptr, ev_end = nullptr; float* DY0 = nullptr; float* DH1 = nullptr; float* DF = nullptr;
    // statistics arenas (doubles). fwd: [sum | sumsq] halves; bwd: [s1 | s2] halves
    double* fstat = nullptr; int64_t fstat_span = 0;
    double* bstat = nullptr; int64_t bstat_span = 0;
    StatArr st_stem, st_X[4], st_F, st_H1; std::vector<StatArr> st_Bt[4];
    StatArr bs_stem, bs_X[4], bs_F, bs_H1; std::vector<StatArr> bs_Bt[4];
    // packed weights: split bf16 units for the MFMA GEMMs (packed_u) and fp32 K-major layouts for the halo 3x3
    // kernels / the value convolution (packed_f)
    u32x4* packed_u = nullptr; int64_t packed_units = 0;
    float* packed_f = nullptr; int64_t packed_floats = 0;
    PackDesc* d_pack = nullptr; std::vector<PackDesc> h_pack[3]; std::vector<PackDesc> h_pack_head[3];
    int pack_stride = 0, bnupd_stride = 0, n_bnupd = 0;   // d_pack / d_bnupd hold one table per (trunk, head)
    int64_t pk_conv0 = 0, pk_head0 = 0, pk_hd0 = 0, pk_head1 = 0;
    std::vector<int64_t> pk_c1[4], pk_d1[4], pk_g3f[4], pk_g3d[4], pk_hf[4], pk_hd[4]; int64_t pk_t[3] = {}, pk_td[3] = {};
    int max_pack = 0;
    // BN statistics as fp32 tables (mean | invstd, [rows][C] each): one per dense-block buffer, one per bottleneck, one
    // for the head's features; written by the first consumer of a channel (BnTab, gemm.cuh), kept until the backward
    float* stab = nullptr; int64_t stab_floats = 0;
    int64_t sx_tab[4] = {}, sb_tab[4][24] = {}, sf_tab = 0;
    // bn update descriptors
    BnUpdDesc* d_bnupd = nullptr;
    // last forward
    bool have_fwd = false; int f_trunk = 0, f_head = 0, f_streams = 0, f_pairs = 0;
    int* d_stream_image = nullptr; int* d_stream_rot = nullptr; int* d_pair_a = nullptr; int* d_pair_b = nullptr;
    int* d_seq_t = nullptr; int* d_seq_h = nullptr; int* d_user_ptr = nullptr; int* d_user_pair = nullptr; int* d_user_slot = nullptr;
    float* d_affine = nullptr;
    // batch description staging: one pinned ping-pong host block -> one device block per forward
    int* d_stage = nullptr; int* h_stage[2] = {}; hipEvent_t ev_stage[2] = {}; int stage_ints = 0, stage_turn = 0;
    int so_image = 0, so_rot = 0, so_pa = 0, so_pb = 0, so_seq_t = 0, so_seq_h = 0, so_uptr = 0, so_upair = 0, so_uslot = 0, so_aff = 0, so_ma = 0, so_mb = 0;
    int64_t workspace_bytes = 0;
    int n_cu = 256;            // compute units of the device (persistent-launch sizing)
    int prec = 0;              // operand precision of the matrix products: 0 fp32-class split, 1 bf16, 2 fp16 (smg_engine_set_precision)
    bool serialize = false;       // smg_engine_set_option("serialize"): every launch on the caller's stream in issue order (profiling: a trace's
                                  // per-kernel durations are not inflated by a kernel of the other chain sharing the chip)
    bool deterministic = false;   // smg_engine_set_option("deterministic"): 1x1 weight gradients as partial tiles + fixed-order reduce instead of fp32 atomics
    bool generic3x3 = false;   // SMG_GENERIC_3X3=1: dense-layer 3x3 convs through the generic implicit GEMM (A/B testing)
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
    if (p.H % 16 || p.W % 16) return 8;
    static const int min16 = getenv("SMG_HALO16_MIN") ? atoi(getenv("SMG_HALO16_MIN")) : 320;      // dev A/B
    return (int64_t)(p.H / 16) * (p.W / 16) * n_streams >= min16 ? 16 : 8;
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
    while (tpw_best < n_tiles && (int64_t)((n_tiles + tpw_best - 1) / tpw_best) * n_streams * 9 * 32 * kBottleneck > part_floats) ++tpw_best;
    return tpw_best;
}

static Plane make_plane(int H, int W) {
    Plane p; p.H = H; p.W = W; p.HW = H * W; p.HWp = (p.HW + 63) / 64 * 64; return p;
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

// Runs CALL with a compile-time PREC equal to the engine's run-time precision setting.
#define PREC_DISPATCH(e, CALL)                                            \
    switch ((e)->prec) {                                                  \
        case 1: { constexpr int PREC = 1; CALL; } break;                  \
        case 2: { constexpr int PREC = 2; CALL; } break;                  \
        default: { constexpr int PREC = 0; CALL; } break;                 \
    }

// Dev instrumentation: per-workgroup phase stamps of ONE launch (SMG_TRACE_KIND = kernel class, SMG_TRACE_SKIP = how many
// launches of that class to skip).  The kernels store s_memtime at up to five points (slots 0..4) and the device-wide
// 100 MHz counter at start / end (slots 5, 6) through g_smg_trace; the scope prints the mean phase lengths.
struct TraceScope {
    hipStream_t st; int kind; dim3 grid; unsigned long long* tbuf = nullptr; size_t n_wg = 0;
    TraceScope(hipStream_t s, int k, dim3 g) : st(s), kind(k), grid(g) {
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
        double sum[4] = {0, 0, 0, 0}, life = 0; size_t live = 0; unsigned long long t_min = ~0ull, t_max = 0;
        for (size_t w = 0; w < n_wg; ++w) {
            const unsigned long long* r = &h[w * 8];
            if (!r[4]) continue;
            ++live;
            for (int k = 0; k < 4; ++k) sum[k] += (double)(r[k + 1] - r[k]);
            t_min = std::min(t_min, r[5]); t_max = std::max(t_max, r[6]);
            life += (double)(r[6] - r[5]);
        }
        const double span = (double)(t_max - t_min) * 0.01, resid = life / (double)(t_max - t_min) / 256.0;
        fprintf(stderr, "[smg trace] kind %d grid %ux%ux%u live %zu: init %.0f | first tile %.0f | k-loop %.0f | epilogue %.0f cycles/WG (mean); span %.1f us, %.2f workgroups resident per CU, mean life %.1f us\n",
                kind, grid.x, grid.y, grid.z, live, sum[0] / live, sum[1] / live, sum[2] / live, sum[3] / live, span, resid, life / live * 0.01);
    }
};

template <class P>
static void launch_gemm(smg_engine* e, hipStream_t st, P p, dim3 grid, int kind, double flops) {
    const size_t smem = (size_t)(P::Cfg::TILE_FLOATS + p.param_floats()) * sizeof(float);
    if (smem > 64 * 1024) {      // more dynamic LDS than the default limit: raise it once per (instantiation, device)
        static bool raised[64][3] = {};
        if (!raised[e->device & 63][e->prec]) {
            PREC_DISPATCH(e, (void)hipFuncSetAttribute((const void*)gemm_kernel<P, PREC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            raised[e->device & 63][e->prec] = true;
        }
    }
    p.tm = TileMap{0, 0, 0};
    if constexpr (P::kSwizzle == 1) {            // x = M tiles, y = N tiles sharing one A operand
        if (grid.y > 1 && grid.z == 1) {
            p.tm = TileMap{(int)grid.x, (int)grid.y, 0};
            grid = dim3(8 * ((grid.x + 7) / 8) * grid.y, 1, 1);
        }
    } else {                                     // weight gradient: z = pixel chunks, x*y = tiles sharing them
        p.gx = grid.x; p.gy = grid.y;
        const int tiles = grid.x * grid.y;
        if (tiles > 1) {
            p.tm = TileMap{(int)grid.z, tiles, (int)grid.x};
            grid = dim3(8 * ((grid.z + 7) / 8) * tiles, 1, 1);
        }
    }
    TraceScope ts(st, kind, grid);
    {
        ProfScope ps(e, st, kind, flops);
        PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_kernel<P, PREC>), dim3((unsigned)(grid.x * grid.y * grid.z)), dim3(256), smem, st, p, (int)grid.x, (int)grid.y));
    }
}

// Weight-gradient launch: partial tiles to the workspace + one reduce kernel (falls back to
// atomics if the workspace is too small for this launch).
template <class P>
static void launch_wgrad(smg_engine* e, hipStream_t st, P& p, dim3 grid, int kind, double flops, int taps, int cmap, bool use_part = true) {
    using C = typename P::Cfg;
    const int64_t ldp = (int64_t)grid.y * C::BN, rowsp = (int64_t)grid.x * C::BM;
    const int64_t need = (int64_t)grid.z * rowsp * ldp;
    p.part = (use_part && need <= e->part_floats) ? e->part : nullptr;
    launch_gemm(e, st, p, grid, kind, flops);
    if (p.part) {
        ReduceArgs r;
        r.part = e->part; r.Z = p.n_chunks; r.taps = taps; r.rows = p.MA; r.cols = p.NB; r.ldp = (int)ldp;
        r.z_stride = rowsp * ldp; r.tap_stride = (int64_t)p.n_chunks * rowsp * ldp;
        r.dw = p.dw; r.ldw_out = p.ldw_out; r.cmap = cmap;
        const int total = taps * p.MA * p.NB;
        ProfScope ps(e, st, kind, 0);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((total + 255) / 256), dim3(256), 0, st, r);
    }
}

// ------------------------------------------------------------------------------------
// creation
// ------------------------------------------------------------------------------------
static int engine_build(smg_engine* e) {
    const Layout& L = *e->L;
    const int NS = e->max_streams, NP = e->max_pairs;
    const int S = e->S;
    const int H1 = (S - 1) / 2 + 1, H2 = (H1 - 1) / 2 + 1;
    e->p_img = make_plane(S, S);
    e->p_stem = make_plane(H1, H1);
    int h = H2;
    for (int b = 0; b < 4; ++b) { e->p_blk[b] = make_plane(h, h); h /= 2; }
    e->OH = e->OW = e->p_blk[3].H - kHeadKernel + 1;
    if (e->OH < 1) return fail(-22, "input_size too small for the 20x20 value head");

    // the 16x16 data-gradient halo kernel needs more than the default 64 KB of dynamic LDS
    HIP_OK(hipFuncSetAttribute((const void*)conv3x3_halo_dgrad_kernel<16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, HaloDgradSGeo<16>::smem_bytes(kBottleneck)));
    HIP_OK(hipFuncSetAttribute((const void*)conv3x3_halo_dgrad_kernel<16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, HaloDgradSGeo<16>::smem_bytes(kBottleneck)));
    HIP_OK(hipFuncSetAttribute((const void*)conv3x3_halo_dgrad_kernel<16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, HaloDgradSGeo<16>::smem_bytes(kBottleneck)));
    const char* g3 = getenv("SMG_GENERIC_3X3");
    e->generic3x3 = g3 && g3[0] == '1';

    ALLOC(e->img4, (int64_t)NS * e->p_img.HWp * 4);
    ALLOC(e->stem, (int64_t)NS * e->p_stem.HWp * 64);
    ALLOC(e->DY0, (int64_t)NS * e->p_stem.HWp * 64);
    ALLOC(e->argmax, (int64_t)NS * e->p_blk[0].HWp * 64);
    int64_t bt_total = 0;
    for (int b = 0; b < 4; ++b) {
        ALLOC(e->X[b], (int64_t)NS * e->p_blk[b].HWp * kBlockCtot[b]);
        ALLOC(e->G[b], (int64_t)NS * e->p_blk[b].HWp * kBlockCtot[b]);
        for (int i = 0; i < kBlockLayers[b]; ++i) {
            e->bt_off[b].push_back(bt_total);
            bt_total += (int64_t)NS * e->p_blk[b].HWp * kBottleneck;
        }
    }
    ALLOC(e->Bt, bt_total);
    for (int k = 0; k < kRing; ++k) {   // ring: see kRing
        ALLOC(e->D2[k], (int64_t)NS * e->p_blk[0].HWp * kBottleneck);
        ALLOC(e->GS[k], (int64_t)NS * e->p_blk[0].HWp * kGrowth);
        HIP_OK(hipEventCreateWithFlags(&e->ev_gs[k], hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&e->ev_d2[k], hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&e->ev_side[k], hipEventDisableTiming));
    }
    HIP_OK(hipEventCreateWithFlags(&e->ev_misc, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&e->ev_end, hipEventDisableTiming));
    HIP_OK(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
    e->part_floats = (int64_t)24 << 20;   // partial weight-gradient tiles: 96 MB, or what the 3x3 launches of a full batch want
    for (int b = 0; b < 4; ++b) {
        const Plane& pl = e->p_blk[b];
        for (int ts : {halo_tile(pl), 8}) {
            const int th = 8;                                    // weight-gradient tiles are ts x 8 pixels
            const int nt = ((pl.H + th - 1) / th) * ((pl.W + ts - 1) / ts);
            const int tpw = w3_tiles_per_wg(nt, ts, NS, INT64_MAX, (double)ts / th);
            e->part_floats = std::max<int64_t>(e->part_floats, (int64_t)((nt + tpw - 1) / tpw) * NS * 9 * 32 * kBottleneck);
        }
    }
    ALLOC(e->part, e->part_floats);
    ALLOC(e->F, (int64_t)NP * e->p_blk[3].HWp * 2 * kFeat);
    ALLOC(e->DF, (int64_t)NP * e->p_blk[3].HWp * 2 * kFeat);
    ALLOC(e->H1, (int64_t)NP * e->p_blk[3].HWp * kHeadMid);
    ALLOC(e->DH1, (int64_t)NP * e->p_blk[3].HWp * kHeadMid);

    // statistic arenas: identical carving for forward (sum, sumsq) and backward (s1, s2)
    int64_t off = 0;
    auto carve = [&](int n, int stride) { StatArr s{off, stride}; off += (int64_t)n * stride; return s; };
    e->st_stem = carve(NS, 64);
    for (int b = 0; b < 4; ++b) e->st_X[b] = carve(NS, kBlockCtot[b]);
    for (int b = 0; b < 4; ++b)
        for (int i = 0; i < kBlockLayers[b]; ++i) e->st_Bt[b].push_back(carve(NS, kBottleneck));
    e->st_F = carve(NP, 2 * kFeat);
    e->st_H1 = carve(NP, kHeadMid);
    e->fstat_span = e->bstat_span = off;
    e->bs_stem = e->st_stem; e->bs_F = e->st_F; e->bs_H1 = e->st_H1;
    for (int b = 0; b < 4; ++b) { e->bs_X[b] = e->st_X[b]; e->bs_Bt[b] = e->st_Bt[b]; }
    ALLOC(e->fstat, 2 * off);
    ALLOC(e->bstat, 2 * off);
    {   // BN statistic tables: mean | invstd, [rows][C] each
        int64_t po = 0;
        auto carve_t = [&](int rows, int C) { int64_t at = po; po += (int64_t)2 * rows * C; return at; };
        for (int b = 0; b < 4; ++b) {
            e->sx_tab[b] = carve_t(NS, kBlockCtot[b]);
            for (int i = 0; i < kBlockLayers[b]; ++i) e->sb_tab[b][i] = carve_t(NS, kBottleneck);
        }
        e->sf_tab = carve_t(NP, 2 * kFeat);
        e->stab_floats = po;
        ALLOC(e->stab, po);
    }

    // packed weights + descriptor tables (one table per trunk, one per head)
    int64_t pku = 0, pkf = 0;
    auto add_units = [&](std::vector<PackDesc>& v, int64_t src, int cout, int cin, int mode, int K, int N) {
        PackDesc d{}; d.src = src; d.dst = pku; d.cout = cout; d.cin = cin; d.mode = mode; d.K8tot = K / 8; d.N = N; d.count = 0;
        v.push_back(d); int64_t at = pku; pku += (int64_t)NPIECE * (K / 8) * N; return at;
    };
    auto add_f32 = [&](std::vector<PackDesc>& v, int64_t src, int cout, int cin, int mode, int64_t count) {
        PackDesc d{}; d.src = src; d.dst = pkf; d.cout = cout; d.cin = cin; d.mode = mode; d.count = (int)count;
        v.push_back(d); int64_t at = pkf; pkf += count; return at;
    };
    for (int t = 0; t < 3; ++t) {
        pku = pkf = 0;   // every trunk packs into the same region (only one trunk is active per forward)
        const TrunkRef& T = L.trunk[t];
        std::vector<PackDesc>& v = e->h_pack[t];
        e->pk_conv0 = add_units(v, T.conv0.w, 64, 3, PK_STEM, 224, 64);
        for (int b = 0; b < 4; ++b) {
            if (t == 0) { e->pk_c1[b].clear(); e->pk_d1[b].clear(); e->pk_g3f[b].clear(); e->pk_g3d[b].clear(); e->pk_hf[b].clear(); e->pk_hd[b].clear(); }
            for (size_t i = 0; i < T.layers[b].size(); ++i) {
                const DenseLayerRef& d = T.layers[b][i];
                int64_t a1 = add_units(v, d.c1.w, kBottleneck, d.cin, PK_T1, d.cin, kBottleneck);
                int64_t a1d = add_units(v, d.c1.w, kBottleneck, d.cin, PK_D1, kBottleneck, d.cin);
                int64_t g2 = 0, g3 = 0;
                const int64_t hf = add_units(v, d.c2.w, kGrowth, kBottleneck, PK_HF, 9 * kBottleneck, kGrowth);
                const int64_t hd = add_units(v, d.c2.w, kGrowth, kBottleneck, PK_HD, 9 * kGrowth, kBottleneck);
                if (e->generic3x3) {
                    g2 = add_units(v, d.c2.w, kGrowth, kBottleneck, PK_3F, 9 * kBottleneck, kGrowth);
                    g3 = add_units(v, d.c2.w, kGrowth, kBottleneck, PK_3D, 9 * kGrowth, kBottleneck);
                }
                if (t == 0) {
                    e->pk_c1[b].push_back(a1); e->pk_d1[b].push_back(a1d);
                    e->pk_g3f[b].push_back(g2); e->pk_g3d[b].push_back(g3); e->pk_hf[b].push_back(hf); e->pk_hd[b].push_back(hd);
                }
            }
            if (b < 3) {
                e->pk_t[b] = add_units(v, T.tconv[b].w, T.tconv[b].cout, T.tconv[b].cin, PK_T1, T.tconv[b].cin, T.tconv[b].cout);
                e->pk_td[b] = add_units(v, T.tconv[b].w, T.tconv[b].cout, T.tconv[b].cin, PK_D1, T.tconv[b].cout, T.tconv[b].cin);
            }
        }
    }
    const int64_t trunk_pku = pku, trunk_pkf = pkf;
    for (int hd = 0; hd < 3; ++hd) {
        pku = trunk_pku; pkf = trunk_pkf;
        const HeadRef& H = L.head[hd];
        std::vector<PackDesc>& v = e->h_pack_head[hd];
        e->pk_head0 = add_units(v, H.c0.w, kHeadMid, 2 * kFeat, PK_T1, 2 * kFeat, kHeadMid);
        e->pk_hd0 = add_units(v, H.c0.w, kHeadMid, 2 * kFeat, PK_D1, kHeadMid, 2 * kFeat);
        e->pk_head1 = add_f32(v, H.c1.w, e->head_out, kHeadMid, PK_HEAD, H.c1.count());
    }
    e->packed_units = pku + 4096;        // slack: tiles wider than N over-read whole units behind the array (never stored)
    e->packed_floats = pkf;
    ALLOC(e->packed_u, e->packed_units);
    HIP_OK(hipMemset(e->packed_u, 0, (size_t)e->packed_units * sizeof(u32x4)));
    ALLOC(e->packed_f, pkf + 4);
    e->max_pack = (int)(e->h_pack[0].size() + e->h_pack_head[0].size());
    // Descriptor tables are static per (trunk, head): upload all nine once, so a forward
    // never has to wait on a host->device copy of them.
    e->pack_stride = e->max_pack;
    ALLOC(e->d_pack, 9 * e->pack_stride);
    for (int t = 0; t < 3; ++t)
        for (int hd = 0; hd < 3; ++hd) {
            std::vector<PackDesc> v = e->h_pack[t];
            v.insert(v.end(), e->h_pack_head[hd].begin(), e->h_pack_head[hd].end());
            HIP_OK(hipMemcpy(e->d_pack + (t * 3 + hd) * e->pack_stride, v.data(), v.size() * sizeof(PackDesc), hipMemcpyHostToDevice));
        }
    e->bnupd_stride = 128;
    ALLOC(e->d_bnupd, 9 * e->bnupd_stride);
    for (int t = 0; t < 3; ++t)
        for (int hd = 0; hd < 3; ++hd) {
            const TrunkRef& T = L.trunk[t];
            const HeadRef& Hd = L.head[hd];
            std::vector<BnUpdDesc> v;   // in the reference's module order
            auto add = [&](const BnRef& r, const StatArr& s, int count, int head) {
                BnUpdDesc d; d.rm = r.rm; d.rv = r.rv; d.nbt = r.nbt; d.stat_off = s.off; d.stride = s.stride; d.coff = 0;
                d.C = r.C; d.count = count; d.head = head; v.push_back(d);
            };
            add(T.norm0, e->st_stem, e->p_stem.HW, 0);
            for (int b = 0; b < 4; ++b) {
                for (size_t i = 0; i < T.layers[b].size(); ++i) {
                    add(T.layers[b][i].n1, e->st_X[b], e->p_blk[b].HW, 0);
                    add(T.layers[b][i].n2, e->st_Bt[b][i], e->p_blk[b].HW, 0);
                }
                if (b < 3) add(T.tnorm[b], e->st_X[b], e->p_blk[b].HW, 0);
            }
            add(T.norm5, e->st_X[3], e->p_blk[3].HW, 0);
            add(Hd.n0, e->st_F, e->p_blk[3].HW, 1);
            add(Hd.n1, e->st_H1, e->p_blk[3].HW, 1);
            if ((int)v.size() > e->bnupd_stride) return fail(-22, "bn descriptor table overflow");
            e->n_bnupd = (int)v.size();
            HIP_OK(hipMemcpy(e->d_bnupd + (t * 3 + hd) * e->bnupd_stride, v.data(), v.size() * sizeof(BnUpdDesc), hipMemcpyHostToDevice));
        }

    // batch description: one device block, filled by one async copy from pinned memory
    const int R = NS > NP ? NS : NP;
    int so = 0;
    auto carve_i = [&](int n) { int at = so; so += (n + 3) / 4 * 4; return at; };
    e->so_image = carve_i(NS); e->so_rot = carve_i(NS); e->so_pa = carve_i(NP); e->so_pb = carve_i(NP);
    e->so_seq_t = carve_i(4 * R + 16); e->so_seq_h = carve_i(4 * R + 16);
    e->so_uptr = carve_i(NS + 1); e->so_upair = carve_i(2 * NP); e->so_uslot = carve_i(2 * NP);
    e->so_aff = carve_i(6 * NS); e->so_ma = carve_i(NS); e->so_mb = carve_i(NS);
    e->stage_ints = so;
    ALLOC(e->d_stage, so);
    e->d_stream_image = e->d_stage + e->so_image; e->d_stream_rot = e->d_stage + e->so_rot;
    e->d_pair_a = e->d_stage + e->so_pa; e->d_pair_b = e->d_stage + e->so_pb;
    e->d_seq_t = e->d_stage + e->so_seq_t; e->d_seq_h = e->d_stage + e->so_seq_h;
    e->d_user_ptr = e->d_stage + e->so_uptr; e->d_user_pair = e->d_stage + e->so_upair; e->d_user_slot = e->d_stage + e->so_uslot;
    e->d_affine = (float*)(e->d_stage + e->so_aff);
    for (int k = 0; k < 2; ++k) {
        HIP_OK(hipHostMalloc((void**)&e->h_stage[k], (size_t)so * sizeof(int), hipHostMallocDefault));
        HIP_OK(hipEventCreateWithFlags(&e->ev_stage[k], hipEventDisableTiming));
    }
    return 0;
}

// ------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------
static inline double* fsum(smg_engine* e, const StatArr& s) { return e->fstat + s.off; }
static inline double* fsq(smg_engine* e, const StatArr& s) { return e->fstat + e->fstat_span + s.off; }
static inline double* b1(smg_engine* e, const StatArr& s) { return e->bstat + s.off; }
static inline double* b2(smg_engine* e, const StatArr& s) { return e->bstat + e->bstat_span + s.off; }

static const float kEps = 1e-5f;

// BN statistics table at float offset `at` of the table arena ([rows_max][C] mean, then invstd), from row r0 on, with the
// affine parameters of the consuming BatchNorm
static BnTab bn_table(smg_engine* e, int64_t at, int rows_max, int r0, int C, const float* gamma, const float* beta) {
    BnTab t;
    t.mean = e->stab + at + (int64_t)r0 * C; t.invstd = t.mean + (int64_t)rows_max * C; t.ld = C; t.gamma = gamma; t.beta = beta;
    return t;
}

static int do_forward(smg_engine* e, const smg_net* net, int trunk_id, int head_id, const smg_batch* B,
                      float* q_out, hipStream_t st) {
    const Layout& L = *e->L;
    const TrunkRef& T = L.trunk[trunk_id];
    const HeadRef& Hd = L.head[head_id];
    const int NS = B->n_streams, NP = B->n_pairs;
    if (NS < 1 || NS > e->max_streams || NP < 1 || NP > e->max_pairs) return fail(-22, "batch exceeds engine capacity");
    if (!B->images_nchw_dev && !B->heightmaps_dev) return fail(-22, "no input images");
    for (int s = 0; s < NS; ++s)
        if (B->stream_image[s] < 0 || B->stream_image[s] >= B->n_images) return fail(-22, "stream_image out of range");
    if (B->masks_dev) {
        if (!B->heightmaps_dev || !B->stream_mask_a || !B->stream_mask_b) return fail(-22, "device masks need the heightmap input form and both index arrays");
        for (int s = 0; s < NS; ++s)
            if (B->stream_mask_a[s] >= B->n_masks || B->stream_mask_b[s] >= B->n_masks || (B->stream_mask_a[s] < 0 && B->stream_mask_b[s] >= 0))
                return fail(-22, "stream mask index out of range");
    }
    for (int j = 0; j < NP; ++j)
        if (B->pair_a[j] < 0 || B->pair_a[j] >= NS || B->pair_b[j] < 0 || B->pair_b[j] >= NS) return fail(-22, "pair index out of range");
    const int n_seq_t = B->bn_seq_trunk ? B->n_bn_seq_trunk : 0, n_seq_h = B->bn_seq_head ? B->n_bn_seq_head : 0;
    const int R = e->max_streams > e->max_pairs ? e->max_streams : e->max_pairs;
    if (n_seq_t > 4 * R + 16 || n_seq_h > 4 * R + 16) return fail(-22, "bn sequence too long");
    int pad = 0;
    if (B->heightmaps_dev) {
        pad = (e->S - 2 * B->hm_size) / 2;
        if (pad < 0 || 2 * B->hm_size + 2 * pad != e->S) return fail(-22, "heightmap size does not match engine input_size");
    }

    {   // batch description -> device: no host synchronisation on the forward path
        const int turn = e->stage_turn; e->stage_turn ^= 1;
        HIP_OK(hipEventSynchronize(e->ev_stage[turn]));   // the copy issued two forwards ago (long done)
        int* h = e->h_stage[turn];
        memcpy(h + e->so_image, B->stream_image, NS * sizeof(int));
        memcpy(h + e->so_rot, B->stream_rotated, NS * sizeof(int));
        memcpy(h + e->so_aff, B->stream_affine, 6 * NS * sizeof(float));
        if (B->masks_dev) {
            memcpy(h + e->so_ma, B->stream_mask_a, NS * sizeof(int));
            memcpy(h + e->so_mb, B->stream_mask_b, NS * sizeof(int));
        }
        memcpy(h + e->so_pa, B->pair_a, NP * sizeof(int));
        memcpy(h + e->so_pb, B->pair_b, NP * sizeof(int));
        if (n_seq_t) memcpy(h + e->so_seq_t, B->bn_seq_trunk, n_seq_t * sizeof(int));
        if (n_seq_h) memcpy(h + e->so_seq_h, B->bn_seq_head, n_seq_h * sizeof(int));
        // users of each stream's features (CSR), for the backward
        int* ptr = h + e->so_uptr; int* up = h + e->so_upair; int* us = h + e->so_uslot; int n = 0;
        ptr[0] = 0;
        for (int s = 0; s < NS; ++s) {
            for (int j = 0; j < NP; ++j) {
                if (B->pair_a[j] == s) { up[n] = j; us[n] = 0; ++n; }
                if (B->pair_b[j] == s) { up[n] = j; us[n] = 1; ++n; }
            }
            ptr[s + 1] = n;
        }
        HIP_OK(hipMemcpyAsync(e->d_stage, h, (size_t)e->stage_ints * sizeof(int), hipMemcpyHostToDevice, st));
        HIP_OK(hipEventRecord(e->ev_stage[turn], st));
    }
    HIP_OK(hipMemsetAsync(e->fstat, 0, 2 * e->fstat_span * sizeof(double), st));

    // weights -> K-major packs
    {
        const unsigned n_pack = (unsigned)(e->h_pack[trunk_id].size() + e->h_pack_head[head_id].size());
        ProfScope ps(e, st, K_OTHER, 0);
        hipLaunchKernelGGL(pack_weights_kernel, dim3(64, n_pack), dim3(256), 0, st,
                           e->d_pack + (trunk_id * 3 + head_id) * e->pack_stride, net->params, e->packed_u, e->packed_f, e->prec);
    }
    const float* P = net->params;

    // The trunk of streams [s0, s0 + ns).  Streams are independent up to the head (BN statistics are per stream), so
    // the batch is run as TWO chains on two HIP streams: the tail of one chain's kernel overlaps the other chain's
    // next kernel (two full sweeps side by side take 83 % of their serial time, tests/gpu_concurrency_probe.py).
    // BN table of one consumer layer (rows [r0, r0 + rows) of a [max rows][C] table at float offset `at`) + the launch that fills it
    // table entries of channels [c0, c0 + C) whose producer is not a dense layer (block inputs, head features)
    auto bn_stat = [&](hipStream_t cs, const BnTab& t, int rows, const double* sum, const double* sq, int sstride, int c0, int C, int count) {
        BnStatArgs a;
        a.sum = sum; a.sq = sq; a.sstride = sstride; a.eps = kEps; a.inv_count = 1.0 / (double)count;
        a.mean = const_cast<float*>(t.mean); a.invstd = const_cast<float*>(t.invstd); a.ld = t.ld; a.c0 = c0; a.C = C; a.rows = rows;
        ProfScope ps(e, cs, K_OTHER, 0);
        hipLaunchKernelGGL(bn_stat_kernel, dim3((rows * C + 255) / 256), dim3(256), 0, cs, a);
    };
    // Units [u_lo, u_hi) of the chain: unit 0 = input preparation + stem + pool0, then one unit per dense layer and per
    // transition.  The caller alternates the chains unit by unit, so that both have work queued from the start (a chain
    // enqueued whole keeps the host busy for ~1.3 ms, during which the other chain's HIP stream sits empty).
    auto trunk_chain = [&](const int s0, const int ns, hipStream_t cs, const int u_lo, const int u_hi) -> int {
        int unit = 0;
        auto on = [&]() { const bool r = unit >= u_lo && unit < u_hi; ++unit; return r; };
        auto xs = [&](int b) { return e->X[b] + (int64_t)s0 * e->p_blk[b].HWp * kBlockCtot[b]; };
        auto st_off = [&](double* base, int stride) { return base + (int64_t)s0 * stride; };
        float* img4 = e->img4 + (int64_t)s0 * e->p_img.HWp * 4;
        float* stem = e->stem + (int64_t)s0 * e->p_stem.HWp * 64;
        const bool head_unit = on();
        if (head_unit) {   // K1 input preparation
            PrepArgs a;
            a.images_nchw = B->images_nchw_dev; a.heightmaps = B->heightmaps_dev; a.hm = B->hm_size; a.pad = pad; a.S = e->S;
            a.mean = B->image_mean; a.stdv = B->image_std;
            a.stream_image = e->d_stream_image + s0; a.stream_affine = e->d_affine + 6 * s0; a.stream_rotated = e->d_stream_rot + s0;
            a.img4 = img4; a.HWp = e->p_img.HWp;
            a.masks = B->masks_dev; a.stream_mask_a = e->d_stage + e->so_ma + s0; a.stream_mask_b = e->d_stage + e->so_mb + s0;
            ProfScope ps(e, cs, K_OTHER, 0);
            hipLaunchKernelGGL(prep_rotate_kernel, dim3((e->S * e->S + 255) / 256, ns), dim3(256), 0, cs, a);
        }
        if (head_unit) {   // stem conv0 7x7/2
            auto run = [&](auto tag) {
                using Cfg = decltype(tag);
                FwdConvP<Cfg, F_STEM> p{};
                p.src = img4; p.lds_ = 4; p.ps = e->p_img; p.po = e->p_stem; p.K = 0;
                p.wp = e->packed_u + e->pk_conv0; p.K8tot = 224 / 8; p.N = 64;
                p.dst = stem; p.ldd = 64; p.dcoff = 0;
                p.dsum = st_off(fsum(e, e->st_stem), 64); p.dsq = st_off(fsq(e, e->st_stem), 64); p.dstride = 64;
                BY(e, 4.0 * ns * ((double)e->p_img.HW * 4 + (double)e->p_stem.HW * 64));
                launch_gemm(e, cs, p, dim3(ns * e->p_stem.HWp / Cfg::BM, 1), K_STEM, 2.0 * ns * e->p_stem.HW * 64 * 147);
            };
            if (e->p_stem.HWp % 128 == 0) run(CfgP128x64{}); else run(CfgP64x64{});
        }
        if (head_unit) {   // norm0 + relu0 + pool0
            Pool0Args a;
            a.stem = stem; a.ps = e->p_stem; a.ssum = st_off(fsum(e, e->st_stem), 64); a.ssq = st_off(fsq(e, e->st_stem), 64);
            a.gamma = P + T.norm0.w; a.beta = P + T.norm0.b; a.eps = kEps;
            a.x1 = xs(0); a.ldx = kBlockCtot[0]; a.po = e->p_blk[0];
            a.dsum = st_off(fsum(e, e->st_X[0]), kBlockCtot[0]); a.dsq = st_off(fsq(e, e->st_X[0]), kBlockCtot[0]); a.dstride = kBlockCtot[0];
            a.argmax = e->argmax + (int64_t)s0 * e->p_blk[0].HWp * 64;
            ProfScope ps(e, cs, K_OTHER, 0);
            hipLaunchKernelGGL(pool0_kernel, dim3(e->p_blk[0].HWp / 64, ns), dim3(256), 0, cs, a);
        }
        for (int b = 0; b < 4; ++b) {
            e->prof_stage = b;
            const Plane pl = e->p_blk[b];
            const int Ct = kBlockCtot[b];
            double* xsum = st_off(fsum(e, e->st_X[b]), Ct); double* xsq = st_off(fsq(e, e->st_X[b]), Ct);
            for (size_t i = 0; i < T.layers[b].size(); ++i) {
                if (!on()) continue;
                const DenseLayerRef& d = T.layers[b][i];
                float* bt = e->Bt + e->bt_off[b][i] + (int64_t)s0 * pl.HWp * kBottleneck;
                double* bsum = st_off(fsum(e, e->st_Bt[b][i]), kBottleneck); double* bsq = st_off(fsq(e, e->st_Bt[b][i]), kBottleneck);
                {   // norm1 + relu + conv1 (1x1, cin -> 128)
                    const BnTab t1 = bn_table(e, e->sx_tab[b], e->max_streams, s0, Ct, P + d.n1.w, P + d.n1.b);
                    if (i == 0) bn_stat(cs, t1, ns, xsum, xsq, Ct, 0, d.cin, pl.HW);     // block input: from pool0 / the transition
                    auto run = [&](auto tag) {
                        using Cfg = decltype(tag);
                        FwdConvP<Cfg, F_ONE> p{};
                        p.src = xs(b); p.lds_ = Ct; p.ps = pl; p.po = pl; p.K = d.cin;
                        p.bt = t1; p.fresh0 = i == 0 ? d.cin : d.cin - kGrowth; p.fsum = xsum; p.fsq = xsq; p.fstride = Ct; p.eps = kEps;
                        p.tw_mean = const_cast<float*>(t1.mean); p.tw_invstd = const_cast<float*>(t1.invstd);
                        p.wp = e->packed_u + e->pk_c1[b][i]; p.K8tot = d.cin / 8; p.N = kBottleneck;
                        p.dst = bt; p.ldd = kBottleneck; p.dcoff = 0;
                        p.dsum = bsum; p.dsq = bsq; p.dstride = kBottleneck;
                        BY(e, 4.0 * ns * pl.HW * (d.cin + kBottleneck));
                        launch_gemm(e, cs, p, dim3(ns * pl.HWp / Cfg::BM, kBottleneck / Cfg::BN), K_C1, 2.0 * ns * pl.HW * d.cin * kBottleneck);
                    };
                    // 128x128 tiles where the plane tiles by 128 rows and the launch still fills the chip; else 64x64 (BK = 32) -
                    // and when even that leaves most CUs idle (few streams per call, or the 20x20 planes), 32x64 tiles with
                    // the k-tile split over wave pairs: twice the workgroups, half the serial K chain.
                    static const int small_wgs = getenv("SMG_C1_SMALL") ? atoi(getenv("SMG_C1_SMALL")) : 320;   // 512 / 1024 measured slower on the 17-stream step
                    const int wg128 = ns * pl.HWp / 128, wg64 = ns * pl.HWp / 64 * 2;
                    static const int k16 = getenv("SMG_C1_K16") ? atoi(getenv("SMG_C1_K16")) : 1 << 30;      // dev A/B: BK = 16 past this many channels
                    static const int mid = getenv("SMG_C1_MID") ? atoi(getenv("SMG_C1_MID")) : 0;                     // dev A/B
                    static const int deep_min = getenv("SMG_C1_DEEP") ? atoi(getenv("SMG_C1_DEEP")) : 1 << 30;       // dev A/B
                    static const bool ws_on = !(getenv("SMG_C1_WS") && atoi(getenv("SMG_C1_WS")) == 0);      // wave-specialised 64x64x32 (ws.cuh); SMG_C1_WS=0: the generic kernel (A/B, cross-check)
                    if (ws_on && !(pl.HWp % 128 == 0 && wg128 >= small_wgs) && wg64 >= small_wgs && d.cin % 32 == 0 && pl.HWp % 64 == 0) {
                        Fwd1x1WsArgs a{};
                        a.src = xs(b); a.lds_ = Ct; a.pl = pl; a.K = d.cin;
                        a.bt = t1; a.fresh0 = i == 0 ? d.cin : d.cin - kGrowth; a.fsum = xsum; a.fsq = xsq; a.fstride = Ct; a.eps = kEps;
                        a.tw_mean = const_cast<float*>(t1.mean); a.tw_invstd = const_cast<float*>(t1.invstd);
                        a.wp = e->packed_u + e->pk_c1[b][i]; a.N = kBottleneck;
                        a.dst = bt; a.ldd = kBottleneck; a.dsum = bsum; a.dsq = bsq; a.dstride = kBottleneck;
                        const int nM = ns * pl.HWp / 64, nN = kBottleneck / 64;
                        a.tm = TileMap{nM, nN, 0};
                        BY(e, 4.0 * ns * pl.HW * (d.cin + kBottleneck));
                        ProfScope ps(e, cs, K_C1, 2.0 * ns * pl.HW * d.cin * kBottleneck);
                        const size_t smem = WsGeo::smem_bytes(d.cin);
                        static bool raised[64][3] = {};
                        if (!raised[e->device & 63][e->prec]) {
                            PREC_DISPATCH(e, (void)hipFuncSetAttribute((const void*)conv1x1_fwd_ws_kernel<PREC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                            raised[e->device & 63][e->prec] = true;
                        }
                        PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1x1_fwd_ws_kernel<PREC>), dim3(8 * ((nM + 7) / 8) * nN), dim3(512), smem, cs, a));
                    } else
                    if (pl.HWp % 128 == 0 && wg128 >= small_wgs) run(CfgP128x128{});
                    else if (pl.HWp % 128 == 0 && wg128 >= deep_min && d.cin % 32 == 0) run(CfgP128x128d{});
                    else if (wg64 < small_wgs) run(CfgP32x64{});
                    else if (mid == 3) run(CfgP64x64w{});
                    else if (mid == 1) run(CfgP64x128{});
                    else if (mid == 2 && d.cin % 32 == 0) run(CfgP64x128d{});
                    else if (d.cin > k16) run(CfgP64x64k16{});
                    else run(CfgP64x64{});
                }
                const BnTab t2 = bn_table(e, e->sb_tab[b][i], e->max_streams, s0, kBottleneck, P + d.n2.w, P + d.n2.b);
                if (e->generic3x3) bn_stat(cs, t2, ns, bsum, bsq, kBottleneck, 0, kBottleneck, pl.HW);
                if (!e->generic3x3) {
                    // norm2 + relu + conv2 (3x3, 128 -> 32) with an LDS-resident input halo (halo.cuh)
                    Halo3x3FwdArgs a;
                    a.src = bt; a.lds_ = kBottleneck; a.pl = pl; a.C = kBottleneck;
                    a.ssum = bsum; a.ssq = bsq; a.sstride = kBottleneck; a.gamma = P + d.n2.w; a.beta = P + d.n2.b; a.eps = kEps;
                    a.tw_mean = const_cast<float*>(t2.mean); a.tw_invstd = const_cast<float*>(t2.invstd);
                    a.dst = xs(b); a.ldd = Ct; a.dcoff = d.cin;
                    a.dsum = xsum; a.dsq = xsq; a.dstride = Ct;
                    a.wu = e->packed_u + e->pk_hf[b][i];
                    BY(e, 4.0 * ns * pl.HW * (kBottleneck + kGrowth));
                    ProfScope ps(e, cs, K_C3, 2.0 * ns * pl.HW * 9 * kBottleneck * kGrowth);
                    if (halo_tile(pl, ns) == 16) {
                        a.tiles_x = pl.W / 16;
                        PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_halo_fwd_kernel<16, PREC>), dim3((pl.H / 16) * a.tiles_x, ns), dim3(256),
                                           HaloFwdSGeo<16>::smem_bytes(kBottleneck), cs, a));
                    } else {
                        a.tiles_x = (pl.W + 7) / 8;
                        PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_halo_fwd_kernel<8, PREC>), dim3(((pl.H + 7) / 8) * a.tiles_x, ns), dim3(256),
                                           HaloFwdSGeo<8>::smem_bytes(kBottleneck), cs, a));
                    }
                } else {   // norm2 + relu + conv2 (3x3, 128 -> 32), appended to the block buffer (generic implicit GEMM)
                    auto run = [&](auto tag) {
                        using Cfg = decltype(tag);
                        FwdConvP<Cfg, F_THREE> p{};
                        p.src = bt; p.lds_ = kBottleneck; p.ps = pl; p.po = pl; p.K = kBottleneck;
                        p.bt = t2; p.fresh0 = kBottleneck; p.fsum = bsum; p.fsq = bsq; p.fstride = kBottleneck; p.eps = kEps;
                        p.tw_mean = const_cast<float*>(t2.mean); p.tw_invstd = const_cast<float*>(t2.invstd);
                        p.wp = e->packed_u + e->pk_g3f[b][i]; p.K8tot = 9 * kBottleneck / 8; p.N = kGrowth;
                        p.dst = xs(b); p.ldd = Ct; p.dcoff = d.cin;
                        p.dsum = xsum; p.dsq = xsq; p.dstride = Ct;
                        BY(e, 4.0 * ns * pl.HW * (kBottleneck + kGrowth));
                        launch_gemm(e, cs, p, dim3(ns * pl.HWp / Cfg::BM, 1), K_C3, 2.0 * ns * pl.HW * 9 * kBottleneck * kGrowth);
                    };
                    if (pl.HWp % 128 == 0) run(CfgP128x32{}); else run(CfgP64x32{});
                }
            }
            if (b < 3 && on()) {   // transition: norm + relu + (avgpool2 commuted in front of) conv 1x1
                const Plane pn = e->p_blk[b + 1];
                const int Cn = kBlockCtot[b + 1];
                const BnTab tt = bn_table(e, e->sx_tab[b], e->max_streams, s0, Ct, P + T.tnorm[b].w, P + T.tnorm[b].b);
                auto run = [&](auto tag) {
                    using Cfg = decltype(tag);
                    FwdConvP<Cfg, F_POOL> p{};
                    p.src = xs(b); p.lds_ = Ct; p.ps = pl; p.po = pn; p.K = Ct;
                    p.bt = tt; p.fresh0 = Ct - kGrowth; p.fsum = xsum; p.fsq = xsq; p.fstride = Ct; p.eps = kEps;     // the block's last layer
                    p.tw_mean = const_cast<float*>(tt.mean); p.tw_invstd = const_cast<float*>(tt.invstd);
                    p.wp = e->packed_u + e->pk_t[b]; p.K8tot = Ct / 8; p.N = Ct / 2;
                    p.dst = xs(b + 1); p.ldd = Cn; p.dcoff = 0;
                    p.dsum = st_off(fsum(e, e->st_X[b + 1]), Cn); p.dsq = st_off(fsq(e, e->st_X[b + 1]), Cn); p.dstride = Cn;
                    BY(e, 4.0 * ns * ((double)pl.HW * Ct + (double)pn.HW * (Ct / 2)));
                    launch_gemm(e, cs, p, dim3(ns * pn.HWp / Cfg::BM, (Ct / 2) / Cfg::BN), K_TRANS, 2.0 * ns * pn.HW * Ct * (Ct / 2));
                };
                if (pn.HWp % 128 == 0) run(CfgP128x128{}); else run(CfgP64x128{});
            }
        }
        return 0;
    };
    static const bool one_chain = getenv("SMG_FWD_ONE_CHAIN") != nullptr;      // dev: A/B switch
    if (NS >= 2 && !e->prof && !e->serialize && !one_chain) {
        const int h = NS / 2;
        HIP_OK(hipEventRecord(e->ev_misc, st));                 // packed weights + batch description are ready
        HIP_OK(hipStreamWaitEvent(e->side, e->ev_misc, 0));
        int n_units = 1 + 3;
        for (int b = 0; b < 4; ++b) n_units += (int)T.layers[b].size();
        for (int u = 0; u < n_units; ++u) {
            if (trunk_chain(0, h, st, u, u + 1)) return -5;
            if (trunk_chain(h, NS - h, e->side, u, u + 1)) return -5;
        }
        HIP_OK(hipEventRecord(e->ev_end, e->side));             // join before the head reads every stream's features
        HIP_OK(hipStreamWaitEvent(st, e->ev_end, 0));
    } else {
        if (trunk_chain(0, NS, st, 0, 1 << 30)) return -5;      // profiling: one chain, per-kernel times stay per layer
    }
    e->prof_stage = -1;
    const Plane p4 = e->p_blk[3];
    {   // norm5 + two-stream concat
        FeatArgs a;
        a.x4 = e->X[3]; a.p4 = p4; a.xsum = fsum(e, e->st_X[3]); a.xsq = fsq(e, e->st_X[3]);
        a.gamma = P + T.norm5.w; a.beta = P + T.norm5.b; a.eps = kEps;
        a.pair_a = e->d_pair_a; a.pair_b = e->d_pair_b; a.F = e->F;
        a.fsum = fsum(e, e->st_F); a.fsq = fsq(e, e->st_F); a.chunk = 64;
        ProfScope ps(e, st, K_OTHER, 0);
        hipLaunchKernelGGL(feat_kernel, dim3(2, NP, (p4.HW + 63) / 64), dim3(256), 0, st, a);
    }
    {   // head norm0 + relu + conv0 (1x1, 2048 -> 64)
        const BnTab th = bn_table(e, e->sf_tab, e->max_pairs, 0, 2 * kFeat, P + Hd.n0.w, P + Hd.n0.b);
        bn_stat(st, th, NP, fsum(e, e->st_F), fsq(e, e->st_F), 2 * kFeat, 0, 2 * kFeat, p4.HW);
        auto run = [&](auto tag) {
                using Cfg = decltype(tag);
                FwdConvP<Cfg, F_ONE> p{};
        p.src = e->F; p.lds_ = 2 * kFeat; p.ps = p4; p.po = p4; p.K = 2 * kFeat;
        p.bt = th; p.fresh0 = 2 * kFeat; p.fsum = fsum(e, e->st_F); p.fsq = fsq(e, e->st_F); p.fstride = 2 * kFeat; p.eps = kEps;
        p.tw_mean = const_cast<float*>(th.mean); p.tw_invstd = const_cast<float*>(th.invstd);
        p.wp = e->packed_u + e->pk_head0; p.K8tot = 2 * kFeat / 8; p.N = kHeadMid;
        p.dst = e->H1; p.ldd = kHeadMid; p.dcoff = 0;
        p.dsum = fsum(e, e->st_H1); p.dsq = fsq(e, e->st_H1); p.dstride = kHeadMid;
        BY(e, 4.0 * NP * p4.HW * (2 * kFeat + kHeadMid));
        launch_gemm(e, st, p, dim3(NP * p4.HWp / Cfg::BM, 1), K_HEAD0, 2.0 * NP * p4.HW * 2 * kFeat * kHeadMid);
            };
            if (p4.HWp % 128 == 0) run(CfgP128x64{}); else run(CfgP64x64{});
    }
    {   // head norm1 + relu + conv1 (20x20 valid)
        ValueArgs a;
        a.h1 = e->H1; a.p4 = p4; a.hsum = fsum(e, e->st_H1); a.hsq = fsq(e, e->st_H1);
        a.gamma = P + Hd.n1.w; a.beta = P + Hd.n1.b; a.eps = kEps;
        a.w2p = e->packed_f + e->pk_head1; a.q = q_out; a.out_ch = e->head_out; a.OH = e->OH; a.OW = e->OW;
        ProfScope ps(e, st, K_OTHER, 0);
        hipLaunchKernelGGL(value_conv_kernel, dim3(NP * e->head_out * e->OH * e->OW), dim3(256), 0, st, a);
    }
    if (n_seq_t || n_seq_h) {   // BN running statistics, in the reference's update order
        ProfScope ps(e, st, K_OTHER, 0);
        hipLaunchKernelGGL(bn_update_kernel, dim3(8, (unsigned)e->n_bnupd), dim3(256), 0, st,
                           e->d_bnupd + (trunk_id * 3 + head_id) * e->bnupd_stride,
                           e->fstat, e->fstat + e->fstat_span, net->bufs, net->nbt, e->d_seq_t, n_seq_t, e->d_seq_h, n_seq_h);
    }
    HIP_OK(hipGetLastError());
    e->have_fwd = true; e->f_trunk = trunk_id; e->f_head = head_id; e->f_streams = NS; e->f_pairs = NP;
    return 0;
}

// ------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------
// Pixel-chunk size of a weight-gradient launch: enough workgroups to fill the chip
// (~768) but no more - every workgroup ends with one fp32 atomicAdd per output element.
static void pick_chunk(const Plane& pl, int n_planes, int tiles_per_chunk, int& chunk, int& cps, int target = 768) {
    const int want = (target + tiles_per_chunk - 1) / tiles_per_chunk;
    cps = (want + n_planes - 1) / n_planes;
    if (cps < 1) cps = 1;
    chunk = ((pl.HWp + cps - 1) / cps + 63) / 64 * 64;
    cps = (pl.HWp + chunk - 1) / chunk;
}

static int do_backward(smg_engine* e, const smg_net* net, const float* dq, hipStream_t st) {
    if (!e->have_fwd) return fail(-22, "smg_backward without a preceding smg_forward");
    if (!net->grads) return fail(-22, "net.grads is NULL");
    const Layout& L = *e->L;
    const TrunkRef& T = L.trunk[e->f_trunk];
    const HeadRef& Hd = L.head[e->f_head];
    const int NS = e->f_streams, NP = e->f_pairs;
    const float* P = net->params;
    float* Gr = net->grads;
    const Plane p4 = e->p_blk[3];
    HIP_OK(hipMemsetAsync(e->bstat, 0, 2 * e->bstat_span * sizeof(double), st));
    // Weight-gradient kernels only read what the data-gradient chain produces and write disjoint
    // gradient ranges, so they run on a second stream beside it (their MFMA/L2-bound phases overlap the
    // HBM-bound epilogues of the data-gradient kernels).  While profiling everything is serialised on
    // `st` so that per-kernel durations stay clean.
    // (a lowest-priority stream for the weight gradients gains 0.2 ms per step with one engine alive, and LOSES 10 ms as soon
    // as a second engine - two more streams - exists in the process: the streams then share hardware queues and serialise)
    const hipStream_t s2 = (e->prof || e->serialize) ? st : e->side;
    auto fork = [&](hipEvent_t ev) -> int {      // side stream continues after everything enqueued on st so far
        HIP_OK(hipEventRecord(ev, st));
        HIP_OK(hipStreamWaitEvent(s2, ev, 0));
        return 0;
    };
    int layer_no = 0;

    {   // value conv backward + relu1 + norm1 sums
        ValueBwdArgs a;
        a.h1 = e->H1; a.p4 = p4; a.hsum = fsum(e, e->st_H1); a.hsq = fsq(e, e->st_H1);
        a.gamma = P + Hd.n1.w; a.beta = P + Hd.n1.b; a.eps = kEps; a.w2p = e->packed_f + e->pk_head1;
        a.dq = dq; a.out_ch = e->head_out; a.OH = e->OH; a.OW = e->OW; a.dh1 = e->DH1;
        a.o1 = b1(e, e->bs_H1); a.o2 = b2(e, e->bs_H1); a.dbeta = Gr + Hd.n1.b; a.dgamma = Gr + Hd.n1.w; a.dw2 = Gr + Hd.c1.w;
        ProfScope ps(e, st, K_OTHER, 0);
        hipLaunchKernelGGL(value_bwd_kernel, dim3((p4.HW + 63) / 64, NP), dim3(256), 0, st, a);
    }
    int chunk4, cps4;
    pick_chunk(p4, NP, 2 * kFeat / 64, chunk4, cps4);
    {   // head conv0 weight gradient
        BwdWeightP<CfgW64x64, W_ONE, C_IDENT> p{};
        p.gbuf = e->DH1; p.ldg = kHeadMid; p.gcoff = 0; p.xbuf = e->H1; p.ldx = kHeadMid; p.xcoff = 0; p.pa = p4; p.MA = kHeadMid;
        p.xsum = fsum(e, e->st_H1); p.xsq = fsq(e, e->st_H1); p.xstride = kHeadMid;
        p.s1 = b1(e, e->bs_H1); p.s2 = b2(e, e->bs_H1); p.sstride = kHeadMid; p.scoff = 0; p.agamma = P + Hd.n1.w;
        p.bbuf = e->F; p.ldb = 2 * kFeat; p.pb = p4; p.NB = 2 * kFeat;
        p.bsum = fsum(e, e->st_F); p.bsq = fsq(e, e->st_F); p.bstride = 2 * kFeat; p.bgamma = P + Hd.n0.w; p.bbeta = P + Hd.n0.b;
        p.eps = kEps; p.chunk = chunk4; p.chunks_per_stream = cps4; p.n_chunks = NP * cps4;
        p.dw = Gr + Hd.c0.w; p.ldw_out = 2 * kFeat;
        if (fork(e->ev_misc)) return -5;
        BY(e, 4.0 * NP * p4.HW * (2 * kHeadMid + 2 * kFeat));
        launch_wgrad(e, s2, p, dim3(1, 2 * kFeat / 64, NP * cps4), K_HW0, 2.0 * NP * p4.HW * 2 * kFeat * kHeadMid, 1, C_IDENT);
    }
    {   // head conv0 data gradient + relu0 + norm0 sums
        auto run = [&](auto tag) {
                using Cfg = decltype(tag);
                BwdDataP<Cfg, false, E_STORE> p{};
        p.gbuf = e->DH1; p.ldg = kHeadMid; p.gcoff = 0; p.xbuf = e->H1; p.ldx = kHeadMid; p.xcoff = 0; p.pa = p4; p.KA = kHeadMid;
        p.xsum = fsum(e, e->st_H1); p.xsq = fsq(e, e->st_H1); p.xstride = kHeadMid;
        p.s1 = b1(e, e->bs_H1); p.s2 = b2(e, e->bs_H1); p.sstride = kHeadMid; p.scoff = 0; p.agamma = P + Hd.n1.w;
        p.wp = e->packed_u + e->pk_hd0; p.K8tot = kHeadMid / 8; p.ldn = 2 * kFeat; p.wcol0 = 0; p.N = 2 * kFeat;
        p.mbuf = e->F; p.ldm = 2 * kFeat; p.mcoff = 0; p.pm = p4;
        p.msum = fsum(e, e->st_F); p.msq = fsq(e, e->st_F); p.mstride = 2 * kFeat; p.egamma = P + Hd.n0.w; p.ebeta = P + Hd.n0.b;
        p.dst = e->DF; p.ldd = 2 * kFeat; p.dcoff = 0;
        p.o1 = b1(e, e->bs_F); p.o2 = b2(e, e->bs_F); p.ostride = 2 * kFeat; p.ocoff = 0;
        p.dbeta = Gr + Hd.n0.b; p.dgamma = Gr + Hd.n0.w; p.eps = kEps;
        BY(e, 4.0 * NP * p4.HW * (2 * kHeadMid + 2 * 2 * kFeat));
        launch_gemm(e, st, p, dim3(NP * p4.HWp / Cfg::BM, 2 * kFeat / Cfg::BN), K_HD0, 2.0 * NP * p4.HW * 2 * kFeat * kHeadMid);
            };
            if (p4.HWp % 128 == 0) run(CfgP128x128{}); else run(CfgP64x128{});
    }
    {   // head norm0 backward + concat backward + norm5 backward -> G'_4
        Norm5BwdArgs a;
        a.DF = e->DF; a.F = e->F; a.p4 = p4; a.fsum = fsum(e, e->st_F); a.fsq = fsq(e, e->st_F);
        a.f1 = b1(e, e->bs_F); a.f2 = b2(e, e->bs_F); a.hgamma = P + Hd.n0.w;
        a.x4 = e->X[3]; a.xsum = fsum(e, e->st_X[3]); a.xsq = fsq(e, e->st_X[3]); a.gamma5 = P + T.norm5.w; a.eps = kEps;
        a.user_ptr = e->d_user_ptr; a.user_pair = e->d_user_pair; a.user_slot = e->d_user_slot;
        a.G4 = e->G[3]; a.SA = b1(e, e->bs_X[3]); a.SB = b2(e, e->bs_X[3]);
        a.dbeta5 = Gr + T.norm5.b; a.dgamma5 = Gr + T.norm5.w; a.chunk = 16;
        ProfScope ps(e, st, K_OTHER, 0);
        hipLaunchKernelGGL(norm5_bwd_kernel, dim3(1, NS, (p4.HW + 15) / 16), dim3(256), 0, st, a);
    }
    for (int b = 3; b >= 0; --b) {
        e->prof_stage = b;
        const Plane pl = e->p_blk[b];
        const int Ct = kBlockCtot[b];
        for (int i = (int)T.layers[b].size() - 1; i >= 0; --i) {
            const DenseLayerRef& d = T.layers[b][i];
            float* bt = e->Bt + e->bt_off[b][i];
            const int db = layer_no % kRing;
            float* GSb = e->GS[db];
            float* D2b = e->D2[db];
            if (layer_no >= kRing) HIP_OK(hipStreamWaitEvent(st, e->ev_side[db], 0));   // side stream done with these buffers (kRing layers ago)
            ++layer_no;
            // This layer's finished output-slice gradient GS = invstd*(G' - SA/n - xhat*SB/n), materialised once (dense
            // [px][32]) for the 3x3 data- and weight-gradient kernels.  They can also apply it while loading the G' / X
            // slices (GradSrc with x set; SMG_GS_FUSED=1): one launch less on the dependency chain, but measured 0.5 ms
            // per step slower - two strided 128-B-per-pixel reads replace one dense one in both consumers.
            GradSrc gsrc{};
            gsrc.g = e->G[b] + d.cin; gsrc.ldg = Ct; gsrc.x = e->X[b] + d.cin; gsrc.ldx = Ct;
            gsrc.xsum = fsum(e, e->st_X[b]) + d.cin; gsrc.xsq = fsq(e, e->st_X[b]) + d.cin;
            gsrc.s1 = b1(e, e->bs_X[b]) + d.cin; gsrc.s2 = b2(e, e->bs_X[b]) + d.cin; gsrc.sstride = Ct; gsrc.eps = kEps;
            static const bool gs_env_fused = getenv("SMG_GS_FUSED") != nullptr;
            static const int gs_fused_hw = getenv("SMG_GS_FUSED_HW") ? atoi(getenv("SMG_GS_FUSED_HW")) : 0;   // dev A/B: fuse on planes up to this many pixels
            const bool gs_mat = !gs_env_fused && NS > 4 && pl.HW > gs_fused_hw;      // few streams: launch-bound, the fused form wins (8.78 -> 8.56 ms per sample)
            if (e->generic3x3 || gs_mat) {
                BnBwdApplyArgs a{};
                a.g = e->G[b]; a.ldg = Ct; a.gcoff = d.cin; a.x = e->X[b]; a.ldx = Ct; a.xcoff = d.cin; a.pl = pl; a.C = kGrowth;
                a.xsum = fsum(e, e->st_X[b]); a.xsq = fsq(e, e->st_X[b]); a.xstride = Ct;
                a.s1 = b1(e, e->bs_X[b]); a.s2 = b2(e, e->bs_X[b]); a.sstride = Ct; a.scoff = d.cin; a.gamma = nullptr; a.eps = kEps;
                a.out = GSb; a.ldo = kGrowth;
                BY(e, 4.0 * NS * pl.HW * 3 * kGrowth);
                ProfScope ps(e, st, K_OTHER, 0);
                hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(pl.HWp / 64, NS), dim3(256), 0, st, a);
                if (gs_mat) { gsrc = GradSrc{}; gsrc.g = GSb; gsrc.ldg = kGrowth; }
            }
            if (fork(e->ev_gs[db])) return -5;
            if (!e->generic3x3) {
                // conv2 (3x3) data gradient with the gradient halo resident in LDS (halo.cuh)
                Halo3x3DgradArgs a;
                a.g = gsrc; a.pl = pl; a.C = kBottleneck;
                a.mbuf = bt;
                a.dst = D2b; a.o1 = b1(e, e->bs_Bt[b][i]); a.o2 = b2(e, e->bs_Bt[b][i]); a.ostride = kBottleneck;
                a.wu = e->packed_u + e->pk_hd[b][i]; a.bt = bn_table(e, e->sb_tab[b][i], e->max_streams, 0, kBottleneck, P + d.n2.w, P + d.n2.b);
                BY(e, 4.0 * NS * pl.HW * (kGrowth + 2 * kBottleneck));      // gradient in, mask source in, dy out
                ProfScope ps(e, st, K_D3, 2.0 * NS * pl.HW * 9 * kBottleneck * kGrowth);
                TraceScope ts(st, K_D3, halo_tile(pl, NS) == 16 ? dim3((pl.H / 16) * (pl.W / 16), NS) : dim3(((pl.H + 7) / 8) * ((pl.W + 7) / 8), NS, kBottleneck / 64));
                if (halo_tile(pl, NS) == 16) {
                    a.tiles_x = pl.W / 16; a.cg_per_wg = kBottleneck / 32;
                    PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_halo_dgrad_kernel<16, PREC>), dim3((pl.H / 16) * a.tiles_x, NS), dim3(256),
                                       HaloDgradSGeo<16>::smem_bytes(kBottleneck), st, a));
                } else {
                    a.tiles_x = (pl.W + 7) / 8; a.cg_per_wg = 1;      // small planes: one 64-channel group per workgroup
                    PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_halo_dgrad_kernel<8, PREC>), dim3(((pl.H + 7) / 8) * a.tiles_x, NS, kBottleneck / 64), dim3(256),
                                       HaloDgradSGeo<8>::smem_bytes(kBottleneck), st, a));
                }
            } else {   // conv2 (3x3) data gradient -> dy of relu2/norm2 (D2) + norm2 sums (generic implicit GEMM)
                auto run = [&](auto tag) {
                    using Cfg = decltype(tag);
                    BwdDataP<Cfg, true, E_STORE, false> p{};
                    p.gbuf = GSb; p.ldg = kGrowth; p.gcoff = 0; p.xbuf = nullptr; p.pa = pl; p.KA = kGrowth;
                    p.wp = e->packed_u + e->pk_g3d[b][i]; p.K8tot = 9 * kGrowth / 8; p.ldn = kBottleneck; p.wcol0 = 0; p.N = kBottleneck;
                    p.mbuf = bt; p.ldm = kBottleneck; p.mcoff = 0; p.pm = pl;
                    p.msum = fsum(e, e->st_Bt[b][i]); p.msq = fsq(e, e->st_Bt[b][i]); p.mstride = kBottleneck;
                    p.egamma = P + d.n2.w; p.ebeta = P + d.n2.b;
                    p.dst = D2b; p.ldd = kBottleneck; p.dcoff = 0;
                    p.o1 = b1(e, e->bs_Bt[b][i]); p.o2 = b2(e, e->bs_Bt[b][i]); p.ostride = kBottleneck; p.ocoff = 0;
                    p.dbeta = Gr + d.n2.b; p.dgamma = Gr + d.n2.w; p.eps = kEps;
                    BY(e, 4.0 * NS * pl.HW * (kGrowth + 2 * kBottleneck));
                    launch_gemm(e, st, p, dim3(NS * pl.HWp / Cfg::BM, kBottleneck / Cfg::BN), K_D3, 2.0 * NS * pl.HW * 9 * kBottleneck * kGrowth);
                };
                if (pl.HWp % 128 == 0) run(CfgP128x128{}); else run(CfgP64x128{});
            }
            if (!e->generic3x3) {
                // conv2 weight gradient with the activation halo resident in LDS (halo.cuh)
                const int ts = halo_tile(pl, NS);
                Halo3x3WgradArgs a;
                a.g = gsrc; a.pl = pl; a.src = bt; a.C = kBottleneck;
                const int th = 8;                              // tiles are ts x 8 pixels
                a.bt = bn_table(e, e->sb_tab[b][i], e->max_streams, 0, kBottleneck, P + d.n2.w, P + d.n2.b);
                a.part = e->part; a.tiles_x = (pl.W + ts - 1) / ts; a.n_tiles = ((pl.H + th - 1) / th) * a.tiles_x;
                a.tiles_per_wg = w3_tiles_per_wg(a.n_tiles, ts, NS, e->part_floats, (double)ts / th);
                const int groups = (a.n_tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
                if ((int64_t)groups * NS * 9 * 32 * kBottleneck > e->part_floats) return fail(-12, "partial-gradient workspace too small");
                {
                    BY(e, 4.0 * NS * pl.HW * (kGrowth + kBottleneck));
                    ProfScope ps(e, s2, K_W3, 2.0 * NS * pl.HW * 9 * kBottleneck * kGrowth);
                    if (ts == 16) {
                        PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_halo_wgrad_kernel<16, PREC>), dim3(groups, kBottleneck / 32, NS), dim3(256),
                                                            HaloWgradSGeo<16>::smem_bytes(), s2, a));
                    } else {
                        PREC_DISPATCH(e, hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_halo_wgrad_kernel<8, PREC>), dim3(groups, kBottleneck / 32, NS), dim3(256),
                                                            HaloWgradSGeo<8>::smem_bytes(), s2, a));
                    }
                }
                ReduceArgs r;
                r.part = e->part; r.Z = groups * NS; r.taps = 9; r.rows = kGrowth; r.cols = kBottleneck; r.ldp = kBottleneck;
                r.z_stride = (int64_t)9 * kGrowth * kBottleneck; r.tap_stride = (int64_t)kGrowth * kBottleneck;
                r.dw = Gr + d.c2.w; r.ldw_out = kBottleneck * 9; r.cmap = C_3x3;
                ProfScope ps(e, s2, K_W3, 0);
                hipLaunchKernelGGL(reduce_partials_kernel, dim3((9 * kGrowth * kBottleneck + 255) / 256), dim3(256), 0, s2, r);
            } else {   // conv2 weight gradient (generic implicit GEMM, one launch slice per tap)
                const int chunk = 512, cps = (pl.HWp + chunk - 1) / chunk;   // latency-bound: many short workgroups
                BwdWeightP<CfgW32x128, W_THREE, C_3x3, SMG_PD_WGRAD, false> p{};
                p.gbuf = GSb; p.ldg = kGrowth; p.gcoff = 0; p.xbuf = nullptr; p.pa = pl; p.MA = kGrowth;
                p.bbuf = bt; p.ldb = kBottleneck; p.pb = pl; p.NB = kBottleneck;
                p.bsum = fsum(e, e->st_Bt[b][i]); p.bsq = fsq(e, e->st_Bt[b][i]); p.bstride = kBottleneck;
                p.bgamma = P + d.n2.w; p.bbeta = P + d.n2.b; p.eps = kEps;
                p.chunk = chunk; p.chunks_per_stream = cps; p.n_chunks = NS * cps;
                p.dw = Gr + d.c2.w; p.ldw_out = kBottleneck * 9;
                BY(e, 4.0 * NS * pl.HW * (kGrowth + kBottleneck));
                launch_wgrad(e, s2, p, dim3(1, 1, 9 * NS * cps), K_W3, 2.0 * NS * pl.HW * 9 * kBottleneck * kGrowth, 9, C_3x3);
            }
            {   // norm2 backward applied once, in place: D2 <- gamma2*invstd*(dy - s1/n - xhat*s2/n)
                BnBwdApplyArgs a{};
                a.g = D2b; a.ldg = kBottleneck; a.gcoff = 0; a.x = bt; a.ldx = kBottleneck; a.xcoff = 0; a.pl = pl; a.C = kBottleneck;
                a.xsum = fsum(e, e->st_Bt[b][i]); a.xsq = fsq(e, e->st_Bt[b][i]); a.xstride = kBottleneck;
                a.s1 = b1(e, e->bs_Bt[b][i]); a.s2 = b2(e, e->bs_Bt[b][i]); a.sstride = kBottleneck; a.scoff = 0;
                a.gamma = P + d.n2.w; a.eps = kEps; a.out = D2b; a.ldo = kBottleneck;
                if (!e->generic3x3) { a.dbeta = Gr + d.n2.b; a.dgamma = Gr + d.n2.w; }   // the halo dgrad leaves these to us
                BY(e, 4.0 * NS * pl.HW * 3 * kBottleneck);
                ProfScope ps(e, st, K_OTHER, 0);
                hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(pl.HWp / 64, NS), dim3(256), 0, st, a);
            }
            if (fork(e->ev_d2[db])) return -5;
            // conv1 (1x1) data gradient -> relu1/norm1 backward accumulated into G'.  Layers are grouped (kGroup,
            // from the top of the block): inside a group only the channels the group itself produced - needed by
            // the very next layer - are accumulated per layer; everything below the group's lowest layer is done
            // once for the whole group by BwdDataGroupP (gemm.cuh), which touches G' and x once instead of once
            // per layer.
            const int L = (int)T.layers[b].size();
            const int g_lo = i - ((L - 1 - i) % kGroup == kGroup - 1 ? 0 : std::min(i, kGroup - 1 - (L - 1 - i) % kGroup));
            const int cs = T.layers[b][g_lo].cin;                       // channels below the group
            if (d.cin > cs) {                                           // [cs, cin): per-layer accumulate
                auto run = [&](auto tag) {
                    using Cfg = decltype(tag);
                    BwdDataP<Cfg, false, E_ACCUM, false> p{};
                    p.gbuf = D2b; p.ldg = kBottleneck; p.gcoff = 0; p.xbuf = nullptr; p.pa = pl; p.KA = kBottleneck;
                    p.wp = e->packed_u + e->pk_d1[b][i]; p.K8tot = kBottleneck / 8; p.ldn = d.cin; p.wcol0 = cs; p.N = d.cin - cs;
                    p.mbuf = e->X[b]; p.ldm = Ct; p.mcoff = cs; p.pm = pl;
                    p.msum = fsum(e, e->st_X[b]); p.msq = fsq(e, e->st_X[b]); p.mstride = Ct;
                    p.egamma = P + d.n1.w + cs; p.ebeta = P + d.n1.b + cs;
                    p.dst = e->G[b]; p.ldd = Ct; p.dcoff = cs;
                    p.o1 = b1(e, e->bs_X[b]); p.o2 = b2(e, e->bs_X[b]); p.ostride = Ct; p.ocoff = cs;
                    p.dbeta = Gr + d.n1.b + cs; p.dgamma = Gr + d.n1.w + cs; p.eps = kEps;
                    BY(e, 4.0 * NS * pl.HW * (kBottleneck + 3.0 * p.N));          // dy in; x in, G' read + written
                    launch_gemm(e, st, p, dim3(NS * pl.HWp / Cfg::BM, (p.N + Cfg::BN - 1) / Cfg::BN), K_D1, 2.0 * NS * pl.HW * p.N * kBottleneck);
                };
                if (pl.HWp % 128 == 0) run(CfgP128x64{}); else run(CfgP64x64{});
            }
            if (i == g_lo) {                                            // [0, cs): the whole group at once
                auto run = [&](auto tag) {
                    using Cfg = decltype(tag);
                    BwdDataGroupP<Cfg> p{};
                    const int g_hi = L - 1 - ((L - 1 - g_lo) / kGroup) * kGroup;      // top layer of this group
                    p.nseg = g_hi - g_lo + 1;
                    for (int k = 0; k < p.nseg; ++k) {                  // layer g_lo + k ran (k layers) before this one
                        const DenseLayerRef& dk = T.layers[b][g_lo + k];
                        const int slot = (layer_no - 1 - k + kRing * 4) % kRing;
                        p.seg[k].g = e->D2[slot]; p.seg[k].wp = e->packed_u + e->pk_d1[b][g_lo + k]; p.seg[k].ldn = dk.cin;
                        p.seg[k].gamma = P + dk.n1.w; p.seg[k].beta = P + dk.n1.b;
                        p.seg[k].dbeta = Gr + dk.n1.b; p.seg[k].dgamma = Gr + dk.n1.w;
                    }
                    p.ldg = kBottleneck; p.pa = pl; p.KA = kBottleneck; p.N = cs;
                    p.mbuf = e->X[b]; p.ldm = Ct;
                    p.msum = fsum(e, e->st_X[b]); p.msq = fsq(e, e->st_X[b]); p.mstride = Ct;
                    p.dst = e->G[b]; p.ldd = Ct;
                    p.o1 = b1(e, e->bs_X[b]); p.o2 = b2(e, e->bs_X[b]); p.ostride = Ct; p.eps = kEps;
                    BY(e, 4.0 * NS * pl.HW * ((double)p.nseg * kBottleneck + 3.0 * cs));
                    launch_gemm(e, st, p, dim3(NS * pl.HWp / Cfg::BM, (cs + Cfg::BN - 1) / Cfg::BN), K_D1, 2.0 * NS * pl.HW * cs * kBottleneck * p.nseg);
                };
                if (pl.HWp % 128 == 0) run(CfgP128x64{}); else run(CfgP64x64{});
            }
            {   // conv1 weight gradient.  ~320 workgroups: it shares the chip with the data-gradient chain on the other
                // stream, and every workgroup ends with 128 x 64 fp32 atomics (measured: atomics beat partial tiles here)
                using Cfg = CfgW128x64;
                const int nt = (d.cin + Cfg::BN - 1) / Cfg::BN;
                int chunk, cps;
                static const int w1_target = getenv("SMG_W1_WGS") ? atoi(getenv("SMG_W1_WGS")) : 320;            // dev A/B
                pick_chunk(pl, NS, nt, chunk, cps, w1_target);   // 256..384 measure the same (22.46 ms per step), 512: 22.6, 768: 22.8, 1024: 23.2
                BwdWeightP<Cfg, W_ONE, C_IDENT, SMG_PD_WGRAD, false> p{};
                p.gbuf = D2b; p.ldg = kBottleneck; p.gcoff = 0; p.xbuf = nullptr; p.pa = pl; p.MA = kBottleneck;
                p.bbuf = e->X[b]; p.ldb = Ct; p.pb = pl; p.NB = d.cin;
                p.bsum = fsum(e, e->st_X[b]); p.bsq = fsq(e, e->st_X[b]); p.bstride = Ct;
                p.bgamma = P + d.n1.w; p.bbeta = P + d.n1.b; p.eps = kEps;
                p.chunk = chunk; p.chunks_per_stream = cps; p.n_chunks = NS * cps;
                p.dw = Gr + d.c1.w; p.ldw_out = d.cin;
                BY(e, 4.0 * NS * pl.HW * (kBottleneck + d.cin));
                launch_wgrad(e, s2, p, dim3(1, nt, NS * cps), K_W1, 2.0 * NS * pl.HW * d.cin * kBottleneck, 1, C_IDENT, e->deterministic);
                HIP_OK(hipEventRecord(e->ev_side[db], s2));
            }
        }
        if (b > 0) {   // transition b-1: X[b-1] (all channels) -> X[b][:, 0:C0]
            const Plane pp = e->p_blk[b - 1];
            const int Cp = kBlockCtot[b - 1], C0 = kBlockCin[b];
            {
                int chunk, cps;
                pick_chunk(pl, NS, (C0 / 128) * (Cp / 128), chunk, cps);
                BwdWeightP<CfgW128x128, W_POOL, C_IDENT, 1> p{};      // (one k-tile in flight: the pooling fetch holds 4 float4 per slot, three tiles of them leave one workgroup per CU)
                p.gbuf = e->G[b]; p.ldg = Ct; p.gcoff = 0; p.xbuf = e->X[b]; p.ldx = Ct; p.xcoff = 0; p.pa = pl; p.MA = C0;
                p.xsum = fsum(e, e->st_X[b]); p.xsq = fsq(e, e->st_X[b]); p.xstride = Ct;
                p.s1 = b1(e, e->bs_X[b]); p.s2 = b2(e, e->bs_X[b]); p.sstride = Ct; p.scoff = 0; p.agamma = nullptr;
                p.bbuf = e->X[b - 1]; p.ldb = Cp; p.pb = pp; p.NB = Cp;
                p.bsum = fsum(e, e->st_X[b - 1]); p.bsq = fsq(e, e->st_X[b - 1]); p.bstride = Cp;
                p.bgamma = P + T.tnorm[b - 1].w; p.bbeta = P + T.tnorm[b - 1].b; p.eps = kEps;
                p.chunk = chunk; p.chunks_per_stream = cps; p.n_chunks = NS * cps;
                p.dw = Gr + T.tconv[b - 1].w; p.ldw_out = Cp;
                if (fork(e->ev_misc)) return -5;
                BY(e, 4.0 * NS * (2.0 * pl.HW * C0 + (double)pp.HW * Cp));
                launch_wgrad(e, s2, p, dim3(C0 / 128, Cp / 128, NS * cps), K_TW, 2.0 * NS * pl.HW * Cp * C0, 1, C_IDENT);
            }
            if (pp.H != 2 * pl.H || pp.W != 2 * pl.W) {
                ProfScope ps(e, st, K_OTHER, 0);
                hipLaunchKernelGGL(zero_uncovered_kernel, dim3(256, NS), dim3(256), 0, st, e->G[b - 1], Cp, pp, 2 * pl.H, 2 * pl.W, Cp);
            }
            {
                auto run = [&](auto tag) {
                using Cfg = decltype(tag);
                BwdDataP<Cfg, false, E_UNPOOL> p{};
                p.gbuf = e->G[b]; p.ldg = Ct; p.gcoff = 0; p.xbuf = e->X[b]; p.ldx = Ct; p.xcoff = 0; p.pa = pl; p.KA = C0;
                p.xsum = fsum(e, e->st_X[b]); p.xsq = fsq(e, e->st_X[b]); p.xstride = Ct;
                p.s1 = b1(e, e->bs_X[b]); p.s2 = b2(e, e->bs_X[b]); p.sstride = Ct; p.scoff = 0; p.agamma = nullptr;
                p.wp = e->packed_u + e->pk_td[b - 1]; p.K8tot = C0 / 8; p.ldn = Cp; p.wcol0 = 0; p.N = Cp;
                p.mbuf = e->X[b - 1]; p.ldm = Cp; p.mcoff = 0; p.pm = pp;
                p.msum = fsum(e, e->st_X[b - 1]); p.msq = fsq(e, e->st_X[b - 1]); p.mstride = Cp;
                p.egamma = P + T.tnorm[b - 1].w; p.ebeta = P + T.tnorm[b - 1].b;
                p.dst = e->G[b - 1]; p.ldd = Cp; p.dcoff = 0;
                p.o1 = b1(e, e->bs_X[b - 1]); p.o2 = b2(e, e->bs_X[b - 1]); p.ostride = Cp; p.ocoff = 0;
                p.dbeta = Gr + T.tnorm[b - 1].b; p.dgamma = Gr + T.tnorm[b - 1].w; p.eps = kEps;
                BY(e, 4.0 * NS * (2.0 * pl.HW * C0 + 2.0 * pp.HW * Cp));
                launch_gemm(e, st, p, dim3(NS * pl.HWp / Cfg::BM, Cp / Cfg::BN), K_TD, 2.0 * NS * pl.HW * Cp * C0);
            };
            run(CfgP64x128{});   // the 128-row variant of the unpool epilogue spills registers
            }
        }
    }
    e->prof_stage = -1;
    {   // pool0 / relu0 backward + norm0 sums
        Pool0BwdArgs a;
        a.G1 = e->G[0]; a.X1 = e->X[0]; a.ld1 = kBlockCtot[0]; a.p1 = e->p_blk[0];
        a.xsum = fsum(e, e->st_X[0]); a.xsq = fsq(e, e->st_X[0]); a.xstride = kBlockCtot[0];
        a.SA = b1(e, e->bs_X[0]); a.SB = b2(e, e->bs_X[0]); a.sstride = kBlockCtot[0];
        a.argmax = e->argmax; a.stem = e->stem; a.ps = e->p_stem;
        a.ssum = fsum(e, e->st_stem); a.ssq = fsq(e, e->st_stem);
        a.gamma = P + T.norm0.w; a.beta = P + T.norm0.b; a.eps = kEps;
        a.DY0 = e->DY0; a.o1 = b1(e, e->bs_stem); a.o2 = b2(e, e->bs_stem);
        a.dbeta = Gr + T.norm0.b; a.dgamma = Gr + T.norm0.w;
        ProfScope ps(e, st, K_OTHER, 0);
        if (e->p_stem.H % 8 || e->p_stem.W % 8) return fail(-22, "stem plane must tile by 8 (input_size multiple of 16)");
        a.tiles_per_wg = 8;          // 4..20 measure the same; 1 costs 0.7 ms per step in atomics
        const int n_t = (e->p_stem.H / 8) * (e->p_stem.W / 8);
        hipLaunchKernelGGL(pool0_bwd_kernel, dim3((n_t + a.tiles_per_wg - 1) / a.tiles_per_wg, NS), dim3(256), 0, st, a);
    }
    {   // conv0 weight gradient (no data gradient: the image needs none)
        const Plane ps_ = e->p_stem;
        int chunk, cps;
        pick_chunk(ps_, NS, 1, chunk, cps);
        BwdWeightP<CfgW64x256, W_STEM, C_STEM> p{};
        p.gbuf = e->DY0; p.ldg = 64; p.gcoff = 0; p.xbuf = e->stem; p.ldx = 64; p.xcoff = 0; p.pa = ps_; p.MA = 64;
        p.xsum = fsum(e, e->st_stem); p.xsq = fsq(e, e->st_stem); p.xstride = 64;
        p.s1 = b1(e, e->bs_stem); p.s2 = b2(e, e->bs_stem); p.sstride = 64; p.scoff = 0; p.agamma = P + T.norm0.w;
        p.bbuf = e->img4; p.ldb = 4; p.pb = e->p_img; p.NB = 196;
        p.eps = kEps; p.chunk = chunk; p.chunks_per_stream = cps; p.n_chunks = NS * cps;
        p.dw = Gr + T.conv0.w; p.ldw_out = 147;
        if (fork(e->ev_misc)) return -5;
        BY(e, 4.0 * NS * (2.0 * ps_.HW * 64 + (double)e->p_img.HW * 4));
        launch_wgrad(e, s2, p, dim3(1, 1, NS * cps), K_SW, 2.0 * NS * ps_.HW * 64 * 147, 1, C_STEM);
    }
    HIP_OK(hipEventRecord(e->ev_end, s2));          // join: everything after the backward sees every gradient
    HIP_OK(hipStreamWaitEvent(st, e->ev_end, 0));
    HIP_OK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------
extern "C" {

const char* smg_last_error(void) { return g_err.c_str(); }
int smg_version(void) { return SMG_ABI_VERSION; }
int smg_abi_struct_bytes(int which) {
    if (which == 0) return (int)sizeof(smg_batch);
    if (which == 1) return (int)sizeof(smg_net);
    return fail(-22, "smg_abi_struct_bytes: 0 = smg_batch, 1 = smg_net");
}

int smg_layout_count(int head_out) { return (int)layout_for(head_out).entries.size(); }
int64_t smg_layout_param_floats(int head_out) { return layout_for(head_out).n_params; }
int64_t smg_layout_buffer_floats(int head_out) { return layout_for(head_out).n_bufs; }
int64_t smg_layout_nbt_count(int head_out) { return layout_for(head_out).n_nbt; }
int smg_layout_entry(int head_out, int index, char* name, int name_cap, int* kind, int64_t* offset, int* ndim, int64_t shape[4]) {
    const Layout& L = layout_for(head_out);
    if (index < 0 || index >= (int)L.entries.size()) return fail(-22, "layout index out of range");
    const LayoutEntry& en = L.entries[index];
    if (name && name_cap > 0) { std::strncpy(name, en.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (kind) *kind = en.kind;
    if (offset) *offset = en.offset;
    if (ndim) *ndim = en.ndim;
    if (shape) for (int i = 0; i < 4; ++i) shape[i] = en.shape[i];
    return 0;
}
int smg_layout_trunk_range(int head_out, int trunk_id, int64_t* offset, int64_t* count) {
    if (trunk_id < 0 || trunk_id > 2) return fail(-22, "trunk_id");
    const TrunkRef& T = layout_for(head_out).trunk[trunk_id];
    *offset = T.p_begin; *count = T.p_feat_end - T.p_begin; return 0;
}
int smg_layout_head_range(int head_out, int head_id, int64_t* offset, int64_t* count) {
    if (head_id < 0 || head_id > 2) return fail(-22, "head_id");
    const HeadRef& H = layout_for(head_out).head[head_id];
    *offset = H.p_begin; *count = H.p_end - H.p_begin; return 0;
}

int smg_engine_create(int device, int input_size, int max_streams, int max_pairs, int head_out, smg_engine** out) {
    if (!out) return fail(-22, "out is NULL");
    if (head_out != 1 && head_out != 3) return fail(-22, "head_out must be 1 or 3");
    if (input_size < 640 || max_streams < 1 || max_pairs < 1) return fail(-22, "bad engine dimensions");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(-19, "no HIP device available: the affordance engine needs an MI355X");
    if (device < 0 || device >= ndev) return fail(-22, "device index out of range");
    HIP_OK(hipSetDevice(device));
    smg_engine* e = new smg_engine();
    e->device = device; e->S = input_size; e->max_streams = max_streams; e->max_pairs = max_pairs; e->head_out = head_out;
    e->L = &layout_for(head_out);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) e->n_cu = cus;
    }
    int r = engine_build(e);
    if (r) { smg_engine_destroy(e); return r; }
    *out = e;
    return 0;
}

void smg_engine_destroy(smg_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    void* ptrs[] = {e->img4, e->stem, e->DY0, e->argmax, e->X[0], e->X[1], e->X[2], e->X[3], e->G[0], e->G[1], e->G[2], e->G[3],
                    e->Bt, e->part, e->F, e->DF, e->H1, e->DH1, e->fstat, e->bstat, e->packed_u, e->packed_f, e->stab, e->d_pack, e->d_bnupd,
                    e->d_stage};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (int k = 0; k < 2; ++k) { if (e->h_stage[k]) (void)hipHostFree(e->h_stage[k]); if (e->ev_stage[k]) (void)hipEventDestroy(e->ev_stage[k]); }
    for (int k = 0; k < kRing; ++k) {
        if (e->D2[k]) (void)hipFree(e->D2[k]);
        if (e->GS[k]) (void)hipFree(e->GS[k]);
        if (e->ev_gs[k]) (void)hipEventDestroy(e->ev_gs[k]); if (e->ev_d2[k]) (void)hipEventDestroy(e->ev_d2[k]); if (e->ev_side[k]) (void)hipEventDestroy(e->ev_side[k]);
    }
    if (e->ev_misc) (void)hipEventDestroy(e->ev_misc);
    if (e->ev_end) (void)hipEventDestroy(e->ev_end);
    if (e->side) (void)hipStreamDestroy(e->side);
    for (auto& r : e->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
    delete e;
}

int64_t smg_engine_workspace_bytes(const smg_engine* e) { return e ? e->workspace_bytes : 0; }

int smg_engine_geometry(const smg_engine* e, int H[6], int HWp[6]) {
    if (!e) return fail(-22, "engine is NULL");
    const Plane* pl[6] = {&e->p_img, &e->p_stem, &e->p_blk[0], &e->p_blk[1], &e->p_blk[2], &e->p_blk[3]};
    for (int i = 0; i < 6; ++i) { H[i] = pl[i]->H; HWp[i] = pl[i]->HWp; }
    return 0;
}

int smg_forward(smg_engine* e, const smg_net* net, int trunk_id, int head_id, const smg_batch* batch, float* q_out_dev, void* stream) {
    if (!e || !net || !batch || !q_out_dev) return fail(-22, "NULL argument");
    if (!net->params || !net->bufs || !net->nbt) return fail(-22, "net arrays are NULL");
    if (trunk_id < 0 || trunk_id > 2 || head_id < 0 || head_id > 2) return fail(-22, "trunk_id / head_id out of range");
    HIP_OK(hipSetDevice(e->device));
    return do_forward(e, net, trunk_id, head_id, batch, q_out_dev, (hipStream_t)stream);
}

int smg_loss(smg_engine* e, int mode, const float* q_dev, const float* labels_dev, int n_pairs, float* loss_dev, float* dq_dev, void* stream) {
    if (!e || !q_dev || !labels_dev || !loss_dev || !dq_dev) return fail(-22, "NULL argument");
    if (mode == 1 && e->head_out != 3) return fail(-22, "cross-entropy loss needs a 3-class head");
    if (mode != 0 && mode != 1) return fail(-22, "loss mode must be 0 (Huber) or 1 (cross entropy)");
    if (n_pairs < 1 || n_pairs > e->max_pairs) return fail(-22, "n_pairs exceeds the engine's max_pairs");
    HIP_OK(hipSetDevice(e->device));
    const int per_pair = e->head_out * e->OH * e->OW;
    hipLaunchKernelGGL(loss_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, (hipStream_t)stream, mode, q_dev, labels_dev, n_pairs, per_pair, loss_dev, dq_dev);
    HIP_OK(hipGetLastError());
    return 0;
}

int smg_backward(smg_engine* e, const smg_net* net, const float* dq_dev, void* stream) {
    if (!e || !net || !dq_dev) return fail(-22, "NULL argument");
    HIP_OK(hipSetDevice(e->device));
    return do_backward(e, net, dq_dev, (hipStream_t)stream);
}

int smg_engine_set_precision(smg_engine* e, int precision) {
    if (!e) return fail(-22, "engine is NULL");
    if (precision < 0 || precision > 2) return fail(-22, "precision must be 0 (fp32-class split), 1 (bf16 operands) or 2 (fp16 operands)");
    e->prec = precision;
    e->have_fwd = false;        // activations saved by a forward of another precision are not backward-compatible
    return 0;
}

int smg_engine_set_option(smg_engine* e, const char* name, int value) {
    if (!e || !name) return fail(-22, "NULL argument");
    const std::string s(name);
    if (s == "deterministic") { e->deterministic = value != 0; return 0; }
    if (s == "serialize") { e->serialize = value != 0; return 0; }
    return fail(-22, "unknown engine option '" + s + "'");
}

int smg_heightmap(const double* depth_img_dev, int h, int w, const double* intrinsics3x3, const double* cam_pose4x4,
                  const double* inv_homography3x3, int out_w, int out_h, double* out_dev, void* stream) {
    if (!depth_img_dev || !intrinsics3x3 || !cam_pose4x4 || !inv_homography3x3 || !out_dev || h < 1 || w < 1 || out_w < 1 || out_h < 1)
        return fail(-22, "bad heightmap arguments");
    hipPointerAttribute_t attr;
    HIP_OK(hipPointerGetAttributes(&attr, depth_img_dev));
    HIP_OK(hipSetDevice(attr.device));
    HeightmapArgs a;
    a.depth = depth_img_dev; a.h = h; a.w = w;
    a.fx = intrinsics3x3[0]; a.fy = intrinsics3x3[4]; a.cx = intrinsics3x3[2]; a.cy = intrinsics3x3[5];
    a.r20 = cam_pose4x4[8]; a.r21 = cam_pose4x4[9]; a.r22 = cam_pose4x4[10]; a.t2 = cam_pose4x4[11];
    for (int i = 0; i < 9; ++i) a.mi[i] = inv_homography3x3[i];
    a.out = out_dev; a.ow = out_w; a.oh = out_h;
    hipLaunchKernelGGL(heightmap_warp_kernel, dim3((out_w + 255) / 256, out_h), dim3(256), 0, (hipStream_t)stream, a);
    HIP_OK(hipGetLastError());
    return 0;
}

int smg_argmax(const float* values_dev, int n, int* idx_out_dev, float* val_out_dev, void* stream) {
    if (!values_dev || !idx_out_dev || !val_out_dev || n < 1) return fail(-22, "bad argmax arguments");
    hipPointerAttribute_t attr;
    HIP_OK(hipPointerGetAttributes(&attr, values_dev));
    HIP_OK(hipSetDevice(attr.device));
    hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, values_dev, n, idx_out_dev, val_out_dev);
    HIP_OK(hipGetLastError());
    return 0;
}

int smg_adam_step(float* params, const float* grads, float* m, float* v, int64_t offset, int64_t count, int step, float lr,
                  float beta1, float beta2, float eps, void* stream) {
    if (!params || !grads || !m || !v || count < 0 || step < 1) return fail(-22, "bad Adam arguments");
    const double bc1 = 1.0 - std::pow((double)beta1, step), bc2 = 1.0 - std::pow((double)beta2, step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)std::sqrt(bc2);
    if (count == 0) return 0;
    {   // no engine argument: launch on the device that owns the parameter array
        hipPointerAttribute_t attr;
        HIP_OK(hipPointerGetAttributes(&attr, params));
        HIP_OK(hipSetDevice(attr.device));
    }
    int blocks = (int)((count + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params + offset, grads + offset, m + offset, v + offset,
                       count, step_size, beta1, beta2, eps, 1.0f, bc2_sqrt);
    HIP_OK(hipGetLastError());
    return 0;
}

int64_t smg_debug_read(smg_engine* e, const char* name, float* host_out, int64_t cap, void* stream) {
    if (!e || !name) return fail(-22, "NULL argument");
    const int NS = e->max_streams, NP = e->max_pairs;
    const float* src = nullptr; int64_t n = 0;
    std::string s(name);
    if (s == "img") { src = e->img4; n = (int64_t)NS * e->p_img.HWp * 4; }
    else if (s == "stem") { src = e->stem; n = (int64_t)NS * e->p_stem.HWp * 64; }
    else if (s == "dy0") { src = e->DY0; n = (int64_t)NS * e->p_stem.HWp * 64; }
    else if (s == "feat") { src = e->F; n = (int64_t)NP * e->p_blk[3].HWp * 2 * kFeat; }
    else if (s == "h1") { src = e->H1; n = (int64_t)NP * e->p_blk[3].HWp * kHeadMid; }
    else if (s == "df") { src = e->DF; n = (int64_t)NP * e->p_blk[3].HWp * 2 * kFeat; }
    else if (s == "dh1") { src = e->DH1; n = (int64_t)NP * e->p_blk[3].HWp * kHeadMid; }
    else if (s.size() == 2 && (s[0] == 'x' || s[0] == 'g') && s[1] >= '1' && s[1] <= '4') {
        const int b = s[1] - '1';
        src = s[0] == 'x' ? e->X[b] : e->G[b]; n = (int64_t)NS * e->p_blk[b].HWp * kBlockCtot[b];
    } else if (s.size() >= 4 && s.substr(0, 2) == "bt") {   // "bt<block>_<layer>" 1-based
        int b = 0, i = 0;
        if (std::sscanf(name, "bt%d_%d", &b, &i) != 2 || b < 1 || b > 4 || i < 1 || i > kBlockLayers[b - 1]) return fail(-22, "bad bt name");
        src = e->Bt + e->bt_off[b - 1][i - 1]; n = (int64_t)NS * e->p_blk[b - 1].HWp * kBottleneck;
    } else return fail(-22, "unknown debug buffer");
    if (!host_out) return n;
    if (cap < n) n = cap;
    HIP_OK(hipSetDevice(e->device));
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    HIP_OK(hipMemcpy(host_out, src, n * sizeof(float), hipMemcpyDeviceToHost));
    return n;
}

int smg_profile_enable(smg_engine* e, int on) {
    if (!e) return fail(-22, "engine is NULL");
    e->prof = on != 0;
    for (auto& r : e->recs) { e->ev_pool.push_back(r.a); e->ev_pool.push_back(r.b); }
    e->recs.clear();
    for (int s = 0; s < 5; ++s)
        for (int k = 0; k < K_COUNT; ++k) { e->prof_ms[s][k] = 0; e->prof_n[s][k] = 0; e->prof_flops[s][k] = 0; e->prof_bytes[s][k] = 0; }
    return 0;
}

int smg_profile_kinds(void) { return K_COUNT; }
const char* smg_profile_kind_name(int kind) { return (kind >= 0 && kind < K_COUNT) ? kKindNames[kind] : ""; }

// Drains the recorded events (synchronises them) and returns the totals of one kind.
int smg_profile_read(smg_engine* e, int kind, double* ms, int64_t* launches, double* flops) {
    if (!e || kind < 0 || kind >= 5 * K_COUNT) return fail(-22, "bad profile query");
    for (auto& r : e->recs) {
        float t = 0.f;
        (void)hipEventSynchronize(r.b);
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            const int slots[2] = {0, r.stage + 1};
            for (int q = 0; q < (r.stage >= 0 ? 2 : 1); ++q) {
                const int s = slots[q];
                e->prof_ms[s][r.kind] += t; e->prof_n[s][r.kind] += 1; e->prof_flops[s][r.kind] += r.flops; e->prof_bytes[s][r.kind] += r.bytes;
            }
        }
        e->ev_pool.push_back(r.a); e->ev_pool.push_back(r.b);
    }
    e->recs.clear();
    const int slot = kind / K_COUNT; kind %= K_COUNT;
    if (ms) *ms = e->prof_ms[slot][kind];
    if (launches) *launches = e->prof_n[slot][kind];
    if (flops) *flops = e->prof_flops[slot][kind];
    return 0;
}

// Algorithmic HBM bytes (see BY()) accumulated for one class since smg_profile_enable; call after smg_profile_read.
int smg_profile_read_bytes(smg_engine* e, int kind, double* bytes) {
    if (!e || kind < 0 || kind >= 5 * K_COUNT || !bytes) return fail(-22, "bad profile query");
    *bytes = e->prof_bytes[kind / K_COUNT][kind % K_COUNT];
    return 0;
}

}  // extern "C"
