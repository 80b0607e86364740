// halo.cuh - 3x3 convolutions of the dense layers with an LDS-resident input halo.
//
// The generic implicit GEMM (gemm.cuh, F_THREE) re-fetches every input pixel once per
// tap: with only growth = 32 output channels per layer that is ~20 B/clk/CU of
// global->LDS traffic at full MFMA rate, and the kernel saturates the load path at a
// third of the fp32 MFMA peak.  Here a workgroup owns a TSxTS tile of output pixels,
// stages the (TS+2)^2 halo of BN+ReLU'd input ONCE per 16-channel chunk, and walks the
// 9 taps by shifting the LDS read address: 9x fewer global loads, BN transforms and
// LDS writes per MFMA.
//
//   out[p][n] = sum_{tap, c} relu(bn(in[p + d(tap)][c])) * W[n][c][tap]
//
// Arithmetic and LDS images: the split-precision schemes of gemm.cuh - in the fp32-class mode operand kind 3 (two scaled fp16
// pieces per fp32 operand, three v_mfma_f32_32x32x16_f16 per product); kind 0 (three bf16 pieces per fp32 operand, six
// v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate; 16-byte units of 8 consecutive k).
//
// Two tile sizes:
//   TS = 16  planes with enough 16 x 16 tiles to fill the chip (160^2, 80^2 at S = 640; 456^2, 228^2, 114^2 at S = 1824 - tiles
//            may hang over the edge, masked).  4 waves, wave w owns pixel rows 4w..4w+3 (two 32x32 MFMA tiles: 2 rows x 16 cols each).
//   TS = 8   everything else (40^2, 20^2): 64 pixels per workgroup so
//            that a 17-stream launch still has hundreds of workgroups.  Two waves split
//            the pixels (4 rows x 8 cols each, one MFMA tile); the other factor of two
//            splits the taps (forward, folded through LDS at the end), the output channels
//            (data gradient) or the pixels again (weight gradient, 16 x 8 / 8 x 8 tiles).
#pragma once
#include "gemm.cuh"

// waves per SIMD the register allocator is held to (512 / n registers per lane); three for the 16 x 16 forward: 168 VGPRs with 128 bytes
// of scratch, 64 -> 111 us per launch
namespace smg { constexpr int kHaloFwdWaves = 2, kHaloWgradWaves = 2; }
#ifndef SMG_HALO_REGFRAG
#define SMG_HALO_REGFRAG 1      // dev A/B: 0 = every activation fragment of the TS = 16 forward / data gradient read from LDS (rounds 2-5)
#endif
#ifndef SMG_D3_ST8
#define SMG_D3_ST8 1            // dev A/B: 3 = the 8 x 8 data gradient stages a whole kernel row's weights (rounds 4-5: 65 KB of LDS, two workgroups per CU)
#endif
#ifndef SMG_HALO_DMA
#define SMG_HALO_DMA 1          // dev A/B: 0 = the TS = 16 data gradient's weights register-staged through two LDS buffers, one stage ahead (rounds 2-5)
#endif

namespace smg {

template <int TS, int TH_ = TS>
struct HaloGeo {
    static_assert(TS == 16 || TS == 8, "tile side");
    static_assert(TH_ == TS || (TS == 16 && TH_ == 8), "tile height: square, or 16 wide x 8 high");
    static constexpr int T = TS, TH = TH_, W = TS + 2, HH = TH + 2, PX = W * HH, NPIX = TS * TH;
    static constexpr int MT = NPIX == 256 ? 2 : 1;     // 32-pixel MFMA tiles per wave
    static constexpr int WQ = NPIX / (32 * MT);        // waves that split the pixels (4 / 2)
    static constexpr int WX = 4 / WQ;                  // the other wave factor (1 / 2)
    static constexpr int ROWS = 32 / TS;               // pixel rows per MFMA tile (2 / 4)
    static constexpr int RP = MT + 2;                  // row pairs a wave's register-resident fragments span (rowi): 4 (16 x 16) / 3 (16 x 8)
    static constexpr int RSTEP = TS == 16 ? (TH == 16 ? 2 : 4) : 0;      // distance of the two rows of a pair
    // pixel (row, col) inside the tile of MFMA-tile row i (0..31) of tile m of pixel-wave wq
    __device__ static __forceinline__ int row(int wq, int m, int i) { return (wq * MT + m) * ROWS + i / TS; }
    __device__ static __forceinline__ int col(int i) { return i % TS; }
    // TS = 16, kernels that derive their activation fragments in registers (round 6): the two pixel rows of a wave's tile m are rows
    // 4 wq + m and 4 wq + m + 2 - INTERLEAVED with the other tile's - so that the fragment of kernel row dy of tile m, input rows
    // (4 wq + m + dy, 4 wq + m + dy + 2), is the row pair i = m + dy of FOUR pairs (i = 0..3) that serve all six (tile, kernel row)
    // combinations of the wave (consecutive rows would need six).  16 x 8 tiles (one MFMA tile per wave): rows wq and wq + 4, pairs
    // i = dy of THREE.
    __device__ static __forceinline__ int rowi(int wq, int m, int i) {
        return TS == 16 ? (TH == 16 ? 4 * wq + m + 2 * (i / TS) : wq + 4 * (i / TS)) : row(wq, m, i);
    }
    __device__ static __forceinline__ int pair0(int wq) { return TH == 16 ? 4 * wq : wq; }      // first halo row of the wave's pair 0
};

// compile-time loop: f(integral_constant<int, B>) ... f(integral_constant<int, N - 1>) (DPP controls must be immediates)
template <int B, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (B < N) { f(std::integral_constant<int, B>{}); sfor<B + 1, N>(f); }
}

// Fragment of kernel column DX (1 or 2) of row pair I, derived from the pair's column-0 fragment `p` (lane c of a 16-lane DPP row =
// halo column c of one pixel row, one k-half) and the wave's edge fragment `e` (lanes 2 I, 2 I + 1 of every DPP row = halo columns
// 16, 17 of pair I): a row shift left by DX whose vacated lanes 16 - DX .. 15 keep what the first move put there - the edge columns,
// shifted right into place.  Two v_mov_b32_dpp per dword instead of one ds_read_b128 per fragment: the taps of a 16-channel chunk
// read 10 activation fragments from LDS instead of 36 (semantics pinned on the hardware with tools/dpp_probe.hip).
template <int DX, int I>
__device__ __forceinline__ u32x4 dpp_col_shift(const u32x4& p, const u32x4& e) {
    if constexpr (DX == 0) return p;
    else {
        u32x4 r;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int edge = __builtin_amdgcn_update_dpp(0, (int)e[d], 0x110 + (16 - DX - 2 * I), 0xF, 0xF, false);      // row_shr
            r[d] = (unsigned)__builtin_amdgcn_update_dpp(edge, (int)p[d], 0x100 + DX, 0xF, 0xF, false);                   // row_shl, bound_ctrl off
        }
        return r;
    }
}

struct Halo3x3FwdArgs {
    const void* src; int lds_; Plane pl;            // [n][HWp][C] raw bottleneck output (fp32 / bf16 / fp16 by mode)
    int C;                                          // input channels (128)
    const double* ssum; const double* ssq; int sstride;     // fp64 statistics of src (norm2 input)
    const float* gamma; const float* beta; float eps;
    float* tw_mean; float* tw_invstd;               // [n][C]: the first tile of every stream stores mean / invstd for the backward
    const u32x4* wu;                                // weight units [chunk][piece][tap][k8][n] (PK_HF); operand kind 3: header unit in front
    const float* asc;                               // operand kind 3: {s, 1 / s} of the BN + ReLU operand
    void* dst; int ldd, dcoff;
    double* dsum; double* dsq; int dstride;
    int tiles_x, n_tiles, streams;                   // 1-D grid of banded_grid(n_tiles, streams) workgroups (banded_tile)
};

// Channels per chunk of the forward: one k16-step per tap for the fp32-class split (three pieces per operand), two for the
// single-piece 16-bit modes (the same 16-byte staging slots per halo pixel: 4 x 4 fp32 or 4 x 8 16-bit channels).
constexpr int halo_ck(int prec) { return prec ? 32 : 16; }

// ------------------------------------------------------------------------------------
// Forward.  Per chunk the halo lands in LDS as units [piece][k8][halo pixel] of 16 bytes (BN + ReLU + split / 16-bit pack
// applied once, at the store), the chunk's weights as units [piece][tap][k8][n] - pack_weights_kernel writes exactly that
// image per chunk (PK_HF), so their staging is a copy.  A fragment is one ds_read_b128 per piece: lanes 0..15 / 16..31 read
// two runs of 16 consecutive halo pixels.  The 128 BN parameters of the stream are derived from the fp64 sums in the
// prologue (one channel per thread) and stored once per stream for the backward kernels.
//   TS = 16: 4 waves x 2 pixel tiles, 43 KB LDS (2 workgroups per CU by registers).
//   TS = 8 : 2 waves split the pixels, the other factor of two splits the TAPS (5 + 4).
// One (halo, weights) buffer: the next chunk is transformed and stored between two barriers while the workgroup's MFMAs pause, and
// the CU's second workgroup fills the pause.  Per-workgroup stamps of round 5 (TS = 16, 17 streams of 160^2: 1700 workgroups, 3.3
// rounds of 2 per CU, 34 us of life each): prologue 9.8k cycles (fp64 moments of the 128 channels, first chunk staged), then per
// 16-channel chunk 4.6k for the nine taps (54 MFMAs per wave = 1.7k cycles of matrix pipe, shared with the CU's other workgroup)
// and 3.2k for barrier + BN / ReLU / split / store of the next chunk + barrier, 3k of epilogue - of 66k.  TS = 8: 4.1k + 8 x 1.9k of 19k.  (Round 5 measured the alternative - TWO buffers, the next chunk's conversions and LDS
// stores dealt out one slice per tap into the shadow of that tap's MFMAs, one barrier per chunk, 84 KB and therefore ONE workgroup
// per CU at TS = 16: 65.8 -> 93.8 us per launch on the 160^2 / 80^2 planes, 16.8 -> 16.4 us at TS = 8 - rejected.)
// ------------------------------------------------------------------------------------
template <int TS, int PREC> struct HaloFwdSGeo : HaloGeo<TS> {
    using G = HaloGeo<TS>;
    static constexpr int NP = np_of(fwd_op(PREC)), CK = halo_ck(PREC), K8C = CK / 8;
    static constexpr int LDH = G::PX;
    static constexpr int A_UNITS = NP * K8C * LDH;
    static constexpr int A_N = (G::PX * 4 + 255) / 256;                  // 16-byte slots per thread: 6 / 2
    static constexpr int BU = NP * 9 * K8C * 32;                         // weight units per chunk: 1728 / 1152
    static constexpr int B_N = (BU + 255) / 256;                         // (the last copies of a round read / write padding)
    __host__ __device__ static constexpr int smem_bytes(int C) { return (A_UNITS + B_N * 256) * 16 + 3 * C * 4; }
    __host__ __device__ static constexpr int smem_bytes_ws(int C) { return 2 * (A_UNITS + BU) * 16 + 3 * C * 4; }      // wave-specialised form: two buffers
};

// RAG: tiles may hang over the plane's edge (stores, statistics and mask loads are bounds-checked) - always at TS = 8; at TS = 16 only
// for planes that do not tile by 16 (S = 1824), so that the exact planes of the headline keep their check-free epilogues.
// (tile, stream) of this workgroup of a 1-D grid of 8 * ceil(n_tiles * streams / 8): workgroups go to the eight XCDs round-robin by
// linear index, and each XCD walks ONE contiguous run of the (stream, tile) sequence - the halo rows and columns that neighbouring
// tiles share are read from HBM once and from that XCD's L2 after (round 5: 141 -> 113 MB fetched per launch at TS = 16, 16.2 -> 11.4
// at TS = 8).
__device__ __forceinline__ bool banded_tile(int n_tiles, int streams, int& tile, int& n) {
    const int total = n_tiles * streams, per = (total + 7) / 8;
    const int w = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || w >= total) return false;
    n = w / n_tiles;
    tile = w - n * n_tiles;
    return true;
}
static inline unsigned banded_grid(int n_tiles, int streams) { return 8u * (unsigned)((n_tiles * streams + 7) / 8); }

// WS (wave-specialised form, 512 threads): waves 0..3 are the kernel's four matrix waves and do nothing but fragment reads and
// MFMAs; waves 4..7 fetch, transform and store the NEXT chunk into a second (halo, weights) buffer meanwhile - one barrier per chunk.
// Measured on the plain kernel (round 5, per launch at TS = 16 / 8): 67.5 / 17.3 us as built, 38.7 / 12.1 with the loads and the
// transform removed, 37.1 / 12.2 with the taps removed - its two halves run one after the other even at two workgroups per CU.
template <int TS, int PREC = 0, bool RAG = false, bool WS = false>
static __global__ __launch_bounds__(WS ? 512 : 256, WS ? 4 : kHaloFwdWaves) void conv3x3_halo_fwd_kernel(const Halo3x3FwdArgs a) {
    constexpr bool kEdge = TS == 8 || RAG;
    using G = HaloFwdSGeo<TS, PREC>;
    using ST = act_t<PREC>;
    constexpr int OP = fwd_op(PREC), NP = G::NP, CK = G::CK, K8C = G::K8C, KSTEP = CK / 16;
    constexpr int MT = G::MT, A_N = G::A_N, B_N = G::B_N, LDH = G::LDH, ESZ = ST::size, E = 16 / ESZ;
    constexpr int BUF_BYTES = WS ? (G::A_UNITS + G::BU) * 16 : 0;       // WS: two (halo, weights) buffers, the weights exactly BU units
    constexpr bool kRegFrag = TS == 16 && OP != 0 && SMG_HALO_REGFRAG;                       // activation fragments of a chunk from four row pairs + DPP column shifts (taps16 below)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* As = reinterpret_cast<char*>(smem);                // [piece][k8][LDH] units          (WS: of the buffer being read)
    char* Bs = As + G::A_UNITS * 16;                         // [piece][tap][k8][32] units
    float* prm = reinterpret_cast<float*>(WS ? reinterpret_cast<char*>(smem) + 2 * BUF_BYTES : Bs + B_N * 256 * 16);      // mean | scale | beta, C each
    const int role = WS ? (int)(threadIdx.x >> 8) : 0;       // WS: 0 = matrix waves, 1 = staging waves
    const int t = WS ? (int)(threadIdx.x & 255) : (int)threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, half = lane >> 5;
    const int wq = wave % G::WQ, wk = wave / G::WQ;          // pixel slice, tap slice
    const int tap0 = (G::WX > 1 && wk) ? 5 : 0, tap1 = (G::WX > 1 && !wk) ? 5 : 9;
    int n, tile;
    if (!banded_tile(a.n_tiles, a.streams, tile, n)) return;
    // dev stamps (SMG_TRACE_KIND=2): start | prologue (parameters, first chunk staged) | first chunk's taps | chunk loop | epilogue
    // (instrumentation build -DSMG_TRACE_FWD3 only: round 5 measured 10 % for live stamps in this kernel)
#ifdef SMG_TRACE_FWD3
    unsigned long long* trace = (g_smg_trace && threadIdx.x == 0) ? g_smg_trace + 8 * (size_t)blockIdx.x : nullptr;
    if (trace) { trace[0] = smg_stamp(); trace[5] = __builtin_amdgcn_s_memrealtime(); }
#else
    constexpr unsigned long long* trace = nullptr;
#endif
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int y0 = ty * TS, x0 = tx * TS;
    const int C = a.C, kq = t & 3;                           // this thread's 16-byte slot inside every chunk (E channels)
    const float sa = OP == 3 ? a.asc[0] : 1.f;               // operand kind 3: the activation scale rides on gamma * invstd and beta

    int a_off[A_N];       // global pixel index, -1 = outside the image / unused slot
    int a_hp[A_N];        // halo pixel of the slot, -1 = unused
#pragma unroll
    for (int i = 0; i < A_N; ++i) {
        const int idx = t + 256 * i;
        const int hp = idx >> 2;
        const int hy = hp / G::W, hx = hp - hy * G::W;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = idx < G::PX * 4 && (unsigned)iy < (unsigned)a.pl.H && (unsigned)ix < (unsigned)a.pl.W;
        a_off[i] = ok ? (iy * a.pl.W + ix) : -1;
        a_hp[i] = (idx < G::PX * 4) ? hp : -1;
    }
    const char* src_n = static_cast<const char*>(a.src) + (int64_t)ESZ * n * a.pl.HWp * a.lds_;
    const u32x4* wu = a.wu;
    constexpr int NSET = 1;              // register sets of loads in flight (WS with two sets, loads two chunks ahead: 63.2 -> 72.8 us per launch at TS = 16 - the
                                         // 128-register budget of the 16-wave CU is gone - and 15.3 -> 15.2 at TS = 8: one chunk of taps covers the loads)
    float4 ra_[NSET][A_N]; u32x4 rb_[NSET][B_N];
    using Set0 = std::integral_constant<int, 0>;
    auto g_load = [&](int chunk, auto SET) {
        float4 (&ra)[A_N] = ra_[decltype(SET)::value]; u32x4 (&rb)[B_N] = rb_[decltype(SET)::value];
        const int c0 = chunk * CK + E * kq;
#pragma unroll
        for (int i = 0; i < A_N; ++i)                               // unconditional loads from clamped addresses
            ra[i] = ld16(src_n, (int64_t)ESZ * ((int64_t)(a_off[i] < 0 ? 0 : a_off[i]) * a.lds_ + c0));
#pragma unroll
        for (int i = 0; i < B_N; ++i) rb[i] = wu[(int64_t)chunk * G::BU + t + 256 * i];      // (slack behind the packed array)
    };
    auto s_store = [&](int chunk, char* As, char* Bs, auto SET) {      // (WS: the buffer being filled)
        const float4 (&ra)[A_N] = ra_[decltype(SET)::value]; const u32x4 (&rb)[B_N] = rb_[decltype(SET)::value];
        const float* pq = prm + chunk * CK + E * kq;
#pragma unroll
        for (int i = 0; i < A_N; ++i) {
            if (a_hp[i] < 0) continue;
            if constexpr (E == 4) {
                const Split4 s = split4<OP>(a_off[i] >= 0 ? bnrelu4(ra[i], pq, C) : zero4());   // conv zero padding applies AFTER bn+relu
#pragma unroll
                for (int pc = 0; pc < NP; ++pc)
                    *reinterpret_cast<uint2*>(As + ((pc * K8C + (kq >> 1)) * LDH + a_hp[i]) * 16 + (kq & 1) * 8) = s.p[pc];
            } else {
                const bool in = a_off[i] >= 0;
                const float4 lo = in ? bnrelu4(slot_quad<ST>(ra[i], 0), pq, C) : zero4();
                const float4 hi = in ? bnrelu4(slot_quad<ST>(ra[i], 1), pq + 4, C) : zero4();
                *reinterpret_cast<u32x4*>(As + (kq * LDH + a_hp[i]) * 16) = pack_unit<OP>(lo, hi);
            }
        }
#pragma unroll
        for (int i = 0; i < B_N; ++i)
            if (!WS || t + 256 * i < G::BU) *reinterpret_cast<u32x4*>(Bs + (t + 256 * i) * 16) = rb[i];
    };

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    int abase[MT];        // halo pixel of this lane's tile row at tap (0, 0)
#pragma unroll
    for (int m = 0; m < MT; ++m) abase[m] = G::row(wq, m, l31) * G::W + G::col(l31);      // (TS = 8; TS = 16 takes its fragments from row pairs, below)
    auto fa = [&](int m, int tap, int ks, int pc) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(As + ((pc * K8C + 2 * ks + half) * LDH + abase[m] + (tap / 3) * G::W + tap % 3) * 16);
    };
    auto fb = [&](int tap, int ks, int pc) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(Bs + (((pc * 9 + tap) * K8C + 2 * ks + half) * 32 + l31) * 16);
    };

    const int NCH = C / CK;
    if (!WS || role == 1)
    g_load(0, Set0{});                       // (the first chunk's loads in front of the parameter prologue, which waits for its own fp64 sums: measured, no change -
                                     //  66.5 / 16.6 us per launch both ways; the prologue's 9.8k cycles are the moments' arithmetic and the first store)
    for (int k = threadIdx.x; k < C; k += (WS ? 512 : 256)) {            // BN parameters of this stream
        float mean, invstd;
        bn_moments(a.ssum, a.ssq, (int64_t)n * a.sstride + k, 1.0 / (double)a.pl.HW, a.eps, mean, invstd);
        prm[k] = mean;
        prm[C + k] = a.gamma[k] * invstd * sa;
        prm[2 * C + k] = a.beta[k] * sa;
        if (tile == 0) { a.tw_mean[(int64_t)n * C + k] = mean; a.tw_invstd[(int64_t)n * C + k] = invstd; }
    }
    __syncthreads();                 // prm visible
    if (!WS || role == 1) s_store(0, As, Bs, Set0{});
    if (WS && role == 1 && NCH > 1) g_load(1, Set0{});
    __syncthreads();
    // TS = 16 (round 6): the wave's two tiles are the row pairs (4 wq + m, 4 wq + m + 2).  Per chunk it reads FOUR row-pair fragments
    // (pair i = halo rows 4 wq + i and 4 wq + i + 2, halo columns 0..15) and one edge fragment (columns 16, 17 of the four pairs) per
    // piece and k16-step - 10 ds_read_b128 instead of 36 - and forms the fragment of tap (dy, dx) of tile m from pair m + dy with a
    // DPP row shift by dx (dpp_col_shift).  The weights' 18 fragment reads per chunk are unchanged.
    const int pbase = (4 * wq + 2 * (l31 >> 4)) * G::W + (l31 & 15);
    const int ebase = (4 * wq + (((lane & 15) >> 1) & 3) + 2 * (l31 >> 4)) * G::W + 16 + (lane & 1);
    auto taps16 = [&]() {
        u32x4 P[4][KSTEP][NP], Eg[KSTEP][NP], bf[2][KSTEP][NP];
#pragma unroll
        for (int ks = 0; ks < KSTEP; ++ks)
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
                const char* pl_ = As + ((pc * K8C + 2 * ks + half) * LDH) * 16;
#pragma unroll
                for (int i = 0; i < 4; ++i) P[i][ks][pc] = *reinterpret_cast<const u32x4*>(pl_ + (pbase + i * G::W) * 16);
                Eg[ks][pc] = *reinterpret_cast<const u32x4*>(pl_ + ebase * 16);
            }
        auto load_b = [&](int set, int tap) {
#pragma unroll
            for (int ks = 0; ks < KSTEP; ++ks)
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) bf[set][ks][pc] = fb(tap, ks, pc);
        };
        load_b(0, 0);
        sfor<0, 9>([&](auto T) {
            constexpr int tap = decltype(T)::value, dy = tap / 3, dx = tap % 3, set = tap & 1;
            if constexpr (tap + 1 < 9) load_b(set ^ 1, tap + 1);
            u32x4 af[MT][KSTEP][NP];
            sfor<0, MT>([&](auto M) {
                constexpr int m = decltype(M)::value;
#pragma unroll
                for (int ks = 0; ks < KSTEP; ++ks)
#pragma unroll
                    for (int pc = 0; pc < NP; ++pc) af[m][ks][pc] = dpp_col_shift<dx, m + dy>(P[m + dy][ks][pc], Eg[ks][pc]);
            });
            if constexpr (OP == 3) {                        // h*l, l*h, h*h (small terms first; tiles innermost: consecutive MFMAs never share an accumulator)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = mfma_f16(af[m][0][0], bf[set][0][1], acc[m]);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = mfma_f16(af[m][0][1], bf[set][0][0], acc[m]);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = mfma_f16(af[m][0][0], bf[set][0][0], acc[m]);
            } else {
#pragma unroll
                for (int ks = 0; ks < KSTEP; ++ks)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] = mfma_1p<OP>(af[m][ks][0], bf[set][ks][0], acc[m]);
            }
        });
    };
    auto taps_of_chunk = [&]() {
        if constexpr (kRegFrag) { taps16(); return; }
        // per tap: hi and lo pieces, the two hi x lo groups, then the mid pieces (fetched under those MFMAs) and the rest;
        // the next tap's hi / lo pieces are requested before the last four groups of this one
        u32x4 ah[2][KSTEP][MT], al[2][MT], bh[2][KSTEP], bl[2];
        auto load_hl = [&](int set, int tap) {
#pragma unroll
            for (int ks = 0; ks < KSTEP; ++ks) {
#pragma unroll
                for (int m = 0; m < MT; ++m) ah[set][ks][m] = fa(m, tap, ks, 0);
                bh[set][ks] = fb(tap, ks, 0);
            }
            if constexpr (OP == 0 || OP == 3) {             // the lowest piece: index 2 of three, 1 of two
#pragma unroll
                for (int m = 0; m < MT; ++m) al[set][m] = fa(m, tap, 0, NP - 1);
                bl[set] = fb(tap, 0, NP - 1);
            }
        };
        auto tap_body = [&](int set, int tap, bool more) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (OP == 3) {                        // two fp16 pieces: h*l, l*h, h*h; the next tap's fragments under them
                if (more) load_hl(set ^ 1, tap + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = mfma_f16(ah[set][0][m], bl[set], acc[m]);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = mfma_f16(al[set][m], bh[set][0], acc[m]);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = mfma_f16(ah[set][0][m], bh[set][0], acc[m]);
                return;
            } else
            if constexpr (OP != 0) {                        // single-piece operands: one term per k16-step
                if (more) load_hl(set ^ 1, tap + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KSTEP; ++ks)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] = mfma_1p<OP>(ah[set][ks][m], bh[set][ks], acc[m]);
                return;
            } else {
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = mfma_bf16(ah[set][0][m], bl[set], acc[m]);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = mfma_bf16(al[set][m], bh[set][0], acc[m]);
                u32x4 am[MT], bm;
#pragma unroll
                for (int m = 0; m < MT; ++m) am[m] = fa(m, tap, 0, 1);
                bm = fb(tap, 0, 1);
                if (more) load_hl(set ^ 1, tap + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = mfma_bf16(am[m], bm, acc[m]);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = mfma_bf16(ah[set][0][m], bm, acc[m]);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = mfma_bf16(am[m], bh[set][0], acc[m]);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = mfma_bf16(ah[set][0][m], bh[set][0], acc[m]);
            }
        };
        load_hl(0, tap0);
#pragma unroll
        for (int tp = 0; tp < 5; ++tp)
            if (tap0 + tp < tap1) tap_body(tp & 1, tap0 + tp, tap0 + tp + 1 < tap1);
        if constexpr (TS == 16) {      // taps 5..8 of the single tap slice
#pragma unroll
            for (int tp = 5; tp < 9; ++tp) tap_body(tp & 1, tp, tp + 1 < 9);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (WS) {
        // each role runs its own loop (one register allocation per role, not their union), one barrier per chunk
        if (role == 1) {             // staging waves: chunk ch + 1 into the other buffer (its loads went out one chunk ago), then chunk ch + 2's loads
            for (int ch = 0; ch < NCH; ++ch) {
                if (ch + 1 < NCH) {
                    char* Ad = reinterpret_cast<char*>(smem) + ((ch + 1) & 1) * BUF_BYTES;
                    s_store(ch + 1, Ad, Ad + G::A_UNITS * 16, Set0{});
                    if (ch + 2 < NCH) g_load(ch + 2, Set0{});
                }
                __syncthreads();
            }
            return;                  // (barriers count live waves only)
        }
        for (int ch = 0; ch < NCH; ++ch) {
            As = reinterpret_cast<char*>(smem) + (ch & 1) * BUF_BYTES;
            Bs = As + G::A_UNITS * 16;
            taps_of_chunk();
            __syncthreads();         // this chunk is read, the next one is in the other buffer
        }
    } else {
        if (trace) trace[1] = smg_stamp();
        for (int ch = 0; ch < NCH; ++ch) {
            if (ch + 1 < NCH) g_load(ch + 1, Set0{});
            taps_of_chunk();
            if (trace && ch == 0) trace[2] = smg_stamp();
            __syncthreads();                          // every wave is done reading this chunk
            if (ch + 1 < NCH) {
                s_store(ch + 1, As, Bs, Set0{});
                __syncthreads();
            }
        }
    }
    if (trace) trace[3] = smg_stamp();
    if constexpr (G::WX > 1) {                    // fold the tap-split partial tiles into the wk == 0 waves
        float* r = smem + (wq * 16) * 64 + lane;  // [WQ][16][64], MT == 1
        if (wk > 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) r[q * 64] = acc[0][q];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[0][q] += r[q * 64];
        }
        __syncthreads();
    }

    // epilogue: raw output + per-(stream, channel) sum / sum of squares (fp64)
    if constexpr (OP == 3) {                      // products of scaled operands: exact power-of-two correction
        const float inv = a.asc[1] * pack_inv_scale(a.wu);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] *= inv;
    }
    double s = 0.0, ss = 0.0;
    if (wk == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = 4 * g + k, i = k + 8 * g + 4 * half;
                    const int py = y0 + (kRegFrag ? G::rowi(wq, m, i) : G::row(wq, m, i)), px = x0 + G::col(i);
                    v[k] = acc[m][r];
                    if (!kEdge || (py < a.pl.H && px < a.pl.W)) {      // tiles may hang over the edge (planes that do not tile exactly)
                        const double xd = (double)v[k];
                        s += xd;
                        ss += xd * xd;
                    }
                }
                // four pixels x four channels of the quad, transposed: this lane stores ONE pixel's four consecutive channels
                quad_transpose4(v[0], v[1], v[2], v[3]);
                const int i = (lane & 3) + 8 * g + 4 * half;
                const int py = y0 + (kRegFrag ? G::rowi(wq, m, i) : G::row(wq, m, i)), px = x0 + G::col(i);
                if (!kEdge || (py < a.pl.H && px < a.pl.W))
                    stq<ST>(a.dst, ((int64_t)n * a.pl.HWp + py * a.pl.W + px) * a.ldd + a.dcoff + 4 * (l31 >> 2), make_float4(v[0], v[1], v[2], v[3]));
            }
    }
    s += __shfl_xor(s, 32);
    ss += __shfl_xor(ss, 32);
    double* red = reinterpret_cast<double*>(smem);      // [2][4][32]
    __syncthreads();
    if (half == 0) { red[wave * 32 + l31] = s; red[128 + wave * 32 + l31] = ss; }
    __syncthreads();
    if (t < 64) {
        const int q = t >> 5, c = t & 31;
        const double tot = red[q * 128 + c] + red[q * 128 + 32 + c] + red[q * 128 + 64 + c] + red[q * 128 + 96 + c];
        atomicAdd((q ? a.dsq : a.dsum) + (int64_t)n * a.dstride + a.dcoff + c + fstat_rep(), tot);
    }
    if (trace) { trace[4] = smg_stamp(); trace[6] = __builtin_amdgcn_s_memrealtime(); }
}


// The finished gradient of a layer's 32 output channels, as the 3x3 backward kernels read it: either a dense
// [n][HWp][32] array (x == nullptr), or the G' and X slices of the block buffers with the deferred BN backward
// applied on load (what bn_bwd_apply_kernel would have written):
//   g = invstd * ((G' - SA/n) - (x - mean) * invstd * SB/n)
// Statistics pointers are already offset to the slice's first channel; sstride = floats per stream.
struct GradSrc {
    const unsigned* amax;           // operand kind 3 (x == nullptr only): [streams][kAmaxRep] recorded maxima of g (bn_bwd_apply_kernel)
    const void* g; int ldg;         // gradient storage: fp32 / bf16 by mode
    const void* x; int ldx;         // activation storage: fp32 / bf16 / fp16 by mode
    const double* xsum; const double* xsq; const double* s1; const double* s2; int sstride;
    float eps;
};
// parameters a | q1 | mean | k of affine2() for the 32 channels -> gp[4][32]  (threads 0..31)
__device__ __forceinline__ void grad_src_params(const GradSrc& s, int n, int hw, float* gp) {
    const int t = threadIdx.x;
    if (s.x && t < 32) {
        const double inv = 1.0 / (double)hw;
        float mean, invstd;
        bn_moments(s.xsum, s.xsq, (int64_t)n * s.sstride + t, inv, s.eps, mean, invstd);
        gp[t] = invstd;
        gp[32 + t] = (float)(stat_get(s.s1, (int64_t)n * s.sstride + t) * inv);
        gp[64 + t] = mean;
        gp[96 + t] = invstd * (float)(stat_get(s.s2, (int64_t)n * s.sstride + t) * inv);
    }
}

struct Halo3x3DgradArgs {
    GradSrc g; Plane pl;                             // finished output gradient (32 channels)
    const u32x4* wu;                                 // weight units [c/32][tap][piece][k8][c%32] (PK_HD)
    BnTab bt;                                        // norm2 statistics of this layer (stored by the forward) + gamma / beta
    int C;                                           // bottleneck channels (128)
    const void* mbuf;                                // raw bottleneck [n][HWp][C] (mask + xhat source)
    void* dst;                                       // dy [n][HWp][C]
    double* o1; double* o2; int ostride;             // per-stream sums [n][C]
    int tiles_x;
    int cg_per_wg;                                   // output-channel groups (NCW chunks of 32) per workgroup; blockIdx.z picks the run
};

// ------------------------------------------------------------------------------------
// Split-precision data gradient.  The (TS+2)^2 x 32 gradient halo is split ONCE per workgroup into units
// [piece][k8 (4)][halo pixel] (62 KB at TS = 16: two workgroups per CU); the weights of one stage = one tap x NCW
// 32-channel output chunks ([piece][k8][c] units, 6 KB per chunk; pack mode PK_HD) stream through a double buffer, one
// barrier per stage, prefetched into registers under the stage's MFMAs.  A stage is two k16-steps (32 gradient channels)
// per pixel tile; after the ninth tap of a chunk the epilogue applies the ReLU mask and collects the norm2 sums.
// ------------------------------------------------------------------------------------
template <int TS, int PREC, int TH = TS> struct HaloDgradSGeo : HaloGeo<TS, TH> {
    using G = HaloGeo<TS, TH>;
    static constexpr int NP = np_of(bwd_op(PREC));
    static constexpr int BU = NP * 4 * 32;                               // weight units per (tap, 32-channel chunk): 384 / 256 / 128
    static constexpr int LDH = G::PX;
    static constexpr int A_UNITS_ = NP * 4 * LDH;
    static constexpr int A_N = (G::PX * (PREC ? 4 : 8) + 255) / 256;     // 16-byte slots per thread (32 gradient channels per pixel)
    static constexpr int NCW = G::WX;                                    // output-channel chunks per stage
    // taps per stage.  16 x 16 tiles: one kernel row (a stage per tap spent more on its barrier than on its 12 MFMAs).  8 x 8 tiles: ONE tap
    // (round 6) - two buffers of a whole row's weights for 64 output channels made 65 KB of LDS, two workgroups per CU, and the 850
    // workgroups of a 17-stream launch on the 40^2 planes ran in TWO rounds (per-workgroup stamps: the second starts 10 us in; span 18.5-20.8 us
    // for lives of 8-9); a tap per stage is 33 KB - four per CU, one round.
    static constexpr int ST = TS == 8 ? SMG_D3_ST8 : 3;
    static constexpr int B_UNITS = NCW * ST * BU;                        // per buffer
    static constexpr int B_N = (B_UNITS + 255) / 256;
    static constexpr int B_PAD = B_N * 256;                              // units per buffer incl. the padding the last copy round touches
    // (16 x 8 tiles: the halo area is rounded up to two ring buffers - it hosts buffers 2 and 3 of the weights' LDS-DMA ring once the
    //  gradient fragments are in registers; 52.7 KB, three workgroups per CU)
    static constexpr int A_UNITS = (TH != TS && A_UNITS_ < 2 * B_PAD) ? 2 * B_PAD : A_UNITS_;
    __host__ __device__ static constexpr int smem_bytes(int C) { return (A_UNITS + 2 * B_PAD) * 16 + (4 * C + 256 + 128) * 4; }
};

// TH = 8 with TS = 16 (round 6): 16 x 8 tiles, one MFMA tile per wave - half the accumulators, fragments and mask prefetch per wave,
// 52.7 KB of LDS: THREE workgroups per CU where the 16 x 16 form (252 registers, 66 KB) holds two.
template <int TS, int PREC = 0, bool RAG = false, int TH = TS>
static __global__ __launch_bounds__(256, TS == 8 ? 4 : (TH == TS ? 2 : 3)) void conv3x3_halo_dgrad_kernel(const Halo3x3DgradArgs a) {
    constexpr bool kEdge = TS == 8 || RAG;
    using G = HaloDgradSGeo<TS, PREC, TH>;
    using GT = grd_t<PREC>;
    using XT = act_t<PREC>;
    constexpr int OP = bwd_op(PREC), NP = G::NP, HDS_BU = G::BU, GSZ = GT::size, XSZ = XT::size, E = 16 / GSZ, SPP = 32 / E;   // SPP: slots per pixel
    constexpr int MT = G::MT, NCW = G::NCW, A_N = G::A_N, B_N = G::B_N, LDH = G::LDH;
    // gradient fragments resident in registers: four row pairs + DPP column shifts (below).  Precision mode 0 only: in the 16-bit modes the
    // resident fragments take the bounds-checked instantiation from 161 to 169 registers - two waves per SIMD instead of three, 62.9 -> 70.6 us
    // per launch on config 5's planes - and gain nothing where they fit (45.6 us both ways on config 3's)
    constexpr bool kRegFrag = TS == 16 && OP == 3 && SMG_HALO_REGFRAG;
    // ... and with the halo out of LDS after the prologue, the weights of the stages stream through a RING of four LDS buffers by LDS-DMA
    // (two behind the halo, two in the halo's own area once the fragments are in registers), three stages ahead of the MFMAs - see below
    constexpr bool kDma = kRegFrag && G::NCW == 1 && G::B_UNITS % 256 == 0 && 2 * G::B_PAD <= G::A_UNITS && SMG_HALO_DMA;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* As = reinterpret_cast<char*>(smem);                            // [piece][k8][LDH] units
    char* Bs = As + G::A_UNITS * 16;                                     // [2][NCW][3 taps][piece][k8][32] units
    float* prm = reinterpret_cast<float*>(Bs + 2 * G::B_PAD * 16);       // scale | beta | mean | invstd, C each
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, half = lane >> 5;
    const int wq = wave % G::WQ, wc = wave / G::WQ;     // pixel slice, output-channel chunk of the stage
    const int n = blockIdx.y;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
    const int y0 = ty * TH, x0 = tx * TS;
    const int C = a.C;
    // dev stamps (SMG_TRACE_KIND=5): start | halo staged | first output-channel group's 9 stages | its epilogue | end
    unsigned long long* trace = (g_smg_trace && t == 0) ? g_smg_trace + 8 * ((size_t)blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z)) : nullptr;
    if (trace) { trace[0] = __builtin_amdgcn_s_memtime(); trace[5] = __builtin_amdgcn_s_memrealtime(); }
    const int cg0 = blockIdx.z * a.cg_per_wg;
    constexpr int SPR = 3 / G::ST;                     // stages per kernel row: 1 (a row's three taps per stage) or 3 (a tap per stage, 8 x 8 tiles)
    const int NSTAGE = a.cg_per_wg * 3 * SPR;          // channel-chunk groups x kernel rows x stages per row
    auto ring = [&](int k) -> char* { return k < 2 ? Bs + k * G::B_PAD * 16 : As + (k - 2) * G::B_PAD * 16; };
    auto dma_stage = [&](int stage, int k) {           // the weights of `stage` (one kernel row of one 32-channel group: B_UNITS straight 16-byte copies) -> ring buffer k
        const int cg = cg0 + stage / 3, dyy = stage % 3;      // (ring form: TS = 16, a kernel row per stage)
        const unsigned dst = (unsigned)(uintptr_t)ring(k) + 16u * 64u * (unsigned)wave;
#pragma unroll
        for (int i = 0; i < B_N; ++i)
            dma_unit(UnitSrc{a.wu, kWholeBuf, 16u * (unsigned)(t + 256 * i), 16u * (unsigned)((cg * NCW * 9 + G::ST * dyy) * HDS_BU)}, dst + 16u * 256u * (unsigned)i);
    };
    if constexpr (kDma) {                              // the first two stages' weights go out in front of everything else
        dma_stage(0, 0);
        dma_stage(NSTAGE > 1 ? 1 : 0, 1);
    }
    // The gradient halo's loads go out HERE, in front of the parameter loop (round 6: per-workgroup stamps showed 7.5k cycles from the
    // kernel's first instruction to these loads being ISSUED - the table loads of the parameter loop, the recorded maxima's scalar loads
    // and the kernel arguments each cost a dependent memory round trip in front of them - and another 2k until they had landed)
    const char* g_n = static_cast<const char*>(a.g.g) + (int64_t)GSZ * n * a.pl.HWp * a.g.ldg;
    const char* x_n = a.g.x ? static_cast<const char*>(a.g.x) + (int64_t)XSZ * n * a.pl.HWp * a.g.ldx : nullptr;
    // ... and in front of them the stream's recorded gradient maxima (operand kind 3, materialised GS): their vector loads used to sit
    // behind the parameter loop with a vmcnt(0) of their own - a third round trip
    u32x4 am4[kAmaxRep / 4];
    if constexpr (OP == 3) {
        if (!a.g.x) {
#pragma unroll
            for (int r = 0; r < kAmaxRep / 4; ++r) am4[r] = *reinterpret_cast<const u32x4*>(a.g.amax + (int64_t)n * kAmaxRep + 4 * r);
        }
    }
    float4 rv[A_N], rx[A_N];
#pragma unroll
    for (int i = 0; i < A_N; ++i) {                // every load in flight before the first LDS store
        const int idx = t + 256 * i;
        const int hp = idx / SPP, q = idx % SPP;                        // halo pixel, 16-byte slot (E channels) of its 32
        const int hy = hp / G::W, hx = hp - hy * G::W;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = idx < G::PX * SPP && (unsigned)iy < (unsigned)a.pl.H && (unsigned)ix < (unsigned)a.pl.W;
        const int64_t pix = ok ? iy * a.pl.W + ix : 0;                 // unconditional loads, clamped address
        rv[i] = ld16(g_n, (int64_t)GSZ * (pix * a.g.ldg + E * q));
        if (x_n) rx[i] = ld16(x_n, (int64_t)XSZ * (pix * a.g.ldx + E * q));
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int k = t; k < C; k += 256) {                  // norm2 parameters of this stream (layer table of the forward)
        const float invstd = tab_invstd(a.bt, n)[k];
        prm[k] = a.bt.gamma[k] * invstd;
        prm[C + k] = a.bt.beta[k];
        prm[2 * C + k] = tab_mean(a.bt, n)[k];
        prm[3 * C + k] = invstd;
    }
    float* gp = prm + 4 * C + 256;                     // GradSrc parameters [4][32]
    grad_src_params(a.g, n, a.pl.HW, gp);
    // operand kind 3: scale of the gradient operand and inverse of the (gradient x weight) scale - from the stream's recorded maximum
    // (materialised GS), or, when the BN backward is applied on load (a.g.x set), from the largest magnitude of THIS workgroup's halo
    ActScale gsc{1.f, 1.f};
    if constexpr (OP == 3) {
        if (!a.g.x) {
            unsigned m = 0;
#pragma unroll
            for (int r = 0; r < kAmaxRep / 4; ++r) m = max(max(m, max(am4[r].x, am4[r].y)), max(am4[r].z, am4[r].w));
            gsc = scale_of_max(m);
            gsc.inv *= pack_inv_scale(a.wu);
        }
    }
    // Weight stages: stage = cgroup * 3 + kernel row (a cgroup holds NCW chunks, a stage the row's three taps of each: they are
    // contiguous in the pack).  Two LDS buffers; the loads of stage s + 1 are issued at the top of stage s, fly under its 36 MFMAs
    // per wave and are stored at its end - ONE barrier per kernel row (round 3: one per tap - 1.6k cycles per stage for 0.77k
    // of MFMA issue; with three MFMA terms a tap is 0.38k).  Unconditional, tail-clamped loads: the same loads on every path.
    constexpr int ST = G::ST;
    u32x4 rb[B_N];
    unsigned b_voff[B_N];
#pragma unroll
    for (int i = 0; i < B_N; ++i) {
        const int idx = t + 256 * i;                                       // past B_UNITS: padding (slack behind the packed array)
        const int j = min(idx / (ST * HDS_BU), NCW - 1), rem = idx - j * ST * HDS_BU;
        b_voff[i] = 16u * (unsigned)(j * 9 * HDS_BU + rem);
    }
    auto g_load = [&](int stage) {
        const int cg = cg0 + stage / (3 * SPR), tap0 = (stage % (3 * SPR)) * ST;       // first tap of the stage
#pragma unroll
        for (int i = 0; i < B_N; ++i) rb[i] = bload_u4(a.wu, kWholeBuf, b_voff[i], 16u * (unsigned)((cg * NCW * 9 + tap0) * HDS_BU));
    };
    // gradient halo (zero outside the image), split at the store
    {
#ifdef SMG_TRACE_INIT
        if (trace) trace[1] = smg_stamp();
#endif
        __syncthreads();                           // gp (and prm) visible
#ifdef SMG_TRACE_INIT
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (trace) trace[2] = smg_stamp();
#endif
        if constexpr (OP == 3) {
            if (x_n) {                             // (launch-uniform) BN backward on load: the halo's own maximum sets the scale
                float vmax = 0.f;
#pragma unroll
                for (int i = 0; i < A_N; ++i) {
                    const int idx = t + 256 * i;
                    const int hp = idx / SPP, q = idx % SPP;
                    const int hy = hp / G::W, hx = hp - hy * G::W;
                    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
                    const bool ok = idx < G::PX * SPP && (unsigned)iy < (unsigned)a.pl.H && (unsigned)ix < (unsigned)a.pl.W;
                    const float4 v = ok ? affine2(rv[i], rx[i], gp + 4 * (q % 8), 32) : zero4();
                    rv[i] = v;
                    vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
                float* red = prm + 4 * C;          // (the epilogue's scratch: free here)
                if (lane == 0) red[wave] = vmax;
                __syncthreads();
                vmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
                if (vmax > 0.f) gsc = scale_of_max(__float_as_uint(vmax));
                gsc.inv *= pack_inv_scale(a.wu);
            }
        }
#pragma unroll
        for (int i = 0; i < A_N; ++i) {
            const int idx = t + 256 * i;
            if (idx < G::PX * SPP) {
                const int hp = idx / SPP, q = idx % SPP;
                const int hy = hp / G::W, hx = hp - hy * G::W;
                const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
                const bool ok = (unsigned)iy < (unsigned)a.pl.H && (unsigned)ix < (unsigned)a.pl.W;
                if constexpr (E == 4) {
                    float4 v = rv[i];
                    if (OP != 3 && x_n) v = affine2(rv[i], rx[i], gp + 4 * q, 32);      // (operand kind 3: applied above, with the maximum)
                    if constexpr (OP == 3) v = mul4(v, gsc.s);
                    if (!ok) v = zero4();
                    const Split4 sp = split4<OP>(v);
#pragma unroll
                    for (int pc = 0; pc < NP; ++pc)
                        *reinterpret_cast<uint2*>(As + ((pc * 4 + (q >> 1)) * LDH + hp) * 16 + (q & 1) * 8) = sp.p[pc];
                } else {
                    u32x4 u = as_u4(rv[i]);                                  // finished bf16 gradient: a copy
                    if (x_n) u = pack_unit<OP>(affine2(slot_quad<GT>(rv[i], 0), slot_quad<XT>(rx[i], 0), gp + 8 * q, 32),
                                               affine2(slot_quad<GT>(rv[i], 1), slot_quad<XT>(rx[i], 1), gp + 8 * q + 4, 32));
                    if (!ok) u = u32x4{0u, 0u, 0u, 0u};
                    *reinterpret_cast<u32x4*>(As + (q * LDH + hp) * 16) = u;
                }
            }
        }
    }
    auto s_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < B_N; ++i) *reinterpret_cast<u32x4*>(Bs + (buf * G::B_PAD + t + 256 * i) * 16) = rb[i];
    };
    int abase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) abase[m] = G::row(wq, m, l31) * G::W + G::col(l31);

    if constexpr (!kDma) {
        g_load(0);
        s_store(0);
    }
    __syncthreads();
#ifdef SMG_TRACE_INIT
    if (trace) trace[3] = smg_stamp();
#else
    if (trace) trace[1] = __builtin_amdgcn_s_memtime();
#endif
    // TS = 16 (round 6): the wave's two tiles are the row pairs (4 wq + m, 4 wq + m + 2) (HaloGeo::rowi), and its whole gradient operand
    // - four row-pair fragments (pair i = halo rows 4 wq + i, 4 wq + i + 2; halo columns 0..15) and one edge fragment (columns 16, 17
    // of the four pairs) per piece and k16-step, 80 registers - is read from the LDS halo ONCE.  Tap (dy, dx) of tile m takes pair
    // m + 2 - dy shifted left by 2 - dx lanes (dpp_col_shift): the stages read nothing but weights from LDS (12 instead of 36
    // ds_read_b128 per kernel row).
    constexpr int RP = kRegFrag ? G::RP : 1;
    u32x4 P[RP][2][NP], Eg[2][NP];
    if constexpr (kRegFrag) {
        const int ej = ((lane & 15) >> 1) < RP ? ((lane & 15) >> 1) : RP - 1;      // edge fragment: lanes 2 j, 2 j + 1 of a DPP row carry pair j
        const int pbase = (G::pair0(wq) + G::RSTEP * (l31 >> 4)) * G::W + (l31 & 15);
        const int ebase = (G::pair0(wq) + ej + G::RSTEP * (l31 >> 4)) * G::W + 16 + (lane & 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
                const char* pl_ = As + ((pc * 4 + 2 * ks + half) * LDH) * 16;
#pragma unroll
                for (int i = 0; i < RP; ++i) P[i][ks][pc] = *reinterpret_cast<const u32x4*>(pl_ + (pbase + i * G::W) * 16);
                Eg[ks][pc] = *reinterpret_cast<const u32x4*>(pl_ + ebase * 16);
            }
        if constexpr (kDma) {
            // every wave holds its fragments: the halo's LDS area becomes ring buffers 2 and 3
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            dma_stage(NSTAGE > 2 ? 2 : NSTAGE - 1, 2);
            dma_stage(NSTAGE > 3 ? 3 : NSTAGE - 1, 3);
        }
    }
#ifdef SMG_TRACE_INIT
    if (trace) trace[4] = smg_stamp();
#endif
    f32x16 acc[MT];
    // Mask / xhat source of this wave's output tile (32 output channels from c0), prefetched: one row segment (a pixel's four
    // consecutive channels, 16 / 8 bytes per lane; finish_acc_rows turns it into accumulator layout) per four accumulator rows,
    // MT x 4 of them per output-channel group - issued behind the weight loads of the group's first two kernel rows (every path
    // issues the same loads: exact vmcnt waits), consumed by the epilogue after the third.  (Fetching them only in the epilogue
    // cost 5.5 - 6k of a group's 21k cycles.)
    rawq_t<XT> xq[MT][4];
    auto load_mask_seg = [&](int c0, int m, int g) {
        const int i = (lane & 3) + 8 * g + 4 * half;
        const int py = y0 + (kRegFrag ? G::rowi(wq, m, i) : G::row(wq, m, i)), px = x0 + G::col(i);
        const bool ok = !kEdge || (py < a.pl.H && px < a.pl.W);     // tiles may hang over the edge: clamped address
        const int64_t idx = ((int64_t)n * a.pl.HWp + (ok ? py * a.pl.W + px : 0)) * C + c0 + 4 * (l31 >> 2);
        if constexpr (std::is_same<XT, e_f32>::value) xq[m][g] = *reinterpret_cast<const float4*>(static_cast<const float*>(a.mbuf) + idx);
        else xq[m][g] = *reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(a.mbuf) + idx);
    };
    for (int s3 = 0; s3 < NSTAGE; s3 += 3 * SPR) {     // one output-channel group (three kernel rows) per trip
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
      const int cmask0 = ((cg0 + s3 / (3 * SPR)) * NCW + wc) * 32;
      sfor<0, 3>([&](auto DY) {
       sfor<0, SPR>([&](auto SUB) {
        constexpr int dy = decltype(DY)::value, sub = decltype(SUB)::value;
        const int stage = s3 + dy * SPR + sub, buf = stage & 1;
        if constexpr (kDma) {
            // Ring of four buffers, stage s in buffer s % 4, requested three stages ahead.  Behind this wait at most the two younger
            // stages' requests (3 per wave and stage) are still out - the ones of stages s + 1, s + 2; vmcnt counts in order, and any
            // other vector memory operation the compiler put behind them only makes the wait stricter - and behind the barrier every
            // wave's pieces of stage s have landed AND every wave has left stage s - 1, whose buffer takes stage s + 3.
            dma_wait<2 * B_N>();
            dma_barrier();
            if (stage > 0) dma_stage(stage + 3 < NSTAGE ? stage + 3 : NSTAGE - 1, (stage + 3) & 3);      // (tail: dead re-requests of the last stage into buffers nobody reads again)
        } else
        g_load(stage + 1 < NSTAGE ? stage + 1 : NSTAGE - 1);             // (tail: a clamped re-load, stored dead)
        if (dy < 2 && sub == 0) {                         // (compile-time after unrolling) the group's mask segments: half behind each of the first two rows' weight loads
            constexpr int SEGS = MT * 4;
#pragma unroll
            for (int sg = 0; sg < SEGS / 2; ++sg) { const int q = dy * (SEGS / 2) + sg; load_mask_seg(cmask0, q / 4, q % 4); }
        }
        sfor<0, ST>([&](auto DX) {
        constexpr int dxi = decltype(DX)::value, dx = sub * ST + dxi;      // tap inside the stage, kernel column
        // output pixel (ry, rx), tap (dy, dx) reads g at halo (ry + 2 - dy, rx + 2 - dx)
        const int toff = (2 - dy) * G::W + (2 - dx);
        const char* Bw = kDma ? ring(stage & 3) + ((wc * ST + dxi) * HDS_BU) * 16 : Bs + (buf * G::B_PAD + (wc * ST + dxi) * HDS_BU) * 16;
        auto fa = [&](int m, int ks, int pc) -> u32x4 {
            return *reinterpret_cast<const u32x4*>(As + ((pc * 4 + 2 * ks + half) * LDH + abase[m] + toff) * 16);
        };
        auto fb = [&](int ks, int pc) -> u32x4 {
            return *reinterpret_cast<const u32x4*>(Bw + ((pc * 4 + 2 * ks + half) * 32 + l31) * 16);
        };
        if constexpr (kRegFrag) {                           // fragments of the gradient from the resident row pairs; weights from LDS
            u32x4 af[2][MT][NP], bf[2][NP];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) {
                    bf[ks][pc] = fb(ks, pc);
                    sfor<0, MT>([&](auto M) {
                        constexpr int m = decltype(M)::value;
                        af[ks][m][pc] = dpp_col_shift<2 - dx, m + 2 - dy>(P[m + 2 - dy][ks][pc], Eg[ks][pc]);
                    });
                }
            if constexpr (OP == 3) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int g = 0; g < 3; ++g)
#pragma unroll
                        for (int m = 0; m < MT; ++m) acc[m] = mfma_f16(af[ks][m][g == 1 ? 1 : 0], bf[ks][g == 0 ? 1 : 0], acc[m]);
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] = mfma_1p<OP>(af[ks][m][0], bf[ks][0], acc[m]);
            }
        } else
        if constexpr (OP == 3) {                            // two fp16 pieces: every fragment of the tap up front, then h*l, l*h, h*h per k16-step
            u32x4 af[2][MT][2], bf[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) af[ks][m][pc] = fa(m, ks, pc);
                    bf[ks][pc] = fb(ks, pc);
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] = mfma_f16(af[ks][m][g == 1 ? 1 : 0], bf[ks][g == 0 ? 1 : 0], acc[m]);
        } else
        if constexpr (OP != 0) {                            // single-piece operands: one term per k16-step
            u32x4 ah[2][MT], bh[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int m = 0; m < MT; ++m) ah[ks][m] = fa(m, ks, 0);
                bh[ks] = fb(ks, 0);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = mfma_1p<OP>(ah[ks][m], bh[ks], acc[m]);
        } else {
            // every fragment of the tap is requested up front (18 ds_read_b128, 72 registers): one LDS round trip per tap
            // instead of four read -> wait -> MFMA phases; hipcc waits per operand (lgkmcnt(N)) as the MFMAs come up
            u32x4 af[2][MT][NPIECE], bf[2][NPIECE];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pc = 0; pc < NPIECE; ++pc) {
                    const int pp = pc == 1 ? 2 : pc == 2 ? 1 : 0;            // hi, lo, mid: the order the terms need them
#pragma unroll
                    for (int m = 0; m < MT; ++m) af[ks][m][pp] = fa(m, ks, pp);
                    bf[ks][pp] = fb(ks, pp);
                }
            constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};     // hi*lo, lo*hi, mid*mid, hi*mid, mid*hi, hi*hi
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int g = 0; g < 6; ++g)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] = mfma_bf16(af[ks][m][PA[g]], bf[ks][PB[g]], acc[m]);
        }
        });   // dx
        if constexpr (!kDma) s_store(buf ^ 1);          // that buffer was last read one stage ago, behind that stage's barrier (at the very end: a dead store)
        if (dy == 2 && sub == SPR - 1) {
#ifndef SMG_TRACE_INIT
            if (trace && s3 == 0) trace[2] = __builtin_amdgcn_s_memtime();
#endif
            // epilogue of this wave's output-channel chunk: ReLU mask, store dy, BN(norm2) backward sums
            const int c = ((cg0 + stage / (3 * SPR)) * NCW + wc) * 32 + l31;
            const float sc = prm[c], be = prm[C + c], mean = prm[2 * C + c], invstd = prm[3 * C + c];
            float s1 = 0.f, s2 = 0.f;
            auto finish = [&](rawq_t<XT> (&xq)[MT][4]) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float xv[4], o[4];
                        finish_acc_rows<XT>(xq[m][g], xv);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int i = k + 8 * g + 4 * half;
                            const int py = y0 + (kRegFrag ? G::rowi(wq, m, i) : G::row(wq, m, i)), px = x0 + G::col(i);
                            const bool in = !kEdge || (py < a.pl.H && px < a.pl.W);
                            const float dyv = (in && bn1(xv[k], mean, sc, be) > 0.f) ? acc[m][4 * g + k] * gsc.inv : 0.f;      // (gsc.inv == 1 unless operand kind 3)
                            o[k] = dyv;
                            s1 += dyv;
                            s2 += dyv * ((xv[k] - mean) * invstd);
                        }
                        quad_transpose4(o[0], o[1], o[2], o[3]);
                        const int i = (lane & 3) + 8 * g + 4 * half;
                        const int py = y0 + (kRegFrag ? G::rowi(wq, m, i) : G::row(wq, m, i)), px = x0 + G::col(i);
                        if (!kEdge || (py < a.pl.H && px < a.pl.W))
                            stq<GT>(a.dst, ((int64_t)n * a.pl.HWp + py * a.pl.W + px) * C + c - l31 + 4 * (l31 >> 2), make_float4(o[0], o[1], o[2], o[3]));
                    }
            };
            finish(xq);
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            float* red = prm + 4 * C;                 // [2][4][32]
            if (half == 0) { red[wave * 32 + l31] = s1; red[128 + wave * 32 + l31] = s2; }
            __syncthreads();
            if (t < 64 * NCW) {                       // waves j*WQ .. j*WQ + WQ-1 hold chunk j of the stage
                const int j = t >> 6, q = (t >> 5) & 1, cc = t & 31;
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < G::WQ; ++w) tot += red[q * 128 + (j * G::WQ + w) * 32 + cc];
                const int ch = ((cg0 + stage / (3 * SPR)) * NCW + j) * 32 + cc;
                atomicAdd((q ? a.o2 : a.o1) + (int64_t)n * a.ostride + ch + stat_rep(), (double)tot);
            }
#ifndef SMG_TRACE_INIT
            if (trace && s3 == 0) trace[3] = __builtin_amdgcn_s_memtime();
#endif
        }
        if constexpr (!kDma) __syncthreads();
       });
      });
    }
    if constexpr (kDma) dma_wait<0>();                  // no request may still be on its way into this workgroup's LDS when it ends
#ifdef SMG_TRACE_INIT
    if (trace) trace[6] = __builtin_amdgcn_s_memrealtime();
#else
    if (trace) { trace[4] = __builtin_amdgcn_s_memtime(); trace[6] = __builtin_amdgcn_s_memrealtime(); }
#endif
}


struct Halo3x3WgradArgs {
    GradSrc g; Plane pl;                             // finished output gradient (32 channels)
    const void* src; int C;                          // raw bottleneck [n][HWp][C]
    BnTab bt;                                        // norm2 statistics of this layer (stored by the forward) + gamma / beta
    const float* asc;                                // operand kind 3: {s, 1 / s} of the BN + ReLU operand
    float* part;                                     // partial sums [groups * streams][9][32][C]
    int tiles_x, n_tiles, tiles_per_wg;
    int groups, streams;                             // launch geometry: 1-D grid of 8 * ceil(groups * streams / 8) * (C / 32) workgroups
};

// ------------------------------------------------------------------------------------
// Split-precision weight gradient.  The reduction runs over pixels, so both operands stay row-major
// [pixel][32 channels] in LDS (bf16 pieces, split at the store) and the MFMA fragments are gathered with
// ds_read_b64_tr_b16: a 16-lane group fetches 4 consecutive pixels x 16 channels and every lane receives its channel
// of all four (gemm.cuh, weight-gradient form).  The activation halo serves all nine taps by shifting the pixel a lane
// points at; the gradient fragment is read once per k16-step.
//   TW = 16: tile 16 x 8 pixels, a k16-step is one tile row; each wave reduces two rows.   59 KB LDS, 2 workgroups / CU
//   TW = 8 : tile  8 x 8 pixels, a k16-step is two tile rows; each wave reduces one step.
// Three taps (one kernel row) are in flight at a time: 18 MFMAs on three independent accumulators per group.
// ------------------------------------------------------------------------------------
template <int TW, int PREC> struct HaloWgradSGeo {
    static_assert(TW == 16 || TW == 8, "tile width");
    static constexpr int NP = np_of(bwd_op(PREC)), SPP = PREC ? 4 : 8;    // pieces per operand; 16-byte slots per pixel (32 channels)
    static constexpr int TH = 8, HW_ = TW + 2, HH = TH + 2, PX = HW_ * HH, NPIX = TW * TH;
    static constexpr int KSTEPS = NPIX / 16 / 4;                          // k16-steps per wave per tile: 2 / 1
    static constexpr int B_BYTES = NP * PX * 64, A_BYTES = NP * NPIX * 64;
    static constexpr int B_N = (PX * SPP + 255) / 256, A_N = (NPIX * SPP + 255) / 256;   // slots per thread: halo, gradient
    static constexpr int RED_BYTES = 4 * 16 * 64 * 4;
    __host__ __device__ static constexpr int smem_bytes() {
        return (B_BYTES + A_BYTES > RED_BYTES ? B_BYTES + A_BYTES : RED_BYTES) + (96 + 128) * 4;
    }
};

template <int TW, int PREC = 0>
static __global__ __launch_bounds__(256, kHaloWgradWaves) void conv3x3_halo_wgrad_kernel(const Halo3x3WgradArgs a) {
    using G = HaloWgradSGeo<TW, PREC>;
    using GT = grd_t<PREC>;
    using XT = act_t<PREC>;
    constexpr int OP = bwd_op(PREC), NP = G::NP, SPP = G::SPP, E = 32 / SPP, GSZ = GT::size, XSZ = XT::size;
    constexpr int B_N = G::B_N, A_N = G::A_N;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* Bh = reinterpret_cast<char*>(smem);           // [piece][PX][32] bf16: activation halo
    char* Ag = Bh + G::B_BYTES;                         // [piece][NPIX][32] bf16: gradient tile
    float* prm = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + G::smem_bytes()) - 96 - 128;    // mean | scale | beta (32 each)
    float* gp = prm + 96;                               // GradSrc parameters [4][32]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, half = lane >> 5;
    // Workgroup -> (tile group, stream, channel group).  The C / 32 channel groups of one (tile group, stream) read the SAME gradient
    // tiles: they sit on ONE XCD (workgroups go to the eight XCDs round-robin by linear index) in consecutive slots, so three of the
    // four reads hit that XCD's L2 instead of HBM (round 5: 1.9 GB per step of this kernel's 4.4 GB were such re-reads).
    const int C = a.C;
    const int ncg = C / 32, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int gz = (slot / ncg) * 8 + xcd;             // (tile group, stream) index
    if (gz >= a.groups * a.streams) return;
    const int n = gz / a.groups, bx = gz - n * a.groups, cc0 = (slot % ncg) * 32;
    // operand kind 3: scale of this stream's gradient operand (from its recorded maxima) and inverse of the (gradient x activation) scale.
    // The maxima's loads go out here and are looked at behind the first tile's loads (round 6: amax_scale's own vmcnt(0) in front of
    // everything else was one more dependent memory round trip per workgroup - the 8 x 8 launches are ONE workgroup life long).
    u32x4 am4[kAmaxRep / 4];
    float sa = 1.f;
    if constexpr (OP == 3) {
#pragma unroll
        for (int r = 0; r < kAmaxRep / 4; ++r) am4[r] = *reinterpret_cast<const u32x4*>(a.g.amax + (int64_t)n * kAmaxRep + 4 * r);
        sa = a.asc[0];
    }
    auto grad_scale = [&]() -> ActScale {
        ActScale g{1.f, 1.f};
        if constexpr (OP == 3) {
            unsigned m = 0;
#pragma unroll
            for (int r = 0; r < kAmaxRep / 4; ++r) m = max(max(m, max(am4[r].x, am4[r].y)), max(am4[r].z, am4[r].w));
            g = scale_of_max(m);
            g.inv *= a.asc[1];
        }
        return g;
    };
    if (t < 32) {
        prm[t] = tab_mean(a.bt, n)[cc0 + t];
        prm[32 + t] = a.bt.gamma[cc0 + t] * tab_invstd(a.bt, n)[cc0 + t] * sa;
        prm[64 + t] = a.bt.beta[cc0 + t] * sa;
    }
    grad_src_params(a.g, n, a.pl.HW, gp);
    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    const char* src_n = static_cast<const char*>(a.src) + (int64_t)XSZ * ((int64_t)n * a.pl.HWp * C + cc0);
    const char* g_n = static_cast<const char*>(a.g.g) + (int64_t)GSZ * n * a.pl.HWp * a.g.ldg;
    const char* x_n = a.g.x ? static_cast<const char*>(a.g.x) + (int64_t)XSZ * n * a.pl.HWp * a.g.ldx : nullptr;
    // transposing-read geometry: lane i of a 16-lane group fetches pixel-row (k) i/4, channel quad i%4 and receives channel i
    const int tr_row = (lane & 15) >> 2, tr_col = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    const int tile0 = bx * a.tiles_per_wg;
    const int tile1 = min(tile0 + a.tiles_per_wg, a.n_tiles);
    for (int tile = tile0; tile < tile1; ++tile) {
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const int y0 = ty * G::TH, x0 = tx * TW;
        __syncthreads();                              // previous tile fully consumed (and prm visible)
        // (the next tile's loads issued behind this tile's LDS stores, in flight under its MFMAs: 68.4 -> 69.2 / 15.6 -> 16.0 us per launch -
        //  two workgroups per CU already cover the round trip, the 40 staging registers cost more)
        {
            float4 rv[B_N], rg[A_N], rgx[E == 4 ? 1 : A_N];
            bool okv[B_N], okg[A_N];
#pragma unroll
            for (int i = 0; i < B_N; ++i) {           // all loads in flight, then transform + split + store
                const int idx = t + 256 * i;
                const int hp = idx / SPP, q = idx % SPP;
                const int hy = hp / G::HW_, hx = hp - hy * G::HW_;
                const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
                okv[i] = idx < G::PX * SPP && (unsigned)iy < (unsigned)a.pl.H && (unsigned)ix < (unsigned)a.pl.W;
                rv[i] = ld16(src_n, (int64_t)XSZ * ((int64_t)(okv[i] ? iy * a.pl.W + ix : 0) * C + E * q));   // unconditional, clamped; zeroed at the store
            }
#pragma unroll
            for (int i = 0; i < A_N; ++i) {
                const int idx = t + 256 * i;
                const int px = idx / SPP, q = idx % SPP;
                const int py = y0 + px / TW, pxx = x0 + px % TW;
                okg[i] = idx < G::NPIX * SPP && py < a.pl.H && pxx < a.pl.W;          // tiles may hang over the edge
                const int64_t pix = okg[i] ? (int64_t)py * a.pl.W + pxx : 0;
                rg[i] = ld16(g_n, (int64_t)GSZ * (pix * a.g.ldg + E * q));
                if constexpr (E == 4) {
                    if (x_n) rg[i] = affine2(rg[i], ld16(x_n, (int64_t)XSZ * (pix * a.g.ldx + E * q)), gp + 4 * q, 32);
                } else {
                    if (x_n) rgx[i] = ld16(x_n, (int64_t)XSZ * (pix * a.g.ldx + E * q));
                }
            }
            const ActScale gsc = grad_scale();       // (behind this tile's loads; a few VALU per tile)
#pragma unroll
            for (int i = 0; i < B_N; ++i) {
                const int idx = t + 256 * i;
                if (idx < G::PX * SPP) {
                    const int q = idx % SPP, hp = idx / SPP;
                    if constexpr (E == 4) {
                        const Split4 sp = split4<OP>(okv[i] ? bnrelu4(rv[i], prm + 4 * q, 32) : zero4());   // zero padding AFTER bn+relu
#pragma unroll
                        for (int pc = 0; pc < NP; ++pc)
                            *reinterpret_cast<uint2*>(Bh + ((pc * G::PX + hp) * 32 + 4 * q) * 2) = sp.p[pc];
                    } else {
                        const float4 lo = okv[i] ? bnrelu4(slot_quad<XT>(rv[i], 0), prm + 8 * q, 32) : zero4();
                        const float4 hi = okv[i] ? bnrelu4(slot_quad<XT>(rv[i], 1), prm + 8 * q + 4, 32) : zero4();
                        *reinterpret_cast<u32x4*>(Bh + (hp * 32 + 8 * q) * 2) = pack_unit<OP>(lo, hi);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < A_N; ++i) {
                const int idx = t + 256 * i;
                if (idx < G::NPIX * SPP) {
                    const int q = idx % SPP, px = idx / SPP;
                    if constexpr (E == 4) {
                        const Split4 sp = split4<OP>(okg[i] ? (OP == 3 ? mul4(rg[i], gsc.s) : rg[i]) : zero4());
#pragma unroll
                        for (int pc = 0; pc < NP; ++pc)
                            *reinterpret_cast<uint2*>(Ag + ((pc * G::NPIX + px) * 32 + 4 * q) * 2) = sp.p[pc];
                    } else {
                        u32x4 u = as_u4(rg[i]);                                  // finished bf16 gradient: a copy
                        if (x_n) u = pack_unit<OP>(affine2(slot_quad<GT>(rg[i], 0), slot_quad<XT>(rgx[i], 0), gp + 8 * q, 32),
                                                   affine2(slot_quad<GT>(rg[i], 1), slot_quad<XT>(rgx[i], 1), gp + 8 * q + 4, 32));
                        if (!okg[i]) u = u32x4{0u, 0u, 0u, 0u};
                        *reinterpret_cast<u32x4*>(Ag + (px * 32 + 8 * q) * 2) = u;
                    }
                }
            }
        }
        __syncthreads();
        // wave w reduces over its k16-steps; this lane's 4-pixel run of the step starts at k = 8*half + tr_row (+4 for the second read)
#pragma unroll
        for (int ks = 0; ks < G::KSTEPS; ++ks) {
            const int k = 8 * half + tr_row;                     // pixel inside the step (second read: + 4)
            int ry, rx;
            if constexpr (TW == 16) { ry = wave * 2 + ks; rx = k; }
            else { ry = wave * 2 + (k >> 3); rx = k & 7; }
            auto tr2 = [&](const char* p0, int stride) -> u32x4 {        // 8 consecutive k of this lane's channel
                const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0));
                const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + stride)));
                return u32x4{lo.x, lo.y, hi.x, hi.y};
            };
            // second read = pixels k + 4: the same tile row for TW == 16 and TW == 8 (k + 4 < 8 within a row of 8)
            const char* ga = Ag + ((ry * TW + rx) * 32 + tr_col) * 2;
            const char* hb = Bh + ((ry * G::HW_ + rx) * 32 + tr_col) * 2;
            u32x4 af[NPIECE];
#pragma unroll
            for (int pc = 0; pc < NPIECE; ++pc)
                if (pc < NP) af[pc] = tr2(ga + pc * G::NPIX * 64, 4 * 64);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                u32x4 bf[3][NPIECE];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int pc = 0; pc < NPIECE; ++pc)
                        if (pc < NP) bf[dx][pc] = tr2(hb + pc * G::PX * 64 + (dy * G::HW_ + dx) * 64, 4 * 64);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (OP == 3) {                    // two fp16 pieces: h*l, l*h, h*h over the three taps of the kernel row
#pragma unroll
                    for (int g = 0; g < 3; ++g)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) acc[dy * 3 + dx] = mfma_f16(af[g == 1 ? 1 : 0], bf[dx][g == 0 ? 1 : 0], acc[dy * 3 + dx]);
                } else
                if constexpr (OP != 0) {                    // single-piece operands: one term per tap
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) acc[dy * 3 + dx] = mfma_1p<OP>(af[0], bf[dx][0], acc[dy * 3 + dx]);
                } else {
                    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                    for (int g = 0; g < 6; ++g)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) acc[dy * 3 + dx] = mfma_bf16(af[PA[g]], bf[dx][PB[g]], acc[dy * 3 + dx]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // flush: per tap, fold the four waves' tiles through LDS and add into the gradient
    const ActScale gsc = grad_scale();
    float* red = smem;                                  // [4][16][64]
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[tap][r];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = t + 256 * k;                  // e = r*64 + lane'
            const int r = e >> 6, ln = e & 63;
            const float v = (red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e]) * gsc.inv;      // (1 unless operand kind 3: per-stream scale, removed before streams are summed)
            const int nn = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), cc = ln & 31;
            a.part[((((int64_t)bx * a.streams + n) * 9 + tap) * 32 + nn) * C + cc0 + cc] = v;
        }
    }
}

}  // namespace smg
