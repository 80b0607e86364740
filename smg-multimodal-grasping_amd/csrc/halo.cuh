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
// Two tile sizes:
//   TS = 16  planes that tile by 16 (160^2, 80^2 at S = 640).  4 waves, wave w owns pixel
//            rows 4w..4w+3 (two 32x32 MFMA tiles: 2 rows x 16 cols each).
//   TS = 8   everything else (40^2, 20^2, ragged edges masked): 64 pixels per workgroup so
//            that a 17-stream launch still has hundreds of workgroups.  Two waves split
//            the pixels (4 rows x 8 cols each, one MFMA tile); the other factor of two
//            splits the reduction (forward: halves of every channel chunk, folded through
//            LDS at the end), the output channels (data gradient) or the pixels again
//            (weight gradient).
// Forward LDS: A[(TS+2)^2 px][17] (pixel-major, odd stride -> conflict-free fragment
// reads), B[9*16][32]: 42 KB / 27 KB, so several workgroups share a CU and one's
// staging / epilogue hides under another's MFMAs.  The next chunk is prefetched into
// registers during the MFMAs; operand fragments of tap t+1 are fetched while the MFMAs
// of tap t run.
#pragma once
#include "gemm.cuh"

namespace smg {

template <int TS>
struct HaloGeo {
    static_assert(TS == 16 || TS == 8, "tile side");
    static constexpr int T = TS, W = TS + 2, PX = W * W, NPIX = TS * TS;
    static constexpr int MT = TS == 16 ? 2 : 1;        // 32-pixel MFMA tiles per wave
    static constexpr int WQ = NPIX / (32 * MT);        // waves that split the pixels (4 / 2)
    static constexpr int WX = 4 / WQ;                  // the other wave factor (1 / 2)
    static constexpr int ROWS = 32 / TS;               // pixel rows per MFMA tile (2 / 4)
    // pixel (row, col) inside the tile of MFMA-tile row i (0..31) of tile m of pixel-wave wq
    __device__ static __forceinline__ int row(int wq, int m, int i) { return (wq * MT + m) * ROWS + i / TS; }
    __device__ static __forceinline__ int col(int i) { return i % TS; }
};

struct Halo3x3FwdArgs {
    const float* src; int lds_; Plane pl;           // [n][HWp][C] raw bottleneck output
    int C;                                          // input channels (128)
    const double* ssum; const double* ssq; int sstride;
    const float* gamma; const float* beta; float eps;
    const float* w;                                 // packed [(tap*C + c)][32]
    float* dst; int ldd, dcoff;
    double* dsum; double* dsq; int dstride;
    int tiles_x;
};

constexpr int HALO_T = 16;                 // tile side of the big-plane variant
constexpr int HALO_CK = 16;                // channels per chunk
constexpr int HALO_LDA = HALO_CK + 1;      // 17
constexpr int HALO_B_FLOATS = 9 * HALO_CK * 32;            // 4608
constexpr int HALO_B_N = (9 * HALO_CK * 8 + 255) / 256;    // 5

template <int TS> struct HaloFwdGeo : HaloGeo<TS> {
    using G = HaloGeo<TS>;
    static constexpr int A_FLOATS = (G::PX * HALO_LDA + 7) / 8 * 8;      // keeps B 32-byte aligned
    static constexpr int A_N = (G::PX * (HALO_CK / 4) + 255) / 256;      // float4 per thread: 6 / 2
    static constexpr int KS = G::WX;                                     // waves splitting each chunk's channels
    static constexpr int KK = 8 / KS;                                    // MFMA k-steps per wave per (chunk, tap)
    __host__ __device__ static constexpr int smem_floats(int C) { return A_FLOATS + HALO_B_FLOATS + 3 * C; }
};

template <int TS>
__global__ __launch_bounds__(256, TS == 16 ? 3 : 4) void conv3x3_halo_fwd_kernel(const Halo3x3FwdArgs a) {
    using G = HaloFwdGeo<TS>;
    constexpr int MT = G::MT, KK = G::KK, A_N = G::A_N;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                   // [PX][17] (+pad to 16 B)
    float* Bs = smem + G::A_FLOATS;                     // [HALO_B_FLOATS]
    float* prm = Bs + HALO_B_FLOATS;                    // mean | scale | beta, C each
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, half = lane >> 5;
    const int wq = wave % G::WQ, wk = wave / G::WQ;     // pixel slice, channel-split slice
    const int n = blockIdx.y;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
    const int y0 = ty * TS, x0 = tx * TS;
    const int C = a.C;

    {   // BN parameters of this stream
        const double inv = 1.0 / (double)a.pl.HW;
        for (int k = t; k < C; k += 256) {
            float mean, invstd;
            bn_moments(a.ssum, a.ssq, (int64_t)n * a.sstride + k, inv, a.eps, mean, invstd);
            prm[k] = mean;
            prm[C + k] = a.gamma[k] * invstd;
            prm[2 * C + k] = a.beta[k];
        }
    }
    // staging slots of this thread (fixed across chunks)
    int a_off[A_N];       // global float offset of the pixel (without channel), -1 = outside the image / unused
    int a_lds[A_N];       // LDS float offset hp*17 + 4*kq
    int a_kq[A_N];
#pragma unroll
    for (int i = 0; i < A_N; ++i) {
        const int idx = t + 256 * i;
        const int hp = idx >> 2, kq = idx & 3;
        const int hy = hp / G::W, hx = hp - hy * G::W;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = idx < G::PX * 4 && (unsigned)iy < (unsigned)a.pl.H && (unsigned)ix < (unsigned)a.pl.W;
        a_off[i] = ok ? (iy * a.pl.W + ix) : -1;
        a_lds[i] = (idx < G::PX * 4) ? hp * HALO_LDA + 4 * kq : -1;
        a_kq[i] = kq;
    }
    const float* src_n = a.src + (int64_t)n * a.pl.HWp * a.lds_;
    float4 ra[A_N], rb[HALO_B_N];
    auto g_load = [&](int chunk) {
        const int c0 = chunk * HALO_CK;
#pragma unroll
        // Unconditional loads from clamped addresses (out-of-image slots read pixel 0 and are zeroed at the
        // LDS store): a branch around a load makes hipcc drain vmcnt(0) in the middle of the load group.
        for (int i = 0; i < A_N; ++i)
            ra[i] = ld4(src_n + (int64_t)(a_off[i] < 0 ? 0 : a_off[i]) * a.lds_ + c0 + 4 * a_kq[i]);
#pragma unroll
        for (int i = 0; i < HALO_B_N; ++i) {
            const int idx = min(t + 256 * i, 9 * HALO_CK * 8 - 1);
            const int row = idx >> 3, q = idx & 7;                 // row = tap*16 + cc
            const int tap = row >> 4, cc = row & 15;
            rb[i] = ld4(a.w + ((int64_t)(tap * C + c0 + cc)) * 32 + 4 * q);
        }
    };
    auto s_store = [&](int chunk) {
        float* A = As;
        float* B = Bs;
        const int c0 = chunk * HALO_CK;
#pragma unroll
        for (int i = 0; i < A_N; ++i) {
            if (a_lds[i] < 0) continue;
            float4 v = zero4();                                     // conv zero padding applies AFTER bn+relu
            if (a_off[i] >= 0) v = bnrelu4(ra[i], prm + c0 + 4 * a_kq[i], C);
            float* d = A + a_lds[i];
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
#pragma unroll
        for (int i = 0; i < HALO_B_N; ++i) {
            const int idx = t + 256 * i;
            if (idx < 9 * HALO_CK * 8) *reinterpret_cast<float4*>(B + idx * 4) = rb[i];
        }
    };

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    // fragment base addresses: MFMA tile m, tile row l31 -> pixel (G::row, G::col); this wave's
    // share of every chunk starts at channel 2*KK*wk
    int abase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) abase[m] = (G::row(wq, m, l31) * G::W + G::col(l31)) * HALO_LDA + half + 2 * KK * wk;
    const int bbase = (half + 2 * KK * wk) * 32 + l31;

    const int NCH = C / HALO_CK;
    __syncthreads();                 // prm visible
    g_load(0);
    s_store(0);
    __syncthreads();
    for (int ch = 0; ch < NCH; ++ch) {
        if (ch + 1 < NCH) g_load(ch + 1);
        const float* A = As;
        const float* B = Bs;
        float fa[2][KK][MT], fb[2][KK];
        auto frag = [&](int set, int tap) {
            const int toff = ((tap / 3) * G::W + (tap % 3)) * HALO_LDA;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[set][kk][m] = A[abase[m] + toff + 2 * kk];
                fb[set][kk] = B[(tap * HALO_CK + 2 * kk) * 32 + bbase];
            }
        };
        frag(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int set = tap & 1;
            if (tap + 1 < 9) frag(set ^ 1, tap + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][kk][m], fb[set][kk], acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                          // every wave is done reading this chunk
        if (ch + 1 < NCH) {
            s_store(ch + 1);
            __syncthreads();
        }
    }
    if constexpr (G::KS > 1) {                    // fold the channel-split partial tiles into the wk == 0 waves
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
    double s = 0.0, ss = 0.0;
    if (wk == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int py = y0 + G::row(wq, m, i), px = x0 + G::col(i);
                if (TS == 16 || (py < a.pl.H && px < a.pl.W)) {      // TS == 8 tiles may hang over the edge
                    const float x = acc[m][r];
                    a.dst[((int64_t)n * a.pl.HWp + py * a.pl.W + px) * a.ldd + a.dcoff + l31] = x;
                    const double xd = (double)x;
                    s += xd;
                    ss += xd * xd;
                }
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
        atomicAdd((q ? a.dsq : a.dsum) + (int64_t)n * a.dstride + a.dcoff + c, tot);
    }
}


// ------------------------------------------------------------------------------------
// 3x3 data gradient (transposed convolution) with the gradient halo resident in LDS:
//   dc[p][c] = sum_{tap, n} g[p - d(tap)][n] * W[n][c][tap]          (g: [pixel][32])
// followed by the ReLU mask and BN(norm2) backward sums of the bottleneck (the same
// epilogue as BwdDataP/E_STORE).  The (TS+2)^2 x 32 halo of g is staged once per
// workgroup; the weights of one (32-channel output chunk, tap row) - 12 KB - are streamed
// through LDS, prefetched into registers under the MFMAs.  TS = 16: 12 stages, 57 KB LDS
// -> 2 workgroups/CU.  TS = 8: the two wave pairs own different output-channel chunks
// (NCW = 2 chunks per stage, 6 stages), 39 KB.
// ------------------------------------------------------------------------------------
// The finished gradient of a layer's 32 output channels, as the 3x3 backward kernels read it: either a dense
// [n][HWp][32] array (x == nullptr), or the G' and X slices of the block buffers with the deferred BN backward
// applied on load (what bn_bwd_apply_kernel would have written):
//   g = invstd * ((G' - SA/n) - (x - mean) * invstd * SB/n)
// Statistics pointers are already offset to the slice's first channel; sstride = floats per stream.
struct GradSrc {
    const float* g; int ldg;
    const float* x; int ldx;
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
        gp[32 + t] = (float)(s.s1[(int64_t)n * s.sstride + t] * inv);
        gp[64 + t] = mean;
        gp[96 + t] = invstd * (float)(s.s2[(int64_t)n * s.sstride + t] * inv);
    }
}

struct Halo3x3DgradArgs {
    GradSrc g; Plane pl;                             // finished output gradient (32 channels)
    const float* w;                                  // packed [(tap*32 + n)][C]
    int C;                                           // bottleneck channels (128)
    const float* mbuf;                               // raw bottleneck [n][HWp][C] (mask + xhat source)
    const double* msum; const double* msq; int mstride;
    const float* gamma; const float* beta; float eps;
    float* dst;                                      // dy [n][HWp][C]
    double* o1; double* o2; int ostride;             // per-stream sums [n][C]
    int tiles_x;
    int cg_per_wg;                                   // output-channel groups (NCW chunks of 32) per workgroup; blockIdx.z picks the run
};

constexpr int HD_LDA = 33;
constexpr int HD_B_CHUNK = 3 * 32 * 32;                     // one tap row x 32 n x 32 c

template <int TS> struct HaloDgradGeo : HaloGeo<TS> {
    using G = HaloGeo<TS>;
    static constexpr int A_FLOATS = (G::PX * HD_LDA + 3) / 4 * 4;
    static constexpr int A_N = (G::PX * 8 + 255) / 256;                  // float4 per thread: 11 / 4
    static constexpr int NCW = G::WX;                                    // output-channel chunks per stage
    static constexpr int B_FLOATS = NCW * HD_B_CHUNK;
    static constexpr int B_N = B_FLOATS / 4 / 256;                       // 3 / 6
    __host__ __device__ static constexpr int smem_floats(int C) { return A_FLOATS + B_FLOATS + 4 * C + 256 + 128; }
};

template <int TS>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_dgrad_kernel(const Halo3x3DgradArgs a) {
    using G = HaloDgradGeo<TS>;
    constexpr int MT = G::MT, NCW = G::NCW, A_N = G::A_N, B_N = G::B_N;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                   // [PX][33]
    float* Bs = smem + G::A_FLOATS;                     // [NCW][96][32]
    float* prm = Bs + G::B_FLOATS;                      // scale | beta | mean | invstd, C each
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, half = lane >> 5;
    const int wq = wave % G::WQ, wc = wave / G::WQ;     // pixel slice, output-channel chunk of the stage
    const int n = blockIdx.y;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
    const int y0 = ty * TS, x0 = tx * TS;
    const int C = a.C;
    {
        const double inv = 1.0 / (double)a.pl.HW;
        for (int k = t; k < C; k += 256) {
            float mean, invstd;
            bn_moments(a.msum, a.msq, (int64_t)n * a.mstride + k, inv, a.eps, mean, invstd);
            prm[k] = a.gamma[k] * invstd;
            prm[C + k] = a.beta[k];
            prm[2 * C + k] = mean;
            prm[3 * C + k] = invstd;
        }
    }
    float* gp = prm + 4 * C + 256;                     // GradSrc parameters [4][32]
    grad_src_params(a.g, n, a.pl.HW, gp);
    const int cg0 = blockIdx.z * a.cg_per_wg;
    const int NSTAGE = a.cg_per_wg * 3;       // channel-chunk groups x 3 tap rows
    float4 rb[B_N];
    auto g_load = [&](int stage) {            // stage = cgroup*3 + tap row
        const int cg = cg0 + stage / 3, dy = stage % 3;
#pragma unroll
        for (int i = 0; i < B_N; ++i) {
            const int idx = t + 256 * i;      // NCW x 768 float4: j = chunk of the stage, row = dx*32 + nn, q
            const int j = idx / 768, rem = idx - j * 768;
            const int row = rem >> 3, q = rem & 7;
            rb[i] = ld4(a.w + (int64_t)((dy * 3) * 32 + row) * C + (cg * NCW + j) * 32 + 4 * q);
        }
    };
    // (issuing g_load(0) here, under the halo staging, costs registers: spills at 3 waves/SIMD)
    // gradient halo (zero outside the image)
    const float* g_n = a.g.g + (int64_t)n * a.pl.HWp * a.g.ldg;
    const float* x_n = a.g.x ? a.g.x + (int64_t)n * a.pl.HWp * a.g.ldx : nullptr;
    {
        float4 rv[A_N], rx[A_N];
#pragma unroll
        for (int i = 0; i < A_N; ++i) {            // every load in flight before the first LDS store
            const int idx = t + 256 * i;
            const int hp = idx >> 3, q = idx & 7;
            const int hy = hp / G::W, hx = hp - hy * G::W;
            const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            const bool ok = idx < G::PX * 8 && (unsigned)iy < (unsigned)a.pl.H && (unsigned)ix < (unsigned)a.pl.W;
            const int64_t pix = ok ? iy * a.pl.W + ix : 0;                 // unconditional loads, clamped address
            rv[i] = ld4(g_n + pix * a.g.ldg + 4 * q);
            if (x_n) rx[i] = ld4(x_n + pix * a.g.ldx + 4 * q);
        }
        __syncthreads();                           // gp (and prm) visible
#pragma unroll
        for (int i = 0; i < A_N; ++i) {
            const int idx = t + 256 * i;
            if (idx < G::PX * 8) {
                const int hp = idx >> 3, q = idx & 7;
                const int hy = hp / G::W, hx = hp - hy * G::W;
                const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
                const bool ok = (unsigned)iy < (unsigned)a.pl.H && (unsigned)ix < (unsigned)a.pl.W;
                float4 v = rv[i];
                if (x_n) v = affine2(rv[i], rx[i], gp + 4 * q, 32);
                if (!ok) v = zero4();
                float* d = As + hp * HD_LDA + 4 * q;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        }
    }
    auto s_store = [&]() {
#pragma unroll
        for (int i = 0; i < B_N; ++i) *reinterpret_cast<float4*>(Bs + (t + 256 * i) * 4) = rb[i];
    };
    int abase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) abase[m] = (G::row(wq, m, l31) * G::W + G::col(l31)) * HD_LDA + half;
    const int bbase = wc * HD_B_CHUNK + half * 32 + l31;

    g_load(0);
    s_store();
    __syncthreads();
    f32x16 acc[MT];
    float xvp[MT][16];                        // TS == 8: prefetched one stage ahead
    auto load_mask = [&](int c, float (&xv)[MT][16]) {   // mask / xhat source of this wave's output tile
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int py = y0 + G::row(wq, m, i), px = x0 + G::col(i);
                const bool ok = TS == 16 || (py < a.pl.H && px < a.pl.W);     // TS == 8 tiles may hang over the edge
                xv[m][r] = ok ? a.mbuf[((int64_t)n * a.pl.HWp + py * a.pl.W + px) * C + c] : 0.f;
            }
    };
    for (int stage = 0; stage < NSTAGE; ++stage) {
        const int dy = stage % 3;
        if (dy == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        }
        if (stage + 1 < NSTAGE) g_load(stage + 1);
        if constexpr (TS == 8) {
            if (dy == 2) load_mask(((cg0 + stage / 3) * NCW + wc) * 32 + l31, xvp);   // in flight under the last MFMA block
        }
        float fa[2][8][MT], fb[2][8];
        // output pixel (ry, rx), tap (dy, dx) reads g at halo (ry + 2 - dy, rx + 2 - dx)
        auto frag = [&](int set, int step) {          // step = dx*2 + (n half): 16 n per step
            const int dx = step >> 1, nh = step & 1;
            const int toff = ((2 - dy) * G::W + (2 - dx)) * HD_LDA + nh * 16;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
                for (int m = 0; m < MT; ++m) fa[set][kk][m] = As[abase[m] + toff + 2 * kk];
                fb[set][kk] = Bs[(dx * 32 + nh * 16 + 2 * kk) * 32 + bbase];
            }
        };
        frag(0, 0);
#pragma unroll
        for (int step = 0; step < 6; ++step) {
            const int set = step & 1;
            if (step + 1 < 6) frag(set ^ 1, step + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][kk][m], fb[set][kk], acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                          // B of this stage fully consumed
        if (stage + 1 < NSTAGE) s_store();
        if (dy == 2) {
            // epilogue of this wave's output-channel chunk: ReLU mask, store dy, BN(norm2) backward sums
            const int c = ((cg0 + stage / 3) * NCW + wc) * 32 + l31;
            const float sc = prm[c], be = prm[C + c], mean = prm[2 * C + c], invstd = prm[3 * C + c];
            float s1 = 0.f, s2 = 0.f;
            auto finish = [&](float (&xv)[MT][16]) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
                        const int py = y0 + G::row(wq, m, i), px = x0 + G::col(i);
                        if (TS == 8 && (py >= a.pl.H || px >= a.pl.W)) continue;
                        const float dyv = bn1(xv[m][r], mean, sc, be) > 0.f ? acc[m][r] : 0.f;
                        a.dst[((int64_t)n * a.pl.HWp + py * a.pl.W + px) * C + c] = dyv;
                        s1 += dyv;
                        s2 += dyv * ((xv[m][r] - mean) * invstd);
                    }
            };
            if constexpr (TS == 16) {
                float xv[MT][16];
                load_mask(c, xv);             // all loads before the stores (which may alias them)
                finish(xv);
            } else {
                finish(xvp);
            }
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
                const int ch = ((cg0 + stage / 3) * NCW + j) * 32 + cc;
                atomicAdd((q ? a.o2 : a.o1) + (int64_t)n * a.ostride + ch, (double)tot);
                // dgamma / dbeta = the same sums over streams: bn_bwd_apply_kernel adds them from o1 / o2 (one
                // atomic per stream and channel; here every tile of every stream would hit the same 2*C addresses)
            }
        }
        __syncthreads();
    }
}


// ------------------------------------------------------------------------------------
// 3x3 weight gradient with the activation halo resident in LDS:
//   dW[n][c][tap] += sum_p g[p][n] * relu(bn(in[p + d(tap)][c]))
// A workgroup owns (a run of TSxTS pixel tiles of one stream) x (a 32-channel chunk of
// c).  Per tile it stages the (TS+2)^2 x 32 activation halo (BN+ReLU applied once) and
// the TS^2 x 32 gradient tile; every wave then walks its own quarter of the pixels as the
// MFMA reduction dimension and accumulates ALL nine taps (9 accumulator tiles) by shifting
// the halo read address - the operands are fetched once for 9 taps instead of once per tap.
// The accumulators persist across the run of tiles; at the end the four waves' tiles
// are folded through LDS and written (plain stores) to a partial buffer that
// reduce_partials_kernel sums over workgroups - fp32 atomics here cost ~3 ms/step.
// ------------------------------------------------------------------------------------
struct Halo3x3WgradArgs {
    GradSrc g; Plane pl;                             // finished output gradient (32 channels)
    const float* src; int C;                         // raw bottleneck [n][HWp][C]
    const double* ssum; const double* ssq; int sstride;
    const float* gamma; const float* beta; float eps;
    float* part;                                     // partial sums [gridDim.x*gridDim.z][9][32][C]
    int tiles_x, n_tiles, tiles_per_wg;
};

template <int TS> struct HaloWgradGeo : HaloGeo<TS> {
    using G = HaloGeo<TS>;
    static constexpr int B_FLOATS = G::PX * 32;                          // activation halo
    static constexpr int A_FLOATS = G::NPIX * 32;                        // gradient tile
    static constexpr int B_N = (G::PX * 8 + 255) / 256;                  // 11 / 4
    static constexpr int A_N = G::NPIX * 8 / 256;                        // 8 / 2
    static constexpr int KSTEPS = G::NPIX / 4 / 2;                       // MFMA k-steps per wave per tile: 32 / 8
    static constexpr int WROWS = TS / 4;                                 // pixel rows per wave: 4 / 2
    static constexpr int NPH = TS == 16 ? 4 : 1;                         // staging phases per tile
    static constexpr int RED_FLOATS = 4 * 16 * 64;                       // flush area
    __host__ __device__ static constexpr int smem_floats() {
        return (B_FLOATS + A_FLOATS > RED_FLOATS ? B_FLOATS + A_FLOATS : RED_FLOATS) + 96 + 128;
    }
};

template <int TS>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_wgrad_kernel(const Halo3x3WgradArgs a) {
    using G = HaloWgradGeo<TS>;
    constexpr int B_N = G::B_N, A_N = G::A_N;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Bh = smem;                                   // [PX][32] activation halo
    float* Ag = smem + G::B_FLOATS;                     // [TS*TS][32] gradient tile
    float* prm = smem + G::smem_floats() - 96 - 128;    // mean | scale | beta (32 each)
    float* gp = prm + 96;                               // GradSrc parameters [4][32]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, half = lane >> 5;
    const int n = blockIdx.z, cc0 = blockIdx.y * 32;
    const int C = a.C;
    if (t < 32) {
        float mean, invstd;
        bn_moments(a.ssum, a.ssq, (int64_t)n * a.sstride + cc0 + t, 1.0 / (double)a.pl.HW, a.eps, mean, invstd);
        prm[t] = mean;
        prm[32 + t] = a.gamma[cc0 + t] * invstd;
        prm[64 + t] = a.beta[cc0 + t];
    }
    grad_src_params(a.g, n, a.pl.HW, gp);
    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    const float* src_n = a.src + (int64_t)n * a.pl.HWp * C + cc0;
    const float* g_n = a.g.g + (int64_t)n * a.pl.HWp * a.g.ldg;
    const float* x_n = a.g.x ? a.g.x + (int64_t)n * a.pl.HWp * a.g.ldx : nullptr;
    const int tile0 = blockIdx.x * a.tiles_per_wg;
    const int tile1 = min(tile0 + a.tiles_per_wg, a.n_tiles);
    for (int tile = tile0; tile < tile1; ++tile) {
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const int y0 = ty * TS, x0 = tx * TS;
        __syncthreads();                              // previous tile fully consumed (and prm visible)
#pragma unroll
        for (int ph = 0; ph < G::NPH; ++ph) {         // staging in NPH phases: fewer registers live -> 2 workgroups / CU
            constexpr int BP = (B_N + G::NPH - 1) / G::NPH, AP = A_N / G::NPH;
            float4 rv[BP], rg[AP];
            bool okv[BP];
#pragma unroll
            for (int i = 0; i < BP; ++i) {            // all loads of the phase in flight, then transform + store
                const int idx = t + 256 * (ph * BP + i);
                const int hp = idx >> 3, q = idx & 7;
                const int hy = hp / G::W, hx = hp - hy * G::W;
                const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
                okv[i] = idx < G::PX * 8 && (unsigned)iy < (unsigned)a.pl.H && (unsigned)ix < (unsigned)a.pl.W;
                rv[i] = ld4(src_n + (int64_t)(okv[i] ? iy * a.pl.W + ix : 0) * C + 4 * q);   // unconditional, clamped; zeroed at the store
            }
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const int idx = t + 256 * (ph * AP + i);
                const int px = idx >> 3, q = idx & 7;
                const int py = y0 + px / TS, pxx = x0 + px % TS;
                const bool ok = TS == 16 || (py < a.pl.H && pxx < a.pl.W);     // TS == 8 tiles may hang over the edge
                const int64_t pix = ok ? (int64_t)py * a.pl.W + pxx : 0;
                float4 v = ld4(g_n + pix * a.g.ldg + 4 * q);
                if (x_n) v = affine2(v, ld4(x_n + pix * a.g.ldx + 4 * q), gp + 4 * q, 32);
                rg[i] = ok ? v : zero4();
            }
#pragma unroll
            for (int i = 0; i < BP; ++i) {
                const int idx = t + 256 * (ph * BP + i);
                if (idx < G::PX * 8) {
                    const int q = idx & 7;
                    const float4 v = okv[i] ? bnrelu4(rv[i], prm + 4 * q, 32) : zero4();   // zero padding AFTER bn+relu
                    *reinterpret_cast<float4*>(Bh + (idx >> 3) * 32 + 4 * q) = v;
                }
            }
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const int idx = t + 256 * (ph * AP + i);
                *reinterpret_cast<float4*>(Ag + (idx >> 3) * 32 + 4 * (idx & 7)) = rg[i];
            }
        }
        __syncthreads();
        // wave w reduces over its quarter of the pixels (rows WROWS*w ..); lane half selects the pixel parity
        float fa[2], fb[2][9];
        auto frag = [&](int set, int kk) {
            const int pw = 2 * kk + half;
            const int ry = G::WROWS * wave + pw / TS, rx = pw % TS;
            fa[set] = Ag[(ry * TS + rx) * 32 + l31];
            const float* b = Bh + (ry * G::W + rx) * 32 + l31;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) fb[set][tap] = b[((tap / 3) * G::W + (tap % 3)) * 32];
        };
        frag(0, 0);
#pragma unroll 2
        for (int kk = 0; kk < G::KSTEPS; ++kk) {
            const int set = kk & 1;
            if (kk + 1 < G::KSTEPS) frag(set ^ 1, kk + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set], fb[set][tap], acc[tap], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // flush: per tap, fold the four waves' tiles through LDS and add into the gradient
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
            const float v = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
            const int nn = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), cc = ln & 31;
            a.part[((((int64_t)blockIdx.x * gridDim.z + n) * 9 + tap) * 32 + nn) * C + cc0 + cc] = v;
        }
    }
}

}  // namespace smg
