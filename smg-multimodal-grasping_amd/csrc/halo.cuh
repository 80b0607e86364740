// halo.cuh - 3x3 convolutions of the dense layers with an LDS-resident input halo.
//
// The generic implicit GEMM (gemm.cuh, F_THREE) re-fetches every input pixel once per
// tap: with only growth = 32 output channels per layer that is ~20 B/clk/CU of
// global->LDS traffic at full MFMA rate, and the kernel saturates the load path at a
// third of the fp32 MFMA peak.  Here a workgroup owns a 16x16 tile of output pixels,
// stages the 18x18 halo of BN+ReLU'd input ONCE per 16-channel chunk, and walks the
// 9 taps by shifting the LDS read address: 9x fewer global loads, BN transforms and
// LDS writes per MFMA.
//
//   out[p][n] = sum_{tap, c} relu(bn(in[p + d(tap)][c])) * W[n][c][tap]
//
// 4 waves, wave w owns pixel rows 4w..4w+3 (two 32x32 MFMA tiles: 2 rows x 16 cols
// each) x all 32 output channels.  LDS: A[2][324 px][17] (pixel-major, odd stride ->
// conflict-free fragment reads), B[2][9*16][32].  Double-buffered over channel chunks;
// operand fragments of tap t+1 are fetched while the MFMAs of tap t run.
#pragma once
#include "gemm.cuh"

namespace smg {

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

constexpr int HALO_T = 16;                 // tile side
constexpr int HALO_W = HALO_T + 2;         // 18
constexpr int HALO_PX = HALO_W * HALO_W;   // 324
constexpr int HALO_CK = 16;                // channels per chunk
constexpr int HALO_LDA = HALO_CK + 1;      // 17
constexpr int HALO_A_FLOATS = HALO_PX * HALO_LDA;          // 5508
constexpr int HALO_B_FLOATS = 9 * HALO_CK * 32;            // 4608
constexpr int HALO_A_N = (HALO_PX * (HALO_CK / 4) + 255) / 256;   // float4 per thread: 6
constexpr int HALO_B_N = (9 * HALO_CK * 8 + 255) / 256;           // 5

__global__ __launch_bounds__(256) void conv3x3_halo_fwd_kernel(const Halo3x3FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                   // [2][HALO_A_FLOATS] (+pad to 16 B)
    float* Bs = smem + 2 * 5512;                        // [2][HALO_B_FLOATS]
    float* prm = Bs + 2 * HALO_B_FLOATS;                // mean | scale | beta, C each
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, half = lane >> 5;
    const int n = blockIdx.y;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
    const int y0 = ty * HALO_T, x0 = tx * HALO_T;
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
    int a_off[HALO_A_N];       // global float offset of the pixel (without channel), -1 = outside the image / unused
    int a_lds[HALO_A_N];       // LDS float offset hp*17 + 4*kq
    int a_kq[HALO_A_N];
#pragma unroll
    for (int i = 0; i < HALO_A_N; ++i) {
        const int idx = t + 256 * i;
        const int hp = idx >> 2, kq = idx & 3;
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = idx < HALO_PX * 4 && (unsigned)iy < (unsigned)a.pl.H && (unsigned)ix < (unsigned)a.pl.W;
        a_off[i] = ok ? (iy * a.pl.W + ix) : -1;
        a_lds[i] = (idx < HALO_PX * 4) ? hp * HALO_LDA + 4 * kq : -1;
        a_kq[i] = kq;
    }
    const float* src_n = a.src + (int64_t)n * a.pl.HWp * a.lds_;
    float4 ra[HALO_A_N], rb[HALO_B_N];
    auto g_load = [&](int chunk) {
        const int c0 = chunk * HALO_CK;
#pragma unroll
        for (int i = 0; i < HALO_A_N; ++i)
            ra[i] = (a_off[i] >= 0) ? ld4(src_n + (int64_t)a_off[i] * a.lds_ + c0 + 4 * a_kq[i]) : zero4();
#pragma unroll
        for (int i = 0; i < HALO_B_N; ++i) {
            const int idx = t + 256 * i;
            const int row = idx >> 3, q = idx & 7;                 // row = tap*16 + cc
            const int tap = row >> 4, cc = row & 15;
            rb[i] = (idx < 9 * HALO_CK * 8) ? ld4(a.w + ((int64_t)(tap * C + c0 + cc)) * 32 + 4 * q) : zero4();
        }
    };
    auto s_store = [&](int buf, int chunk) {
        float* A = As + buf * 5512;
        float* B = Bs + buf * HALO_B_FLOATS;
        const int c0 = chunk * HALO_CK;
#pragma unroll
        for (int i = 0; i < HALO_A_N; ++i) {
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

    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    // fragment base addresses: MFMA tile m covers pixel rows 4*wave + 2m + (i >> 4), col i & 15
    int abase[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) abase[m] = ((4 * wave + 2 * m + (l31 >> 4)) * HALO_W + (l31 & 15)) * HALO_LDA + half;
    const int bbase = half * 32 + l31;

    const int NCH = C / HALO_CK;
    __syncthreads();                 // prm visible
    g_load(0);
    s_store(0, 0);
    __syncthreads();
    for (int ch = 0; ch < NCH; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < NCH) g_load(ch + 1);
        const float* A = As + buf * 5512;
        const float* B = Bs + buf * HALO_B_FLOATS;
        float fa[2][8][2], fb[2][8];
        auto frag = [&](int set, int tap) {
            const int toff = ((tap / 3) * HALO_W + (tap % 3)) * HALO_LDA;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                fa[set][kk][0] = A[abase[0] + toff + 2 * kk];
                fa[set][kk][1] = A[abase[1] + toff + 2 * kk];
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
            for (int kk = 0; kk < 8; ++kk) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][kk][0], fb[set][kk], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][kk][1], fb[set][kk], acc[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ch + 1 < NCH) s_store(buf ^ 1, ch + 1);
        __syncthreads();
    }

    // epilogue: raw output + per-(stream, channel) sum / sum of squares (fp64)
    double s = 0.0, ss = 0.0;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int py = y0 + 4 * wave + 2 * m + (i >> 4), px = x0 + (i & 15);
            const float x = acc[m][r];
            a.dst[((int64_t)n * a.pl.HWp + py * a.pl.W + px) * a.ldd + a.dcoff + l31] = x;
            const double xd = (double)x;
            s += xd;
            ss += xd * xd;
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

}  // namespace smg
