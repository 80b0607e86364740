// ws.cuh - wave-specialised 1x1 forward convolution (norm1 + relu + conv1 of a dense layer) for launches that cannot fill
// the chip with 128-row tiles (the 40^2 / 20^2 planes).
//
// The generic kernel (gemm.cuh) gives every wave the same program: fragment reads -> MFMAs -> wait for the next tile's
// loads -> BN + ReLU + split -> LDS stores -> barrier.  Inside one workgroup those phases are serial, the chip overlaps them
// only across workgroups, and LDS admits two of those per CU: the per-workgroup stamps show 1500 cycles per 64x64x32
// k-tile against 384 of MFMA issue.  Here a workgroup is EIGHT waves with two roles:
//   waves 0..3  consumers: fragment reads + the three-term (operand kind 3; six-term: kind 0) MFMA blocks of k-tile kt (LDS buffer kt & 1), the epilogue
//   waves 4..7  producers: global loads (two k-tiles in flight), BN + ReLU + split, LDS stores of k-tile kt + 1
// One barrier per k-tile.  Every SIMD holds one consumer and one producer wave of the workgroup, so the MFMA pipe works
// while the VALU splits the next tile - inside one workgroup, whatever else is resident.
// Same arithmetic, LDS images and epilogue as FwdConvP<GemmCfg<64, 64, 32, 2, 2, 1, true>, F_ONE> (results equal to fp32
// summation order: the k order per accumulator is unchanged).
#pragma once
#include "gemm.cuh"

namespace smg {

struct Fwd1x1WsArgs {
    const float* src; int lds_; Plane pl;          // [n][HWp][lds_] block buffer, K = first K channels
    int K;
    BnTab bt; int fresh0; const double* fsum; const double* fsq; int fstride; float eps;
    float* tw_mean; float* tw_invstd;
    const u32x4* wp; int N;                        // packed weight units [piece][K/8][N]; operand kind 3: header unit in front
    const float* asc;                              // operand kind 3: {s, 1 / s} of the BN + ReLU operand
    float* dst; int ldd;                           // raw output [n][HWp][ldd]
    double* dsum; double* dsq; int dstride;        // per-(stream, channel) sum / sum of squares
    TileMap tm;                                    // XCD-aware order: the N tiles of one M tile are consecutive on one XCD
};

template <int NP_, int BN_ = 64>
struct WsGeoT {
    static constexpr int NP = NP_;                                    // pieces per operand (3: bf16 split, 2: fp16 split)
    static constexpr int BM = 64, BN = BN_, BK = 32, K8 = BK / 8;
    static constexpr int TN = BN / 64;                                // 32-column accumulator tiles per consumer wave (2 x 2 waves)
    static constexpr int LDUA = BM + 2, LDUB = BN;                    // A rows padded as in GemmCfg (64 / BK units)
    static constexpr int A_BYTES = NP * K8 * LDUA * 16, B_BYTES = NP * K8 * LDUB * 16;
    static constexpr int A_N = BM * (BK / 4) / 256;                   // 2 float4 per producer thread
    static constexpr int B_N = NP * K8 * BN / 256;                    // units per producer thread
    static constexpr int TILE_BYTES = 2 * (A_BYTES + B_BYTES);
    __host__ __device__ static constexpr int smem_bytes(int K) { return TILE_BYTES + 3 * K * 4; }
};

using WsGeo = WsGeoT<np_of(fwd_op(0))>;
// BN = 128 (round 6): ONE workgroup per 64-row tile covers all 128 output columns - the activations are loaded, BN + ReLU'd and split
// once instead of once per 64-column tile (the producers' instruction work per output halves; the small planes' waves are
// issue-bound), each consumer wave carries two accumulator tiles.  62 KB of LDS at K = 992: two per CU,
// 425 workgroups of a 17-stream 40^2 launch in one round.
using WsGeo128 = WsGeoT<np_of(fwd_op(0)), 128>;

template <int PREC = 0, int BN = 64>
static __global__ __launch_bounds__(512, 1) void conv1x1_fwd_ws_kernel(const Fwd1x1WsArgs a) {
    static_assert(PREC == 0, "fp32 storage only");
    using G = WsGeoT<np_of(fwd_op(0)), BN>;
    constexpr int TN = G::TN;
    constexpr int OP = fwd_op(PREC), NP = G::NP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* As = reinterpret_cast<char*>(smem);
    char* Bs = As + 2 * G::A_BYTES;
    float* sp = reinterpret_cast<float*>(As + G::TILE_BYTES);         // mean | gamma*invstd | beta, K each
    const int t = threadIdx.x, role = t >> 8, tp = t & 255, lane = t & 63, wave = (t >> 6) & 3;
    const int l31 = lane & 31, half = lane >> 5;
    int mt, nt;
    if (!tile_decode(a.tm, blockIdx.x, mt, nt)) return;
    const int m0 = sgpr(mt * G::BM), n0 = sgpr(nt * G::BN);
    const int n = sgpr(m0 / a.pl.HWp);
    const int pbase = m0 - n * a.pl.HWp;
    if (pbase >= a.pl.HW) return;                                     // tile made of padding rows only
    const int K = a.K, KT = K / G::BK;

    // ---- producers: staging geometry (loop-invariant lane offsets; descriptor loads, see gemm.cuh)
    const int aq = tp % (G::BK / 4), al = tp / (G::BK / 4);          // channel quad, first row; rows al, al + 32
    const float* a_base = a.src + (int64_t)n * a.pl.HWp * a.lds_;
    unsigned a_voff[G::A_N];
#pragma unroll
    for (int i = 0; i < G::A_N; ++i) a_voff[i] = 4u * (unsigned)((pbase + al + 32 * i) * a.lds_ + 4 * aq);
    unsigned b_voff[G::B_N];
#pragma unroll
    for (int i = 0; i < G::B_N; ++i) {
        const int id = tp + 256 * i, r = id % G::BN, pk = id / G::BN;   // pk = piece * K8 + k8
        b_voff[i] = 16u * (unsigned)(((pk / G::K8) * (K / 8) + pk % G::K8) * a.N + r);
    }
    float4 ra[2][G::A_N]; u32x4 rb[2][G::B_N];
    auto g_load = [&](int kt, float4 (&xa)[G::A_N], u32x4 (&xb)[G::B_N]) {
#pragma unroll
        for (int i = 0; i < G::A_N; ++i) xa[i] = bload4(a_base, kWholeBuf, a_voff[i], 4u * (unsigned)(kt * G::BK));
#pragma unroll
        for (int i = 0; i < G::B_N; ++i) xb[i] = bload_u4(a.wp, kWholeBuf, b_voff[i], 16u * (unsigned)(kt * G::K8 * a.N + n0));
    };
    auto s_store = [&](int buf, int kt, const float4 (&xa)[G::A_N], const u32x4 (&xb)[G::B_N]) {
        char* A = As + buf * G::A_BYTES;
        char* B = Bs + buf * G::B_BYTES;
        const int ch = kt * G::BK + 4 * aq;
        KPrm3 f;
        f.mean = ldv4(sp + ch); f.scale = ldv4(sp + K + ch); f.beta = ldv4(sp + 2 * K + ch);
#pragma unroll
        for (int i = 0; i < G::A_N; ++i) {
            const Split4 s = split4<OP>(bnrelu4<OP == 3>(xa[i], f));
            const int row = al + 32 * i;
#pragma unroll
            for (int pc = 0; pc < NP; ++pc)
                *reinterpret_cast<uint2*>(A + ((pc * G::K8 + (aq >> 1)) * G::LDUA + row) * 16 + (aq & 1) * 8) = s.p[pc];
        }
#pragma unroll
        for (int i = 0; i < G::B_N; ++i) *reinterpret_cast<u32x4*>(B + (tp + 256 * i) * 16) = xb[i];
    };
    if (role == 1) {                 // the first two tiles' loads go out before the parameter prologue
        g_load(0, ra[0], rb[0]);
        g_load(KT > 1 ? 1 : 0, ra[1], rb[1]);
    }
    // ---- BN parameters of the K input channels -> LDS (all 512 threads; the fresh channels from the fp64 sums)
    {
        const float* tmean = tab_mean(a.bt, n);
        const float* tinv = tab_invstd(a.bt, n);
        const float sa = OP == 3 ? a.asc[0] : 1.f;                       // operand kind 3: the activation scale rides on gamma * invstd and beta
        for (int ch = 4 * t; ch < K && ch < a.fresh0; ch += 2048) {
            const f32x4 m = ldv4(tmean + ch), iv = ldv4(tinv + ch), g = ldv4(a.bt.gamma + ch), be = ldv4(a.bt.beta + ch);
            *reinterpret_cast<f32x4*>(sp + ch) = m;
            *reinterpret_cast<f32x4*>(sp + K + ch) = g * iv * sa;
            *reinterpret_cast<f32x4*>(sp + 2 * K + ch) = be * sa;
        }
        if (a.fresh0 < K && t < 32) {
            const int ch = a.fresh0 + t;
            float mean, invstd;
            bn_moments(a.fsum, a.fsq, (int64_t)n * a.fstride + ch, 1.0 / (double)a.pl.HW, a.eps, mean, invstd);
            sp[ch] = mean;
            sp[K + ch] = a.bt.gamma[ch] * invstd * sa;
            sp[2 * K + ch] = a.bt.beta[ch] * sa;
            if (n0 == 0 && pbase == 0) {
                a.tw_mean[(int64_t)n * a.bt.ld + ch] = mean;
                a.tw_invstd[(int64_t)n * a.bt.ld + ch] = invstd;
            }
        }
    }
    __syncthreads();
    if (role == 1) s_store(0, 0, ra[0], rb[0]);
    __syncthreads();

    // ---- consumers: accumulator and fragment geometry (2 x 2 waves, one 32 x 32 tile each)
    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32 * TN;
    auto compute = [&](int buf) {
        const char* A = As + buf * G::A_BYTES;
        const char* B = Bs + buf * G::B_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int k8 = 2 * s + half;
            u32x4 af[NPIECE], bf[TN][NPIECE];
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
                af[pc] = *reinterpret_cast<const u32x4*>(A + ((pc * G::K8 + k8) * G::LDUA + wm0 + l31) * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j][pc] = *reinterpret_cast<const u32x4*>(B + ((pc * G::K8 + k8) * G::LDUB + wn0 + 32 * j + l31) * 16);
            }
            if constexpr (OP == 3) {             // term groups outermost, tiles innermost: consecutive MFMAs never share an accumulator (TN = 2)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = mfma_f16(af[0], bf[j][1], acc[j]);
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = mfma_f16(af[1], bf[j][0], acc[j]);
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = mfma_f16(af[0], bf[j][0], acc[j]);
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[j] = mfma_bf16(af[0], bf[j][2], acc[j]);
                    acc[j] = mfma_bf16(af[2], bf[j][0], acc[j]);
                    acc[j] = mfma_bf16(af[1], bf[j][1], acc[j]);
                    acc[j] = mfma_bf16(af[0], bf[j][1], acc[j]);
                    acc[j] = mfma_bf16(af[1], bf[j][0], acc[j]);
                    acc[j] = mfma_bf16(af[0], bf[j][0], acc[j]);
                }
            }
        }
    };
    // ---- the k-loop: two k-tiles per trip (static buffers and register slots), one barrier per k-tile.  Each role runs its
    // OWN loop (the same number of barriers): inside the producers' loop every path issues the same number of loads, so hipcc
    // counts vmcnt exactly and a store waits only for its own tile, not for the tile requested half a trip ago.
    if (role == 0) {
        int kt = 0;
        for (; kt + 2 <= KT; kt += 2) {
            compute(0);
            __syncthreads();
            compute(1);
            __syncthreads();
        }
        if (kt < KT) { compute(0); __syncthreads(); }       // odd KT: the last tile sits in buffer 0
    } else {
        g_load(KT > 2 ? 2 : KT - 1, ra[0], rb[0]);         // slot 0 is free again (tile 0 is in LDS)
        int kt = 0;
        for (; kt + 2 <= KT; kt += 2) {
            s_store(1, kt + 1, ra[1], rb[1]);                                   // tile kt + 1 (< KT here)
            g_load(kt + 3 < KT ? kt + 3 : KT - 1, ra[1], rb[1]);                // (tail: clamped re-loads)
            __syncthreads();
            s_store(0, kt + 2 < KT ? kt + 2 : KT - 1, ra[0], rb[0]);            // at kt + 2 == KT: a dead store
            g_load(kt + 4 < KT ? kt + 4 : KT - 1, ra[0], rb[0]);
            __syncthreads();
        }
        if (kt < KT) __syncthreads();
    }

    // ---- epilogue (consumers): raw output + per-(stream, channel) sum / sum of squares (fp64)
    double v0[TN], v1[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) v0[j] = v1[j] = 0.0;
    if (role == 0) {
        if constexpr (OP == 3) {               // products of scaled operands: exact power-of-two correction
            const float inv = a.asc[1] * pack_inv_scale(a.wp);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] *= inv;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cj = wn0 + 32 * j + l31, col = n0 + cj;
            if (pbase + G::BM <= a.pl.HW) {
                // whole tile inside the plane: shifted fp32 sums per 16-row strip, widened once (see FwdConvP::epilogue)
                float* tb = a.dst + (int64_t)m0 * a.ldd + n0;
                unsigned o = (unsigned)((wm0 + 4 * half) * a.ldd + cj);
                const float s = acc[j][0];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float x = acc[j][r];
                    tb[o] = x;
                    o += (r & 3) == 3 ? 5u * (unsigned)a.ldd : (unsigned)a.ldd;
                    const float dx = x - s;
                    s1 += dx;
                    s2 = fmaf(dx, dx, s2);
                }
                const double sd = (double)s, s1d = (double)s1;
                v0[j] = s1d + 16.0 * sd;
                v1[j] = (double)s2 + 2.0 * sd * s1d + 16.0 * sd * sd;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (pbase + row < a.pl.HW) {
                        const float x = acc[j][r];
                        a.dst[(int64_t)(m0 + row) * a.ldd + col] = x;
                        const double xd = (double)x;
                        v0[j] += xd;
                        v1[j] += xd * xd;
                    }
                }
            }
            v0[j] += __shfl_xor(v0[j], 32);
            v1[j] += __shfl_xor(v1[j], 32);
        }
    }
    double* red = reinterpret_cast<double*>(smem);       // [2 quantities][2 row-waves][BN columns]
    if (role == 0 && half == 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            red[((0 * 2) + (wave >> 1)) * G::BN + wn0 + 32 * j + l31] = v0[j];
            red[((1 * 2) + (wave >> 1)) * G::BN + wn0 + 32 * j + l31] = v1[j];
        }
    }
    __syncthreads();
    if (t < 2 * G::BN) {
        const int q = t / G::BN, c = t % G::BN;
        const double tot = red[(q * 2) * G::BN + c] + red[(q * 2 + 1) * G::BN + c];
        atomicAdd((q ? a.dsq : a.dsum) + (int64_t)n * a.dstride + n0 + c + fstat_rep(), tot);
    }
}

}  // namespace smg
