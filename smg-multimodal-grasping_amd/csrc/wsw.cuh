// wsw.cuh - wave-specialised 1x1 weight gradient of the dense layers (conv1: dW[128][cin] = sum over pixels of D2^T x relu(bn1(x))),
// precision mode 0 / operand kind 3.
//
// The generic kernel (gemm.cuh, BwdWeightP<128 x 64 x 16>) walks 16-pixel k-tiles - six MFMAs per wave between two barriers - with
// every wave doing everything in turn: wait for loads -> BN + ReLU + split -> LDS stores -> barrier -> transposing reads -> MFMAs.
// Its workgroups pull 12 KB per 1250-cycle k-tile, 8 KB of them the gradient operand that every 64-column tile of the same pixels
// re-reads: about 12 bytes per cycle and CU, which is what bounds it (0.39 of the HBM roof, the matrix pipe 16 % busy).
// Here a workgroup is EIGHT waves with two roles (the structure of ws.cuh):
//   waves 0..3  consumers: transposing fragment reads + the three-term MFMA blocks of k-tile kt (LDS buffer kt & 1), the epilogue
//   waves 4..7  producers: global loads two k-tiles ahead; the gradient operand arrives in UNIT form (gemm.cuh, kD2K8: finished
//               fp16 pieces, a straight 16-byte copy), the activation operand gets BN + ReLU + split; LDS stores of k-tile kt + 1
// on a 128 x 128 tile (the gradient operand is staged once per 128 columns instead of once per 64) with 32-pixel k-tiles: 24 MFMAs
// per consumer wave between two barriers.  Every SIMD holds a consumer and a producer wave of the workgroup.
// Same products as the generic kernel (same operands, per-block scales, k order per accumulator: summation order within a pixel chunk
// unchanged); partial tiles + reduce_partials_kernel or fp32 atomics as there.
#pragma once
#include "gemm.cuh"

namespace smg {

struct Wgrad1x1WsArgs {
    const u32x4* d2; const float* binv;            // gradient operand: units of the ring slot + [streams][HWp / 64] inverse block scales
    const float* x; int ldb; Plane pl; int NB;     // block buffer [n][HWp][ldb], the layer's first NB channels
    StatTab btab; const float* bgamma; const float* bbeta; const float* basc;      // norm1 of the layer: statistics table, affine, {s, 1 / s}
    int chunk, chunks_per_stream, n_chunks;        // pixel chunks (multiples of 64) per stream / in all
    float* dw; int ldw_out;                        // gradient [128][cin] (atomics) ...
    float* part; int ldp;                          // ... or partial tiles [chunk][128][ldp] for the fixed-order reduce
    TileMap tm;                                    // major = pixel chunk, minor = 128-column tile: the column tiles of a chunk share one XCD's L2
};

struct WswGeo {
    static constexpr int BM = 128, BN = 128, BK = 32;
    static constexpr int PLANE = BK + 4;                                  // A: [piece][channel / 8][PLANE] units (see gemm_tile, AU_PLANE)
    static constexpr int LDTB = BN + 32;                                  // B: row-major [piece][pixel][LDTB] fp16, row stride = 64 (mod 128) bytes
    static constexpr int A_BYTES = 2 * (BM / 8) * PLANE * 16, B_BYTES = 2 * BK * LDTB * 2;
    static constexpr int A_N = 2 * (BM / 8) * BK / 256, B_N = BK * (BN / 4) / 256;      // producer slots per k-tile: 4 units, 4 float4
    static constexpr int TILE_BYTES = 2 * (A_BYTES + B_BYTES);
    __host__ __device__ static constexpr int smem_bytes(int chunk) { return TILE_BYTES + (3 * BN + chunk / kScaleBlock + 1) * 4; }
};

static __global__ __launch_bounds__(512, 1) void conv1x1_wgrad_ws_kernel(const Wgrad1x1WsArgs a) {
    using G = WswGeo;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* As = reinterpret_cast<char*>(smem);
    char* Bs = As + 2 * G::A_BYTES;
    float* bp = reinterpret_cast<float*>(As + G::TILE_BYTES);             // mean | gamma * invstd * s | beta * s of the 128 columns
    float* sinv = bp + 3 * G::BN;                                         // inverse scales of the chunk's 64-pixel blocks
    const int t = threadIdx.x, role = t >> 8, tp = t & 255, lane = t & 63, wave = (t >> 6) & 3, half = lane >> 5;
    int z, nt;
    tile_decode(a.tm, blockIdx.x, z, nt);
    const int n = sgpr(z / a.chunks_per_stream);
    const int p0 = sgpr((z - n * a.chunks_per_stream) * a.chunk);
    if (p0 >= a.pl.HW) return;
    const int n0 = sgpr(nt * G::BN);
    int len = a.pl.HWp - p0;
    len = len < a.chunk ? len : a.chunk;
    const int KT = sgpr(len / G::BK);

    // ---- producers: loop-invariant lane offsets (descriptor loads: uniform base, per-k-tile scalar offset)
    const u32x4* d2n = a.d2 + d2_stream_units(n, a.pl.HWp);
    const float* xn = a.x + (int64_t)n * a.pl.HWp * a.ldb;
    const unsigned d2_bytes = 16u * (unsigned)(2 * kD2K8 * a.pl.HWp), x_bytes = 4u * (unsigned)(a.pl.HW * a.ldb);      // rows past the plane read as zero
    unsigned a_voff[G::A_N], b_voff[G::B_N];
#pragma unroll
    for (int i = 0; i < G::A_N; ++i) {
        const int id = tp + 256 * i, kr = id % G::BK, q = id / G::BK;      // q = piece * 16 + k8: consecutive lanes -> consecutive pixels of one plane
        a_voff[i] = 16u * (unsigned)(q * a.pl.HWp + kr);
    }
    const int bq = tp % (G::BN / 4), bl = tp / (G::BN / 4);              // channel quad of the tile, first pixel row; rows bl + 8 i
#pragma unroll
    for (int i = 0; i < G::B_N; ++i) b_voff[i] = 4u * (unsigned)((bl + 8 * i) * a.ldb + n0 + 4 * bq);
    u32x4 ra[2][G::A_N]; float4 rb[2][G::B_N];
    auto g_load = [&](int kt, u32x4 (&xa)[G::A_N], float4 (&xb)[G::B_N]) {
#pragma unroll
        for (int i = 0; i < G::A_N; ++i) xa[i] = bload_u4(d2n, d2_bytes, a_voff[i], 16u * (unsigned)(p0 + kt * G::BK));
#pragma unroll
        for (int i = 0; i < G::B_N; ++i) xb[i] = bload4(xn, x_bytes, b_voff[i], 4u * (unsigned)((p0 + kt * G::BK) * a.ldb));
    };
    KPrm3 bfix{};
    auto s_store = [&](int buf, const u32x4 (&xa)[G::A_N], const float4 (&xb)[G::B_N]) {
        char* A = As + buf * G::A_BYTES;
        char* B = Bs + buf * G::B_BYTES;
#pragma unroll
        for (int i = 0; i < G::A_N; ++i) {
            const int id = tp + 256 * i;
            *reinterpret_cast<u32x4*>(A + ((id / G::BK) * G::PLANE + id % G::BK) * 16) = xa[i];
        }
#pragma unroll
        for (int i = 0; i < G::B_N; ++i) {
            const Split4 s = split4<3>(bnrelu4<true>(xb[i], bfix));
            const int kr = bl + 8 * i;
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) *reinterpret_cast<uint2*>(B + ((pc * G::BK + kr) * G::LDTB + 4 * bq) * 2) = s.p[pc];
        }
    };
    if (role == 1) {                 // the first two tiles' loads go out before the parameter prologue
        g_load(0, ra[0], rb[0]);
        g_load(KT > 1 ? 1 : 0, ra[1], rb[1]);
    }
    // ---- parameters: BN + ReLU of the 128 columns (from the forward's table), the chunk's block scales
    if (t < G::BN) {
        const int ch = n0 + t;
        float mean = 0.f, sc = 0.f, be = 0.f;
        if (ch < a.NB) {
            const float sa = a.basc[0];
            mean = a.btab.mean[(int64_t)n * a.btab.ld + ch];
            sc = a.bgamma[ch] * a.btab.invstd[(int64_t)n * a.btab.ld + ch] * sa;
            be = a.bbeta[ch] * sa;
        }
        bp[t] = mean; bp[G::BN + t] = sc; bp[2 * G::BN + t] = be;
    }
    {
        const float* bi = a.binv + (int64_t)n * (a.pl.HWp / kScaleBlock) + p0 / kScaleBlock;
        for (int j = t; j * kScaleBlock < KT * G::BK; j += 512) sinv[j] = bi[j];
    }
    __syncthreads();
    if (role == 1) {
        bfix.mean = ldv4(bp + 4 * bq); bfix.scale = ldv4(bp + G::BN + 4 * bq); bfix.beta = ldv4(bp + 2 * G::BN + 4 * bq);
        s_store(0, ra[0], rb[0]);
    }
    __syncthreads();

    // ---- consumers: 2 x 2 waves, a 64 x 64 tile each (2 x 2 MFMA tiles)
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    const int tr_row = (lane & 15) >> 2, tr_col = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    auto tr2 = [&](const char* p, int stride) -> u32x4 {
        const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p));
        const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + stride)));
        return u32x4{lo.x, lo.y, hi.x, hi.y};
    };
    auto compute = [&](int buf) {
        const char* A = As + buf * G::A_BYTES;
        const char* B = Bs + buf * G::B_BYTES;
#pragma unroll
        for (int s = 0; s < G::BK / 16; ++s) {
            const int k0 = s * 16 + 8 * half + tr_row;
            u32x4 af[2][2], bf[2][2];        // [piece][tile]
#pragma unroll
            for (int pc = 0; pc < 2; ++pc)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ch = wm0 + i * 32 + tr_col;
                    af[pc][i] = tr2(A + ((pc * (G::BM / 8) + (ch >> 3)) * G::PLANE + k0) * 16 + ((ch >> 2) & 1) * 8, 4 * 16);
                    bf[pc][i] = tr2(B + ((pc * G::BK + k0) * G::LDTB + wn0 + i * 32 + tr_col) * 2, 4 * G::LDTB * 2);
                }
#pragma unroll
            for (int g = 0; g < 3; ++g)      // h*l, l*h, h*h (small terms first, tiles innermost: consecutive MFMAs never share an accumulator)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma_f16(af[g == 1 ? 1 : 0][i], bf[g == 0 ? 1 : 0][j], acc[i][j]);
        }
    };
    // the accumulators carry the scale of the 64-pixel block being reduced: at a block boundary bring them to the next block's.
    // A block whose scale lies kSkipBinades or more ABOVE the accumulators' (its largest gradient that many binades below the previous
    // block's) is skipped instead: the upward factor could push the sums past fp32's range (advisor, round 5), and such a block's
    // products sit below the last bit of any sum of comparable terms (an fp32 chain would lose them the same way; where every earlier
    // term of an output happened to be zero the plain chain keeps them - an error of 2^-64 of the larger blocks' magnitude).
    // (The generic kernel - few-stream calls, block 1's first layer - keeps the plain rescale: the same skip as a workgroup-uniform branch
    //  around its MFMA block measured +2 % on the single-sample step, 5.55-5.61 -> 5.68-5.71 ms; its factor overflows only when two
    //  neighbouring 64-pixel blocks of one stream differ by ~90 binades.)
    constexpr int kSkipBinades = 64;
    float cur_inv = sinv[0];
    bool skip = false;                       // (workgroup-uniform: sinv is the same for every lane)
    auto rescale = [&](int kt) {             // end of k-tile kt
        const int nx = (kt + 1) * G::BK;
        if (nx % kScaleBlock || kt + 1 >= KT) return;
        const float inv = sinv[nx / kScaleBlock];
        // (positive powers of two: exponent fields.  Not behind a block of scale 1 - that is what a block of zeros carries, and the sums
        //  it leaves are rescaled for whatever follows)
        skip = cur_inv != 1.f && (int)(__float_as_uint(cur_inv) >> 23) - (int)(__float_as_uint(inv) >> 23) >= kSkipBinades;
        if (inv != cur_inv && !skip) {
            const float f = cur_inv * __uint_as_float((254u << 23) - __float_as_uint(inv));      // cur_inv / inv, exact
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] *= f;
            cur_inv = inv;
        }
    };
    // ---- the k-loop: two k-tiles per trip (static buffers and register slots), one barrier per k-tile; each role runs its own loop
    if (role == 0) {
        int kt = 0;
        for (; kt + 2 <= KT; kt += 2) {
            if (!skip) compute(0);
            rescale(kt);
            __syncthreads();
            if (!skip) compute(1);
            rescale(kt + 1);
            __syncthreads();
        }
        if (kt < KT) { if (!skip) compute(0); __syncthreads(); }
    } else {
        g_load(KT > 2 ? 2 : KT - 1, ra[0], rb[0]);         // slot 0 is free again (tile 0 is in LDS)
        int kt = 0;
        for (; kt + 2 <= KT; kt += 2) {
            s_store(1, ra[1], rb[1]);                                           // tile kt + 1 (< KT here)
            g_load(kt + 3 < KT ? kt + 3 : KT - 1, ra[1], rb[1]);                // (tail: clamped re-loads)
            __syncthreads();
            s_store(0, ra[0], rb[0]);                                           // tile kt + 2 (at kt + 2 == KT: a dead store)
            g_load(kt + 4 < KT ? kt + 4 : KT - 1, ra[0], rb[0]);
            __syncthreads();
        }
        if (kt < KT) __syncthreads();
        return;
    }

    // ---- epilogue (consumers): exact power-of-two correction, then partial tile or atomics
    const float gi = cur_inv * a.basc[1];
    const int l31 = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn0 + j * 32 + l31;
            if (a.part) {        // (columns past NB are never read by the reduce: not stored)
                if (col >= a.NB) continue;
                float* ob = a.part + ((int64_t)z * G::BM + wm0 + i * 32 + 4 * half) * a.ldp + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) ob[(int64_t)((r & 3) + 8 * (r >> 2)) * a.ldp] = acc[i][j][r] * gi;
            } else if (col < a.NB) {
                float* ob = a.dw + (int64_t)(wm0 + i * 32 + 4 * half) * a.ldw_out + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) atomicAdd(ob + (int64_t)((r & 3) + 8 * (r >> 2)) * a.ldw_out, acc[i][j][r] * gi);
            }
        }
}

}  // namespace smg
