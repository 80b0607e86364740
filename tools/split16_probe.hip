// Development probe (not part of the product): numerics of the 2-piece fp16 split (x*s = h + l, three v_mfma_f32_32x32x16_f16 terms
// hh + hl + lh) against the exact fp32 FMA chain (v_mfma_f32_32x32x2_f32) and the 3-piece bf16 split (six terms), all against fp64
// on the host; plus whether the fp16 MFMA honours SUBNORMAL inputs (the low piece lives there for small elements).
//   hipcc --offload-arch=gfx950 -O3 tools/split16_probe.hip -o gpurun_out/split16_probe && ./gpurun_out/split16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf_rne(float x) { unsigned u = __float_as_uint(x); u += 0x7FFFu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
__device__ __forceinline__ float bf_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ f32x16 mma_bf(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// V: 0 fp32 chain, 1 bf16x3 (6 terms), 2 fp16x2 RNE (3 terms), 3 fp16x2 RTZ pieces (3 terms), 4 fp16x2 RNE 4 terms (+ l*l)
template <int V>
__global__ void probe(const float* A, const float* B, float* D, int K, float sa, float sb) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (V == 0) {
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k + half], B[(k + half) * 32 + l31], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            float xa[8], xb[8];
            for (int j = 0; j < 8; ++j) { xa[j] = A[l31 * K + k + 8 * half + j]; xb[j] = B[(k + 8 * half + j) * 32 + l31]; }
            if (V == 1) {
                s16x8 ah, am, al, bh, bm, bl;
                for (int j = 0; j < 8; ++j) {
                    unsigned short a = bf_rne(xa[j]); float r1 = xa[j] - bf_f(a); unsigned short b = bf_rne(r1); float r2 = r1 - bf_f(b);
                    ah[j] = (short)a; am[j] = (short)b; al[j] = (short)bf_rne(r2);
                    a = bf_rne(xb[j]); r1 = xb[j] - bf_f(a); b = bf_rne(r1); r2 = r1 - bf_f(b);
                    bh[j] = (short)a; bm[j] = (short)b; bl[j] = (short)bf_rne(r2);
                }
                acc = mma_bf(ah, bl, acc); acc = mma_bf(al, bh, acc); acc = mma_bf(am, bm, acc);
                acc = mma_bf(ah, bm, acc); acc = mma_bf(am, bh, acc); acc = mma_bf(ah, bh, acc);
                continue;
            }
            f16x8 ah, al, bh, bl;
            for (int j = 0; j < 8; ++j) {
                const float ya = xa[j] * sa, yb = xb[j] * sb;
                if (V == 3) {
                    ah[j] = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ya, 0.f)) & 0xFFFF));
                    al[j] = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ya - (float)ah[j], 0.f)) & 0xFFFF));
                    bh[j] = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(yb, 0.f)) & 0xFFFF));
                    bl[j] = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(yb - (float)bh[j], 0.f)) & 0xFFFF));
                } else {
                    ah[j] = (_Float16)ya; al[j] = (_Float16)(ya - (float)ah[j]);
                    bh[j] = (_Float16)yb; bl[j] = (_Float16)(yb - (float)bh[j]);
                }
            }
            if (V == 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        }
    }
    const float inv = V >= 2 ? 1.0f / (sa * sb) : 1.0f;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        D[row * 32 + l31] = acc[r] * inv;
    }
}

// subnormal operands: A = 2^-20 (fp16 subnormal: 16 ulps of 2^-24), B = 2^10 -> every product 2^-10, sum over K=16: 2^-6
__global__ void subnormal_probe(float* out) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)9.5367431640625e-07f; b[j] = (_Float16)1024.0f; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; }
    // smallest subnormal 2^-24 times 2^14
    for (int j = 0; j < 8; ++j) { a[j] = __builtin_bit_cast(_Float16, (unsigned short)1); b[j] = (_Float16)16384.0f; }
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[2] = acc[0];      // expect 16 * 2^-10 = 0.015625
    // conversion of a value in the subnormal range
    const float tiny = 3.0e-6f;
    if (threadIdx.x == 0) out[3] = (float)(_Float16)tiny;
}

static double urand() { return (double)rand() / RAND_MAX; }
static double nrand() { return sqrt(-2.0 * log(1.0 - urand() * 0.999999)) * cos(6.283185307179586 * urand()); }
static float pow2_scale(const std::vector<float>& v, int target_exp) {      // s = 2^k with max|v| * s in [2^(target-1), 2^target)
    float m = 0; for (float x : v) m = fmaxf(m, fabsf(x));
    int e; frexpf(m, &e);            // m = f * 2^e, f in [0.5, 1)
    return ldexpf(1.0f, target_exp - e);
}

int main() {
    float* dO; hipMalloc(&dO, 64);
    subnormal_probe<<<1, 64>>>(dO);
    float o[4]; hipMemcpy(o, dO, 16, hipMemcpyDeviceToHost);
    printf("subnormal inputs: 16 x (2^-20 * 2^10) = %.9g (expect 0.015625); a[0] back = %.9g; 16 x (2^-24 * 2^14) = %.9g (expect 0.015625); cvt(3e-6) = %.9g\n", o[0], o[1], o[2], o[3]);
    const char* names[5] = {"fp32 chain", "bf16x3 6t", "fp16x2 RNE 3t", "fp16x2 RTZ 3t", "fp16x2 RNE 4t"};
    for (int K : {64, 128, 288, 1152}) {
        for (int mode = 0; mode < 4; ++mode) {
            // 0: relu'd BN activations x kaiming weights (sa = 32);  1: the same with SMALL activations (x 1e-3: the floor regime);
            // 2: gradient-like tiny values x weights (dynamic scale from the maximum);  3: gradient x activation (weight gradient)
            std::vector<float> A(32 * K), B(K * 32);
            srand(1234 + mode + K);
            for (auto& v : A) {
                double x = nrand();
                if (mode == 0) x = x > 0 ? x * 1.3 + 0.1 : 0.0;
                else if (mode == 1) x = x > 0 ? (x * 1.3 + 0.1) * 1e-3 : 0.0;
                else x *= 1e-6 * exp(2.0 * nrand());
                v = (float)x;
            }
            for (auto& v : B) {
                if (mode == 3) { double x = nrand(); v = (float)(x > 0 ? x * 1.3 + 0.1 : 0.0); }
                else v = (float)(nrand() * sqrt(2.0 / K));
            }
            const float sa = mode <= 1 ? 32.0f : pow2_scale(A, 14), sb = mode == 3 ? 32.0f : pow2_scale(B, 14);
            std::vector<double> ref(32 * 32), mag(32 * 32);
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double s = 0, m = 0;
                    for (int k = 0; k < K; ++k) { const double p = (double)A[i * K + k] * (double)B[k * 32 + j]; s += p; m += fabs(p); }
                    ref[i * 32 + j] = s; mag[i * 32 + j] = m;
                }
            float *dA, *dB, *dD;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 32 * 32 * 4);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            double rms0 = 0;
            for (int v = 0; v < 5; ++v) {
                switch (v) {
                    case 0: probe<0><<<1, 64>>>(dA, dB, dD, K, sa, sb); break;
                    case 1: probe<1><<<1, 64>>>(dA, dB, dD, K, sa, sb); break;
                    case 2: probe<2><<<1, 64>>>(dA, dB, dD, K, sa, sb); break;
                    case 3: probe<3><<<1, 64>>>(dA, dB, dD, K, sa, sb); break;
                    case 4: probe<4><<<1, 64>>>(dA, dB, dD, K, sa, sb); break;
                }
                std::vector<float> D(32 * 32);
                hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
                double worst = 0, rms = 0, bias = 0;
                for (int e = 0; e < 32 * 32; ++e) {
                    const double err = ((double)D[e] - ref[e]) / mag[e];
                    worst = fmax(worst, fabs(err)); rms += err * err; bias += err;
                }
                rms = sqrt(rms / 1024);
                if (v == 0) rms0 = rms;
                printf("K %4d mode %d sa %.3g sb %.3g %-14s err/sum|ab|: max %.3e rms %.3e (%.2fx chain) mean %+.3e\n", K, mode, sa, sb, names[v], worst, rms, rms / rms0, bias / 1024);
            }
            hipFree(dA); hipFree(dB); hipFree(dD);
        }
    }
    return 0;
}
