// Development probe (not part of the product): numerics of the split-precision MFMA schemes the convolution kernels
// use, measured on the GPU against an fp64 host reference.
//   V0  v_mfma_f32_32x32x2_f32                       (exact fp32 FMA chain - the round-1 arithmetic)
//   V1  bf16 x3 split (RNE pieces), 6 MFMAs           hh hm mh mm hl lh
//   V2  bf16 x3 split (truncated pieces), 6 MFMAs
//   V3  bf16 x2 split (RNE), 3 MFMAs                  hh hl lh
//   V4  bf16 x3 split (RNE), 9 MFMAs                  all products
//   V5  plain bf16 (RNE), 1 MFMA
//   V6  plain fp16 (RNE), 1 MFMA
//   hipcc --offload-arch=gfx950 -O3 tools/split_probe.hip -o gpurun_out/split_probe && ./gpurun_out/split_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ unsigned short bf_trunc(float x) { return (unsigned short)(__float_as_uint(x) >> 16); }
__device__ __forceinline__ float bf_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

template <bool RNE>
__device__ __forceinline__ void split3(const float (&x)[8], s16x8& h, s16x8& m, s16x8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned short a = RNE ? bf_rne(x[j]) : bf_trunc(x[j]);
        const float r1 = x[j] - bf_f(a);
        const unsigned short b = RNE ? bf_rne(r1) : bf_trunc(r1);
        const float r2 = r1 - bf_f(b);
        const unsigned short c = RNE ? bf_rne(r2) : bf_trunc(r2);
        h[j] = (short)a; m[j] = (short)b; l[j] = (short)c;
    }
}

__device__ __forceinline__ f32x16 mma(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// A [32][K] row-major, B [K][32] row-major, D [32][32]
template <int V>
__global__ void probe(const float* A, const float* B, float* D, int K) {
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
            if (V == 6) {
                f16x8 a, b;
                for (int j = 0; j < 8; ++j) { a[j] = (_Float16)xa[j]; b[j] = (_Float16)xb[j]; }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
                continue;
            }
            s16x8 ah, am, al, bh, bm, bl;
            if (V == 2) { split3<false>(xa, ah, am, al); split3<false>(xb, bh, bm, bl); }
            else { split3<true>(xa, ah, am, al); split3<true>(xb, bh, bm, bl); }
            if (V == 4) { acc = mma(al, bl, acc); acc = mma(am, bl, acc); acc = mma(al, bm, acc); }
            if (V == 1 || V == 2 || V == 4) { acc = mma(ah, bl, acc); acc = mma(al, bh, acc); acc = mma(am, bm, acc); }
            if (V == 3) { acc = mma(ah, bm, acc); acc = mma(am, bh, acc); }      // 2-piece: "m" is the low piece
            if (V == 1 || V == 2 || V == 4) { acc = mma(ah, bm, acc); acc = mma(am, bh, acc); }
            acc = mma(ah, bh, acc);
        }
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        D[row * 32 + l31] = acc[r];
    }
}

static double urand() { return (double)rand() / RAND_MAX; }
static double nrand() { return sqrt(-2.0 * log(1.0 - urand() * 0.999999)) * cos(6.283185307179586 * urand()); }

int main() {
    const char* names[7] = {"fp32 mfma", "bf16x3 RNE 6", "bf16x3 trunc 6", "bf16x2 RNE 3", "bf16x3 RNE 9", "bf16 x1", "fp16 x1"};
    for (int K : {128, 1024}) {
        for (int mode = 0; mode < 3; ++mode) {   // 0: relu'd activations x kaiming weights; 1: signed wide dynamic range; 2: gradient-like tiny values
            std::vector<float> A(32 * K), B(K * 32);
            srand(1234 + mode);
            for (auto& v : A) {
                double x = nrand();
                if (mode == 0) x = x > 0 ? x * 1.3 + 0.1 : 0.0;
                else if (mode == 1) x *= exp(4.0 * nrand());
                else x *= 1e-7 * exp(2.0 * nrand());
                v = (float)x;
            }
            for (auto& v : B) v = (float)(nrand() * sqrt(2.0 / K) * (mode == 1 ? exp(2.0 * nrand()) : 1.0));
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
            for (int v = 0; v < 7; ++v) {
                switch (v) {
                    case 0: probe<0><<<1, 64>>>(dA, dB, dD, K); break;
                    case 1: probe<1><<<1, 64>>>(dA, dB, dD, K); break;
                    case 2: probe<2><<<1, 64>>>(dA, dB, dD, K); break;
                    case 3: probe<3><<<1, 64>>>(dA, dB, dD, K); break;
                    case 4: probe<4><<<1, 64>>>(dA, dB, dD, K); break;
                    case 5: probe<5><<<1, 64>>>(dA, dB, dD, K); break;
                    case 6: probe<6><<<1, 64>>>(dA, dB, dD, K); break;
                }
                std::vector<float> D(32 * 32);
                hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
                double worst = 0, rms = 0, bias = 0;
                for (int e = 0; e < 32 * 32; ++e) {
                    const double err = ((double)D[e] - ref[e]) / mag[e];      // relative to sum |a*b|
                    worst = fmax(worst, fabs(err)); rms += err * err; bias += err;
                }
                printf("K %4d mode %d %-16s err/sum|ab|: max %.3e rms %.3e mean %+.3e\n", K, mode, names[v], worst, sqrt(rms / 1024), bias / 1024);
            }
            hipFree(dA); hipFree(dB); hipFree(dD);
        }
    }
    return 0;
}
