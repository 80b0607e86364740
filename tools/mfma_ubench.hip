// Development microbenchmark (not part of the product): what does the k-tile loop of gemm.cuh cost,
// layer by layer?  V0 pure v_mfma_f32_32x32x2_f32; V1 + LDS fragment reads per k-tile; V2 + LDS
// writes + one barrier per k-tile; V3 + global B-tile load per k-tile.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_ubench.hip -o gpurun_out/mfma_ubench && ./gpurun_out/mfma_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 16, LDA = BM + 1, LDB = BN + 4, KK = BK / 2;

template <int V, int TM, int TN>
__global__ __launch_bounds__(256) void k(const float* __restrict__ w, float* out, int ktiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * BK * LDA;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave >> 1) * TM * 32, wn0 = (wave & 1) * TN * 32;
    for (int i = t; i < 2 * BK * LDA + 2 * BK * LDB; i += 256) smem[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 rb[2];
    const int bq = t % 32, bl = t / 32;
    for (int kt = 0; kt < ktiles; ++kt) {
        const int buf = kt & 1;
        if (V >= 3) {
            for (int i = 0; i < 2; ++i) rb[i] = *reinterpret_cast<const float4*>(w + ((size_t)((kt * BK + bl + 8 * i) & 1023)) * 128 + 4 * bq);
        }
        float af[KK][TM], bf[KK][TN];
        if (V >= 1) {
            const float* A = As + buf * BK * LDA + half * LDA + wm0 + l31;
            const float* B = Bs + buf * BK * LDB + half * LDB + wn0 + l31;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[kk][i] = A[2 * kk * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[kk][j] = B[2 * kk * LDB + j * 32];
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                for (int i = 0; i < TM; ++i) af[kk][i] = (float)(lane + kk + i);
                for (int j = 0; j < TN; ++j) bf[kk][j] = (float)(lane - kk + j);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][i], bf[kk][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (V >= 2) {
            float* A = As + (buf ^ 1) * BK * LDA;
            float* B = Bs + (buf ^ 1) * BK * LDB;
            const int aq = t % 4, al = t / 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = al + 64 * i;
                const float v = acc[0][0][i] * 1e-30f;
                A[(aq * 4 + 0) * LDA + row] = v; A[(aq * 4 + 1) * LDA + row] = v + 1.f;
                A[(aq * 4 + 2) * LDA + row] = v + 2.f; A[(aq * 4 + 3) * LDA + row] = v + 3.f;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float4 v = (V >= 3) ? rb[i] : make_float4(1.f, 2.f, 3.f, 4.f);
                *reinterpret_cast<float4*>(&B[(bl + 8 * i) * LDB + bq * 4]) = v;
            }
            __syncthreads();
        }
    }
    if (V == 4) {          // epilogue A: what gemm.cuh does - one 4-byte store per accumulator element
        float* o = out + (size_t)blockIdx.x * BM * BN;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, col = wn0 + j * 32 + l31;
                    o[row * BN + col] = acc[i][j][r];
                }
        return;
    }
    if (V == 6 || V == 7) {   // scalar stores + fp64 column statistics (+ fp64 atomics for V7), as FwdConvP::epilogue
        float* o = out + (size_t)blockIdx.x * BM * BN;
        double v0[TN], v1[TN];
        for (int j = 0; j < TN; ++j) v0[j] = v1[j] = 0.0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, col = wn0 + j * 32 + l31;
                    const float x = acc[i][j][r];
                    o[row * BN + col] = x;
                    const double xd = (double)x;
                    v0[j] += xd; v1[j] += xd * xd;
                }
        double* red = reinterpret_cast<double*>(smem);
        for (int j = 0; j < TN; ++j) { v0[j] += __shfl_xor(v0[j], 32); v1[j] += __shfl_xor(v1[j], 32); }
        __syncthreads();
        if (half == 0) for (int j = 0; j < TN; ++j) { red[(0 * 2 + (wave >> 1)) * BN + wn0 + j * 32 + l31] = v0[j]; red[(1 * 2 + (wave >> 1)) * BN + wn0 + j * 32 + l31] = v1[j]; }
        __syncthreads();
        if (t < BN) {
            const double s0 = red[t] + red[BN + t], s1 = red[2 * BN + t] + red[3 * BN + t];
            double* st = reinterpret_cast<double*>(out) + (size_t)3400 * BM * BN / 2 + ((blockIdx.x / 200) * 256 + t) * 2;
            if (V == 7) { atomicAdd(st, s0); atomicAdd(st + 1, s1); }
            else if (s0 == 1.2345) { st[0] = s0; st[1] = s1; }
        }
        return;
    }
    if (V == 5) {          // epilogue B: transpose through LDS, 16-byte stores of whole rows
        float* o = out + (size_t)blockIdx.x * BM * BN;
        float* T = smem;                                  // [64][BN + 4]
        constexpr int LDT = BN + 4;
#pragma unroll
        for (int hrow = 0; hrow < 2; ++hrow) {            // 64 tile rows at a time (33 KB of LDS)
            __syncthreads();
            if ((wave >> 1) == hrow) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, col = wn0 + j * 32 + l31;
                            T[row * LDT + col] = acc[i][j][r];
                        }
            }
            __syncthreads();
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                const int idx = t + 256 * k2;             // 64 rows x 32 float4
                const int row = idx >> 5, q = idx & 31;
                *reinterpret_cast<float4*>(o + (size_t)(hrow * 64 + row) * BN + 4 * q) = *reinterpret_cast<const float4*>(T + row * LDT + 4 * q);
            }
        }
        return;
    }
    float s = 0.f;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 123.456f) out[blockIdx.x * 256 + t] = s;
}

template <int V, int TM, int TN>
void run(const char* name, float* w, float* out, int blocks, int ktiles) {
    const size_t smem = (2 * BK * LDA + 2 * BK * LDB) * sizeof(float) + 3 * 1024;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<V, TM, TN>), dim3(blocks), dim3(256), smem, 0, w, out, ktiles);
    hipEventRecord(a, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<V, TM, TN>), dim3(blocks), dim3(256), smem, 0, w, out, ktiles);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    const double flops = (double)blocks * 4 * ktiles * KK * TM * TN * 2.0 * 32 * 32 * 2;
    printf("%-40s blocks %5d ktiles %4d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ktiles, ms, flops / ms / 1e9);
}

int main() {
    float *w, *out;
    hipMalloc(&w, 1024 * 128 * 4); hipMalloc(&out, (size_t)3400 * 128 * 128 * 4 + (1 << 20));
    hipMemset(w, 0, 1024 * 128 * 4);
    for (int blocks : {768, 3400}) {
        for (int kt : {14, 64}) {
            run<0, 2, 2>("V0 pure MFMA 2x2", w, out, blocks, kt);
            run<1, 2, 2>("V1 +LDS fragment reads", w, out, blocks, kt);
            run<2, 2, 2>("V2 +LDS writes +barrier", w, out, blocks, kt);
            run<3, 2, 2>("V3 +global B loads", w, out, blocks, kt);
        }
    }
    for (int kt : {4, 14}) {
        run<3, 2, 2>("V3 (no output)", w, out, 3400, kt);
        run<4, 2, 2>("V4 scalar-store epilogue", w, out, 3400, kt);
        run<5, 2, 2>("V5 LDS-transposed float4 epilogue", w, out, 3400, kt);
        run<6, 2, 2>("V6 scalar stores + fp64 stats", w, out, 3400, kt);
        run<7, 2, 2>("V7 ... + fp64 atomics", w, out, 3400, kt);
    }
    run<0, 1, 1>("V0 pure MFMA 1x1", w, out, 3400, 64);
    run<1, 1, 1>("V1 1x1 +LDS reads", w, out, 3400, 64);
    run<2, 1, 1>("V2 1x1 +writes+barrier", w, out, 3400, 64);
    return 0;
}
