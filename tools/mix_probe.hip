// dev probe: the fp16 residual of the two-piece split through v_fma_mixlo/hi_f16 (one instruction per element) against the
// convert / subtract / convert form - must be bit-identical for every input, subnormal residuals included.
// build: hipcc --offload-arch=gfx950 -O3 tools/mix_probe.hip -o tools/mix_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_f16(float a, float b) { const f32x2 v = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2)); }
__device__ __forceinline__ float f16_lo(unsigned u) { return (float)__builtin_bit_cast(f16x2, u).x; }
__device__ __forceinline__ float f16_hi(unsigned u) { return (float)__builtin_bit_cast(f16x2, u).y; }
__device__ __forceinline__ unsigned resid_mix(float a, float b, unsigned h) {
    unsigned l = 0u;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(a), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "v"(h));
    return l;
}
__global__ void k(const float* x, unsigned* ref, unsigned* mix, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    const unsigned h = pack_f16(a, b);
    ref[i] = pack_f16(a - f16_lo(h), b - f16_hi(h));
    mix[i] = resid_mix(a, b, h);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(2 * n);
    srand(1);
    for (int i = 0; i < 2 * n; ++i) {
        const int e = rand() % 46 - 30;                         // magnitudes 2^-30 .. 2^15 (fp16 overflow excluded: the scales exclude it)
        const float m = 1.f + (float)rand() / (float)RAND_MAX;
        x[i] = ldexpf(m, e) * ((rand() & 1) ? 1.f : -1.f);
        if (i % 97 == 0) x[i] = 0.f;
    }
    float* dx; unsigned *dr, *dm;
    hipMalloc((void**)&dx, 2 * n * 4); hipMalloc((void**)&dr, n * 4); hipMalloc((void**)&dm, n * 4);
    hipMemcpy(dx, x.data(), 2 * n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dr, dm, n);
    std::vector<unsigned> r(n), m(n);
    hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost); hipMemcpy(m.data(), dm, n * 4, hipMemcpyDeviceToHost);
    long bad = 0, sub = 0;
    for (int i = 0; i < n; ++i) {
        if (r[i] != m[i]) { if (bad < 5) printf("mismatch %d: x %a %a ref %08x mix %08x\n", i, x[2 * i], x[2 * i + 1], r[i], m[i]); ++bad; }
        const unsigned lo = r[i] & 0x7fffu; if (lo && lo < 0x400u) ++sub;
    }
    printf("pairs %d mismatches %ld (subnormal residuals seen: %ld)\n", n, bad, sub);
    return bad != 0;
}
