// Development probe: semantics of ds_read_b64_tr_b16 on gfx950 (which LDS element lands in which lane / slot).
//   hipcc --offload-arch=gfx950 -O3 tools/tr_probe.hip -o tools/tr_probe.bin && ./tools/tr_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// mode 0: lane l reads at element offset 4*l (linear).  mode 1: lane i of each 16-group points at row i/4, col quad i%4 of a
// matrix with row stride 160 elements; group g starts at column 16*g.
__global__ void k(short* out, int mode) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    int off = mode == 0 ? 4 * l : (i / 4) * 160 + 16 * g + 4 * (i % 4);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 512);
    short h[256];
    for (int mode = 0; mode < 2; ++mode) {
        k<<<1, 64>>>(d, mode);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    }
    return 0;
}
