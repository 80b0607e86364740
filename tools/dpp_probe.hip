// Development probe: semantics of v_mov_b32_dpp row_shl / row_shr with bound_ctrl = 0 on gfx950, as the 3x3 halo kernels use them to
// derive the dx = 1, 2 activation fragments of a kernel row from the dx = 0 fragment (a tile row = the 16 lanes of one DPP row).
//   hipcc --offload-arch=gfx950 -O3 tools/dpp_probe.hip -o tools/dpp_probe.bin && ./tools/dpp_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ int upd(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, 0xF, false); }
__global__ void k(int* out) {
    const int l = threadIdx.x;
    const int base = 100 + l, ext = 1000 + l;           // ext: lanes 2j, 2j+1 of every row hold columns 16, 17 of pair j
    out[0 * 64 + l] = upd<0x101>(-1, base);             // row_shl:1, old = -1
    out[1 * 64 + l] = upd<0x102>(-1, base);             // row_shl:2
    out[2 * 64 + l] = upd<0x111>(-1, base);             // row_shr:1
    out[3 * 64 + l] = upd<0x11E>(-1, ext);              // row_shr:14  (lane 14 <- lane 0, lane 15 <- lane 1)
    out[4 * 64 + l] = upd<0x102>(upd<0x11E>(-1, ext), base);      // dx = 2 fragment of pair 0
    out[5 * 64 + l] = upd<0x101>(upd<0x11F>(-1, ext), base);      // dx = 1 fragment of pair 0 (lane 15 <- ext lane 0)
    out[6 * 64 + l] = upd<0x102>(upd<0x11A>(-1, ext), base);      // dx = 2 fragment of pair 2 (ext lanes 4, 5: shr 10)
}
int main() {
    int* d; hipMalloc(&d, 7 * 64 * 4);
    int h[7 * 64];
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[7] = {"row_shl:1", "row_shl:2", "row_shr:1", "row_shr:14(ext)", "dx2 pair0", "dx1 pair0", "dx2 pair2"};
    for (int r = 0; r < 7; ++r) {
        printf("%-16s", nm[r]);
        for (int l = 0; l < 32; ++l) printf(" %4d", h[r * 64 + l]);
        printf("\n");
    }
    return 0;
}
