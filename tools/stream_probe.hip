// stream_probe.hip - does the SHAPE of a streaming read decide its HBM rate?  (round 5)
//
// The 1x1 weight gradient (wsw.cuh) with its MFMAs removed still takes 144 us for 613 MB on the 160^2 planes (4.3 TB/s), and what a
// workgroup asks for per 32-pixel k-tile is 64 pieces of 512 contiguous bytes: 32 rows of the block buffer (row stride 1-4 KB) and 32
// planes of gradient units (plane stride = pixels x 16 bytes).  This probe streams the same number of bytes with the same number of
// workgroups, waves and loads in flight per lane, in four shapes:
//   linear    a workgroup's k-tile is 32 KB contiguous
//   rows      64 rows x 512 bytes per k-tile, row stride `stride` bytes (the activation operand; two column tiles share a row)
//   planes    32 planes x 512 bytes + 32 rows x 512 bytes (the kernel's real mix)
//   blocks    the gradient units re-laid as [pixel block of 32][piece][k8][32]: 16 KB contiguous + 32 rows x 512 bytes
// build: hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o tools/stream_probe.bin     run: tools/stream_probe.bin [workgroups] [MB per launch]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
    const uint4* buf; size_t buf_units;      // 16-byte units
    float* out;
    int tiles;                               // 32 KB k-tiles per workgroup
    int mode; int row_units;                 // row stride in units (rows / planes / blocks)
    size_t plane_units;                      // plane stride in units
};

// 256 threads, 8 loads of 16 bytes per thread and k-tile, two k-tiles in flight (like the producer waves of wsw.cuh)
__global__ __launch_bounds__(256) void probe_kernel(const Args a) {
    const int t = threadIdx.x, wg = blockIdx.x;
    const int seg = t >> 5, l = t & 31;      // 8 segments of 512 bytes per load round
    uint4 r[2][8];
    unsigned acc = 0;
    auto addr = [&](int kt, int i) -> size_t {
        const size_t tile = (size_t)wg * a.tiles + kt;
        const int piece = seg + 8 * i;       // 0..63: which 512-byte piece of the k-tile
        size_t u;
        if (a.mode == 0) u = tile * 2048 + (size_t)piece * 32 + l;
        else if (a.mode == 1) u = (tile * 64 + piece) * (size_t)a.row_units + l;                        // 64 rows, 512 bytes of each
        else if (a.mode == 2) u = piece < 32 ? (size_t)piece * a.plane_units + tile * 32 + l              // 32 planes ...
                                             : 32 * a.plane_units + (tile * 32 + (piece - 32)) * (size_t)a.row_units + l;   // ... + 32 rows
        else u = piece < 32 ? tile * 1024 + (size_t)piece * 32 + l
                            : 32 * a.plane_units + (tile * 32 + (piece - 32)) * (size_t)a.row_units + l;
        return u % a.buf_units;
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) r[0][i] = a.buf[addr(0, i)];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[1][i] = a.buf[addr(a.tiles > 1 ? 1 : 0, i)];
    for (int kt = 0; kt < a.tiles; kt += 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += r[0][i].x ^ r[0][i].w;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[0][i] = a.buf[addr(kt + 2 < a.tiles ? kt + 2 : a.tiles - 1, i)];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += r[1][i].y ^ r[1][i].z;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[1][i] = a.buf[addr(kt + 3 < a.tiles ? kt + 3 : a.tiles - 1, i)];
    }
    if (acc == 0x12345678u) a.out[wg] = 1.f;
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 640, mb = argc > 2 ? atoi(argv[2]) : 640;
    const size_t buf_bytes = (size_t)3 << 30;
    uint4* buf; OK(hipMalloc((void**)&buf, buf_bytes)); OK(hipMemset(buf, 1, buf_bytes));
    float* out; OK(hipMalloc((void**)&out, sizeof(float) * wgs));
    hipStream_t st; OK(hipStreamCreate(&st));
    hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
    Args a{}; a.buf = buf; a.buf_units = buf_bytes / 16; a.out = out;
    a.tiles = (int)((size_t)mb * 1024 * 1024 / 32768 / wgs); a.tiles &= ~1;
    const double bytes = (double)wgs * a.tiles * 32768;
    auto timed = [&](int mode, int row_bytes) {
        a.mode = mode; a.row_units = row_bytes / 16;
        a.plane_units = (size_t)wgs * a.tiles * 32;           // a plane = every k-tile's 32 units
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            OK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(probe_kernel, dim3(wgs), dim3(256), 0, st, a);
            OK(hipEventRecord(e1, st)); OK(hipEventSynchronize(e1));
            float ms; OK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        return bytes / best * 1e-9;          // TB/s = bytes / ms * 1e-9
    };
    printf("stream probe: %d workgroups x %d k-tiles of 32 KB = %.0f MB per launch, 2 k-tiles (64 KB) in flight per workgroup\n", wgs, a.tiles, bytes / 1048576.0);
    printf("  linear                          %.2f TB/s\n", timed(0, 0));
    for (int rb : {512, 1024, 2048, 4096}) printf("  rows, stride %4d B              %.2f TB/s\n", rb, timed(1, rb));
    for (int rb : {1024, 4096}) printf("  32 planes + 32 rows (stride %4d) %.2f TB/s\n", rb, timed(2, rb));
    for (int rb : {1024, 4096}) printf("  16 KB block + 32 rows (stride %4d) %.2f TB/s\n", rb, timed(3, rb));
    return 0;
}
