// dma_depth_probe.hip - what do MORE k-tiles in flight per workgroup buy, and does LDS-DMA deliver them?  (round 5)
//
// The 1x1 data gradients stage finished 16-byte units (the gradient D2 and the packed weights: straight copies) global -> registers ->
// LDS with ONE 24 KB k-tile in flight per workgroup; their k-loops take ~2 us per k-tile, the time of one load round trip.  This probe
// runs the skeleton of such a k-loop - 512 workgroups x 256 threads, a k-tile = 24 wave-instructions of 1 KB, then 6 ds_read_b128 per
// lane and ~400 cycles of dependent VALU standing in for the MFMAs - with
//   reg    global -> registers -> LDS, one k-tile in flight (issue at the top of the step, store at its end, one barrier)
//   dma N  buffer_load_dwordx4 ... lds into a ring of N + 1 LDS stages, N k-tiles in flight (wait vmcnt -> barrier -> issue -> compute)
// over a footprint that streams from HBM (every tile new) or re-reads 32 MB (L2 / MALL hits), and prints TB/s and us per k-tile.
// build: hipcc --offload-arch=gfx950 -O3 tools/dma_depth_probe.hip -o tools/dma_depth_probe.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kTileUnits = 1536;             // 24 KB
constexpr int kPerThread = kTileUnits / 256; // 6

struct Args { const u32x4* buf; unsigned buf_units; unsigned foot_tiles; float* out; int tiles; };

__device__ __forceinline__ i32x4 rsrc_of(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFFu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

__device__ __forceinline__ float fake_mfma(const u32x4* lds, int t, float acc) {
    u32x4 f[kPerThread];
#pragma unroll
    for (int i = 0; i < kPerThread; ++i) f[i] = lds[i * 256 + t];
#pragma unroll
    for (int i = 0; i < kPerThread; ++i) acc += __uint_as_float(f[i].x ^ f[i].w);
#pragma unroll
    for (int i = 0; i < 90; ++i) acc = acc * 1.0001f + 0.5f;          // ~360 cycles of dependent VALU
    return acc;
}
__device__ __forceinline__ unsigned tile_soff(const Args& a, int wg, int kt) {
    const unsigned tile = ((unsigned)wg * (unsigned)a.tiles + (unsigned)kt) % a.foot_tiles;
    return tile * (kTileUnits * 16u);
}

template <int NFLY>      // 0: register staging
__global__ __launch_bounds__(256) void probe(const Args a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    const int t = threadIdx.x, wave = t >> 6, wg = blockIdx.x;
    const i32x4 rs = rsrc_of(a.buf, a.buf_units * 16u);
    float acc = 0.f;
    if constexpr (NFLY == 0) {
        u32x4 r[kPerThread];
        auto g_load = [&](int kt) {
#pragma unroll
            for (int i = 0; i < kPerThread; ++i)
                r[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(a.buf), 0, (int)(a.buf_units * 16u), 0x00020000),
                                                                                       (int)(16u * (unsigned)(i * 256 + t)), (int)tile_soff(a, wg, kt), 0));
        };
        auto s_store = [&](int buf) {
#pragma unroll
            for (int i = 0; i < kPerThread; ++i) lds[buf * kTileUnits + i * 256 + t] = r[i];
        };
        g_load(0); s_store(0); __syncthreads();
        for (int kt = 0; kt < a.tiles; ++kt) {
            if (kt + 1 < a.tiles) g_load(kt + 1);
            acc = fake_mfma(lds + (kt & 1) * kTileUnits, t, acc);
            if (kt + 1 < a.tiles) s_store((kt + 1) & 1);
            __syncthreads();
        }
    } else {
        constexpr int NST = NFLY + 1;
        const unsigned base = (unsigned)(size_t)lds;
        auto issue = [&](int kt) {
            const unsigned so = tile_soff(a, wg, kt < a.tiles ? kt : a.tiles - 1);       // (tail: clamped re-loads keep the counts exact)
            const unsigned st = (unsigned)(kt % NST) * (kTileUnits * 16u);
#pragma unroll
            for (int i = 0; i < kPerThread; ++i)
                dma16(rs, 16u * (unsigned)(i * 256 + t), so, __builtin_amdgcn_readfirstlane(base + st + (unsigned)(i * 256 + wave * 64) * 16u));
        };
#pragma unroll
        for (int u = 0; u < NFLY; ++u) issue(u);
        for (int kt = 0; kt < a.tiles; ++kt) {
            wait_vm<(NFLY - 1) * kPerThread>();          // this wave's pieces of k-tile kt have landed
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // everybody's have; stage (kt - 1) % NST is free
            issue(kt + NFLY);
            acc = fake_mfma(lds + (kt % NST) * kTileUnits, t, acc);
        }
        wait_vm<0>();
    }
    if (acc == 12345.678f) a.out[wg] = acc;
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 512, tiles = argc > 2 ? atoi(argv[2]) : 64;
    const size_t buf_bytes = (size_t)2 << 30;
    u32x4* buf; OK(hipMalloc((void**)&buf, buf_bytes)); OK(hipMemset(buf, 1, buf_bytes));
    float* out; OK(hipMalloc((void**)&out, sizeof(float) * wgs));
    hipStream_t st; OK(hipStreamCreate(&st));
    hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
    Args a{}; a.buf = buf; a.buf_units = (unsigned)(buf_bytes / 16); a.out = out; a.tiles = tiles;
    const double bytes = (double)wgs * tiles * kTileUnits * 16;
    auto run = [&](auto kern, int stages, const char* name) {
        OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (unsigned foot_mb : {2048u, 32u}) {
            a.foot_tiles = foot_mb * 1024u * 1024u / (kTileUnits * 16u);
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                OK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), (size_t)stages * kTileUnits * 16, st, a);
                OK(hipEventRecord(e1, st)); OK(hipEventSynchronize(e1));
                float ms; OK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("  %-8s footprint %4u MB: %6.2f TB/s  %5.2f us per k-tile\n", name, foot_mb, bytes / best * 1e-9, best * 1e3 / tiles);
        }
    };
    printf("dma depth probe: %d workgroups x %d k-tiles of 24 KB (%.0f MB per launch)\n", wgs, tiles, bytes / 1048576.0);
    run(probe<0>, 2, "reg");
    run(probe<1>, 2, "dma 1");
    run(probe<2>, 3, "dma 2");
    run(probe<3>, 4, "dma 3");
    run(probe<4>, 5, "dma 4");
    return 0;
}
