// chain_probe.hip - what would ONE persistent kernel per dense block buy on the small planes?  (round 5)
//
// The backward chain of dense block 3 is 24 layers x 4 dependent kernels of 10-17 us (apply -> 3x3 data gradient -> apply -> 1x1 data
// gradient) over 17 streams x 25 tiles = 425 workgroups; between two of them lies a kernel boundary, and what crosses it is small:
// per-(stream, channel) BN sums that every workgroup of the SAME stream adds to and the next phase reads.  This probe times exactly
// that skeleton - P dependent phases, each workgroup streams a few KB of its own data, adds 128 fp64 sums of its stream with
// atomics, and the next phase starts by reading its stream's 128 sums - in three forms:
//   launches   P kernel launches on one stream (what the engine does)
//   stream     ONE launch; the workgroups of a stream meet at a per-stream counter barrier between phases (release fence -> relaxed
//              agent-scope atomic arrive -> relaxed polls -> acquire fence: MI355X_MICROARCH.md, barrier-counter / Guideline 16)
//   grid       ONE launch; every workgroup meets at one grid-wide counter barrier between phases
// and prints microseconds per phase.  All workgroups are co-resident by construction (grid <= 2 per CU, checked against the occupancy
// API minus one); every spin is bounded (a stuck barrier aborts the kernel with a flag instead of hanging the box).
// build: hipcc --offload-arch=gfx950 -O3 tools/chain_probe.hip -o tools/chain_probe.bin      run: tools/chain_probe.bin [phases] [KB per WG]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kStreams = 17, kTiles = 25, kC = 128;

struct Args {
    const float4* data; int vec_per_wg;      // each workgroup's own slice, streamed once per phase
    double* sums;                            // [2 (ping-pong)][streams][kC]
    float* out;                              // [workgroups]: keeps the work alive
    unsigned* counters;                      // [phases][streams] (stream form) / [phases] (grid form)
    unsigned* stuck;
    int phases, mode;                        // mode 0: one phase per launch (phase index = `first`), 1: per-stream barrier, 2: grid barrier
    int first;
};

__device__ __forceinline__ float phase_body(const Args& a, int phase, int wg, int stream) {
    // the previous phase's statistics of this stream (what a BN consumer reads in its prologue), then this workgroup's data
    const double* prev = a.sums + (size_t)((phase + 1) & 1) * kStreams * kC + (size_t)stream * kC;
    float acc = 0.f;
    if (threadIdx.x < kC) acc = (float)__hip_atomic_load(prev + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float4* p = a.data + (size_t)wg * a.vec_per_wg;
    for (int i = threadIdx.x; i < a.vec_per_wg; i += 256) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    // this phase's statistics: one fp64 atomic per (workgroup, channel), like the epilogues of the convolution kernels
    double* cur = a.sums + (size_t)(phase & 1) * kStreams * kC + (size_t)stream * kC;
    if (threadIdx.x < kC) atomicAdd(cur + threadIdx.x, (double)acc * 1e-9);
    return acc;
}

__device__ __forceinline__ bool barrier(unsigned* counter, unsigned expected, unsigned* stuck) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) { atomicExch(stuck, 1u); break; }      // bounded: never hang the box
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return true;
}

__global__ __launch_bounds__(256) void chain_kernel(const Args a) {
    const int wg = blockIdx.x, stream = wg / kTiles;
    float keep = 0.f;
    if (a.mode == 0) {
        keep = phase_body(a, a.first, wg, stream);
    } else {
        for (int ph = 0; ph < a.phases; ++ph) {
            keep += phase_body(a, ph, wg, stream);
            if (ph + 1 < a.phases) {
                if (a.mode == 1) barrier(a.counters + (size_t)ph * kStreams + stream, kTiles, a.stuck);
                else barrier(a.counters + ph, kStreams * kTiles, a.stuck);
                if (*(volatile unsigned*)a.stuck) return;
            }
        }
    }
    if (threadIdx.x == 0) a.out[wg] = keep;
}

int main(int argc, char** argv) {
    const int phases = argc > 1 ? atoi(argv[1]) : 96, kb = argc > 2 ? atoi(argv[2]) : 16;
    const int wgs = kStreams * kTiles;
    int per_cu = 0, cus = 0;
    OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chain_kernel, 256, 0));
    OK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    if ((per_cu - 1) * cus < wgs) { fprintf(stderr, "grid of %d workgroups is not co-resident (%d per CU x %d CUs): refusing to spin\n", wgs, per_cu, cus); return 2; }
    Args a{};
    a.vec_per_wg = kb * 1024 / 16; a.phases = phases;
    float4* data; OK(hipMalloc((void**)&data, (size_t)wgs * a.vec_per_wg * 16)); OK(hipMemset(data, 0, (size_t)wgs * a.vec_per_wg * 16)); a.data = data;
    OK(hipMalloc((void**)&a.sums, sizeof(double) * 2 * kStreams * kC)); OK(hipMemset(a.sums, 0, sizeof(double) * 2 * kStreams * kC));
    OK(hipMalloc((void**)&a.out, sizeof(float) * wgs));
    OK(hipMalloc((void**)&a.counters, sizeof(unsigned) * (size_t)phases * kStreams));
    OK(hipMalloc((void**)&a.stuck, sizeof(unsigned))); OK(hipMemset(a.stuck, 0, sizeof(unsigned)));
    hipStream_t st; OK(hipStreamCreate(&st));
    hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
    auto timed = [&](int mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            OK(hipMemsetAsync(a.counters, 0, sizeof(unsigned) * (size_t)phases * kStreams, st));
            OK(hipEventRecord(e0, st));
            a.mode = mode;
            if (mode == 0) for (int ph = 0; ph < phases; ++ph) { a.first = ph; hipLaunchKernelGGL(chain_kernel, dim3(wgs), dim3(256), 0, st, a); }
            else hipLaunchKernelGGL(chain_kernel, dim3(wgs), dim3(256), 0, st, a);
            OK(hipEventRecord(e1, st));
            OK(hipEventSynchronize(e1));
            float ms; OK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        unsigned stuck = 0; OK(hipMemcpy(&stuck, a.stuck, sizeof(unsigned), hipMemcpyDeviceToHost));
        if (stuck) { fprintf(stderr, "mode %d: a barrier did not complete\n", mode); exit(3); }
        return best * 1e3f / phases;
    };
    const float t_launch = timed(0), t_stream = timed(1), t_grid = timed(2);
    printf("chain probe: %d workgroups (%d streams x %d tiles), %d phases, %d KB per workgroup and phase\n", wgs, kStreams, kTiles, phases, kb);
    printf("  us per phase: %.2f as kernel launches | %.2f persistent, per-stream counter barrier | %.2f persistent, grid-wide counter barrier\n", t_launch, t_stream, t_grid);
    return 0;
}
