// elem.cuh - the HBM-bound kernels around the MFMA convolutions: input
// preparation (zoom/pad/normalise/rotate gather), weight repacking, max-pool forward
// and backward, head feature assembly (norm5 + two-stream concat), the 20x20 value
// convolution and its backward, losses, BN running-statistics update, Adam.
#pragma once
#include "gemm.cuh"

namespace smg {

// ------------------------------------------------------------------------------------
// K1: fused Trainer.forward preprocessing (code/trainer.py:165-191) + per-stream
// rotation (code/models.py:372-382) -> NHWC4 image (4th channel zero).
//
// The sampling arithmetic restates torch's CPU operators bit for bit (pinned by
// golden vector G1): linspace(-1,1,S)[i] = fma(step, i, -1) for i < S/2 else
// fma(-step, S-1-i, 1); affine_grid = fma(y, a01, x*a00) + a02; grid_sample nearest,
// align_corners=True: rint(((g+1)/2)*(S-1)), zero outside [0, S-1].
// ------------------------------------------------------------------------------------
struct PrepArgs {
    const float* images_nchw;   // [n_images][3][S][S] or null
    const double* heightmaps;   // [n_images][hm][hm] or null
    int hm, pad, S;
    double mean, stdv;
    const int* stream_image;
    const float* stream_affine;
    const int* stream_rotated;
    float* img4;                // [streams][HWp][4]
    float* img1;                // if set (heightmap form): ONE channel per pixel [streams][HWp] instead - the three are identical
    int HWp;
    const double* masks;        // [n_masks][hm][hm] or null: object masks applied here (code/main.py:160,187)
    const int* stream_mask_a;   // per stream: mask index or -1
    const int* stream_mask_b;   // per stream: second mask index (ES pairs: mask[g] + mask[s]) or -1
};

__device__ __forceinline__ float lin_coord(int i, int S, float step) {
    return (i < S / 2) ? fmaf(step, (float)i, -1.f) : fmaf(-step, (float)(S - 1 - i), 1.f);
}

static __global__ void prep_rotate_kernel(const PrepArgs a) {
    const int s = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int S = a.S;
    if (p >= S * S) return;
    const int y = p / S, x = p - y * S;
    int sx = x, sy = y;
    bool inside = true;
    if (a.stream_rotated[s]) {
        const float* th = a.stream_affine + 6 * s;
        const float step = 2.0f / (float)(S - 1);
        const float lx = lin_coord(x, S, step), ly = lin_coord(y, S, step);
        const float gx = fmaf(ly, th[1], __fmul_rn(lx, th[0])) + th[2];
        const float gy = fmaf(ly, th[4], __fmul_rn(lx, th[3])) + th[5];
        const float fx = rintf(__fmul_rn(__fdiv_rn(__fadd_rn(gx, 1.f), 2.f), (float)(S - 1)));
        const float fy = rintf(__fmul_rn(__fdiv_rn(__fadd_rn(gy, 1.f), 2.f), (float)(S - 1)));
        inside = fx >= 0.f && fx <= (float)(S - 1) && fy >= 0.f && fy <= (float)(S - 1);
        sx = (int)fx;
        sy = (int)fy;
    }
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (inside) {
        const int img = a.stream_image[s];
        if (a.images_nchw) {
            const float* b = a.images_nchw + (int64_t)img * 3 * S * S + (int64_t)sy * S + sx;
            out.x = b[0];
            out.y = b[(int64_t)S * S];
            out.z = b[(int64_t)2 * S * S];
        } else {
            const int hy = sy - a.pad, hx = sx - a.pad;
            double v = 0.0;
            if (hy >= 0 && hx >= 0 && hy < 2 * a.hm && hx < 2 * a.hm) {
                const int64_t at = (int64_t)(hy >> 1) * a.hm + (hx >> 1);
                v = a.heightmaps[(int64_t)img * a.hm * a.hm + at];
                if (a.masks) {
                    const int ma = a.stream_mask_a[s], mb = a.stream_mask_b[s];
                    if (ma >= 0) {
                        double m = a.masks[(int64_t)ma * a.hm * a.hm + at];
                        if (mb >= 0) m += a.masks[(int64_t)mb * a.hm * a.hm + at];
                        v *= m;                                  // depth * (mask[g] + mask[s]), float64 like numpy
                    }
                }
            }
            const float f = (float)((v - a.mean) / a.stdv);
            out.x = out.y = out.z = f;
        }
    }
    if (a.img1) a.img1[(int64_t)s * a.HWp + p] = out.x;
    else *reinterpret_cast<float4*>(a.img4 + ((int64_t)s * a.HWp + p) * 4) = out;
}

// ------------------------------------------------------------------------------------
// Weight repack: reference layout [cout][cin][kh][kw] -> what the kernels stream.  One launch for a
// whole trunk + head through a descriptor table.
//   split modes: the B operand of the MFMA GEMMs as 16-byte units [piece][K/8][N] of 8 consecutive k of one
//   output column n, already split into its pieces (two scaled fp16 pieces behind a header unit for operand kind 3, three bf16
//   pieces for kind 0; gemm.cuh) - staging them is a plain copy;
//   PK_HF / PK_HD: the per-stage LDS images of the LDS-halo 3x3 kernels; PK_HEAD: the value convolution's fp32 layout.
// ------------------------------------------------------------------------------------
enum { PK_T1 = 0, PK_D1 = 1, PK_3F = 2, PK_3D = 3, PK_STEM = 4, PK_HEAD = 5, PK_HF = 6, PK_HD = 7, PK_STEM1 = 8 };
struct PackDesc { int64_t src, dst; int cout, cin, mode, K8tot, N, count, op0; };   // dst: unit offset (split modes) / float offset (fp32 modes); op0: operand kind of the pack in precision mode 0 (0 = 3-piece bf16 split, 3 = 2-piece fp16 split with a header unit at dst - 1)

// Scales of operand kind 3 (gemm.cuh), one launch per forward in front of pack_weights_kernel:
//   blockIdx.y <  n_pack: the weight tensor of pack descriptor y (kind-3 packs only): s = 2^(13 - floor(log2 max|w|)) -> header
//                         unit {s, 1 / s, 0, 0} in front of the pack
//   blockIdx.y >= n_pack: BatchNorm + ReLU operand y - n_pack: m = max_c hypot(gamma_c, beta_c), s = 2^(4 - floor(log2 m))
//                         (m * s in [16, 32)) -> asc[2 * (y - n_pack)] = {s, 1 / s}
struct ActScaleDesc { int64_t gamma, beta; int C; };
static __global__ __launch_bounds__(1024) void scale_kernel(const PackDesc* descs, int n_pack, const ActScaleDesc* adescs, const float* params, u32x4* packed_u,
                                                           float* asc, int prec) {
    __shared__ float red[16];
    const int t = threadIdx.x, y = blockIdx.y;
    float m = 0.f;
    int target;
    if (y < n_pack) {
        const PackDesc d = descs[y];
        if (prec != 0 || d.op0 != 3) return;               // (workgroup-uniform)
        const int taps = (d.mode == PK_3F || d.mode == PK_3D || d.mode == PK_HF || d.mode == PK_HD) ? 9 : 1;
        const int64_t count = (int64_t)d.cout * d.cin * taps;
        const float* w = params + d.src;
        float m4[4] = {0.f, 0.f, 0.f, 0.f};                 // four independent loads in flight per thread (127 k floats in the largest tensor)
        int64_t i = t;
        if ((d.src & 3) == 0) {                             // 16-byte aligned tensors (all of the layout's): four float4 in flight, 8 trips instead of 31
            const float4* w4 = reinterpret_cast<const float4*>(w);
            const int64_t c4 = count >> 2;
            int64_t j = t;
            for (; j + 3 * 1024 < c4; j += 4 * 1024) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 v = w4[j + u * 1024];
                    m4[u] = fmaxf(m4[u], fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                }
            }
            for (; j < c4; j += 1024) { const float4 v = w4[j]; m4[0] = fmaxf(m4[0], fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)))); }
            i = 4 * c4 + t;                                 // (the up to three floats behind the last whole float4)
        } else
        for (; i + 3 * 1024 < count; i += 4 * 1024) {
#pragma unroll
            for (int u = 0; u < 4; ++u) m4[u] = fmaxf(m4[u], fabsf(w[i + u * 1024]));
        }
        for (; i < count; i += 1024) m4[0] = fmaxf(m4[0], fabsf(w[i]));
        m = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
        target = 13;
    } else {
        const ActScaleDesc d = adescs[y - n_pack];
        for (int c = t; c < d.C; c += 1024) { const float g = params[d.gamma + c], b = params[d.beta + c]; m = fmaxf(m, g * g + b * b); }
        m = sqrtf(m);
        target = 4;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((t & 63) == 0) red[t >> 6] = m;
    __syncthreads();
    if (t == 0) {
        m = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) m = fmaxf(m, red[u]);
        int e = (int)(__float_as_uint(m) >> 23) - 127;     // floor(log2 m); 0, subnormal or non-finite maxima: scale 1
        float sc = 1.f, inv = 1.f;
        if (m > 0.f && e > -100 && e < 100) { sc = __uint_as_float((unsigned)(127 + target - e) << 23); inv = __uint_as_float((unsigned)(127 - target + e) << 23); }
        if (y < n_pack) {
            float* h = reinterpret_cast<float*>(packed_u + descs[y].dst - 1);
            h[0] = sc; h[1] = inv; h[2] = 0.f; h[3] = 0.f;
        } else {
            asc[2 * (y - n_pack)] = sc; asc[2 * (y - n_pack) + 1] = inv;
        }
    }
}

// prec: the engine's precision mode.  Operand kind of a pack: the forward packs (PK_T1, PK_STEM, PK_3F, PK_HF) take the
// forward kind (the desc's split kind op0 / bf16 / fp16), the data-gradient packs (PK_D1, PK_3D, PK_HD) the backward kind (op0 / bf16).
// Single-piece kinds store ONE piece (the kernels copy a third of the bytes); the halo forward image then holds 32 channels
// per chunk instead of 16 (halo_ck).
static __global__ void pack_weights_kernel(const PackDesc* descs, const float* params, u32x4* packed_u, float* packed_f, const int prec) {
    const PackDesc d = descs[blockIdx.y];
    const float* s = params + d.src;
    if (d.mode == PK_HEAD) {                       // fp32 dst[o][tap][c] = src[o][c][tap] (value convolution)
        float* o = packed_f + d.dst;
        const int taps = 400;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.count; i += gridDim.x * blockDim.x) {
            const int o_ = i / (taps * d.cin), r = i - o_ * taps * d.cin;
            const int tap = r / d.cin, c = r - tap * d.cin;
            o[i] = s[((int64_t)o_ * d.cin + c) * taps + tap];
        }
        return;
    }
    const bool bwd = d.mode == PK_D1 || d.mode == PK_3D || d.mode == PK_HD;
    const int op = prec == 0 ? d.op0 : (bwd ? 1 : prec);    // 0 bf16 split (3 pieces), 1 bf16, 2 fp16, 3 fp16 split (2 pieces, scaled)
    const int np = np_of(op);
    const int total = d.K8tot * d.N;               // units per piece
    u32x4* o = packed_u + d.dst;
    const float wsc = op == 3 ? reinterpret_cast<const float*>(o - 1)[0] : 1.f;      // header written by scale_kernel
    auto put = [&](const float (&v)[8], int64_t base, int64_t pstride) {
        const float4 v0 = make_float4(v[0] * wsc, v[1] * wsc, v[2] * wsc, v[3] * wsc), v1 = make_float4(v[4] * wsc, v[5] * wsc, v[6] * wsc, v[7] * wsc);
        const Split4 lo = op == 0 ? split4<0>(v0) : op == 1 ? split4<1>(v0) : op == 2 ? split4<2>(v0) : split4<3>(v0);
        const Split4 hi = op == 0 ? split4<0>(v1) : op == 1 ? split4<1>(v1) : op == 2 ? split4<2>(v1) : split4<3>(v1);
#pragma unroll
        for (int pc = 0; pc < NPIECE; ++pc)
            if (pc < np) o[base + pc * pstride] = u32x4{lo.p[pc].x, lo.p[pc].y, hi.p[pc].x, hi.p[pc].y};
    };
    if (d.mode == PK_HF || d.mode == PK_HD) {
        // LDS-halo 3x3 kernels: the image of one stage, per stage (halo.cuh)
        //   PK_HF forward       [chunk = c/CK][piece][tap][k8 (CK/8)][n (32)]  unit = 8 input channels c of output channel n
        //   PK_HD data gradient [cgroup = c/32][tap][piece][k8 (4)][c (32)]    unit = 8 output channels n of input channel c
        const int k8c = (prec ? 32 : 16) / 8;      // halo_ck(prec) / 8
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
            float v[8];
            int64_t base, pstride;
            if (d.mode == PK_HF) {
                const int n = e & 31, k8 = (e >> 5) % k8c, tap = (e / (32 * k8c)) % 9, chunk = e / (9 * 32 * k8c);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = s[((int64_t)n * d.cin + chunk * 8 * k8c + 8 * k8 + j) * 9 + tap];
                pstride = 9 * k8c * 32;
                base = ((int64_t)chunk * np * 9 + tap) * (k8c * 32) + k8 * 32 + n;
            } else {
                const int c = e & 31, k8 = (e >> 5) & 3, tap = (e >> 7) % 9, cg = e / 1152;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = s[((int64_t)(8 * k8 + j) * d.cin + cg * 32 + c) * 9 + tap];
                base = (((int64_t)cg * 9 + tap) * np) * 128 + k8 * 32 + c; pstride = 128;
            }
            put(v, base, pstride);
        }
        return;
    }
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int k8 = e / d.N, n = e - k8 * d.N;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * k8 + j;
            if (d.mode == PK_T1) {                 // B(k = c, n) = src[n][c]
                v[j] = s[(int64_t)n * d.cin + k];
            } else if (d.mode == PK_D1) {          // B(k = cout index, n = c) = src[k][c]
                v[j] = s[(int64_t)k * d.cin + n];
            } else if (d.mode == PK_3F) {          // B(k = tap*cin + c, n) = src[n][c][tap]
                const int tap = k / d.cin, c = k - tap * d.cin;
                v[j] = s[((int64_t)n * d.cin + c) * 9 + tap];
            } else if (d.mode == PK_3D) {          // B(k = tap*cout + nn, n = c) = src[nn][c][tap]
                const int tap = k / d.cout, nn = k - tap * d.cout;
                v[j] = s[((int64_t)nn * d.cin + n) * 9 + tap];
            } else if (d.mode == PK_STEM1) {       // k = tap (49 padded to 64): the weights summed over the 3 (identical) input channels
                v[j] = k < 49 ? (float)((double)s[((int64_t)n * 3 + 0) * 49 + k] + (double)s[((int64_t)n * 3 + 1) * 49 + k] + (double)s[((int64_t)n * 3 + 2) * 49 + k]) : 0.f;
            } else {                               // PK_STEM: k = tap*4 + c (K padded to 224), src[n][c][tap]
                const int tap = k >> 2, c = k & 3;
                v[j] = (c < 3 && tap < 49) ? s[((int64_t)n * 3 + c) * 49 + tap] : 0.f;
            }
        }
        put(v, e, total);
    }
}

// ------------------------------------------------------------------------------------
// BN statistics table entries for channels whose producer is not a dense layer (block inputs after pool0 / a
// transition, the head's concatenated features): fp64 sums -> fp32 mean | invstd, once.  The channels a dense
// layer appends are finished by their first consumer instead (BnTab, gemm.cuh) - no launch of their own.
// ------------------------------------------------------------------------------------
struct BnStatArgs {
    const double* sum; const double* sq; int sstride;          // [row][sstride] statistics
    float eps; double inv_count;
    float* mean; float* invstd; int ld;                        // tables [row][ld]
    int c0, C, rows;                                           // channels [c0, c0 + C)
};
static __global__ void bn_stat_kernel(const BnStatArgs a) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.rows * a.C) return;
    const int n = idx / a.C, c = a.c0 + idx - n * a.C;
    float mean, invstd;
    bn_moments(a.sum, a.sq, (int64_t)n * a.sstride + c, a.inv_count, a.eps, mean, invstd);
    a.mean[(int64_t)n * a.ld + c] = mean;
    a.invstd[(int64_t)n * a.ld + c] = invstd;
}

// ------------------------------------------------------------------------------------
// pool0: BN(norm0)+ReLU+MaxPool(3,2,1) of the stem output -> channels 0..63 of the
// block-1 buffer, + their per-stream statistics, + argmax (for the backward).
// Workgroup = 64 pooled pixels x 64 channels; thread = (channel quad, pixel slot).
// ------------------------------------------------------------------------------------
struct Pool0Args {
    const float* stem; Plane ps;           // [n][HWp][64]
    const double* ssum; const double* ssq;  // [n][64]
    const float* gamma; const float* beta; float eps;
    void* x1; int ldx; Plane po;            // block-1 buffer (activation storage of the mode)
    double* dsum; double* dsq; int dstride;
    unsigned char* argmax;                  // [n][po.HWp][64]
};

template <int PREC>
static __global__ __launch_bounds__(256) void pool0_kernel(const Pool0Args a) {
    using XT = act_t<PREC>;
    __shared__ float prm[192];
    __shared__ double red[2][16][64];
    const int n = blockIdx.y, t = threadIdx.x, cq = t & 15, slot = t >> 4;
    if (t < 64) {
        float mean, invstd;
        bn_moments(a.ssum, a.ssq, (int64_t)n * 64 + t, 1.0 / (double)a.ps.HW, a.eps, mean, invstd);
        prm[t] = mean;
        prm[64 + t] = a.gamma[t] * invstd;
        prm[128 + t] = a.beta[t];
    }
    __syncthreads();
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        const int p = blockIdx.x * 64 + slot + 16 * i;
        if (p >= a.po.HW) continue;
        const int y = p / a.po.W, x = p - y * a.po.W;
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
        for (int k = 0; k < 9; ++k) {
            const int yy = 2 * y - 1 + k / 3, xx = 2 * x - 1 + k % 3;
            if ((unsigned)yy >= (unsigned)a.ps.H || (unsigned)xx >= (unsigned)a.ps.W) continue;
            const float4 v = bnrelu4(ld4(a.stem + ((int64_t)n * a.ps.HWp + yy * a.ps.W + xx) * 64 + 4 * cq), prm + 4 * cq, 64);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (vv[c] > best[c]) { best[c] = vv[c]; bi[c] = k; }
        }
        const int64_t o = (int64_t)n * a.po.HWp + p;
        stq<XT>(a.x1, o * a.ldx + 4 * cq, make_float4(best[0], best[1], best[2], best[3]));
        if (a.argmax) *reinterpret_cast<uchar4*>(a.argmax + o * 64 + 4 * cq) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
#pragma unroll
        for (int c = 0; c < 4; ++c) { s[c] += (double)best[c]; ss[c] += (double)best[c] * (double)best[c]; }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) { red[0][slot][4 * cq + c] = s[c]; red[1][slot][4 * cq + c] = ss[c]; }
    __syncthreads();
    if (t < 128) {
        const int q = t >> 6, c = t & 63;
        double tot = 0.0;
        for (int k = 0; k < 16; ++k) tot += red[q][k][c];
        atomicAdd((q ? a.dsq : a.dsum) + (int64_t)n * a.dstride + c + fstat_rep(), tot);
    }
}

// ------------------------------------------------------------------------------------
// Head feature assembly: F[pair] = concat(norm5(X4[a]), norm5(X4[b]))  (no ReLU after
// norm5 - the reference calls .features directly, code/models.py:384-386) and the
// per-(pair, channel) statistics the head's norm0 needs.
// ------------------------------------------------------------------------------------
struct FeatArgs {
    const void* x4; Plane p4;               // [n][HWp][1024] (activation storage of the mode)
    const double* xsum; const double* xsq;  // [n][1024]
    const float* gamma; const float* beta; float eps;   // norm5
    const int* pair_a; const int* pair_b;
    float* F;                               // [pair][HWp][2048]
    double* fsum; double* fsq;              // [pair][2048]
    int chunk;
};

template <int PREC>
static __global__ __launch_bounds__(256) void feat_kernel(const FeatArgs a) {
    using XT = act_t<PREC>;
    const int j = blockIdx.y, cq = blockIdx.x * 256 + threadIdx.x;   // channel quad of 2048/4
    const int ch = 4 * cq, slot = ch >> 10, c5 = ch & 1023;
    const int s = slot ? a.pair_b[j] : a.pair_a[j];
    float mu[4], sc[4], be[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float invstd;
        bn_moments(a.xsum, a.xsq, (int64_t)s * 1024 + c5 + c, 1.0 / (double)a.p4.HW, a.eps, mu[c], invstd);
        sc[c] = a.gamma[c5 + c] * invstd;
        be[c] = a.beta[c5 + c];
    }
    double sm[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
    const int p0 = blockIdx.z * a.chunk;
    const int p1 = min(p0 + a.chunk, a.p4.HW);
    for (int p = p0; p < p1; ++p) {
        const float4 v = ldq<XT>(a.x4, ((int64_t)s * a.p4.HWp + p) * 1024 + c5);
        float4 o;
        o.x = bn1(v.x, mu[0], sc[0], be[0]); o.y = bn1(v.y, mu[1], sc[1], be[1]);
        o.z = bn1(v.z, mu[2], sc[2], be[2]); o.w = bn1(v.w, mu[3], sc[3], be[3]);
        *reinterpret_cast<float4*>(a.F + ((int64_t)j * a.p4.HWp + p) * 2048 + ch) = o;
        const double od[4] = {(double)o.x, (double)o.y, (double)o.z, (double)o.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) { sm[c] += od[c]; sq[c] += od[c] * od[c]; }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        atomicAdd(a.fsum + (int64_t)j * 2048 + ch + c + fstat_rep(), sm[c]);
        atomicAdd(a.fsq + (int64_t)j * 2048 + ch + c + fstat_rep(), sq[c]);
    }
}

// ------------------------------------------------------------------------------------
// Value convolution: 20x20 valid conv over BN+ReLU(H1) (code/models.py:322,332).
// One workgroup per output element; fixed summation order -> reproducible argmax.
// ------------------------------------------------------------------------------------
struct ValueArgs {
    const float* h1; Plane p4;               // [pair][HWp][64]
    const double* hsum; const double* hsq;   // [pair][64]
    const float* gamma; const float* beta; float eps;
    const float* w2p;                        // packed [out][400][64]
    float* q; int out_ch, OH, OW;
};

static __global__ __launch_bounds__(256) void value_conv_kernel(const ValueArgs a) {
    __shared__ float prm[192];
    __shared__ float red[256];
    const int t = threadIdx.x, cq = t & 15, slot = t >> 4;
    int id = blockIdx.x;
    const int ox = id % a.OW; id /= a.OW;
    const int oy = id % a.OH; id /= a.OH;
    const int o = id % a.out_ch;
    const int j = id / a.out_ch;
    if (t < 64) {
        float mean, invstd;
        bn_moments(a.hsum, a.hsq, (int64_t)j * 64 + t, 1.0 / (double)a.p4.HW, a.eps, mean, invstd);
        prm[t] = mean;
        prm[64 + t] = a.gamma[t] * invstd;
        prm[128 + t] = a.beta[t];
    }
    __syncthreads();
    float acc = 0.f;
    for (int tap = slot; tap < 400; tap += 16) {
        const int pix = (oy + tap / 20) * a.p4.W + ox + tap % 20;
        const float4 v = bnrelu4(ld4(a.h1 + ((int64_t)j * a.p4.HWp + pix) * 64 + 4 * cq), prm + 4 * cq, 64);
        const float4 w = ld4(a.w2p + ((int64_t)o * 400 + tap) * 64 + 4 * cq);
        acc = fmaf(v.x, w.x, acc); acc = fmaf(v.y, w.y, acc); acc = fmaf(v.z, w.z, acc); acc = fmaf(v.w, w.w, acc);
    }
    red[t] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        __syncthreads();
    }
    if (t == 0) a.q[blockIdx.x] = red[0];
}

// Backward of the value convolution + ReLU + BN(norm1) statistics:
//   dact[p][c] = sum_{o,oy,ox} dq[o][oy][ox] * W[o][c][py-oy][px-ox]
//   dy = dact * [BN(h1) > 0] -> DH1 ; sums of dy, dy*xhat per (pair, c);
//   dW[o][c][tap] += dq * act.
struct ValueBwdArgs {
    const float* h1; Plane p4;
    const double* hsum; const double* hsq;
    const float* gamma; const float* beta; float eps;
    const float* w2p;                         // packed [out][400][64]
    const float* dq; int out_ch, OH, OW;
    float* dh1;                               // [pair][HWp][64]
    double* o1; double* o2;                   // [pair][64]
    float* dbeta; float* dgamma;              // norm1 grads
    float* dw2;                               // native [out][64][20][20]
};

static __global__ __launch_bounds__(256) void value_bwd_kernel(const ValueBwdArgs a) {
    __shared__ float prm[256];
    __shared__ float red[2][16][64];
    const int j = blockIdx.y, t = threadIdx.x, cq = t & 15, slot = t >> 4;
    if (t < 64) {
        float mean, invstd;
        bn_moments(a.hsum, a.hsq, (int64_t)j * 64 + t, 1.0 / (double)a.p4.HW, a.eps, mean, invstd);
        prm[t] = a.gamma[t] * invstd; prm[64 + t] = a.beta[t]; prm[128 + t] = mean; prm[192 + t] = invstd;
    }
    __syncthreads();
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        const int p = blockIdx.x * 64 + slot + 16 * i;
        if (p >= a.p4.HW) continue;
        const int py = p / a.p4.W, px = p - py * a.p4.W;
        const int64_t row = (int64_t)j * a.p4.HWp + p;
        const float4 hv = ld4(a.h1 + row * 64 + 4 * cq);
        const float h[4] = {hv.x, hv.y, hv.z, hv.w};
        float act[4], dact[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 4; ++c) act[c] = fmaxf(bn1(h[c], prm[128 + 4 * cq + c], prm[4 * cq + c], prm[64 + 4 * cq + c]), 0.f);
        for (int o = 0; o < a.out_ch; ++o)
            for (int oy = max(0, py - 19); oy <= min(py, a.OH - 1); ++oy)
                for (int ox = max(0, px - 19); ox <= min(px, a.OW - 1); ++ox) {
                    const float g = a.dq[(((int64_t)j * a.out_ch + o) * a.OH + oy) * a.OW + ox];
                    if (g == 0.f) continue;
                    const int tap = (py - oy) * 20 + (px - ox);
                    const float4 w = ld4(a.w2p + ((int64_t)o * 400 + tap) * 64 + 4 * cq);
                    dact[0] = fmaf(g, w.x, dact[0]); dact[1] = fmaf(g, w.y, dact[1]);
                    dact[2] = fmaf(g, w.z, dact[2]); dact[3] = fmaf(g, w.w, dact[3]);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        atomicAdd(a.dw2 + ((int64_t)o * 64 + 4 * cq + c) * 400 + tap, g * act[c]);
                }
        float dy[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            dy[c] = act[c] > 0.f ? dact[c] : 0.f;
            s1[c] += dy[c];
            s2[c] += dy[c] * ((h[c] - prm[128 + 4 * cq + c]) * prm[192 + 4 * cq + c]);
        }
        *reinterpret_cast<float4*>(a.dh1 + row * 64 + 4 * cq) = make_float4(dy[0], dy[1], dy[2], dy[3]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) { red[0][slot][4 * cq + c] = s1[c]; red[1][slot][4 * cq + c] = s2[c]; }
    __syncthreads();
    if (t < 128) {
        const int q = t >> 6, c = t & 63;
        float tot = 0.f;
        for (int k = 0; k < 16; ++k) tot += red[q][k][c];
        atomicAdd((q ? a.o2 : a.o1) + (int64_t)j * 64 + c + stat_rep(), (double)tot);
        atomicAdd((q ? a.dgamma : a.dbeta) + c, tot);
    }
}

// ------------------------------------------------------------------------------------
// norm5 backward (no ReLU) fused with the head's norm0 backward and the two-stream
// concat backward: for stream s, sum over every (pair, slot) that consumed its
// features the BN-corrected gradient of F, then start the block-4 gradient buffer:
//   G'_4[s] = gamma5 * dy5,  SA/SB += gamma5 * sums,  dbeta5/dgamma5 += sums.
// ------------------------------------------------------------------------------------
struct Norm5BwdArgs {
    const float* DF; const float* F; Plane p4;      // [pair][HWp][2048]
    const double* fsum; const double* fsq;          // F statistics [pair][2048]
    const double* f1; const double* f2;             // sums of dy0, dy0*xhat [pair][2048]
    const float* hgamma;                            // head norm0 gamma [2048]
    const void* x4; const double* xsum; const double* xsq;    // [n][HWp][1024] (activation storage), [n][1024]
    const float* gamma5; float eps;
    const int* user_ptr; const int* user_pair; const int* user_slot;  // CSR over streams
    void* G4;                                       // [n][HWp][1024] (gradient storage of the mode)
    double* SA; double* SB;                         // [n][1024]
    float* dbeta5; float* dgamma5;
    int chunk;
};

template <int PREC>
static __global__ __launch_bounds__(256) void norm5_bwd_kernel(const Norm5BwdArgs a) {
    using XT = act_t<PREC>;
    using GT = grd_t<PREC>;
    // Workgroup = (256 channels, one stream, 16 pixels): 64 channel quads x 4 user phases.  A stream with several users (the masked
    // stream feeds every pair) is a serial chain of parameter + pixel round trips per user; its users are dealt round-robin to the
    // four phases, which meet in LDS in a fixed order (one thread per quad over all users: 175 us, a 16x longer tail than the
    // single-user streams).
    constexpr int CH = 16;                               // pixels per workgroup (= a.chunk, checked at the launch)
    __shared__ float4 red[3][CH][64];
    __shared__ float reds[3][8][64];
    const int s = blockIdx.y, cq = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c5 = 4 * (blockIdx.x * 64 + cq);
    const int u0 = a.user_ptr[s], u1 = a.user_ptr[s + 1];
    const bool multi = u1 - u0 > 1;                      // (workgroup-uniform)
    // single-user streams: kSpan chunks per workgroup, so that the per-channel dbeta / dgamma (one address for every stream and
    // chunk) see a fifth of the atomics - 425 same-address atomics per channel were 38 of the kernel's 110 us
    constexpr int kSpan = 5;
    const int n_sub = multi ? 1 : kSpan;
    const int z0 = multi ? blockIdx.z : blockIdx.z * kSpan;
    if (z0 * a.chunk >= a.p4.HW) return;
    const double inv = 1.0 / (double)a.p4.HW;
    float mean5[4], inv5[4], g5[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bn_moments(a.xsum, a.xsq, (int64_t)s * 1024 + c5 + c, inv, a.eps, mean5[c], inv5[c]);
        g5[c] = a.gamma5[c5 + c];
    }
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    for (int sub = 0; sub < n_sub; ++sub) {
    const int p0 = (z0 + sub) * a.chunk, p1 = min(p0 + a.chunk, a.p4.HW);
    if (p0 >= p1) break;
    // G4 of the chunk's pixels accumulates in registers over the users of the stream (a read-modify-write of G4 per user was a
    // chain of dependent global round trips per pixel)
    float4 gacc[CH];
#pragma unroll
    for (int q = 0; q < CH; ++q) gacc[q] = zero4();
    for (int u = u0 + ph; u < u1; u += 4) {
        const int j = a.user_pair[u], ch = a.user_slot[u] * 1024 + c5;
        float cf[16];   // a[4], q1[4], mean[4], k[4]
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float mean, invstd;
            bn_moments(a.fsum, a.fsq, (int64_t)j * 2048 + ch + c, inv, a.eps, mean, invstd);
            const float q2 = (float)(stat_get(a.f2, (int64_t)j * 2048 + ch + c) * inv);
            cf[c] = a.hgamma[ch + c] * invstd;
            cf[4 + c] = (float)(stat_get(a.f1, (int64_t)j * 2048 + ch + c) * inv);
            cf[8 + c] = mean;
            cf[12 + c] = invstd * q2;
        }
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int p = p0 + q;
            if (p < p1) {
                const int64_t fr = ((int64_t)j * a.p4.HWp + p) * 2048 + ch;
                const float4 d = affine2(ld4(a.DF + fr), ld4(a.F + fr), cf, 4);
                const int64_t xr = ((int64_t)s * a.p4.HWp + p) * 1024 + c5;
                const float4 xv = ldq<XT>(a.x4, xr);
                const float dd[4] = {d.x, d.y, d.z, d.w}, xx[4] = {xv.x, xv.y, xv.z, xv.w};
                gacc[q].x += g5[0] * dd[0]; gacc[q].y += g5[1] * dd[1]; gacc[q].z += g5[2] * dd[2]; gacc[q].w += g5[3] * dd[3];
#pragma unroll
                for (int c = 0; c < 4; ++c) { s1[c] += dd[c]; s2[c] += dd[c] * ((xx[c] - mean5[c]) * inv5[c]); }
            }
        }
    }
    if (multi) {
        if (ph) {
#pragma unroll
            for (int q = 0; q < CH; ++q) red[ph - 1][q][cq] = gacc[q];
#pragma unroll
            for (int c = 0; c < 4; ++c) { reds[ph - 1][c][cq] = s1[c]; reds[ph - 1][4 + c][cq] = s2[c]; }
        }
        __syncthreads();
        if (ph == 0) {
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const float4 r0 = red[0][q][cq], r1 = red[1][q][cq], r2 = red[2][q][cq];
                gacc[q].x = (gacc[q].x + r0.x) + (r1.x + r2.x); gacc[q].y = (gacc[q].y + r0.y) + (r1.y + r2.y);
                gacc[q].z = (gacc[q].z + r0.z) + (r1.z + r2.z); gacc[q].w = (gacc[q].w + r0.w) + (r1.w + r2.w);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s1[c] = (s1[c] + reds[0][c][cq]) + (reds[1][c][cq] + reds[2][c][cq]);
                s2[c] = (s2[c] + reds[0][4 + c][cq]) + (reds[1][4 + c][cq] + reds[2][4 + c][cq]);
            }
        }
    }
    if (ph == 0) {
#pragma unroll
        for (int q = 0; q < CH; ++q)
            if (p0 + q < p1) stq<GT>(a.G4, ((int64_t)s * a.p4.HWp + p0 + q) * 1024 + c5, gacc[q]);
    }
    }   // sub-chunks
    if (ph || u0 == u1) return;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        atomicAdd(a.SA + (int64_t)s * 1024 + c5 + c + stat_rep(), (double)(g5[c] * s1[c]));
        atomicAdd(a.SB + (int64_t)s * 1024 + c5 + c + stat_rep(), (double)(g5[c] * s2[c]));
        atomicAdd(a.dbeta5 + c5 + c, s1[c]);
        atomicAdd(a.dgamma5 + c5 + c, s2[c]);
    }
}

// ------------------------------------------------------------------------------------
// pool0 + relu0 backward: route the (BN-corrected) gradient of block-1 channels 0..63
// back through the 3x3/stride-2 max pool to the stem output, apply the ReLU mask of
// norm0, and collect norm0's backward sums.  Gather form: each stem pixel looks at
// the <= 4 pooled windows covering it and takes those whose argmax points at it.
// ------------------------------------------------------------------------------------
struct Pool0BwdArgs {
    const void* G1; const void* X1; int ld1; Plane p1;       // block-1 buffers (gradient / activation storage of the mode)
    const double* xsum; const double* xsq; int xstride;      // block-1 stats
    const double* SA; const double* SB; int sstride;
    const unsigned char* argmax;                             // [n][p1.HWp][64]
    const float* stem; Plane ps;                             // raw stem output [n][HWp][64]
    const double* ssum; const double* ssq;                   // [n][64]
    const float* gamma; const float* beta; float eps;        // norm0
    float* DY0;                                              // [n][ps.HWp][64]
    double* o1; double* o2;                                  // [n][64]
    float* dbeta; float* dgamma;
    int tiles_per_wg;                                        // consecutive 8x8 tiles per workgroup (one flush of the sums)
};

template <int PREC>
static __global__ __launch_bounds__(256) void pool0_bwd_kernel(const Pool0BwdArgs a) {
    using XT = act_t<PREC>;
    using GT = grd_t<PREC>;
    // Workgroup = an 8x8 tile of stem pixels x 64 channels.  The <= 5x5 pooled pixels whose windows
    // cover the tile are finalised (BN-backward-corrected) ONCE into LDS together with their argmax;
    // every stem pixel then picks its <= 4 windows from LDS.  (The gather straight from global memory
    // chained argmax load -> compare -> conditional gradient load: one memory round trip per window.)
    // A workgroup walks tiles_per_wg tiles and flushes its BN-backward sums once: one tile per workgroup meant
    // 27 200 workgroups x 256 atomics on 2 x 64 x (1 + streams) addresses.
    __shared__ float prm[8 * 64];
    __shared__ float gl[25][64];
    __shared__ unsigned char al[25][64];
    __shared__ float red[2][16][64];
    const int n = blockIdx.y, t = threadIdx.x, cq = t & 15, slot = t >> 4;
    const int tiles_x = a.ps.W / 8, n_tiles = tiles_x * (a.ps.H / 8);
    if (t < 64) {
        float mean, invstd;
        bn_moments(a.ssum, a.ssq, (int64_t)n * 64 + t, 1.0 / (double)a.ps.HW, a.eps, mean, invstd);
        prm[t] = a.gamma[t] * invstd; prm[64 + t] = a.beta[t]; prm[128 + t] = mean; prm[192 + t] = invstd;
        const double inv = 1.0 / (double)a.p1.HW;
        float m1, i1;
        bn_moments(a.xsum, a.xsq, (int64_t)n * a.xstride + t, inv, a.eps, m1, i1);
        const float q1 = (float)(stat_get(a.SA, (int64_t)n * a.sstride + t) * inv);
        const float q2 = (float)(stat_get(a.SB, (int64_t)n * a.sstride + t) * inv);
        prm[256 + t] = i1; prm[320 + t] = q1; prm[384 + t] = m1; prm[448 + t] = i1 * q2;
    }
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    const int tile0 = blockIdx.x * a.tiles_per_wg, tile1 = min(tile0 + a.tiles_per_wg, n_tiles);
    for (int tile = tile0; tile < tile1; ++tile) {
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int y0 = ty * 8, x0 = tx * 8;
    // stem values of this thread's 4 pixels: issue early
    float4 sv[4];
    int64_t srow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int lp = slot + 16 * i;                       // local pixel 0..63
        srow[i] = (int64_t)n * a.ps.HWp + (y0 + (lp >> 3)) * a.ps.W + x0 + (lp & 7);
        sv[i] = ld4(a.stem + srow[i] * 64 + 4 * cq);
    }
    // pooled pixels wy in [y0/2, y0/2 + 4], wx likewise
    const int wy0 = y0 >> 1, wx0 = x0 >> 1;
    float4 gq[2], xq[2];
    uchar4 aq[2];
    bool okq[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int item = t + 256 * k;                       // 25 pooled pixels x 16 quads = 400 items
        const int pp = item >> 4, q = item & 15;
        const int wy = wy0 + pp / 5, wx = wx0 + pp % 5;
        okq[k] = item < 400 && wy < a.p1.H && wx < a.p1.W;
        const int64_t pr = (int64_t)n * a.p1.HWp + (okq[k] ? wy * a.p1.W + wx : 0);
        gq[k] = ldq<GT>(a.G1, pr * a.ld1 + 4 * q);
        xq[k] = ldq<XT>(a.X1, pr * a.ld1 + 4 * q);
        aq[k] = *reinterpret_cast<const uchar4*>(a.argmax + pr * 64 + 4 * q);
    }
    __syncthreads();                                        // prm ready; previous tile's gl / al consumed
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int item = t + 256 * k;
        if (item < 400) {
            const int pp = item >> 4, q = item & 15;
            float4 gv = affine2(gq[k], xq[k], prm + 256 + 4 * q, 64);
            if (!okq[k]) gv = zero4();
            *reinterpret_cast<float4*>(&gl[pp][4 * q]) = gv;
            *reinterpret_cast<uchar4*>(&al[pp][4 * q]) = okq[k] ? aq[k] : make_uchar4(255, 255, 255, 255);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int lp = slot + 16 * i;
        const int y = y0 + (lp >> 3), x = x0 + (lp & 7);
        float g[4] = {0, 0, 0, 0};
        const int py0 = y >> 1, py1 = (y + 1) >> 1, px0 = x >> 1, px1 = (x + 1) >> 1;
        for (int wy = py0; wy <= py1; ++wy)
            for (int wx = px0; wx <= px1; ++wx) {
                const int k = (y - (2 * wy - 1)) * 3 + (x - (2 * wx - 1));   // my index inside that window
                const int pp = (wy - wy0) * 5 + (wx - wx0);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (al[pp][4 * cq + c] == k) g[c] += gl[pp][4 * cq + c];
            }
        const float st[4] = {sv[i].x, sv[i].y, sv[i].z, sv[i].w};
        float dy[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            dy[c] = bn1(st[c], prm[128 + 4 * cq + c], prm[4 * cq + c], prm[64 + 4 * cq + c]) > 0.f ? g[c] : 0.f;
            s1[c] += dy[c];
            s2[c] += dy[c] * ((st[c] - prm[128 + 4 * cq + c]) * prm[192 + 4 * cq + c]);
        }
        *reinterpret_cast<float4*>(a.DY0 + srow[i] * 64 + 4 * cq) = make_float4(dy[0], dy[1], dy[2], dy[3]);
    }
    }   // tiles
#pragma unroll
    for (int c = 0; c < 4; ++c) { red[0][slot][4 * cq + c] = s1[c]; red[1][slot][4 * cq + c] = s2[c]; }
    __syncthreads();
    if (t < 128) {
        const int q = t >> 6, c = t & 63;
        float tot = 0.f;
        for (int k = 0; k < 16; ++k) tot += red[q][k][c];
        atomicAdd((q ? a.o2 : a.o1) + (int64_t)n * 64 + c + stat_rep(), (double)tot);
        // (norm0's dbeta / dgamma are these sums over the streams: db_flush_kernel adds them - 3400 workgroups on one address per
        // channel here were the tail of this kernel)
    }
}

// ------------------------------------------------------------------------------------
// BN backward applied once: out = a*((g - q1) - (x - mean)*k) for a C-channel slice
// of one stream's plane (C = 32: the finalized gradient of a dense layer's output
// slice; C = 128: the gradient of a bottleneck through norm2).  The data- and
// weight-gradient GEMMs that follow then stream ONE array instead of two.
// ------------------------------------------------------------------------------------
struct BnBwdApplyArgs {
    const void* g; int ldg, gcoff;                // gradient storage of the mode
    const void* x; int ldx, xcoff;                // activation storage of the mode
    Plane pl; int C;
    const double* xsum; const double* xsq; int xstride; StatTab xtab;      // xtab: table of the same activation (columns as xcoff + t), or none
    const double* s1; const double* s2; int sstride, scoff;
    const float* gamma; float eps;
    void* out; int ldo;
    float* dbeta; float* dgamma;                  // if set: the affine gradients, dbeta += sum_n s1[n], dgamma += sum_n s2[n]
                                                  // (one fp32 atomic per stream and channel instead of one per producer tile)
    unsigned* amax;                               // if set: [streams][kAmaxRep] - the largest |out| of each stream as float bits (atomicMax per
                                                  // workgroup): the scale of the consumers' fp16-split operand (gemm.cuh, operand kind 3)
};

// rows of a plane per workgroup: the 16-bit modes move half the bytes per row, so a workgroup takes twice the rows
constexpr int bn_apply_rows(int prec) { return prec ? 128 : 64; }
template <int PREC>
static __global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdApplyArgs a) {
    using XT = act_t<PREC>;
    using GT = grd_t<PREC>;
    constexpr int E = 16 / GT::size;              // channels per 16-byte slot: 4 (fp32 storage) or 8 (16-bit storage)
    constexpr int RPW = bn_apply_rows(PREC);      // rows per workgroup
    static_assert(GT::size == XT::size, "one slot geometry");
    __shared__ float prm[4 * 128];
    const int n = blockIdx.y, t = threadIdx.x;
    const int spr = a.C / E;                      // slots per row
    const int rows_per_pass = 256 / spr;
    const int cs = t % spr;
    const int r0 = blockIdx.x * RPW + t / spr;
    // RPW rows per workgroup: up to 8 row slots per thread.  The data loads go out FIRST (unconditional, clamped row) so that
    // they share one memory round trip with the parameter prologue below; all loads before any store (the output may alias
    // the gradient input - it does for the in-place norm2 case).
    float4 gv[8], xv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int r = r0 + k * rows_per_pass;
        if (k * rows_per_pass >= RPW) break;           // (workgroup-uniform: C = 32 uses two slots, C = 128 eight)
        const int64_t pix = (int64_t)n * a.pl.HWp + (r < a.pl.HW ? r : 0);
        gv[k] = ld16(a.g, (int64_t)GT::size * (pix * a.ldg + a.gcoff + E * cs));
        xv[k] = ld16(a.x, (int64_t)XT::size * (pix * a.ldx + a.xcoff + E * cs));
    }
    if (t < a.C) {
        const double inv = 1.0 / (double)a.pl.HW;
        float mean, invstd;
        tab_or_moments(a.xtab, n, a.xcoff + t, a.xsum, a.xsq, (int64_t)n * a.xstride + a.xcoff + t, inv, a.eps, mean, invstd);
        const float q2 = (float)(stat_get(a.s2, (int64_t)n * a.sstride + a.scoff + t) * inv);
        prm[t] = (a.gamma ? a.gamma[t] : 1.f) * invstd;
        prm[a.C + t] = (float)(stat_get(a.s1, (int64_t)n * a.sstride + a.scoff + t) * inv);
        prm[2 * a.C + t] = mean;
        prm[3 * a.C + t] = invstd * q2;
        if (a.dbeta && blockIdx.x == 0) {
            atomicAdd(a.dbeta + t, (float)stat_get(a.s1, (int64_t)n * a.sstride + a.scoff + t));
            atomicAdd(a.dgamma + t, (float)stat_get(a.s2, (int64_t)n * a.sstride + a.scoff + t));
        }
    }
    __syncthreads();
    float vmax = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int r = r0 + k * rows_per_pass;
        if (k * rows_per_pass < RPW && r < a.pl.HW) {
            const int64_t pix = (int64_t)n * a.pl.HWp + r;
            if constexpr (E == 4) {
                const float4 o = affine2(gv[k], xv[k], prm + 4 * cs, a.C);
                stq<GT>(a.out, pix * a.ldo + 4 * cs, o);
                vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            } else {
                const u32x4 u = pack_unit<1>(affine2(slot_quad<GT>(gv[k], 0), slot_quad<XT>(xv[k], 0), prm + 8 * cs, a.C),
                                             affine2(slot_quad<GT>(gv[k], 1), slot_quad<XT>(xv[k], 1), prm + 8 * cs + 4, a.C));   // gradients: bf16
                *reinterpret_cast<u32x4*>(static_cast<char*>(a.out) + (int64_t)GT::size * (pix * a.ldo + 8 * cs)) = u;
            }
        }
    }
    if constexpr (E == 4) {
        if (a.amax) {                             // (launch-uniform) largest |out| of the workgroup -> this stream's replica
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
            __syncthreads();                      // prm is free again
            if ((t & 63) == 0) prm[t >> 6] = vmax;
            __syncthreads();
            if (t == 0) {
                vmax = fmaxf(fmaxf(prm[0], prm[1]), fmaxf(prm[2], prm[3]));
                if (vmax > 0.f) atomicMax(a.amax + (int64_t)n * kAmaxRep + (blockIdx.x % kAmaxRep), __float_as_uint(vmax));
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// The same for the bottleneck gradient of precision mode 0 (C = 128, through norm2), written in UNIT form (gemm.cuh, kD2K8):
// a workgroup owns one 64-pixel scale block of one stream - it computes the block's 64 x 128 outputs in registers, takes their
// largest magnitude, scales by the power of two that puts it into [2^13, 2^14), splits into the two fp16 pieces and stores them
// as 16-byte units [piece][channel / 8][pixel]: what the three consumers' MFMAs read, with no arithmetic left for their k-loops.
// Rows of the plane padding are written as zeros (the consumers need no row mask).  Thread t: channel quad t / 8, rows t % 8 + 8 k -
// eight consecutive pixels of one quad side by side, so loads and stores both move 128-byte runs.
// ------------------------------------------------------------------------------------
struct BnBwdApplySplitArgs {
    const float* g;                               // raw 3x3 data gradient dy [n][HWp][128]
    const float* x;                               // the bottleneck activation norm2 normalised [n][HWp][128]
    Plane pl;
    const double* xsum; const double* xsq; int xstride; StatTab xtab;
    const double* s1; const double* s2; int sstride;
    const float* gamma; float eps;
    u32x4* out;                                   // units of this ring slot
    float* binv;                                  // [streams][HWp / 64] inverse block scales
    float* dbeta; float* dgamma;                  // norm2's affine gradients (one fp32 atomic per stream and channel)
};
static __global__ __launch_bounds__(256) void bn_bwd_apply_split_kernel(const BnBwdApplySplitArgs a) {
    constexpr int C = 8 * kD2K8;
    __shared__ float prm[4 * C];
    __shared__ float red[4];
    const int n = blockIdx.y, t = threadIdx.x;
    const int cs = t >> 3, rr = t & 7;
    const int r0 = blockIdx.x * kScaleBlock + rr;
    float4 gv[8], xv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {                 // the data loads first: one memory round trip with the parameter prologue
        const int r = r0 + 8 * k;
        const int64_t pix = (int64_t)n * a.pl.HWp + (r < a.pl.HW ? r : 0);
        gv[k] = ld4(a.g + pix * C + 4 * cs);
        xv[k] = ld4(a.x + pix * C + 4 * cs);
    }
    if (t < C) {
        const double inv = 1.0 / (double)a.pl.HW;
        float mean, invstd;
        tab_or_moments(a.xtab, n, t, a.xsum, a.xsq, (int64_t)n * a.xstride + t, inv, a.eps, mean, invstd);
        const double s1 = stat_get(a.s1, (int64_t)n * a.sstride + t), s2 = stat_get(a.s2, (int64_t)n * a.sstride + t);
        prm[t] = a.gamma[t] * invstd;
        prm[C + t] = (float)(s1 * inv);
        prm[2 * C + t] = mean;
        prm[3 * C + t] = invstd * (float)(s2 * inv);
        if (a.dbeta && blockIdx.x == 0) {
            atomicAdd(a.dbeta + t, (float)s1);
            atomicAdd(a.dgamma + t, (float)s2);
        }
    }
    __syncthreads();
    float vmax = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float4 o = r0 + 8 * k < a.pl.HW ? affine2(gv[k], xv[k], prm + 4 * cs, C) : zero4();
        gv[k] = o;
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if ((t & 63) == 0) red[t >> 6] = vmax;
    __syncthreads();
    vmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    ActScale sc{1.f, 1.f};
    if (vmax > 0.f) sc = scale_of_max(__float_as_uint(vmax));      // (a block of zeros keeps scale 1)
    if (t == 0) a.binv[(int64_t)n * (a.pl.HWp / kScaleBlock) + blockIdx.x] = sc.inv;
    char* ob = reinterpret_cast<char*>(a.out + d2_stream_units(n, a.pl.HWp));
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const Split4 s = split4<3>(mul4(gv[k], sc.s));
        const int64_t at = ((int64_t)(cs >> 1) * a.pl.HWp + r0 + 8 * k) * 16 + (cs & 1) * 8;
        *reinterpret_cast<uint2*>(ob + at) = s.p[0];
        *reinterpret_cast<uint2*>(ob + at + (int64_t)kD2K8 * a.pl.HWp * 16) = s.p[1];
    }
}

// Sum the replicas of the dbeta / dgamma scratch (engine.h) into the gradient array and leave the scratch zeroed for the next call.
struct DbSegD { int64_t grad_off; int scr_off; int n; };
static __global__ void db_flush_kernel(const DbSegD* segs, float* scr, int rep_stride, int reps, float* grads,
                                       const double* stem1, const double* stem2, int ns, float* dbeta0, float* dgamma0) {
    if (blockIdx.y == gridDim.y - 1) {       // norm0: dbeta / dgamma = pool0_bwd's per-stream sums, summed over the streams
        if (blockIdx.x == 0 && threadIdx.x < 128 && stem1) {
            const int q = threadIdx.x >> 6, c = threadIdx.x & 63;
            double v = 0.0;
            for (int n = 0; n < ns; ++n) v += stat_get(q ? stem2 : stem1, (int64_t)n * 64 + c);
            (q ? dgamma0 : dbeta0)[c] += (float)v;
        }
        return;
    }
    const DbSegD sg = segs[blockIdx.y];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < sg.n; k += gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int r = 0; r < reps; ++r) {
            float* q = scr + (int64_t)r * rep_stride + sg.scr_off + k;
            v += *q;
            *q = 0.f;
        }
        grads[sg.grad_off + k] += v;
    }
}

// Zero the rows/columns of a gradient plane that an odd-sized 2x2/stride-2 average
// pool never reads (they receive no gradient from the transition).
template <int PREC>
static __global__ void zero_uncovered_kernel(void* G, int ld, Plane p, int Hc, int Wc, int C) {
    using GT = grd_t<PREC>;
    const int n = blockIdx.y;
    const int64_t total = (int64_t)p.HW * (C / 4);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int pix = (int)(i / (C / 4)), cq = (int)(i - (int64_t)pix * (C / 4));
        const int y = pix / p.W, x = pix - y * p.W;
        if (y >= Hc || x >= Wc) stq<GT>(G, ((int64_t)n * p.HWp + pix) * ld + 4 * cq, zero4());
    }
}

// ------------------------------------------------------------------------------------
// Losses (code/trainer.py:345-348 Huber on element [0,0,0,0]; :296-299 +
// code/utils.py:306-313 class-weighted cross entropy, weights {1,1,0}).
// ------------------------------------------------------------------------------------
static __global__ void loss_kernel(int mode, const float* q, const float* labels, int n_pairs, int per_pair,
                            float* loss, float* dq) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pairs) return;
    const float* qj = q + (int64_t)j * per_pair;
    float* dj = dq + (int64_t)j * per_pair;
    for (int i = 0; i < per_pair; ++i) dj[i] = 0.f;
    if (mode == 0) {
        const float d = qj[0] - labels[j];
        if (fabsf(d) < 1.f) { loss[j] = 0.5f * (d * d); dj[0] = d; }
        else { loss[j] = fabsf(d) - 0.5f; dj[0] = d > 0.f ? 1.f : -1.f; }
    } else {
        // labels outside {0, 1, 2} count as class 2 ("no loss", weight 0); the reference only ever emits 0 / 1
        // (code/trainer.py:220-234).  A zero class weight gives loss 0 and no gradient - torch's weighted mean
        // would be 0/0 there and poison every gradient and Adam moment with NaN.
        const float lf = labels[j];
        const int y = (lf >= 0.f && lf < 2.5f) ? (int)lf : 2;
        const float m = fmaxf(qj[0], fmaxf(qj[1], qj[2]));
        const float e0 = expf(qj[0] - m), e1 = expf(qj[1] - m), e2 = expf(qj[2] - m);
        const float se = e0 + e1 + e2, lse = logf(se) + m;
        if (y == 2) {
            loss[j] = 0.f;
        } else {
            loss[j] = lse - qj[y];                   // size_average: weighted mean over 1 element of weight 1
            const float sm[3] = {e0 / se, e1 / se, e2 / se};
            for (int c = 0; c < 3; ++c) dj[c] = sm[c] - (c == y ? 1.f : 0.f);
        }
    }
}

// ------------------------------------------------------------------------------------
// Heightmap generation, the step in front of Trainer.forward (utils.get_heightmap, code/utils.py:38-68): camera depth
// image -> robot-frame z of every pixel (get_pointcloud + cam_pose, :12-47) -> perspective warp onto the table
// (cv2.warpPerspective with an INTER_LINEAR / BORDER_CONSTANT inverse map, :62-66), fused: one thread per heightmap
// pixel gathers its four taps and converts them on the fly.  Arithmetic restated in oracle/heightmap.py: coordinates in
// double, rounded to 1/32 pixel, float32 table weights.
// ------------------------------------------------------------------------------------
struct HeightmapArgs {
    const double* depth; int h, w;          // camera depth image [h][w]
    double fx, fy, cx, cy;                  // intrinsics
    double r20, r21, r22, t2;               // third row of the camera pose (robot-frame z)
    double mi[9];                           // INVERSE of the source -> heightmap homography
    double* out; int ow, oh;
};
static __global__ void heightmap_warp_kernel(const HeightmapArgs a) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.ow) return;
    const double xd = (double)x, yd = (double)y;
    const double den = a.mi[6] * xd + a.mi[7] * yd + a.mi[8];
    const double scale = den != 0.0 ? 32.0 / den : 0.0;
    double fxs = (a.mi[0] * xd + a.mi[1] * yd + a.mi[2]) * scale, fys = (a.mi[3] * xd + a.mi[4] * yd + a.mi[5]) * scale;
    fxs = fmin(fmax(fxs, -2147483648.0), 2147483647.0);
    fys = fmin(fmax(fys, -2147483648.0), 2147483647.0);
    const long long ix = (long long)rint(fxs), iy = (long long)rint(fys);      // saturate_cast<int>: round half to even
    const long long x0 = ix >> 5, y0 = iy >> 5;
    const float ax = (float)(ix & 31) * (1.0f / 32.0f), ay = (float)(iy & 31) * (1.0f / 32.0f);
    const float w00 = (1.0f - ay) * (1.0f - ax), w01 = (1.0f - ay) * ax, w10 = ay * (1.0f - ax), w11 = ay * ax;
    auto tap = [&](long long yy, long long xx) -> double {
        if (yy < 0 || yy >= a.h || xx < 0 || xx >= a.w) return 0.0;
        const double d = a.depth[yy * a.w + xx];
        const double px = ((double)xx - a.cx) * (d / a.fx), py = ((double)yy - a.cy) * (d / a.fy);
        return a.r20 * px + a.r21 * py + a.r22 * d + a.t2;
    };
    a.out[(int64_t)y * a.ow + x] = tap(y0, x0) * (double)w00 + tap(y0, x0 + 1) * (double)w01 + tap(y0 + 1, x0) * (double)w10 +
                                   tap(y0 + 1, x0 + 1) * (double)w11;
}

// Largest value and its index, exactly like np.argmax: lowest index on ties, and a NaN beats every number (the FIRST NaN
// wins) - a diverged network surfaces as a NaN best value instead of an arbitrary but valid-looking action.  One workgroup.
__device__ __forceinline__ bool argmax_better(float x, int i, float bx, int bi) {
    if (bi == 0x7fffffff) return true;                  // nothing held yet
    const bool xn = x != x, bn = bx != bx;
    if (xn != bn) return xn;
    if (xn) return i < bi;
    return x > bx || (x == bx && i < bi);
}
static __global__ __launch_bounds__(256) void argmax_kernel(const float* v, int n, int* idx_out, float* val_out) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    const int t = threadIdx.x;
    float best = -INFINITY; int at = 0x7fffffff;
    for (int i = t; i < n; i += 256) {
        const float x = v[i];
        if (argmax_better(x, i, best, at)) { best = x; at = i; }
    }
    bv[t] = best; bi[t] = at;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) {
            const float x = bv[t + s]; const int j = bi[t + s];
            if (j != 0x7fffffff && argmax_better(x, j, bv[t], bi[t])) { bv[t] = x; bi[t] = j; }
        }
        __syncthreads();
    }
    if (t == 0) { *idx_out = bi[0] == 0x7fffffff ? 0 : bi[0]; *val_out = bv[0]; }
}

// ------------------------------------------------------------------------------------
// BN running statistics: momentum 0.1, unbiased variance,
// one update per entry of the reference-order sequence, num_batches_tracked += len.
// ------------------------------------------------------------------------------------
struct BnUpdDesc { int64_t rm, rv, nbt; int64_t stat_off; int stride, coff, C, count, head; };

static __global__ void bn_update_kernel(const BnUpdDesc* descs, const double* stats_sum, const double* stats_sq,
                                 float* bufs, int64_t* nbt, const int* seq_trunk, int n_trunk,
                                 const int* seq_head, int n_head,
                                 const int* pair_a, const int* pair_b, int n_pairs, int per_pair, float* q_out, int n_streams, int update) {
    // update == 0 (a forward that leaves the running statistics alone): only the non-finite check below, over every stream / pair of the batch
    const BnUpdDesc d = descs[blockIdx.y];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int* seq = d.head ? seq_head : seq_trunk;
    const int ns = update ? (d.head ? n_head : n_trunk) : (d.head ? n_pairs : n_streams);
    if (update && c == 0 && ns > 0) nbt[d.nbt] += ns;
    if (c >= d.C || ns == 0) return;
    double rm = bufs[d.rm + c], rv = bufs[d.rv + c];
    const double inv = 1.0 / (double)d.count;
    for (int i = 0; i < ns; ++i) {
        const int sq = update ? seq[i] : i;
        const int64_t idx = d.stat_off + (int64_t)sq * d.stride + d.coff + c;
        const double m = fstat_get(stats_sum, idx) * inv;
        double var = fstat_get(stats_sq, idx) * inv - m * m;
        var = var < 0 ? 0 : var;
        const double unb = var * (double)d.count / (double)(d.count - 1);
        rm = (double)(float)(0.1 * m + 0.9 * rm);
        rv = (double)(float)(0.1 * unb + 0.9 * rv);
        // A non-finite batch statistic (NaN / inf anywhere in that channel of that sample) makes every later activation of the
        // sample NaN in the reference: BN turns the whole channel into NaN, torch's relu / max_pool keep NaN, the next convolution
        // sums over the channel.  The kernels' v_max_f32 ReLU returns the non-NaN operand instead, so the NaN would die at the
        // first ReLU and a valid-looking Q come out - a diverged network, or the released mean = std = 0 constants
        // (code/trainer.py:176-185, golden G2), would go unnoticed.  Restore the reference's result here: Q of every sample that
        // used the stream (trunk statistics) / of the pair (head statistics) becomes NaN.
        if (!(m - m == 0.0) || !(var - var == 0.0)) {
            const int s = sq;
            for (int p = 0; p < n_pairs; ++p)
                if (d.head ? p == s : (pair_a[p] == s || pair_b[p] == s))
                    for (int j = 0; j < per_pair; ++j) q_out[(int64_t)p * per_pair + j] = __builtin_nanf("");
        }
    }
    if (update) {
        bufs[d.rm + c] = (float)rm;
        bufs[d.rv + c] = (float)rv;
    }
}

// ------------------------------------------------------------------------------------
// Adam (torch.optim.Adam, no amsgrad, no weight decay).
// ------------------------------------------------------------------------------------
static __global__ void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                            float eps, float bc1, float bc2_sqrt) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);           // torch: exp_avg.lerp_(grad, 1-beta1)
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;           // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - (lr / bc1) * (mi / denom);
    }
}

// The same step with its two step-dependent scalars (lr / (1 - beta1^t), sqrt(1 - beta2^t)) read from device memory: the form
// a captured training step replays (smg_train_step_graph) - the host refreshes the scalars, the graph's kernel arguments stay.
static __global__ void adam_dev_kernel(float* p, const float* g, float* m, float* v, int64_t n, const float* sc, float b1, float b2, float eps) {
    const float lr = sc[0], bc2_sqrt = sc[1];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - (lr / 1.f) * (mi / denom);
    }
}

}  // namespace smg
