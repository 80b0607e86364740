// gemm.cuh - one split-precision MFMA implicit-GEMM kernel template for gfx950 and the
// policies (operand fetchers + epilogues) that turn it into every convolution
// forward / data-gradient / weight-gradient of the DenseNet-121 affordance path.
//
//   D[i][j] = sum_r A(i, r) * B(r, j)
//
// Arithmetic: fp32 in, fp32 out, on the 16-bit matrix cores, in one of two split schemes ("operand kinds", below):
// kind 3 (the dense layers' products, since round 4): a scaled two-piece fp16 split, three v_mfma_f32_32x32x16_f16 terms;
// kind 0 (rounds 2-3; still the stem's, the transitions' and the head's gradient products): every fp32 operand is split into
// three bf16 pieces  x = hi + mid + lo  (round-to-nearest pieces, v_cvt_pk_bf16_f32; the residuals
// are exact, 3 x 8 significand bits cover the 24 of an fp32), and a product is accumulated from six
// v_mfma_f32_32x32x16_bf16 terms
//   hi*lo + lo*hi + mid*mid + hi*mid + mid*hi + hi*hi            (fp32 accumulate)
// - the three dropped terms are <= 2^-24 of the product.  Measured on the MI355X against fp64
// (tools/split_probe.hip): error / sum|a*b| = 2.9e-8 rms at K = 1024, the same as the exact
// fp32 FMA chain of v_mfma_f32_32x32x2_f32 (2.7e-8) - at 16/6 = 2.7x its issue rate.
//
// Data layout: activations are NHWC fp32, one "plane" of HWp = roundup(H*W, 64)
// pixel rows per stream, so a 64-row (or, where HWp % 128 == 0, 128-row) tile never
// straddles two streams and a dense block is ONE buffer [stream][pixel][Ctot] that
// every layer appends 32 channels to (torch.cat of code/models.py:386 / torchvision
// _DenseBlock disappears).
//
// LDS images.  The bf16 MFMA wants, per lane, 8 CONSECUTIVE k of one row: a "unit" = 8 k of one
// row of one piece = 16 bytes.
//   * forward / data gradient (Cfg::AT): the A operand arrives pixel-major (a thread holds 4
//     consecutive channels = k of one pixel row), gets its BN / ReLU / BN-backward transform and
//     the split at LDS-store time and lands as units [piece][k/8][row] (ds_write_b64); the B operand
//     (weights) is pre-split by pack_weights_kernel into the SAME unit layout in HBM
//     [piece][K/8][N], so its staging is a straight 16-byte copy.  Fragments are one ds_read_b128
//     per (piece, 32-row tile): consecutive lanes read consecutive units, conflict-free.
//   * weight gradient (!Cfg::AT): the reduction runs over PIXELS while memory is channel-major, so
//     both operands are stored as they arrive, row-major [pixel][channel] per piece, and the
//     fragments come out of LDS transposed with ds_read_b64_tr_b16 (two per piece and tile).
//
// Pipeline per k-tile: raw global loads of tile kt+1 are issued first and stay in
// flight across the MFMA block of tile kt; all operand fragments of the tile are read
// from LDS before its MFMAs; the transform + split is applied when tile kt+1 is written to
// the other LDS buffer; one barrier per k-tile.
//
// BatchNorm (training mode, per stream - SURVEY.md section 7) is never a kernel of
// its own: the producer's epilogue accumulates per-(stream, channel) sum / sum of
// squares in fp64 (in-lane, then fp64 atomics); the first consumer of a channel turns them into
// fp32 (mean, invstd) table entries - every workgroup for itself, one per stream for the table
// (BnTab) - and every later consumer's staging threads read the table per k-tile and apply
// BN + ReLU in the centered form (x - mean)*(gamma*invstd) + beta while staging the operand:
// a forward workgroup's parameter prologue is 32 channels, whatever K is.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

// k-tiles of global loads in flight per policy family (register ring depth of the staging pipeline) and the waves per SIMD a few
// kernels are held to - every value below was A/B-measured on the box (profiles/README.md) and is frozen here.
namespace smg {
// k-tiles in flight by LDS-DMA (gemm_tile; 0 = register staging) where both operands are finished 16-byte units.  Serialised ms per step,
// register staging / 1 / 2 / 3 tiles in flight: per-layer data gradient 128 x 64 x 16 0.77-0.82 / 0.74 / 0.69 / 0.69 and 64 x 64 x 32 0.31-0.34 / 0.31 /
// 0.31 / 0.32; grouped 128 x 64 x 32 1.24 / 1.12 / 1.08 / 1.54 (four stages: one workgroup per CU) - and 1.04 with ONE tile in flight at
// three waves per SIMD (two stages of LDS + 168 registers let a third workgroup in); grouped 64 x 64 x 32 0.89 / 0.78 / 0.84 / 1.00.
// What the DMA buys is the staging work (no registers, no ds_write pass), not depth: occupancy decides between the depths.
constexpr int kDmaFlyDgrad = 2, kDmaFlyGroup = 1;
constexpr int kPdFwdSmall = 3;      // one MFMA tile per wave (small planes): 0.1 us of MFMAs per k-tile against ~1 us of memory latency (round 5, 2 -> 3: the 32 x 64
                                    // forward of the 20^2 planes 18.3 -> 17.1 us per launch, the single-sample step's kernels 7.91 -> 7.75 ms; 4 measures like 3)
constexpr int kPdFwdBig = 2;        // 128 x 128 forward: with three-term products 384 cycles of MFMA per k-tile no longer cover a load (72.4 -> 67.6 us per launch)
constexpr int kPdDgrad = 2;
constexpr int kPdDgradBig = 2;
constexpr int kPdWgrad = 3;
constexpr int kPdDgradGroup = 1;    // the layer-grouped 1x1 data gradient with deep k-tiles
constexpr int kFwdBigMinWaves = 3;  // waves per SIMD the 128 x 128 forward is held to (register cap 512 / n)
}
// The next tile's loads are pinned AT the top of the k-tile (sched_barrier): hipcc otherwise sinks them to their first use (the weight
// loads ended up directly in front of their LDS store, the activation loads in the middle of the MFMA block - one exposed memory
// latency each per k-tile; serialised kernel total 19.33 -> 18.94 ms).  The order of fragment reads and MFMAs inside a k-tile is
// left to hipcc (with two k-tiles of loads in flight it interleaves the next tile's split / LDS stores with the MFMAs).

namespace smg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // one 16-byte unit (HIP's uint4 struct keeps register arrays in scratch)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct Plane { int H, W, HW, HWp; };

// ------------------------------------------------------------------------------------
// Split precision.
// ------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr int NPIECE = 3;
// Precision mode of an engine (compile-time PREC of every kernel; smg_engine_set_precision picks the instantiation):
//   0  fp32 storage; fp32-class products (default; what the parity suite gates): operand kind 3 for the dense layers, kind 0 elsewhere
//   1  bf16 storage of activations AND gradients (dense-block buffers, bottlenecks, G', the backward ring), ONE
//      v_mfma_f32_32x32x16_bf16 per product                                              (BASELINE.json config 3)
//   2  fp16 storage of activations with fp16 forward products (v_mfma_f32_32x32x16_f16); gradients are stored and multiplied in
//      bf16 (fp32 exponent range: no loss scaling, no underflow of small gradients)      (config 5)
// In every mode BN statistics, every accumulation, the parameters, their gradients and Adam stay fp32 (fp64 for the sums).
// Operand kind of one kernel's MFMAs (OP): 0 = 3-piece bf16 split (six terms), 1 = bf16, 2 = fp16, 3 = scaled 2-piece fp16 split (three terms).
struct e_f32 { static constexpr int size = 4; };
struct e_bf16 { static constexpr int size = 2; };
struct e_f16 { static constexpr int size = 2; };
template <int PREC> struct ActT { using type = e_f32; };          // storage type of activations
template <> struct ActT<1> { using type = e_bf16; };
template <> struct ActT<2> { using type = e_f16; };
template <int PREC> struct GrdT { using type = e_f32; };          // storage type of activation gradients
template <> struct GrdT<1> { using type = e_bf16; };
template <> struct GrdT<2> { using type = e_bf16; };
template <int PREC> using act_t = typename ActT<PREC>::type;
template <int PREC> using grd_t = typename GrdT<PREC>::type;
// Operand kind 3: the fp32-class products of the HOT classes (dense-layer 1x1 / 3x3 forward, data and weight gradients) as a
// TWO-piece fp16 split  x * s = h + l  (round-to-nearest pieces, v_cvt_pk_f16_f32; s a power of two that places the operand in
// fp16's range: 2 x 11 significand bits) and THREE v_mfma_f32_32x32x16_f16 terms  h*l + l*h + h*h  - half the matrix work, two
// thirds of the LDS bytes and half the split arithmetic of kind 0.  Measured against fp64 (tools/split16_probe.hip): error /
// sum|a*b| 1.03-1.09x the exact fp32 FMA chain at K = 64 / 128, 0.87x at K = 288 (kind 0: 1.06-1.11x), provided the operand's
// typical magnitude sits well above fp16's subnormal floor - which is what the scales are for (the fp16 MFMA honours subnormal
// inputs, measured): activations by a per-BN-layer scale from gamma / beta (scale_kernel), weights per tensor from their
// maximum (header unit in front of every pack), gradients per (tensor, stream) from the maximum their producer recorded
// (amax_scale).  The accumulator is multiplied by the inverse scales (exact powers of two) in the epilogue.
#ifndef SMG_SPLIT16
#define SMG_SPLIT16 1        // 0: every fp32-class product on the 3-piece bf16 split (kind 0), as in rounds 2-3 (A/B)
#endif
constexpr int kSplitOp = SMG_SPLIT16 ? 3 : 0;
constexpr int np_of(int op) { return op == 0 ? NPIECE : (op == 3 ? 2 : 1); }      // pieces per operand
constexpr int fwd_op(int prec) { return prec ? prec : kSplitOp; }         // operand kind of the hot forward kernels
constexpr int bwd_op(int prec) { return prec ? 1 : kSplitOp; }            // ... of the hot backward kernels (gradients are bf16 in both 16-bit modes)
constexpr int fwd_op_plain(int prec) { return prec; }                     // policies that keep kind 0 in mode 0 (stem: unscaled image operand)
constexpr int bwd_op_plain(int prec) { return prec ? 1 : 0; }             // ... (gradient operands transformed on the fly: no recorded maximum)
// bf16 pieces of 4 floats (element 0 in the low half of .x): 8 bytes per piece
struct Split4 { uint2 p[NPIECE]; };
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// round-to-nearest-even pieces: one v_cvt_pk_bf16_f32 per pair
__device__ __forceinline__ unsigned pack_bf16(float lo_elem, float hi_elem) {
    const f32x2 v = {lo_elem, hi_elem};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ unsigned pack_f16(float lo_elem, float hi_elem) {
    const f32x2 v = {lo_elem, hi_elem};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
// the fp16 residuals rn(a - lo(h)), rn(b - hi(h)) of a rounded pair h, ONE instruction per element: v_fma_mix{lo,hi}_f16 evaluates
// a * 1.0 - h in fp32 (exact: h is a rounded to 11 bits) and rounds to fp16 - bit-identical to convert / subtract / convert, subnormal
// residuals included (tools/mix_probe.hip: 4 M pairs over 2^-30 .. 2^15), at half the split's instructions (3 instead of 6 per pair).
__device__ __forceinline__ unsigned resid_f16(float a, float b, unsigned h) {
    unsigned l;                     // (mixlo keeps the destination's high half, which mixhi then overwrites: no initialisation needed)
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "v"(h));
    return l;
}
__device__ __forceinline__ float f16_lo(unsigned u) { return (float)__builtin_bit_cast(f16x2, u).x; }
__device__ __forceinline__ float f16_hi(unsigned u) { return (float)__builtin_bit_cast(f16x2, u).y; }
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }
// x = hi + mid + lo: the residuals x - hi and (x - hi) - mid are exact in fp32, the last one has <= 9 significant bits
template <int OP = 0>
__device__ __forceinline__ Split4 split4(float4 v) {
    Split4 o;
    if constexpr (OP == 3) {                             // two fp16 pieces of an operand the caller has scaled into range
        const unsigned h01 = pack_f16(v.x, v.y), h23 = pack_f16(v.z, v.w);
        o.p[0] = make_uint2(h01, h23);
        o.p[1] = make_uint2(resid_f16(v.x, v.y, h01), resid_f16(v.z, v.w, h23));
        o.p[2] = make_uint2(0u, 0u);
        return o;
    } else
    if constexpr (OP != 0) {                             // single-piece operands
        o.p[0] = OP == 1 ? make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)) : make_uint2(pack_f16(v.x, v.y), pack_f16(v.z, v.w));
        o.p[1] = o.p[2] = make_uint2(0u, 0u);
        return o;
    }
    const unsigned h01 = pack_bf16(v.x, v.y), h23 = pack_bf16(v.z, v.w);
    const float r0 = v.x - bf16_lo(h01), r1 = v.y - bf16_hi(h01), r2 = v.z - bf16_lo(h23), r3 = v.w - bf16_hi(h23);
    const unsigned m01 = pack_bf16(r0, r1), m23 = pack_bf16(r2, r3);
    const float s0 = r0 - bf16_lo(m01), s1 = r1 - bf16_hi(m01), s2 = r2 - bf16_lo(m23), s3 = r3 - bf16_hi(m23);
    o.p[0] = make_uint2(h01, h23);
    o.p[1] = make_uint2(m01, m23);
    o.p[2] = make_uint2(pack_bf16(s0, s1), pack_bf16(s2, s3));
    return o;
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_f16(const u32x4& a, const u32x4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// one term of a single-piece operand kind
template <int OP>
__device__ __forceinline__ f32x16 mfma_1p(const u32x4& a, const u32x4& b, f32x16 c) {
    if constexpr (OP == 2 || OP == 3) return mfma_f16(a, b, c); else return mfma_bf16(a, b, c);
}
// ---- typed element access: `base` + element index -> fp32 values, whatever the array stores ---------------------------
template <class T> __device__ __forceinline__ float2 cvt2(unsigned u) {          // two packed 16-bit elements -> fp32
    if constexpr (std::is_same<T, e_bf16>::value) return make_float2(bf16_lo(u), bf16_hi(u));
    else return make_float2(f16_lo(u), f16_hi(u));
}
template <class T> __device__ __forceinline__ unsigned pack2(float a, float b) {
    if constexpr (std::is_same<T, e_bf16>::value) return pack_bf16(a, b); else return pack_f16(a, b);
}
// 4 consecutive elements from element index idx (idx % 4 == 0)
template <class T> __device__ __forceinline__ float4 ldq(const void* base, int64_t idx) {
    if constexpr (std::is_same<T, e_f32>::value) return *reinterpret_cast<const float4*>(static_cast<const float*>(base) + idx);
    else {
        const uint2 u = *reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(base) + idx);
        const float2 a = cvt2<T>(u.x), b = cvt2<T>(u.y);
        return make_float4(a.x, a.y, b.x, b.y);
    }
}
template <class T> __device__ __forceinline__ void stq(void* base, int64_t idx, float4 v) {
    if constexpr (std::is_same<T, e_f32>::value) *reinterpret_cast<float4*>(static_cast<float*>(base) + idx) = v;
    else *reinterpret_cast<uint2*>(static_cast<unsigned short*>(base) + idx) = make_uint2(pack2<T>(v.x, v.y), pack2<T>(v.z, v.w));
}
template <class T, class I> __device__ __forceinline__ float ld1(const void* base, I idx) {
    if constexpr (std::is_same<T, e_f32>::value) return static_cast<const float*>(base)[idx];
    else if constexpr (std::is_same<T, e_bf16>::value) return __uint_as_float((unsigned)static_cast<const unsigned short*>(base)[idx] << 16);
    else return (float)static_cast<const _Float16*>(base)[idx];
}
template <class T, class I> __device__ __forceinline__ void st1(void* base, I idx, float v) {
    if constexpr (std::is_same<T, e_f32>::value) static_cast<float*>(base)[idx] = v;
    else if constexpr (std::is_same<T, e_bf16>::value) static_cast<unsigned short*>(base)[idx] = (unsigned short)(pack_bf16(v, 0.f) & 0xFFFFu);
    else static_cast<_Float16*>(base)[idx] = (_Float16)v;
}
// a 16-byte slot as fetched: 4 fp32 elements, or 8 16-bit elements (two quads); quad h of it as fp32
template <class T> __device__ __forceinline__ float4 slot_quad(float4 raw, int h) {
    if constexpr (std::is_same<T, e_f32>::value) return raw;
    else {
        const unsigned u0 = __float_as_uint(h ? raw.z : raw.x), u1 = __float_as_uint(h ? raw.w : raw.y);
        const float2 a = cvt2<T>(u0), b = cvt2<T>(u1);
        return make_float4(a.x, a.y, b.x, b.y);
    }
}
// two fp32 quads -> one 16-byte operand unit (8 consecutive k) of a single-piece operand kind
template <int OP> __device__ __forceinline__ u32x4 pack_unit(float4 lo, float4 hi) {
    if constexpr (OP == 2) return u32x4{pack_f16(lo.x, lo.y), pack_f16(lo.z, lo.w), pack_f16(hi.x, hi.y), pack_f16(hi.z, hi.w)};
    else return u32x4{pack_bf16(lo.x, lo.y), pack_bf16(lo.z, lo.w), pack_bf16(hi.x, hi.y), pack_bf16(hi.z, hi.w)};
}
__device__ __forceinline__ u32x4 as_u4(float4 v) { return u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}; }
// ------------------------------------------------------------------------------------
// Accumulator tiles <-> memory in full-width accesses.  A 32x32 MFMA accumulator has lane = column (channel) and
// register r = row (r & 3) + 8 (r >> 2) + 4 half: a group of four registers 4g .. 4g + 3 of the four lanes of a quad is a
// 4 x 4 block of 4 consecutive rows x 4 consecutive channels.  Transposing it inside the quad (two DPP butterfly steps, 12
// VALU per 16 values) leaves every lane with ONE row and four consecutive channels: the epilogues load / store 16 bytes
// (fp32) or 8 bytes (16-bit storage) per lane and instruction instead of one element - a quarter of the memory
// instructions (the epilogue of a 16-bit 1x1 data gradient issued three times the memory instructions of its k-loop).
// in : v[k] = element (row k of the block, this lane's column);  out: v[k] = element (this lane's row = lane & 3, column k)
// - the same call maps a loaded row segment back to the accumulator layout (the transpose is its own inverse).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float dpp_quad(float v, const int ctrl) {
    // quad_perm exchange inside every 4 lanes (ctrl: 0xB1 = xor 1, 0x4E = xor 2)
    return __int_as_float(ctrl == 0xB1 ? __builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)
                                       : __builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ void quad_transpose4(float& a0, float& a1, float& a2, float& a3) {
    const int lane = threadIdx.x;
    const bool o1 = lane & 1, o2 = lane & 2;
    float t;
    t = dpp_quad(o1 ? a0 : a1, 0xB1); if (o1) a0 = t; else a1 = t;
    t = dpp_quad(o1 ? a2 : a3, 0xB1); if (o1) a2 = t; else a3 = t;
    t = dpp_quad(o2 ? a0 : a2, 0x4E); if (o2) a0 = t; else a2 = t;
    t = dpp_quad(o2 ? a1 : a3, 0x4E); if (o2) a1 = t; else a3 = t;
}
// Store the four accumulator rows 4g .. 4g + 3 (values in accumulator layout) of a 32x32 tile whose element (row 0, column 0)
// sits at element index `at` of `base` with row pitch ld: one stq per lane.
template <class T>
__device__ __forceinline__ void store_acc_rows(void* base, int64_t at, int ld, int g, int half, int lane, float v0, float v1, float v2, float v3) {
    quad_transpose4(v0, v1, v2, v3);
    stq<T>(base, at + (int64_t)(8 * g + 4 * half + (lane & 3)) * ld + 4 * ((lane & 31) >> 2), make_float4(v0, v1, v2, v3));
}
// ... and the matching load in two halves, so that a workgroup can issue the loads at its start and leave them in flight across its
// k-loop: fetch_acc_rows returns the row segment as loaded (no instruction depends on it), finish_acc_rows converts and
// transposes it to accumulator layout (this lane's column, rows 4g .. 4g + 3) where the epilogue needs the values.
template <class T> struct RawQ { using type = float4; };
template <> struct RawQ<e_bf16> { using type = uint2; };
template <> struct RawQ<e_f16> { using type = uint2; };
template <class T> using rawq_t = typename RawQ<T>::type;
template <class T>
__device__ __forceinline__ rawq_t<T> fetch_acc_rows(const void* base, int64_t at, int ld, int g, int half, int lane) {
    const int64_t idx = at + (int64_t)(8 * g + 4 * half + (lane & 3)) * ld + 4 * ((lane & 31) >> 2);
    if constexpr (std::is_same<T, e_f32>::value) return *reinterpret_cast<const float4*>(static_cast<const float*>(base) + idx);
    else return *reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(base) + idx);
}
template <class T>
__device__ __forceinline__ void finish_acc_rows(const rawq_t<T>& q, float (&v)[4]) {
    if constexpr (std::is_same<T, e_f32>::value) { v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
    else { const float2 a = cvt2<T>(q.x), b = cvt2<T>(q.y); v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; }
    quad_transpose4(v[0], v[1], v[2], v[3]);
}
template <int BM_, int BN_, int BK_, int WM_, int WN_, int WK_, bool AT_>
struct GemmCfg {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, WK = WK_;
    static constexpr bool AT = AT_;  // forward / data-gradient form (A pixel-major, B packed weight units); false: weight gradient
    // 4 waves: WM x WN of them tile the output, WK of them split every k-tile (in-block
    // split-K for the small late stages, reduced through LDS before the epilogue)
    static_assert(WM * WN * WK == 4, "4 waves per workgroup");
    static constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    static_assert(TM >= 1 && TN >= 1 && TM * WM * 32 == BM && TN * WN * 32 == BN, "tile shape");
    static constexpr int KS = BK / 16 / WK;                  // MFMA k16-steps per wave per k-tile
    static_assert(KS >= 1 && KS * 16 * WK == BK, "k split");
    static constexpr int K8 = BK / 8;
    // ---- AT: unit images [piece][k8][row] of 16-byte units.  The A rows are padded so that the LDS stores of one lane
    // group (ds_write_b64: 16 lanes = 64 / BK rows x BK / 4 quads; ds_write_b128 of the 16-bit modes: 8 lanes x one unit)
    // spread over all 32 banks.
    static constexpr int PADU = 64 / BK > 0 ? 64 / BK : 1;
    static constexpr int LDUA = BM + PADU, LDUB = BN;
    // ---- !AT: row-major [k][channel] bf16 images; row stride = 64 (mod 128) bytes keeps the four rows a transposing
    // read gathers on distinct banks
    static constexpr int LDTA = BM + ((BM * 2) % 128 == 64 ? 0 : 32), LDTB = BN + ((BN * 2) % 128 == 64 ? 0 : 32);
    static constexpr int RED_FLOATS = (WK - 1) * WM * WN * TM * TN * 16 * 64;
    // Sizes that depend on the operand kind (NP pieces per operand: 3 for the fp32-class split, 1 for bf16 / fp16) and on
    // how many elements one 16-byte staging slot of a thread holds (AE for the A operand, BE for the weight gradient's B
    // operand: 4 fp32 or 8 16-bit elements).
    template <int NP, int AE, int BE, int NST = 2>
    struct G {
        static constexpr int A_BYTES = AT ? NP * K8 * LDUA * 16 : NP * BK * LDTA * 2;
        static constexpr int B_BYTES = AT ? NP * K8 * LDUB * 16 : NP * BK * LDTB * 2;
        static_assert(A_BYTES % 16 == 0 && B_BYTES % 16 == 0, "16-byte aligned LDS images");
        // staging: A (and the weight gradient's B) in 16-byte slots, (line, slot)
        static constexpr int A_Q = AT ? BK / AE : BM / AE;       // slots per tile line
        static constexpr int A_LINES = AT ? BM : BK;
        static constexpr int A_STEP = 256 / A_Q;
        static constexpr int A_N = (A_LINES + A_STEP - 1) / A_STEP;
        static constexpr int B_Q = BN / BE;
        static constexpr int B_STEP = 256 / B_Q;
        static constexpr int B_N = AT ? (NP * K8 * BN + 255) / 256 : (BK + B_STEP - 1) / B_STEP;    // AT: 16-byte unit copies
        // every staging slot of a thread maps inside the tile (no run-time range check, which would also make
        // hipcc drain vmcnt between the load groups of one k-tile)
        static constexpr bool A_FULL = A_LINES % A_STEP == 0, B_FULL = AT ? (NP * K8 * BN) % 256 == 0 : BK % B_STEP == 0;
        static constexpr int AB_FLOATS = NST * (A_BYTES + B_BYTES) / 4;      // NST LDS stages (2: register staging; kDmaFly + 1: LDS-DMA ring)
        static constexpr int TILE_FLOATS = AB_FLOATS > RED_FLOATS ? AB_FLOATS : RED_FLOATS;
    };
};
// LDS geometry of policy P's kernel
template <class P> using GeoOf = typename P::Cfg::template G<np_of(P::kOp), P::kAE, P::kBE, (P::kDmaFly > 0 ? P::kDmaFly + 1 : 2)>;

// What a fetch leaves in registers: the untouched global loads (NV of them) and whether the
// element exists at all (conv zero padding / padded pixel rows).  The BN transform is applied later,
// when the tile is written to LDS, so the loads stay in flight across the MFMA block.
template <int NV> struct RawT { float4 v[NV]; bool ok; };
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// Staged operand loads go through buffer descriptors: workgroup-uniform base (SGPRs) + a loop-invariant 32-bit byte offset per
// lane (VGPR) + a per-k-tile scalar byte offset (SGPR) - no address arithmetic on the VALU inside the k-loop (a flat
// global_load needs a 64-bit add per load per k-tile; hipcc widens the lane offset outside the loop and cannot pick the
// scalar-base form).  Bytes at or past `bytes` read as zero: lanes whose element does not exist carry kOOB as offset.
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }   // workgroup-uniform by construction
constexpr unsigned kOOB = 0xC0000000u;          // + any scalar offset (< 1 GiB) stays past every descriptor's extent
constexpr unsigned kWholeBuf = 0xBFFFFFFFu;     // extent of descriptors without a tight bound
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* ubase, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ubase), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 bload_u4(const void* ubase, unsigned bytes, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(ubase, bytes), (int)voff, (int)soff, 0));
}
// The same load as an LDS-DMA: `buffer_load_dwordx4 ... offen lds` writes the 64 lanes' 16-byte units to 64 CONSECUTIVE units of LDS from
// the wave-uniform byte address in M0 (the global side stays per-lane: base + voff + soff).  Nothing crosses the register file; the
// request counts in vmcnt like any load and stays in flight across s_barrier.  hipcc neither counts nor moves these statements: the
// issuing wave waits with dma_wait<N>() (in-order completion), other waves' reads additionally need a barrier behind that wait.
// (M0 is compiler-reserved: saved and restored inside the statement that uses it.)
struct UnitSrc { const void* base; unsigned bytes, voff, soff; };
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma_unit(const UnitSrc& u, unsigned lds_dst) {
    const unsigned long long a = (unsigned long long)u.base;
    i32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    rs.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xFFFFu));
    rs.z = __builtin_amdgcn_readfirstlane((int)u.bytes);
    rs.w = 0x00020000;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(u.voff), "s"(rs), "s"(__builtin_amdgcn_readfirstlane((int)u.soff)), "s"(__builtin_amdgcn_readfirstlane((int)lds_dst)) : "memory");
}
template <int N> __device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void dma_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float4 bload4(const void* ubase, unsigned bytes, unsigned voff, unsigned soff) {
    const u32x4 v = bload_u4(ubase, bytes, voff, soff);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 ld16(const void* base, int64_t byte_off) { return *reinterpret_cast<const float4*>(static_cast<const char*>(base) + byte_off); }
// Per-k-tile BN parameters of a thread's channel quad (forward policies): as fetched (mean, invstd, gamma, beta) and as
// applied (mean, gamma*invstd, beta), 4 channels each
// (native vector types: arrays / structs of HIP's float4 class that cross a branch end up in scratch)
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct KPrm4 { f32x4 mean, invstd, gamma, beta; };
struct KPrm3 { f32x4 mean, scale, beta; };
struct KPrm0 {};
__device__ __forceinline__ f32x4 ldv4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 bloadv4(const void* ubase, unsigned bytes, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, bload_u4(ubase, bytes, voff, soff));
}


// The BACKWARD fp64 statistic arena (s1 / s2 sums) exists in kStatRep replicas kStatRepStride doubles apart: a producer's workgroup
// adds to replica (workgroup index mod kStatRep) - hundreds of same-address fp64 atomics per (stream, channel) and launch serialise
// in the L2 - and every reader sums the replicas (stat_get).  Same box, ms per step, replicas 1 / 2 / 4 / 8: headline (17 streams,
// 160^2 planes) 21.64 / 21.71 / 21.74 / 21.88 - the readers pay; config 5's share (5 streams, 456^2 planes: 800-1600 row tiles per
// stream) 27.2 / 25.7 / 25.4 / 25.6; config 3 27.2 / - / 27.2 / 27.5.  Two it is.  The forward arena stays single: its readers (every
// workgroup's parameter prologue, 32-128 channels each) outnumber its writers (replicated eightfold: headline + 0.6 ms).
// The stride is a compile-time constant so that no kernel needs another argument; the engine checks that its arena fits.
constexpr int kStatRep = 2;
constexpr int64_t kStatRepStride = (int64_t)1 << 23;
__device__ __forceinline__ double stat_get(const double* p, int64_t idx) {
    double v = p[idx];
#pragma unroll
    for (int r = 1; r < kStatRep; ++r) v += p[idx + r * kStatRepStride];
    return v;
}
// Replicas of the FORWARD statistic arena (readers: every parameter prologue).  Same box, 1 / 2 / 4: headline 21.40 / 21.52 / 21.58 ms,
// config 5 share 25.4 / 24.5 / 24.7, config 3 27.0 / 27.1 / - ; with the backward's per-layer prologues on the fp32 tables (StatTab) 2
// replicas still cost the headline 0.15 ms: the forward's own first-consumer prologues pay.  Stays 1.
constexpr int kFStatRep = 1;
__device__ __forceinline__ double fstat_get(const double* p, int64_t idx) {
    double v = p[idx];
#pragma unroll
    for (int r = 1; r < kFStatRep; ++r) v += p[idx + r * kStatRepStride];
    return v;
}
__device__ __forceinline__ int64_t fstat_rep() { return (int64_t)((blockIdx.x + blockIdx.y + blockIdx.z) % kFStatRep) * kStatRepStride; }
__device__ __forceinline__ int64_t stat_rep() { return (int64_t)((blockIdx.x + blockIdx.y + blockIdx.z) % kStatRep) * kStatRepStride; }
// The backward's per-layer kernels take (mean, invstd) of a forward activation from the fp32 table the forward finished (the very
// floats bn_moments would produce from the fp64 sums) when the caller hands one over: two float loads instead of fp64 loads,
// a divide and a square root per channel and workgroup - and no reader of the forward sums left on the per-layer path.
struct StatTab { const float* mean; const float* invstd; int ld; };       // mean == nullptr: no table, use the sums
__device__ __forceinline__ void bn_moments(const double* sum, const double* sq, int64_t idx, double inv_cnt, float eps,
                                           float& mean, float& invstd) {
    const double m = fstat_get(sum, idx) * inv_cnt;
    double var = fstat_get(sq, idx) * inv_cnt - m * m;
    var = var < 0.0 ? 0.0 : var;
    mean = (float)m;
    invstd = (float)(1.0 / sqrt(var + (double)eps));
}
__device__ __forceinline__ void tab_or_moments(const StatTab& t, int n, int col, const double* sum, const double* sq, int64_t idx,
                                               double inv_cnt, float eps, float& mean, float& invstd) {
    if (t.mean) { mean = t.mean[(int64_t)n * t.ld + col]; invstd = t.invstd[(int64_t)n * t.ld + col]; }
    else bn_moments(sum, sq, idx, inv_cnt, eps, mean, invstd);
}

// ---- scales of operand kind 3 ---------------------------------------------------------------------------------------
// Weight packs carry a header unit in FRONT of their first unit: floats {s, 1 / s, 0, 0}, s the power of two that puts the
// tensor's largest |w| into [2^13, 2^14) (scale_kernel); the units hold the fp16 pieces of w * s.
__device__ __forceinline__ float pack_inv_scale(const u32x4* wp) { return reinterpret_cast<const float*>(wp - 1)[1]; }
// Activation scale of one BatchNorm + ReLU operand: {s, 1 / s} with s = 2^5 / (largest hypot(gamma_c, beta_c), rounded up to a
// power of two): relu(gamma * xhat + beta) * s <= 32 (|xhat| + 1) stays below fp16's maximum for |xhat| < 2046 (|xhat| <=
// sqrt(pixels per plane): planes up to 4 M pixels), typical values sit around 2^5 - 17 binades above the subnormal floor.
struct ActScale { float s, inv; };
__device__ __forceinline__ ActScale act_scale(const float* asc) { ActScale a; a.s = asc ? asc[0] : 1.f; a.inv = asc ? asc[1] : 1.f; return a; }
// Gradient scale of one (tensor, stream): its producer recorded the largest |element| as float bits in kAmaxRep replicas
// (atomicMax per workgroup); s = 2^(13 - floor(log2 max)) puts the maximum into [2^13, 2^14): elements down to 2^-17 of it keep
// all 22 bits, smaller ones an absolute error of 2^-39 of it.  The address is workgroup-uniform: scalar loads, scalar maxima.
constexpr int kAmaxRep = 16;
__device__ __forceinline__ ActScale scale_of_max(unsigned m) {      // m: the float bits of the largest |element|
    int e = (int)(m >> 23) - 127;                 // floor(log2 max) of a normal float (0 -> -127)
    e = e < -100 ? -100 : e;
    ActScale a;
    a.s = __uint_as_float((unsigned)(140 - e) << 23);      // 2^(13 - e)
    a.inv = __uint_as_float((unsigned)(114 + e) << 23);    // 2^(e - 13)
    return a;
}
__device__ __forceinline__ ActScale amax_scale(const unsigned* amax_n) {
    unsigned m = 0;
#pragma unroll
    for (int r = 0; r < kAmaxRep; ++r) m = max(m, amax_n[r]);
    return scale_of_max(m);
}
// The bottleneck gradient D2 (the norm2 backward of a dense layer's 3x3 data gradient) reaches its three consumers - the grouped and
// the per-layer 1x1 data gradient, the 1x1 weight gradient - in UNIT form (operand kind 3, precision mode 0): bn_bwd_apply_split_kernel
// (elem.cuh) writes the two fp16 pieces ONCE, as the 16-byte units the MFMAs take (8 consecutive channels of one pixel),
//     unit(stream n, piece pc, k8, pixel p)  at  (((n * 2 + pc) * kD2K8 + k8) * HWp + p)
// scaled per 64-pixel block of a stream by the power of two that puts the block's largest |element| into [2^13, 2^14)
// (inverse scales: [streams][HWp / 64] floats).  The consumers stage the operand with straight 16-byte copies - no scale, no
// split, no conversion in their k-loops (the 1x1 weight gradient used to redo them once per 64 input channels, the grouped data
// gradient once per 64 output channels) - and a block's scale is exact for that block (the per-stream maximum it replaces left
// elements below 2^-17 of the STREAM's maximum with fewer than 22 bits).
constexpr int kD2K8 = 16;                        // 128 bottleneck channels / 8
constexpr int kScaleBlock = 64;                  // pixels per scale block (= the plane padding granule: a tile never straddles two streams)
__device__ __forceinline__ int64_t d2_stream_units(int n, int HWp) { return (int64_t)n * 2 * kD2K8 * HWp; }
__device__ __forceinline__ float4 mul4(float4 v, float s) {      // two elements per instruction (v_pk_mul_f32)
    const f32x2 a = f32x2{v.x, v.y} * s, b = f32x2{v.z, v.w} * s;
    return make_float4(a.x, a.y, b.x, b.y);
}

// BatchNorm statistics of one activation buffer as fp32, finished ONCE per (stream, channel): mean | invstd tables of
// [rows][ld] (rows = streams or pairs), plus the affine parameters of the consuming BN layer.  Who writes them: the first
// consumer of a channel - every workgroup of that launch derives the (few) channels nobody has finished yet from the
// fp64 sums itself, and one designated workgroup per stream stores them for all later layers and for the backward.
struct BnTab { const float* mean; const float* invstd; int ld; const float* gamma; const float* beta; };
__device__ __forceinline__ const float* tab_mean(const BnTab& t, int n) { return t.mean + (int64_t)n * t.ld; }
__device__ __forceinline__ const float* tab_invstd(const BnTab& t, int n) { return t.invstd + (int64_t)n * t.ld; }

// BN + ReLU in the centered form (x - mean) * (gamma * invstd) + beta: no cancellation
// between x*scale and a pre-folded shift on near-constant channels.
// prm points at {mean[4]...}, scale at prm + stride, beta at prm + 2*stride.
__device__ __forceinline__ float bn1(float x, float mean, float sc, float beta) { return fmaf(x - mean, sc, beta); }
__device__ __forceinline__ float4 bnrelu4(float4 v, const float* prm, int stride) {
    const float* sc = prm + stride;
    const float* be = prm + 2 * stride;
    float4 r;
    r.x = fmaxf(bn1(v.x, prm[0], sc[0], be[0]), 0.f);
    r.y = fmaxf(bn1(v.y, prm[1], sc[1], be[1]), 0.f);
    r.z = fmaxf(bn1(v.z, prm[2], sc[2], be[2]), 0.f);
    r.w = fmaxf(bn1(v.w, prm[3], sc[3], be[3]), 0.f);
    return r;
}
// H: the result becomes an fp16 operand (kinds 2 and 3) - clamped to fp16's largest finite value, one v_med3_f32 in place of the
// v_max_f32.  In-range values are untouched; what the clamp catches are the rows of the PLANE PADDING, which are read as they lie
// (zeros): relu(gamma * (0 - mean) * invstd + beta) * s of a near-constant channel (|mean| / std in the thousands) is past 65504,
// would become inf, and in the 1x1 weight gradient - whose other operand is zero on those rows - inf * 0 = NaN in a whole column of dW.
constexpr float kF16Max = 65504.f;
template <bool H = false>
__device__ __forceinline__ float4 bnrelu4(float4 v, const KPrm3& k) {
    // two channels per instruction (v_pk_add_f32 / v_pk_fma_f32); the relu has no packed form
    const f32x2 a = __builtin_elementwise_fma(f32x2{v.x, v.y} - k.mean.xy, k.scale.xy, k.beta.xy);
    const f32x2 b = __builtin_elementwise_fma(f32x2{v.z, v.w} - k.mean.zw, k.scale.zw, k.beta.zw);
    if constexpr (H) return make_float4(__builtin_amdgcn_fmed3f(a.x, 0.f, kF16Max), __builtin_amdgcn_fmed3f(a.y, 0.f, kF16Max),
                                        __builtin_amdgcn_fmed3f(b.x, 0.f, kF16Max), __builtin_amdgcn_fmed3f(b.y, 0.f, kF16Max));
    else return make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(b.x, 0.f), fmaxf(b.y, 0.f));
}
// the same with an upper clamp: cap = +inf -> relu; cap = 0 -> 0 (rows of the plane padding)
__device__ __forceinline__ float4 bnrelu4(float4 v, const KPrm3& k, float cap) {
    float4 r;
    r.x = __builtin_amdgcn_fmed3f(bn1(v.x, k.mean.x, k.scale.x, k.beta.x), 0.f, cap);
    r.y = __builtin_amdgcn_fmed3f(bn1(v.y, k.mean.y, k.scale.y, k.beta.y), 0.f, cap);
    r.z = __builtin_amdgcn_fmed3f(bn1(v.z, k.mean.z, k.scale.z, k.beta.z), 0.f, cap);
    r.w = __builtin_amdgcn_fmed3f(bn1(v.w, k.mean.w, k.scale.w, k.beta.w), 0.f, cap);
    return r;
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// BN backward folded into an operand fetch, centered form:
//   v = a * ((g - q1) - (x - mean) * k),   a = gamma*invstd, q1 = mean(dy), k = invstd * mean(dy*xhat)
// prm -> a[4]; q1 at prm + stride; mean at prm + 2*stride; k at prm + 3*stride.
__device__ __forceinline__ float bnb1(float g, float x, float a, float q1, float mean, float k) {
    return a * fmaf(-(x - mean), k, g - q1);
}
__device__ __forceinline__ float4 affine2(float4 g, float4 x, const float* prm, int stride) {
    const float* q1 = prm + stride;
    const float* mu = prm + 2 * stride;
    const float* k = prm + 3 * stride;
    float4 r;
    r.x = bnb1(g.x, x.x, prm[0], q1[0], mu[0], k[0]);
    r.y = bnb1(g.y, x.y, prm[1], q1[1], mu[1], k[1]);
    r.z = bnb1(g.z, x.z, prm[2], q1[2], mu[2], k[2]);
    r.w = bnb1(g.w, x.w, prm[3], q1[3], mu[3], k[3]);
    return r;
}

// XCD-aware tile order.  Workgroup b is observed to run on XCD b % 8 (MI355X_MICROARCH.md, used for
// speed only): sibling tiles that re-read the same A operand (the N-tiles of one M-tile, or of one
// pixel chunk in a weight gradient) are made consecutive on ONE XCD so the re-reads hit its private L2
// instead of going out to the fabric.  nM == 0 -> plain (x, y, z) block coordinates.
// The grid is EXACT (nM * nN workgroups): the last nM % 8 majors, which do not fill a group of eight, are dealt round-robin behind the
// full groups (their tiles lose the shared XCD, nobody else does) - no padding workgroups that take a slot only to leave at once.
struct TileMap { int nM, nN, gx; };
__host__ __device__ __forceinline__ unsigned tile_grid(const TileMap& tm) { return (unsigned)tm.nM * (unsigned)tm.nN; }
__device__ __forceinline__ bool tile_decode(const TileMap& tm, int b, int& major, int& minor) {
    const int full = 8 * tm.nN * (tm.nM >> 3);
    if (b < full) {
        const int x = b & 7, slot = b >> 3;
        major = (slot / tm.nN) * 8 + x;
        minor = slot % tm.nN;
    } else {
        const int r = b - full, rem = tm.nM & 7;
        major = (tm.nM & ~7) + r % rem;
        minor = r / rem;
    }
    return true;
}

// Sum per-lane column partials over the rows of the whole workgroup tile.
// v[q][tn]: this lane's partial for column (wn0 + tn*32 + l31) of quantity q.
// On return threads t < BN hold the totals of column t in out[q].
template <class C, int NQ, class T>
__device__ __forceinline__ void block_col_reduce(T (&v)[NQ][C::TN], T* red, T (&out)[NQ]) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, half = lane >> 5;
    const int wmn = wave % (C::WM * C::WN);
    const int wm = wmn / C::WN, wn0 = (wmn % C::WN) * C::TN * 32;
    const bool owner = wave < C::WM * C::WN;            // wk == 0
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) v[q][tn] += __shfl_xor(v[q][tn], 32);
    __syncthreads();
    if (half == 0 && owner) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn) red[(q * C::WM + wm) * C::BN + wn0 + tn * 32 + l31] = v[q][tn];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        T s = 0;
        if (t < C::BN) {
#pragma unroll
            for (int w = 0; w < C::WM; ++w) s += red[(q * C::WM + w) * C::BN + t];
        }
        out[q] = s;
    }
}

// Dev instrumentation (SMG_TRACE_* in engine.hip): when set, thread 0 of every workgroup of a gemm_kernel launch
// stores s_memtime at five points: start | parameters ready | first tile staged | k-loop done | epilogue done.
// a cycle stamp the scheduler may not move: hipcc otherwise hoists s_memtime over whole phases
static __device__ __forceinline__ unsigned long long smg_stamp() {
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
static __device__ unsigned long long* g_smg_trace = nullptr;      // (one copy per translation unit; TraceScope sets its own)
#if defined(SMG_TRACE_ITER) || defined(SMG_TRACE_EPI) || defined(SMG_TRACE_PRO)
#define SMG_TRACE(slot) do {} while (0)
#else
#define SMG_TRACE(slot) do { if (trace) trace[slot] = smg_stamp(); } while (0)
#endif

// ------------------------------------------------------------------------------------
// The kernel.
// ------------------------------------------------------------------------------------
// Virtual block coordinates of one tile (what blockIdx was before the kernel became persistent).
struct VBlock { int x, y, z, linear; };

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <class P>
__device__ __forceinline__ void gemm_tile(const P& p, const VBlock vb, float* smem) {
    using C = typename P::Cfg;
    using Z = GeoOf<P>;
    constexpr int OP = P::kOp, NP = np_of(OP), AE = P::kAE, BE = P::kBE;
    static_assert((AE == 4 || AE == 8) && (BE == 4 || BE == 8), "16-byte staging slots");
    static_assert((OP != 0 && OP != 3) || (AE == 4 && BE == 4), "the fp32-class splits read fp32 storage");
    // LDS-DMA form (P::kDmaFly > 0; both operands finished 16-byte units): a ring of NST = kDmaFly + 1 stages, kDmaFly k-tiles in
    // flight per workgroup without a staging register - see the k-loop below.  Otherwise two stages fed through registers.
    constexpr int NFLY = P::kDmaFly, NST = NFLY > 0 ? NFLY + 1 : 2;
    constexpr bool DMA = NFLY > 0;
    char* As = reinterpret_cast<char*>(smem);
    char* Bs = As + NST * Z::A_BYTES;
    float* sp = smem + Z::TILE_FLOATS;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wk = wave / (C::WM * C::WN), wmn = wave % (C::WM * C::WN);
    const int wm0 = (wmn / C::WN) * C::TM * 32, wn0 = (wmn % C::WN) * C::TN * 32;

    unsigned long long* trace = (g_smg_trace && t == 0) ? g_smg_trace + 8 * (size_t)vb.linear : nullptr;
    SMG_TRACE(0);
#ifdef SMG_TRACE_PRO
    if (trace) trace[0] = smg_stamp();
#endif
    if (trace) trace[5] = __builtin_amdgcn_s_memrealtime();   // 100 MHz, one base for the whole device
    typename P::Ctx ctx;
    if (!p.init_ctx(ctx, vb)) return;      // tile made of padding rows only (block-uniform)
    const int KT = p.ktiles(ctx);

    f32x16 acc[C::TM][C::TN];
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int aq = t % Z::A_Q, al = t / Z::A_Q;
    const int bq = t % Z::B_Q, bl = t / Z::B_Q;
    // A operand that arrives as FINISHED units (P::kAUnit: the bottleneck gradient D2, see kD2K8): staged with straight 16-byte
    // copies like the packed weights.  Forward form: the usual image [piece][k8][row]; weight-gradient form: the units as they
    // are, [piece][channel / 8][pixel of the k-tile] with the planes AU_PLANE units apart (the 64 bytes of padding put the four
    // planes a transposing read gathers on distinct banks), read with the same ds_read_b64_tr_b16.
    constexpr bool AU = P::kAUnit;
    constexpr int AU_PLANE = C::BK + 4;
    constexpr int AU_UNITS = C::AT ? NP * C::K8 * C::BM : NP * (C::BM / 8) * C::BK;
    constexpr int A_CNT = AU ? AU_UNITS / 256 : Z::A_N;
    static_assert(!AU || AU_UNITS % 256 == 0, "unit copies: whole rounds of the workgroup");
    static_assert(!AU || C::AT || NP * (C::BM / 8) * AU_PLANE * 16 <= Z::A_BYTES, "the unit image fits the row-major one");
    using ARawT = typename std::conditional<AU, u32x4, typename P::ARaw>::type;
    // Register ring of PD k-tiles in flight (global loads issued PD tiles ahead of their LDS store).
    constexpr int PD = P::kPrefetch;
    ARawT ra[PD][A_CNT];
    typename P::BRaw rb[PD][Z::B_N];
    typename P::KPrm kp[PD];
    typename P::ARow arow[Z::A_N];
    typename P::DRow da[Z::A_N], db[C::AT ? 1 : Z::B_N];
    KPrm3 bfix[BE / 4] = {};   // weight gradient: BN parameters of this thread's (fixed) B channel quad(s), read from LDS once
    if constexpr (C::AT) {
        if constexpr (!AU) {
#pragma unroll
            for (int i = 0; i < Z::A_N; ++i) p.a_row_init(ctx, arow[i], al + i * Z::A_STEP);
        }
    } else {
        if constexpr (!AU) {
#pragma unroll
            for (int i = 0; i < Z::A_N; ++i) p.d_init(ctx, da[i], al + i * Z::A_STEP);
        }
#pragma unroll
        for (int i = 0; i < Z::B_N; ++i) p.d_init(ctx, db[i], bl + i * Z::B_STEP);
    }

    auto g_load = [&](int kt, ARawT (&xa)[A_CNT], typename P::BRaw (&xb)[Z::B_N], typename P::KPrm& xk) {
        if constexpr (C::AT) {
            xk = p.k_fetch(ctx, kt, aq);
            if constexpr (AU) {
#pragma unroll
                for (int i = 0; i < A_CNT; ++i) {            // consecutive lanes -> consecutive rows of one (piece, k8) plane
                    const int id = t + 256 * i;
                    const int r = id % C::BM, pk = id / C::BM;
                    xa[i] = p.a_unit(ctx, kt, pk / C::K8, pk % C::K8, r);
                }
            } else {
#pragma unroll
            for (int i = 0; i < Z::A_N; ++i)
                if (Z::A_FULL || al + i * Z::A_STEP < C::BM) xa[i] = p.a_fetch(ctx, arow[i], kt, aq);
            }
#pragma unroll
            for (int i = 0; i < Z::B_N; ++i) {           // weight units [piece][k8][row]: consecutive lanes -> consecutive rows
                const int id = t + 256 * i;
                const int r = id % C::BN, pk = id / C::BN;                 // pk = piece * K8 + k8
                if (Z::B_FULL || pk < NP * C::K8) xb[i] = p.b_unit(ctx, kt, pk / C::K8, pk % C::K8, r);
            }
        } else {
            if constexpr (AU) {
#pragma unroll
                for (int i = 0; i < A_CNT; ++i) {            // consecutive lanes -> consecutive pixels of one (piece, k8) plane
                    const int id = t + 256 * i;
                    const int kr = id % C::BK, q = id / C::BK;
                    xa[i] = p.a_unit_d(ctx, kt, q / (C::BM / 8), q % (C::BM / 8), kr);
                }
            } else {
#pragma unroll
            for (int i = 0; i < Z::A_N; ++i) {
                const int kr = al + i * Z::A_STEP;
                if (Z::A_FULL || kr < C::BK) xa[i] = p.a_fetch_d(ctx, da[i], kt, kr, aq);
                p.d_next(ctx, da[i]);
            }
            }
#pragma unroll
            for (int i = 0; i < Z::B_N; ++i) {
                const int kr = bl + i * Z::B_STEP;
                if (Z::B_FULL || kr < C::BK) xb[i] = p.b_fetch(ctx, db[i], kt, kr, bq);
                p.d_next(ctx, db[i]);
            }
        }
    };
    // transform (BN / ReLU / BN-backward) + split (or 16-bit pack) + LDS store of k-tile kt.  A staging slot is 16 bytes of
    // the operand's storage: one channel quad (fp32: ds_write_b64 per piece) or two (16-bit storage: the two quads go through
    // the policy's quad transform one after the other and leave as ONE 16-byte unit / row segment, ds_write_b128).
    auto s_store = [&](int buf, int kt, const ARawT (&xa)[A_CNT], const typename P::BRaw (&xb)[Z::B_N], const typename P::KPrm& xk) {
        char* A = As + buf * Z::A_BYTES;
        char* B = Bs + buf * Z::B_BYTES;
        if constexpr (C::AT) {
            if constexpr (AU) {
#pragma unroll
                for (int i = 0; i < A_CNT; ++i) {
                    const int id = t + 256 * i;
                    *reinterpret_cast<u32x4*>(A + ((id / C::BM) * C::LDUA + id % C::BM) * 16) = xa[i];
                }
            } else
            if constexpr (AE == 4) {
                const typename P::KFin kf = p.k_finish(ctx, xk, kt, aq, sp);
#pragma unroll
                for (int i = 0; i < Z::A_N; ++i) {
                    const int row = al + i * Z::A_STEP;
                    if (Z::A_FULL || row < C::BM) {
                        const Split4 s = split4<OP>(p.a_xform(ctx, xa[i], kf, kt, aq, sp));
#pragma unroll
                        for (int pc = 0; pc < NP; ++pc)
                            *reinterpret_cast<uint2*>(A + ((pc * C::K8 + (aq >> 1)) * C::LDUA + row) * 16 + (aq & 1) * 8) = s.p[pc];
                    }
                }
            } else {
                const typename P::KFin kf0 = p.k_finish(ctx, xk, kt, 2 * aq, sp), kf1 = p.k_finish(ctx, xk, kt, 2 * aq + 1, sp);
#pragma unroll
                for (int i = 0; i < Z::A_N; ++i) {
                    const int row = al + i * Z::A_STEP;
                    if (Z::A_FULL || row < C::BM) {
                        u32x4 u;
                        if constexpr (P::kARawCopy) u = as_u4(xa[i].v[0]);          // finished 16-bit operand: a straight copy
                        else u = pack_unit<OP>(p.a_xform(ctx, p.a_quad(xa[i], 0), kf0, kt, 2 * aq, sp),
                                               p.a_xform(ctx, p.a_quad(xa[i], 1), kf1, kt, 2 * aq + 1, sp));
                        *reinterpret_cast<u32x4*>(A + (aq * C::LDUA + row) * 16) = u;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < Z::B_N; ++i) {
                const int id = t + 256 * i;
                if (Z::B_FULL || id < NP * C::K8 * C::BN) *reinterpret_cast<u32x4*>(B + id * 16) = p.b_unit_xform(ctx, xb[i], id % C::BN);
            }
        } else {
            if constexpr (AU) {
#pragma unroll
                for (int i = 0; i < A_CNT; ++i) {
                    const int id = t + 256 * i;
                    *reinterpret_cast<u32x4*>(A + ((id / C::BK) * AU_PLANE + id % C::BK) * 16) = xa[i];
                }
            } else
#pragma unroll
            for (int i = 0; i < Z::A_N; ++i) {
                const int kr = al + i * Z::A_STEP;
                if (Z::A_FULL || kr < C::BK) {
                    if constexpr (AE == 4) {
                        const Split4 s = split4<OP>(p.a_xform(ctx, xa[i], KPrm0{}, kt, aq, sp));
#pragma unroll
                        for (int pc = 0; pc < NP; ++pc)
                            *reinterpret_cast<uint2*>(A + ((pc * C::BK + kr) * C::LDTA + 4 * aq) * 2) = s.p[pc];
                    } else {
                        u32x4 u;
                        if constexpr (P::kARawCopy) u = as_u4(xa[i].v[0]);
                        else u = pack_unit<OP>(p.a_xform(ctx, p.a_quad(xa[i], 0), KPrm0{}, kt, 2 * aq, sp),
                                               p.a_xform(ctx, p.a_quad(xa[i], 1), KPrm0{}, kt, 2 * aq + 1, sp));
                        *reinterpret_cast<u32x4*>(A + (kr * C::LDTA + 8 * aq) * 2) = u;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < Z::B_N; ++i) {
                const int kr = bl + i * Z::B_STEP;
                if (Z::B_FULL || kr < C::BK) {
                    if constexpr (BE == 4) {
                        const Split4 s = split4<OP>(p.b_xform(ctx, xb[i], kt, bq, sp, bfix[0]));
#pragma unroll
                        for (int pc = 0; pc < NP; ++pc)
                            *reinterpret_cast<uint2*>(B + ((pc * C::BK + kr) * C::LDTB + 4 * bq) * 2) = s.p[pc];
                    } else {
                        const u32x4 u = pack_unit<OP>(p.b_xform(ctx, p.b_quad(xb[i], 0), kt, 2 * bq, sp, bfix[0]),
                                                      p.b_xform(ctx, p.b_quad(xb[i], 1), kt, 2 * bq + 1, sp, bfix[BE / 4 - 1]));
                        *reinterpret_cast<u32x4*>(B + (kr * C::LDTB + 8 * bq) * 2) = u;
                    }
                }
            }
        }
    };
    // transposing-read geometry of this lane (weight gradient): its 16-lane group gathers 4 k-rows x 16 channels; lane i of
    // the group fetches row i/4, channel quad i%4 and receives channel i of all four rows (ds_read_b64_tr_b16).
    const int tr_row = (lane & 15) >> 2, tr_col = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    // One operand fragment piece of a 32-row tile for this wave's k16-step s: ds_read_b128 of a unit (forward / data
    // gradient) or two transposing reads (weight gradient).
    auto frag = [&](const char* img, int ldu, int ldt, int r0, int s, int pc) -> u32x4 {
        if constexpr (C::AT) {
            const int k8 = (wk * C::KS + s) * 2 + half;
            return *reinterpret_cast<const u32x4*>(img + ((pc * C::K8 + k8) * ldu + r0 + l31) * 16);
        } else {
            const int k0 = (wk * C::KS + s) * 16 + 8 * half + tr_row;
            const char* a0 = img + ((pc * C::BK + k0) * ldt + r0 + tr_col) * 2;
            const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0));
            const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * ldt * 2)));
            return u32x4{lo.x, lo.y, hi.x, hi.y};
        }
    };
    // the A fragment: as above, or (weight-gradient form, unit image) the same transposing reads with unit addressing - channel
    // quad cq of pixel k sits at ((piece * BM / 8 + cq / 2) * AU_PLANE + k) * 16 + (cq & 1) * 8
    auto fragA = [&](const char* img, int r0, int s, int pc) -> u32x4 {
        if constexpr (AU && !C::AT) {
            const int k0 = (wk * C::KS + s) * 16 + 8 * half + tr_row;
            const int ch = r0 + tr_col;
            const char* a0 = img + ((pc * (C::BM / 8) + (ch >> 3)) * AU_PLANE + k0) * 16 + ((ch >> 2) & 1) * 8;
            const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0));
            const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * 16)));
            return u32x4{lo.x, lo.y, hi.x, hi.y};
        } else return frag(img, C::LDUA, C::LDTA, r0, s, pc);
    };
    auto compute = [&](int buf) {
        const char* A = As + buf * Z::A_BYTES;
        const char* B = Bs + buf * Z::B_BYTES;
        // Per k16-step: the hi and lo pieces of every fragment first (one LDS round trip, not one per MFMA), the two
        // hi x lo product groups, then the mid pieces - fetched under those MFMAs into the registers lo occupied - and the
        // remaining four groups.  Within a group the tiles are innermost: consecutive MFMAs never share an accumulator.
#pragma unroll
        for (int s = 0; s < C::KS; ++s) {
            u32x4 ah[C::TM], bh[C::TN];
#pragma unroll
            for (int i = 0; i < C::TM; ++i) ah[i] = fragA(A, wm0 + i * 32, s, 0);
#pragma unroll
            for (int j = 0; j < C::TN; ++j) bh[j] = frag(B, C::LDUB, C::LDTB, wn0 + j * 32, s, 0);
            if constexpr (OP == 3) {         // two fp16 pieces: h*l, l*h, h*h (small terms first, tiles innermost)
                u32x4 al_[C::TM], bl_[C::TN];
#pragma unroll
                for (int i = 0; i < C::TM; ++i) al_[i] = fragA(A, wm0 + i * 32, s, 1);
#pragma unroll
                for (int j = 0; j < C::TN; ++j) bl_[j] = frag(B, C::LDUB, C::LDTB, wn0 + j * 32, s, 1);
                #pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int i = 0; i < C::TM; ++i)
#pragma unroll
                        for (int j = 0; j < C::TN; ++j)
                            acc[i][j] = mfma_f16(g == 1 ? al_[i] : ah[i], g == 0 ? bl_[j] : bh[j], acc[i][j]);
                            } else
            if constexpr (OP != 0) {         // single-piece operands: one term per tile
                #pragma unroll
                for (int i = 0; i < C::TM; ++i)
#pragma unroll
                    for (int j = 0; j < C::TN; ++j) acc[i][j] = mfma_1p<OP>(ah[i], bh[j], acc[i][j]);
                            } else {
                {
                    u32x4 al_[C::TM], bl_[C::TN];
#pragma unroll
                    for (int i = 0; i < C::TM; ++i) al_[i] = fragA(A, wm0 + i * 32, s, 2);
#pragma unroll
                    for (int j = 0; j < C::TN; ++j) bl_[j] = frag(B, C::LDUB, C::LDTB, wn0 + j * 32, s, 2);
#pragma unroll
                    for (int i = 0; i < C::TM; ++i)
#pragma unroll
                        for (int j = 0; j < C::TN; ++j) acc[i][j] = mfma_bf16(ah[i], bl_[j], acc[i][j]);
#pragma unroll
                    for (int i = 0; i < C::TM; ++i)
#pragma unroll
                        for (int j = 0; j < C::TN; ++j) acc[i][j] = mfma_bf16(al_[i], bh[j], acc[i][j]);
                                    }
                u32x4 am[C::TM], bm[C::TN];
#pragma unroll
                for (int i = 0; i < C::TM; ++i) am[i] = fragA(A, wm0 + i * 32, s, 1);
#pragma unroll
                for (int j = 0; j < C::TN; ++j) bm[j] = frag(B, C::LDUB, C::LDTB, wn0 + j * 32, s, 1);
                #pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < C::TM; ++i)
#pragma unroll
                        for (int j = 0; j < C::TN; ++j)
                            acc[i][j] = mfma_bf16((g & 1) ? ah[i] : am[i], g < 2 ? bm[j] : bh[j], acc[i][j]);   // mid*mid, hi*mid, mid*hi, hi*hi
                            }
        }
    };

    // First tile's global loads go out BEFORE the parameter prologue of the policies that still have one (it waits on its
    // own global loads): one memory round trip per workgroup instead of two.
    // LDS-DMA of k-tile kt into ring stage st: every wave copies its 64-unit runs of the (piece, k8) planes - the lane -> unit mapping
    // of the register path's copies, whose LDS targets are lane-consecutive already
    constexpr int DMA_PER = A_CNT + Z::B_N;                 // DMA instructions per wave and k-tile
    auto dma_tile = [&](int kt, int st) {
        if constexpr (DMA) {
            static_assert(C::AT && AU && C::BM % 64 == 0 && C::BN % 64 == 0 && Z::B_FULL && AU_UNITS % 256 == 0, "LDS-DMA: whole 64-unit runs per wave");
            const unsigned lds0 = (unsigned)(uintptr_t)As;      // (low half of the flat address = the LDS byte address)
            const unsigned a_dst = lds0 + (unsigned)st * Z::A_BYTES, b_dst = lds0 + NST * Z::A_BYTES + (unsigned)st * Z::B_BYTES;
#pragma unroll
            for (int i = 0; i < A_CNT; ++i) {
                const int id = t + 256 * i;
                const int r = id % C::BM, pk = id / C::BM;
                dma_unit(p.a_unit_src(ctx, kt, pk / C::K8, pk % C::K8, r), a_dst + 16u * (unsigned)(pk * C::LDUA + r - lane));
            }
#pragma unroll
            for (int i = 0; i < Z::B_N; ++i) {
                const int id = t + 256 * i;
                const int r = id % C::BN, pk = id / C::BN;
                dma_unit(p.b_unit_src(ctx, kt, pk / C::K8, pk % C::K8, r), b_dst + 16u * (unsigned)(id - lane));
            }
        }
    };
    if constexpr (DMA) {
#pragma unroll
        for (int u = 0; u < NFLY; ++u) dma_tile(u < KT ? u : KT - 1, u);      // (KT < NFLY: clamped re-loads keep the counts exact)
    } else {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
        if constexpr (PD > 1) g_load(u < KT ? u : KT - 1, ra[u], rb[u], kp[u]);   // same load count on every path: exact vmcnt
        else if (u < KT) g_load(u, ra[u], rb[u], kp[u]);
    }
    }
#ifdef SMG_TRACE_PRO     // dev: prologue stamps (start | first loads issued | parameters in LDS | early fetch issued | barrier passed)
    if (trace) trace[1] = smg_stamp();
#endif
    p.init_params(ctx, sp);
#ifdef SMG_TRACE_PRO
    if (trace) trace[2] = smg_stamp();
#endif
    // The epilogue's operands (mask source, old G') go out BEHIND the first k-tiles' loads and the parameter loads: vmcnt is
    // in-order, so issued first they would have to land before the first k-tile could be staged (measured: 15-17k of a
    // workgroup's 98k cycles); now the k-loop runs PD tiles before its waits reach them.
    // The explicit vmcnt(0) in front of them costs nothing (the parameter threads have just waited for younger loads) and tells
    // hipcc that the first tiles ARE in registers on every path: behind the divergent parameter branch and the conditional
    // early fetch it otherwise drains everything - the early fetch included - before the first tile's LDS store.
    if constexpr (P::kEarlyFetch) {
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0) only
        p.early_fetch(ctx);
    }
#ifdef SMG_TRACE_PRO
    if (trace) trace[3] = smg_stamp();
#endif
    if constexpr (P::kHasPrologue) __syncthreads();
#ifdef SMG_TRACE_PRO
    if (trace) trace[4] = smg_stamp();
#endif
    if constexpr (!C::AT) {
#pragma unroll
        for (int h = 0; h < BE / 4; ++h) bfix[h] = p.b_fix(ctx, (BE / 4) * bq + h, sp);
    }
    SMG_TRACE(1);
    if constexpr (DMA) {
        // k-tile kt: wait for this wave's pieces of it (in-order completion: at most the NFLY - 1 younger tiles stay out), barrier
        // (everybody's pieces have landed AND everybody has left k-tile kt - 1, whose stage is free), issue k-tile kt + NFLY into
        // that stage, compute.  One barrier per k-tile, nothing to store, NFLY tiles in flight.  The last NFLY tiles are peeled with
        // their own exact counts.  (The epilogue operands' ordinary loads sit between the first tiles' requests: a count meant for
        // DMA pieces only is then stricter than needed, never laxer.)
        // The epilogue operands' loads are waited for HERE, with the first tiles: hipcc cannot see the requests above and would
        // otherwise guard every later use of those registers (each segment hook) with a vmcnt(0) that drains the ring.
        if constexpr (P::kEarlyFetch) __builtin_amdgcn_s_waitcnt(0x0F70);
        SMG_TRACE(2);
        int kt = 0, st = 0;
        for (; kt + NFLY < KT; ++kt) {
            dma_wait<(NFLY - 1) * DMA_PER>();
            dma_barrier();
            dma_tile(kt + NFLY, st == 0 ? NST - 1 : st - 1);
            compute(st);
            if constexpr (P::kSegmented) p.k_hook(ctx, kt, acc, sp);
            st = st + 1 == NST ? 0 : st + 1;
        }
#pragma unroll
        for (int u = NFLY - 1; u >= 0; --u) {            // u tiles stay in flight behind this one
            if (kt < KT && KT - kt == u + 1) {
                if (u == NFLY - 1) dma_wait<(NFLY - 1) * DMA_PER>(); else if (u == 2) dma_wait<2 * DMA_PER>(); else if (u == 1) dma_wait<DMA_PER>(); else dma_wait<0>();
                dma_barrier();
                compute(st);
                if constexpr (P::kSegmented) p.k_hook(ctx, kt, acc, sp);
                st = st + 1 == NST ? 0 : st + 1;
                ++kt;
            }
        }
        dma_wait<0>();                                   // (KT < NFLY: the clamped re-loads) nothing may land in LDS behind this point
        dma_barrier();                                   // the epilogue reuses the stages
    } else {
    if (KT > 0) s_store(0, 0, ra[0], rb[0], kp[0]);
    __syncthreads();
    SMG_TRACE(2);
    }
    if constexpr (DMA) {
    } else
    if constexpr (PD == 1) {
        // two k-tiles per trip: the LDS buffer of every access is a compile-time constant (immediate offsets, no address VALU)
        auto step = [&](int kt, auto BUF, bool more) {
            constexpr int buf = decltype(BUF)::value;
#ifdef SMG_TRACE_ITER   // dev: sub-phase stamps of k-tile 4 instead of the whole-kernel phases (compute | load wait | store | barrier)
            const bool tr = trace && kt == 4;
            if (tr) trace[0] = smg_stamp();
#endif
            if (more) g_load(kt + 1, ra[0], rb[0], kp[0]);
            __builtin_amdgcn_sched_barrier(0);
            compute(buf);
            if constexpr (P::kSegmented) p.k_hook(ctx, kt, acc, sp);   // end of a K segment: fold acc away
#ifdef SMG_TRACE_ITER
            if (tr) trace[1] = smg_stamp();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tr) trace[2] = smg_stamp();
#endif
            if (more) s_store(buf ^ 1, kt + 1, ra[0], rb[0], kp[0]);
#ifdef SMG_TRACE_ITER
            if (tr) trace[3] = smg_stamp();
#endif
            __syncthreads();
#ifdef SMG_TRACE_ITER
            if (tr) trace[4] = smg_stamp();
#endif
        };
        int kt = 0;
        for (; kt + 2 <= KT; kt += 2) {
            step(kt, std::integral_constant<int, 0>{}, true);
            step(kt + 1, std::integral_constant<int, 1>{}, kt + 2 < KT);
        }
        if (kt < KT) step(kt, std::integral_constant<int, 0>{}, false);
    } else {
        // PD tiles in flight.  Whole groups of PD tiles run branch-free with unconditional (tail-clamped) loads: with the same
        // number of loads issued on every path hipcc counts vmcnt exactly instead of draining to 0 at the loop header; the last
        // KT % PD tiles are peeled (tile kt0 + u sits in ring slot u because kt0 is a multiple of PD).
        int kt0 = 0;
        for (; kt0 + PD <= KT; kt0 += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int kt = kt0 + u, buf = kt & 1;
#ifdef SMG_TRACE_ITER
                const bool tr = trace && kt == 4;
                if (tr) trace[0] = smg_stamp();
#endif
                g_load(kt + PD < KT ? kt + PD : KT - 1, ra[u], rb[u], kp[u]);   // slot u went to LDS one step ago
                __builtin_amdgcn_sched_barrier(0);
                compute(buf);
                if constexpr (P::kSegmented) p.k_hook(ctx, kt, acc, sp);
#ifdef SMG_TRACE_ITER
                if (tr) { trace[1] = smg_stamp(); trace[2] = trace[1]; }
#endif
                s_store(buf ^ 1, kt + 1 < KT ? kt + 1 : KT - 1, ra[(u + 1) % PD], rb[(u + 1) % PD], kp[(u + 1) % PD]);   // at kt + 1 == KT: a dead store of the clamped re-load
#ifdef SMG_TRACE_ITER
                if (tr) trace[3] = smg_stamp();
#endif
                __syncthreads();
#ifdef SMG_TRACE_ITER
                if (tr) trace[4] = smg_stamp();
#endif
            }
        }
#pragma unroll
        for (int u = 0; u + 1 < PD; ++u) {
            const int kt = kt0 + u, buf = kt & 1;
            if (kt < KT) {
                compute(buf);
                if constexpr (P::kSegmented) p.k_hook(ctx, kt, acc, sp);
                if (kt + 1 < KT) s_store(buf ^ 1, kt + 1, ra[u + 1], rb[u + 1], kp[u + 1]);
                __syncthreads();
            }
        }
    }
    if constexpr (C::WK > 1) {              // in-block split-K: fold the partial tiles into the wk == 0 waves
        constexpr int PER = C::TM * C::TN * 16 * 64;
        if (wk > 0) {
            float* r = smem + ((wk - 1) * C::WM * C::WN + wmn) * PER + lane;
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int j = 0; j < C::TN; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) r[((i * C::TN + j) * 16 + q) * 64] = acc[i][j][q];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int w = 1; w < C::WK; ++w) {
                const float* r = smem + ((w - 1) * C::WM * C::WN + wmn) * PER + lane;
#pragma unroll
                for (int i = 0; i < C::TM; ++i)
#pragma unroll
                    for (int j = 0; j < C::TN; ++j)
#pragma unroll
                        for (int q = 0; q < 16; ++q) acc[i][j][q] += r[((i * C::TN + j) * 16 + q) * 64];
            }
        }
        __syncthreads();
    }
    SMG_TRACE(3);
    p.epilogue(ctx, acc, smem, sp, wk == 0);
    SMG_TRACE(4);
    if (trace) trace[6] = __builtin_amdgcn_s_memrealtime();
}

// One workgroup per virtual block (x fastest).  A persistent variant (resident workgroups striding over the virtual
// grid) was measured and rejected: hipcc hoists the per-thread addressing out of the tile loop (+50..70 VGPRs, one
// workgroup less per CU) and the launch is not dispatch-bound.
template <class P>
static __global__ __launch_bounds__(256, P::kMinWaves) void gemm_kernel(const P p, const int vgx, const int vgy) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    VBlock vb;
    vb.linear = blockIdx.x;
    vb.x = vb.linear % vgx;
    const int q = vb.linear / vgx;
    vb.y = q % vgy;
    vb.z = q / vgy;
    gemm_tile<P>(p, vb, smem);
}

// Accumulator element (tm, tn, reg) of this lane sits at tile row / column:
//   row = wm0 + tm*32 + (reg&3) + 8*(reg>>2) + 4*half,  col = wn0 + tn*32 + l31
#define SMG_ACC_ROW(wm0, tm, reg, half) ((wm0) + (tm)*32 + ((reg)&3) + 8 * ((reg) >> 2) + 4 * (half))

// ------------------------------------------------------------------------------------
// Forward convolution policy.
//   F_ONE   1x1 conv over BN+ReLU(src)                      (bottleneck conv1, head conv0)
//   F_THREE 3x3 pad-1 conv over BN+ReLU(src), K-tiles walk (tap, channel)
//   F_POOL  1x1 conv over avgpool2x2(BN+ReLU(src))          (transition; pool commutes
//           with the pointwise conv, so it is applied first: 4x fewer MACs)
//   F_STEM  7x7 stride-2 pad-3 conv over the NHWC4 input image (no BN)
//   F_STEM1 the same over a ONE-channel image plane with the weights summed over the three input channels (K = 49 taps,
//           padded to 64): the reference feeds the stem three identical channels (code/trainer.py:178-181)
// Epilogue: store raw output + per-(stream, channel) sum / sum-of-squares (fp64 atomics).
// ------------------------------------------------------------------------------------
enum { F_ONE = 0, F_THREE = 1, F_POOL = 2, F_STEM = 3, F_STEM1 = 4 };

// PREC: the engine's precision mode (storage of src / dst, operand kind).  F32IO: src and dst are fp32 buffers whatever the
// mode (the head's feature buffers; the stem reads the fp32 image and writes the fp32 stem plane).
template <class Cfg_, int MODE, int PREC = 0, bool F32IO_ = false>
struct FwdConvP {
    using Cfg = Cfg_;
    static_assert(Cfg::AT, "forward form");
    static constexpr bool F32IO = F32IO_ || MODE == 3 || MODE == 4;      // F_STEM, F_STEM1
    using SrcT = typename std::conditional<F32IO, e_f32, act_t<PREC>>::type;
    using DstT = SrcT;
    static constexpr int kOp = (MODE == 3 || MODE == 4) ? fwd_op_plain(PREC) : fwd_op(PREC), kAE = 16 / SrcT::size, kBE = 4, ESZ = SrcT::size;
    static constexpr bool kARawCopy = false, kAUnit = false;
    static constexpr bool kH = kOp == 2 || kOp == 3;      // the BN + ReLU operand becomes fp16: clamped to fp16's range (bnrelu4)
    const float* asc;               // operand kind 3: {s, 1 / s} of the BN + ReLU operand (scale_kernel)
    const void* src; int lds_;
    Plane ps, po;
    int K;
    BnTab bt;                       // statistics + affine parameters of the BN in front of this conv
    // channels [fresh0, K) have no table entry yet (the 32 a dense layer appended last): derived here from the fp64 sums
    int fresh0; const double* fsum; const double* fsq; int fstride; float eps;
    float* tw_mean; float* tw_invstd;       // the table again, writable (designated workgroups store the fresh channels)
    const u32x4* wp; int K8tot;     // packed weight units [piece][K8tot][N] (pack_weights_kernel)
    int N;
    void* dst; int ldd; int dcoff;
    double* dsum; double* dsq; int dstride;
    TileMap tm;
    static constexpr int kSwizzle = 1;
    // k-tiles of global loads in flight: the one-MFMA-tile-per-wave configurations of the small planes do 0.1 us of MFMAs per
    // k-tile against ~1 us of memory latency
    static constexpr int kPrefetch = (Cfg::TM * Cfg::TN == 1) ? kPdFwdSmall : kPdFwdBig;
    static constexpr int kDmaFly = 0;
    static constexpr bool kSegmented = false;
    static constexpr bool kStem = MODE == F_STEM || MODE == F_STEM1;
    static constexpr bool kHasPrologue = !kStem;
    static constexpr bool kEarlyFetch = false;
    static constexpr int kFresh = 32;       // growth rate: at most this many fresh channels
    // waves per SIMD the register allocator is held to: the 128x128 tile (64 accumulator registers) must stay at 3
    static constexpr int kMinWaves = (Cfg::TM * Cfg::TN == 4 && MODE != F_POOL) ? kFwdBigMinWaves : 1;    // (the pooling fetch holds 4 float4 per row)

    struct Ctx { int n, m0, n0; };
    struct ARow { int y, x; bool valid; unsigned off; };     // off: element offset of the row inside its stream (F_ONE)
    struct DRow {};
    using KPrm = KPrm0;
    using KFin = typename std::conditional<kStem, KPrm0, KPrm3>::type;

    __host__ __device__ int param_floats() const { return kStem ? 0 : 3 * K; }       // mean | gamma*invstd | beta of all K channels

    __device__ bool init_ctx(Ctx& c, const VBlock& vb) const {
        int mt = vb.x, nt = vb.y;
        if (tm.nM && !tile_decode(tm, vb.x, mt, nt)) return false;
        c.m0 = sgpr(mt * Cfg::BM);          // (scalar registers for sure: the staged loads take them as scalar offsets)
        c.n0 = sgpr(nt * Cfg::BN);
        c.n = sgpr(c.m0 / po.HWp);
        return c.m0 - c.n * po.HWp < po.HW;
    }
    // BN parameters of this stream's K channels -> LDS, once per workgroup: (mean, gamma*invstd, beta) from the fp32 tables;
    // the fresh channels (the last kFresh) from the fp64 sums, and the first tile of every stream stores those in the table.
    // The staging threads then read their channel quad's parameters from LDS per k-tile (reading them from the tables per
    // k-tile was 16 KB of L1 traffic per workgroup and k-tile, twice the A operand).
    __device__ void init_params(const Ctx& c, float* sp) const {
        if constexpr (!kStem) {
            const int t = threadIdx.x;
            const float* tmean = tab_mean(bt, c.n);
            const float* tinv = tab_invstd(bt, c.n);
            // operand kind 3: relu(s * t) = s * relu(t) - the activation scale is folded into gamma * invstd and beta (no VALU in the k-loop)
            const float sa = kOp == 3 ? asc[0] : 1.f;
            for (int ch = 4 * t; ch < K && ch < fresh0; ch += 1024) {
                const f32x4 m = ldv4(tmean + ch), iv = ldv4(tinv + ch), g = ldv4(bt.gamma + ch), be = ldv4(bt.beta + ch);
                *reinterpret_cast<f32x4*>(sp + ch) = m;
                *reinterpret_cast<f32x4*>(sp + K + ch) = g * iv * sa;
                *reinterpret_cast<f32x4*>(sp + 2 * K + ch) = be * sa;
            }
            if (fresh0 < K && t < kFresh) {
                const int ch = fresh0 + t;
                float mean, invstd;
                bn_moments(fsum, fsq, (int64_t)c.n * fstride + ch, 1.0 / (double)ps.HW, eps, mean, invstd);
                sp[ch] = mean;
                sp[K + ch] = bt.gamma[ch] * invstd * sa;
                sp[2 * K + ch] = bt.beta[ch] * sa;
                if (c.n0 == 0 && c.m0 == c.n * po.HWp) {
                    tw_mean[(int64_t)c.n * bt.ld + ch] = mean;
                    tw_invstd[(int64_t)c.n * bt.ld + ch] = invstd;
                }
            }
        }
    }
    __device__ void d_init(const Ctx&, DRow&, int) const {}
    __device__ void d_next(const Ctx&, DRow&) const {}
    __device__ int ktiles(const Ctx&) const {
        if constexpr (MODE == F_THREE) return 9 * (K / Cfg::BK);
        else if constexpr (MODE == F_STEM) return 224 / Cfg::BK;
        else if constexpr (MODE == F_STEM1) return 64 / Cfg::BK;
        else return K / Cfg::BK;
    }
    __device__ void a_row_init(const Ctx& c, ARow& r, int line) const {
        const int p = c.m0 + line - c.n * po.HWp;
        r.valid = p < po.HW;
        r.y = p / po.W;
        r.x = p - r.y * po.W;
        r.off = (unsigned)ESZ * (unsigned)(p * lds_);    // byte offset of the row inside its stream
    }
    struct RawTaps { float4 v[1]; bool ok; unsigned m; };      // F_STEM1: four gathered taps + which of them exist
    using ARaw = typename std::conditional<MODE == F_STEM1, RawTaps, RawT<(MODE == F_POOL) ? 4 : 1>>::type;
    using BRaw = u32x4;
    // first channel of k-tile kt (workgroup-uniform: the per-thread part of every address below is loop-invariant, so the
    // loads are scalar base + 32-bit lane offset with no address arithmetic per k-tile)
    __device__ int a_chan0(int kt) const {
        if constexpr (MODE == F_THREE) { const int kpt = K / Cfg::BK; return (kt % kpt) * Cfg::BK; }
        else return kt * Cfg::BK;
    }
    __device__ int a_chan(int kt, int q) const { return a_chan0(kt) + 4 * q; }            // first channel of quad q of the k-tile
    __device__ int a_slot_chan(int kt, int q) const { return a_chan0(kt) + kAE * q; }     // ... of 16-byte staging slot q
    // quad h of a fetched slot as a slot of fp32 values (16-bit storage: the staging code transforms the two quads of a slot
    // one after the other)
    __device__ ARaw a_quad(const ARaw& o, int h) const {
        ARaw r = o;
#pragma unroll
        for (int j = 0; j < (MODE == F_POOL ? 4 : 1); ++j) r.v[j] = slot_quad<SrcT>(o.v[j], h);
        return r;
    }
    __device__ KPrm k_fetch(const Ctx&, int, int) const { return KPrm{}; }
    // BN parameters of this thread's channel quad for k-tile kt, from the workgroup's LDS copy
    __device__ KFin k_finish(const Ctx&, const KPrm&, int kt, int q, const float* sp) const {
        KFin f;
        if constexpr (!kStem) {
            const int ch = a_chan(kt, q);
            f.mean = ldv4(sp + ch);
            f.scale = ldv4(sp + K + ch);
            f.beta = ldv4(sp + 2 * K + ch);
        }
        return f;
    }
    // Unconditional loads from clamped addresses (rows of the plane padding / taps outside the image read pixel 0 and are
    // zeroed at the LDS store): a branch around a staged load makes hipcc drain vmcnt(0) before the next load group.
    __device__ ARaw a_fetch(const Ctx& c, const ARow& r, int kt, int q) const {
        ARaw o;
        if constexpr (MODE == F_ONE) {
            // rows of the plane padding are read as they are (they exist) and produce output rows nobody stores or sums
            o.ok = true;
            o.v[0] = bload4(static_cast<const char*>(src) + (int64_t)ESZ * c.n * ps.HWp * lds_, kWholeBuf, r.off + 16u * (unsigned)q, (unsigned)ESZ * (unsigned)(kt * Cfg::BK));
        } else if constexpr (MODE == F_THREE) {
            const int tap = kt / (K / Cfg::BK);
            const int yy = r.y + tap / 3 - 1, xx = r.x + tap % 3 - 1;
            o.ok = r.valid && (unsigned)yy < (unsigned)ps.H && (unsigned)xx < (unsigned)ps.W;
            o.v[0] = ld16(src, (int64_t)ESZ * (((int64_t)c.n * ps.HWp + (o.ok ? yy * ps.W + xx : 0)) * lds_ + a_slot_chan(kt, q)));
        } else if constexpr (MODE == F_POOL) {
            o.ok = r.valid;
            const int64_t b = (int64_t)ESZ * (((int64_t)c.n * ps.HWp + (o.ok ? (2 * r.y) * ps.W + 2 * r.x : 0)) * lds_ + a_slot_chan(kt, q));
            o.v[0] = ld16(src, b);
            o.v[1] = ld16(src, b + (int64_t)ESZ * lds_);
            o.v[2] = ld16(src, b + (int64_t)ESZ * ps.W * lds_);
            o.v[3] = ld16(src, b + (int64_t)ESZ * (ps.W + 1) * lds_);
        } else if constexpr (MODE == F_STEM1) {
            // slot q of the k-tile = taps 4 q' .. 4 q' + 3 of the 7x7 window: four unconditional scalar loads from clamped
            // addresses of the one-channel plane; which taps exist (inside the window and the image) travels as a bit mask
            const float* img = static_cast<const float*>(src) + (int64_t)c.n * ps.HWp;
            float tv[4];
            o.ok = true; o.m = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tap = kt * Cfg::BK + 4 * q + j;
                const int yy = 2 * r.y + tap / 7 - 3, xx = 2 * r.x + tap % 7 - 3;
                const bool in = r.valid && tap < 49 && (unsigned)yy < (unsigned)ps.H && (unsigned)xx < (unsigned)ps.W;
                tv[j] = img[in ? yy * ps.W + xx : 0];
                o.m |= in ? (1u << j) : 0u;
            }
            o.v[0] = make_float4(tv[0], tv[1], tv[2], tv[3]);
        } else {
            const int tap = kt * (Cfg::BK / 4) + q;
            const int yy = 2 * r.y + tap / 7 - 3, xx = 2 * r.x + tap % 7 - 3;
            o.ok = r.valid && tap < 49 && (unsigned)yy < (unsigned)ps.H && (unsigned)xx < (unsigned)ps.W;
            o.v[0] = ld16(src, 16 * ((int64_t)c.n * ps.HWp + (o.ok ? yy * ps.W + xx : 0)));
        }
        return o;
    }
    __device__ float4 a_xform(const Ctx&, const ARaw& o, const KFin& k, int, int, const float*) const {
        if constexpr (MODE == F_STEM1) {
            return make_float4((o.m & 1u) ? o.v[0].x : 0.f, (o.m & 2u) ? o.v[0].y : 0.f, (o.m & 4u) ? o.v[0].z : 0.f, (o.m & 8u) ? o.v[0].w : 0.f);
        } else if constexpr (MODE == F_STEM) {
            return o.ok ? o.v[0] : zero4();
        } else {
            if constexpr (MODE == F_ONE) return bnrelu4<kH>(o.v[0], k);
            if (!o.ok) return zero4();                   // zero padding applies AFTER bn + relu
            if constexpr (MODE == F_POOL) {
                float4 s = bnrelu4<kH>(o.v[0], k);
                s = add4(s, bnrelu4<kH>(o.v[1], k));
                s = add4(s, bnrelu4<kH>(o.v[2], k));
                s = add4(s, bnrelu4<kH>(o.v[3], k));
                return make_float4(s.x * 0.25f, s.y * 0.25f, s.z * 0.25f, s.w * 0.25f);
            } else {
                return bnrelu4<kH>(o.v[0], k);
            }
        }
    }
    __device__ ARaw a_fetch_d(const Ctx&, const DRow&, int, int, int) const { return ARaw{}; }
    // weight unit (piece, k8 of this k-tile, tile row r): rows past N read a neighbouring unit (the packed array has
    // slack behind it); the columns they feed are never stored
    __device__ BRaw b_unit(const Ctx& c, int kt, int piece, int k8, int r) const {
        return bload_u4(wp, kWholeBuf, 16u * (unsigned)((piece * K8tot + k8) * N + r), 16u * (unsigned)(kt * Cfg::K8 * N + c.n0));
    }
    __device__ u32x4 b_unit_xform(const Ctx&, const BRaw& o, int) const { return o; }
    __device__ void epilogue(const Ctx& c, f32x16 (&acc)[Cfg::TM][Cfg::TN], float* smem, float*, bool active) const {
        const int t = threadIdx.x, lane = t & 63, wmn = (t >> 6) % (Cfg::WM * Cfg::WN), l31 = lane & 31, half = lane >> 5;
        const int wm0 = (wmn / Cfg::WN) * Cfg::TM * 32, wn0 = (wmn % Cfg::WN) * Cfg::TN * 32;
        // fp64 in-lane accumulation: E[x^2] - mean^2 must survive near-constant channels
        // (the masked stream is mostly background).
        double v[2][Cfg::TN];
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) v[0][j] = v[1][j] = 0.0;
        if constexpr (kOp == 3) {          // the products were formed on scaled operands: exact power-of-two correction
            const float inv = asc[1] * pack_inv_scale(wp);
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        acc[i][j][r] *= inv;
                    }
        }
        const int pbase = c.m0 - c.n * po.HWp;
        if (pbase + Cfg::BM <= po.HW && c.n0 + Cfg::BN <= N) {
            // Whole tile inside the plane (every tile of the 160^2 / 80^2 / 40^2 stages): straight-line stores, and the
            // statistics as fp32 sums of (x - s), (x - s)^2 about the lane's first value s of each 16-row strip,
            // widened to fp64 once per strip: sum x = S1 + 16 s, sum x^2 = S2 + 2 s S1 + 16 s^2.  (Shifted sums
            // keep the precision the per-element fp64 path has - no cancellation on near-constant channels.)
            if (active) {
#pragma unroll
                for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                    for (int j = 0; j < Cfg::TN; ++j) {
                        float s1 = 0.f, s2 = 0.f;
                        const float s = acc[i][j][0];
                        if constexpr (DstT::size == 2) {
                            // 16-bit storage: rows 4g .. 4g + 3 of the strip leave as one quad-transposed 8-byte store per lane
                            // (a 2-byte store per element issued four times the memory instructions)
                            const int64_t at = (int64_t)(c.m0 + wm0 + i * 32) * ldd + dcoff + c.n0 + wn0 + j * 32;
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
#pragma unroll
                                for (int r = 4 * g; r < 4 * g + 4; ++r) {
                                    const float dx = acc[i][j][r] - s;
                                    s1 += dx;
                                    s2 = fmaf(dx, dx, s2);
                                }
                                store_acc_rows<DstT>(dst, at, ldd, g, half, lane, acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                            }
                        } else {
                            // uniform tile base + one running 32-bit lane offset (saddr stores): 16 precomputed 64-bit
                            // addresses per strip would cost a workgroup of occupancy
                            char* tb = static_cast<char*>(dst) + (int64_t)DstT::size * ((int64_t)c.m0 * ldd + dcoff + c.n0);
                            unsigned o = (unsigned)((wm0 + i * 32 + 4 * half) * ldd + wn0 + j * 32 + l31);
#pragma unroll
                            for (int r = 0; r < 16; ++r) {           // accumulator rows (r & 3) + 8 * (r >> 2)
                                const float x = acc[i][j][r];
                                st1<DstT>(tb, o, x);
                                o += (r & 3) == 3 ? 5u * (unsigned)ldd : (unsigned)ldd;
                                float dx = x - s;
                                // (opaque to the SLP vectoriser: with -fslp-vectorize hipcc pairs the statistics of DIFFERENT strips into
                                //  v_pk_add_f32 / v_pk_fma_f32 chains - every packed form legal as written - and the fp64 sums of a launch
                                //  then vary from run to run; this barrier, one behind `acc *= inv`, or -fno-slp-vectorize each restore
                                //  bit-reproducible statistics, wait states do not.  The Makefile keeps the flag, this keeps the site safe
                                //  without it, test_forward_is_bit_reproducible_run_to_run watches every other one.)
                                asm volatile("" : "+v"(dx));
                                s1 += dx;
                                s2 = fmaf(dx, dx, s2);
                            }
                        }
                        const double sd = (double)s, s1d = (double)s1;
                        v[0][j] += s1d + 16.0 * sd;
                        v[1][j] += (double)s2 + 2.0 * sd * s1d + 16.0 * sd * sd;
                        __builtin_amdgcn_sched_barrier(0);     // one 32x32 strip at a time: keeps the address math of the
                                                               // other strips out of the live set (occupancy 3, not 2)
                    }
            }
        } else
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) {
                const int col = c.n0 + wn0 + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = SMG_ACC_ROW(wm0, i, r, half);
                    if (active && pbase + row < po.HW && col < N) {
                        const float x = acc[i][j][r];
                        st1<DstT>(dst, (int64_t)(c.m0 + row) * ldd + dcoff + col, x);
                        const double xd = (double)x;
                        v[0][j] += xd;
                        v[1][j] += xd * xd;
                    }
                }
            }
        double tot[2];
        block_col_reduce<Cfg, 2, double>(v, reinterpret_cast<double*>(smem), tot);
        if (t < Cfg::BN && c.n0 + t < N) {
            const int64_t si = (int64_t)c.n * dstride + dcoff + c.n0 + t;
            atomicAdd(dsum + si + fstat_rep(), tot[0]);
            atomicAdd(dsq + si + fstat_rep(), tot[1]);
        }
    }
};

// ------------------------------------------------------------------------------------
// Data-gradient policy (pixel-major A, transposed into LDS).
//   A(m, r)  = g*a[r] + x*b[r] + c[r]     the BN-backward-corrected upstream gradient:
//              g from the gradient buffer, x from the raw activation the BN normalised.
//              With SHIFT3 the reduction walks (tap, channel) and the pixel is shifted
//              (transposed 3x3 convolution).
//   B(r, j)  = weights, row-major [K][N].
//   Epilogue = ReLU + BN backward of the BN that FEEDS this convolution's input
//              (mask source `mbuf`), in one of three forms:
//     E_STORE  dst = dy; per-stream sums of dy and dy*xhat           (-> next fetch)
//     E_ACCUM  dst += gamma*dy; per-stream sums weighted by gamma    (dense-block G')
//     E_UNPOOL as E_ACCUM with plain stores to the 4 pixels each pooled pixel covers
//   and always dbeta += sum dy, dgamma += sum dy*xhat.
// ------------------------------------------------------------------------------------
enum { E_STORE = 0, E_ACCUM = 1, E_UNPOOL = 2 };

// PREC: the engine's precision mode; F32IO: every buffer is fp32 whatever the mode (the head).
template <class Cfg_, bool SHIFT3, int EMODE, bool AFF = true, int PREC = 0, bool F32IO = false>
struct BwdDataP {
    using Cfg = Cfg_;
    static_assert(Cfg::AT, "data-gradient form");
    using GT = typename std::conditional<F32IO, e_f32, grd_t<PREC>>::type;      // gradients in (gbuf) and out (dst)
    using XT = typename std::conditional<F32IO, e_f32, act_t<PREC>>::type;      // activations (xbuf, mbuf)
    static_assert(GT::size == XT::size, "one slot geometry for the gradient and its activation");
    // operand kind 3 needs the gradient operand's recorded maximum: the pointwise form on a FINISHED gradient only
    static constexpr int kOp = (AFF || SHIFT3) ? bwd_op_plain(PREC) : bwd_op(PREC), kAE = 16 / GT::size, kBE = 4, GSZ = GT::size, XSZ = XT::size;
    static constexpr bool kARawCopy = !AFF && !SHIFT3 && kAE == 8;              // finished bf16 gradient: copied to LDS as it is
    static constexpr bool kAUnit = !AFF && !SHIFT3 && kOp == 3;                 // finished gradient in unit form (kD2K8): gbuf = units, ldg unused
    const float* binv;              // operand kind 3: [streams][HWp / 64] inverse block scales of gbuf (bn_bwd_apply_split_kernel)
    const void* gbuf; int ldg; int gcoff;
    const void* xbuf; int ldx; int xcoff;
    Plane pa;
    int KA;
    const double* xsum; const double* xsq; int xstride;
    const double* s1; const double* s2; int sstride; int scoff;
    const float* agamma;
    const u32x4* wp; int K8tot; int ldn; int wcol0;    // packed weight units [piece][K8tot][ldn], first output column wcol0
    int N;
    const void* mbuf; int ldm; int mcoff; Plane pm;
    const double* msum; const double* msq; int mstride; StatTab mtab;
    const float* egamma; const float* ebeta;
    void* dst; int ldd; int dcoff;
    double* o1; double* o2; int ostride; int ocoff;
    float* dbeta; float* dgamma; int rep_stride;      // rep_stride != 0: kDbRep replicas of a scratch (engine.h), this workgroup's = index mod 8
    float eps;
    TileMap tm;
    static constexpr int kSwizzle = 1;
    static constexpr int kPrefetch = (Cfg::TM * Cfg::TN <= 2) ? kPdDgrad : kPdDgradBig;
    static constexpr int kDmaFly = kAUnit ? kDmaFlyDgrad : 0;       // unit-form gradient + packed weights: both operands by LDS-DMA (gemm_tile)
    static constexpr bool kSegmented = false;
    static constexpr bool kHasPrologue = true;
    static constexpr bool kEarlyFetch = EMODE != E_UNPOOL;     // early_fetch(): the epilogue's operands, issued behind the first k-tiles' loads
    // the transitions' un-pooling data gradient held to 3 waves per SIMD: 172 -> 152 registers without scratch, 169 -> 163 us per launch
    static constexpr int kMinWaves = (PREC == 0 && EMODE == E_UNPOOL) ? 3 : 1;       // (3 waves per SIMD for the 128 x 64 accumulate form: 168 VGPRs + 88 bytes of scratch, serialised total 19.0 -> 19.4 ms)

    // AFF = false: the gradient operand is finished (xbuf unused).  Pointwise and finished -> descriptor loads, rows outside
    // the plane carry the out-of-range offset and read as zero (no mask, no address arithmetic in the k-loop).
    static constexpr bool kFast = !SHIFT3 && !AFF;
    // Whole tiles inside the plane fetch the epilogue's operands (the mask source x and, for E_ACCUM, the old G') at the START
    // of the workgroup, into registers: the K loop of a 1x1 data gradient is 8 k-tiles, and an epilogue that only then starts
    // its loads costs more cycles than that loop (measured: 10k of 24k per workgroup).
    static constexpr bool kEarly = EMODE != E_UNPOOL;
    // 16-bit storage: the epilogue's loads / stores go through quad transposes (8 bytes per lane instead of 2); fp32 storage
    // keeps one element per lane and instruction (the transposes cost the fp32-class kernels registers they do not have)
    static constexpr bool kWide = GT::size == 2;
    struct Ctx {
        int n, m0, n0; bool whole;
        float ginv[2];              // operand kind 3: inverse of (gradient x weight) scale of the tile's (up to two) 64-row scale blocks
        // mask source x and old G' of a whole tile, fetched at the start of the workgroup: per element in accumulator layout
        // (fp32 storage), or as loaded row segments of four accumulator rows each (16-bit storage; fetch_acc_rows)
        float xv[(kEarly && !kWide) ? Cfg::TM : 1][(kEarly && !kWide) ? Cfg::TN : 1][16];
        float gold[(kEarly && !kWide && EMODE == E_ACCUM) ? Cfg::TM : 1][(kEarly && !kWide && EMODE == E_ACCUM) ? Cfg::TN : 1][16];
        rawq_t<XT> xq[(kEarly && kWide) ? Cfg::TM : 1][(kEarly && kWide) ? Cfg::TN : 1][4];
        rawq_t<GT> gq[(kEarly && kWide && EMODE == E_ACCUM) ? Cfg::TM : 1][(kEarly && kWide && EMODE == E_ACCUM) ? Cfg::TN : 1][4];
    };
    struct ARow { int y, x; bool valid; unsigned off; };
    struct DRow {};
    using KPrm = KPrm0;
    using KFin = KPrm0;
    __device__ KPrm k_fetch(const Ctx&, int, int) const { return KPrm{}; }
    __device__ KFin k_finish(const Ctx&, const KPrm&, int, int, const float*) const { return KFin{}; }

    __host__ __device__ int param_floats() const { return 4 * KA + 5 * Cfg::BN; }

    __device__ void d_init(const Ctx&, DRow&, int) const {}
    __device__ void d_next(const Ctx&, DRow&) const {}
    __device__ bool init_ctx(Ctx& c, const VBlock& vb) const {
        int mt = vb.x, nt = vb.y;
        if (tm.nM && !tile_decode(tm, vb.x, mt, nt)) return false;
        c.m0 = sgpr(mt * Cfg::BM);          // (scalar registers for sure: the staged loads take them as scalar offsets)
        c.n0 = sgpr(nt * Cfg::BN);
        c.n = sgpr(c.m0 / pa.HWp);
        const int pbase = c.m0 - c.n * pa.HWp;
        if (pbase >= pa.HW) return false;
        // whole = every ROW of the tile inside the plane; a last column tile may be partial as long as it ends on a 32-column
        // boundary: an accumulator tile (wave, j) is then entirely inside or entirely outside [0, N) - a wave-uniform test (jok),
        // no per-element predicate.  (The narrow per-layer launches of a layer group have N = 32 / 64 / 96.)
        c.whole = kEarly && pbase + Cfg::BM <= pa.HW && (c.n0 + Cfg::BN <= N || (N & 31) == 0);
        c.ginv[0] = c.ginv[1] = 1.f;
        if constexpr (kOp == 3) {
            static_assert(Cfg::BM == kScaleBlock || (Cfg::BM == 2 * kScaleBlock && Cfg::WM == 2), "a wave's rows lie in one scale block");
            const float* bi = binv + (int64_t)c.n * (pa.HWp / kScaleBlock) + pbase / kScaleBlock;      // (workgroup-uniform: scalar loads)
            const float wi = pack_inv_scale(wp);
            c.ginv[0] = bi[0] * wi;
            c.ginv[1] = (Cfg::BM > kScaleBlock ? bi[1] : bi[0]) * wi;
        }
        return true;
    }
    // unit (piece, k8 of this k-tile, tile row) of the finished gradient: plane-major units, consecutive rows consecutive
    __device__ UnitSrc a_unit_src(const Ctx& c, int kt, int piece, int k8, int row) const {
        return UnitSrc{static_cast<const u32x4*>(gbuf) + d2_stream_units(c.n, pa.HWp), kWholeBuf,
                       16u * (unsigned)((piece * kD2K8 + k8) * pa.HWp + row), 16u * (unsigned)(kt * Cfg::K8 * pa.HWp + (c.m0 - c.n * pa.HWp))};
    }
    __device__ u32x4 a_unit(const Ctx& c, int kt, int piece, int k8, int row) const {
        const UnitSrc u = a_unit_src(c, kt, piece, k8, row);
        return bload_u4(u.base, u.bytes, u.voff, u.soff);
    }
    __device__ u32x4 a_unit_d(const Ctx&, int, int, int, int) const { return u32x4{}; }
    __device__ void early_fetch(Ctx& c) const {
        if constexpr (kEarly) {
            if (c.whole) {
                const int t = threadIdx.x, lane = t & 63, wmn = (t >> 6) % (Cfg::WM * Cfg::WN), l31 = lane & 31, half = lane >> 5;
                const int wm0 = (wmn / Cfg::WN) * Cfg::TM * 32, wn0 = (wmn % Cfg::WN) * Cfg::TN * 32;
                if constexpr (kWide) {
    #pragma unroll
                    for (int i = 0; i < Cfg::TM; ++i)
    #pragma unroll
                        for (int j = 0; j < Cfg::TN; ++j) {
                            if (c.n0 + wn0 + j * 32 + 32 > N) continue;      // (wave-uniform) tile outside [0, N): never read
                            // 16 / 8-byte loads, one per four accumulator rows (fetch_acc_rows, above GemmCfg); transposed in the epilogue
                            const int64_t ax = (int64_t)(c.m0 + wm0 + i * 32) * ldm + mcoff + c.n0 + wn0 + j * 32;
                            const int64_t ag = (int64_t)(c.m0 + wm0 + i * 32) * ldd + dcoff + c.n0 + wn0 + j * 32;
    #pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                c.xq[i][j][g] = fetch_acc_rows<XT>(mbuf, ax, ldm, g, half, lane);
                                if constexpr (EMODE == E_ACCUM) c.gq[i][j][g] = fetch_acc_rows<GT>(dst, ag, ldd, g, half, lane);
                            }
                        }
                } else {
                    const char* xb = static_cast<const char*>(mbuf) + (int64_t)XSZ * ((int64_t)c.m0 * ldm + mcoff + c.n0);
                    const char* gb = static_cast<const char*>(dst) + (int64_t)GSZ * ((int64_t)c.m0 * ldd + dcoff + c.n0);
    #pragma unroll
                    for (int i = 0; i < Cfg::TM; ++i)
    #pragma unroll
                        for (int j = 0; j < Cfg::TN; ++j) {
                            if (c.n0 + wn0 + j * 32 + 32 > N) continue;      // (wave-uniform) tile outside [0, N): never read
                            const int cj = wn0 + j * 32 + l31;
                            unsigned ox = (unsigned)((wm0 + i * 32 + 4 * half) * ldm + cj);
                            unsigned og = (unsigned)((wm0 + i * 32 + 4 * half) * ldd + cj);
    #pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                c.xv[i][j][r] = ld1<XT>(xb, ox);
                                if constexpr (EMODE == E_ACCUM) c.gold[i][j][r] = ld1<GT>(gb, og);
                                ox += (r & 3) == 3 ? 5u * (unsigned)ldm : (unsigned)ldm;
                                og += (r & 3) == 3 ? 5u * (unsigned)ldd : (unsigned)ldd;
                            }
                        }
                }
            }
        }
    }
    __device__ void init_params(const Ctx& c, float* sp) const {
        const double inv = 1.0 / (double)pa.HW;
        for (int k = threadIdx.x; xbuf && k < KA; k += 256) {
            float mean, invstd;
            bn_moments(xsum, xsq, (int64_t)c.n * xstride + xcoff + k, inv, eps, mean, invstd);
            const float g = agamma ? agamma[k] : 1.f;
            const float q1 = (float)(stat_get(s1, (int64_t)c.n * sstride + scoff + k) * inv);
            const float q2 = (float)(stat_get(s2, (int64_t)c.n * sstride + scoff + k) * inv);
            sp[k] = g * invstd;
            sp[KA + k] = q1;
            sp[2 * KA + k] = mean;
            sp[3 * KA + k] = invstd * q2;
        }
        float* ep = sp + 4 * KA;
        const double minv = 1.0 / (double)pm.HW;
        for (int j = threadIdx.x; j < Cfg::BN; j += 256) {
            const int col = c.n0 + j;
            float mean = 0.f, invstd = 0.f, g = 0.f, b = 0.f;
            if (col < N) {
                tab_or_moments(mtab, c.n, mcoff + col, msum, msq, (int64_t)c.n * mstride + mcoff + col, minv, eps, mean, invstd);
                g = egamma[col];
                b = ebeta[col];
            }
            ep[j] = g * invstd;
            ep[Cfg::BN + j] = b;
            ep[2 * Cfg::BN + j] = mean;
            ep[3 * Cfg::BN + j] = invstd;
            ep[4 * Cfg::BN + j] = g;
        }
    }
    __device__ int ktiles(const Ctx&) const { return (SHIFT3 ? 9 : 1) * (KA / Cfg::BK); }
    __device__ void a_row_init(const Ctx& c, ARow& r, int line) const {
        const int p = c.m0 + line - c.n * pa.HWp;
        r.valid = p < pa.HW;
        r.y = p / pa.W;
        r.x = p - r.y * pa.W;
        r.off = r.valid ? (unsigned)GSZ * (unsigned)(p * ldg) : kOOB;
    }
    using ARaw = RawT<2>;       // gradient + (when xbuf is set) the raw activation its BN normalised
    using BRaw = u32x4;
    __device__ int a_chan0(int kt) const {
        if constexpr (SHIFT3) { const int kpt = KA / Cfg::BK; return (kt % kpt) * Cfg::BK; }
        else return kt * Cfg::BK;
    }
    __device__ int a_chan(int kt, int q) const { return a_chan0(kt) + 4 * q; }             // channel quad q of the k-tile
    __device__ ARaw a_quad(const ARaw& o, int h) const {
        ARaw r; r.ok = o.ok;
        r.v[0] = slot_quad<GT>(o.v[0], h);
        r.v[1] = slot_quad<XT>(o.v[1], h);
        return r;
    }
    __device__ ARaw a_fetch(const Ctx& c, const ARow& r, int kt, int q) const {
        ARaw o;
        if constexpr (kFast) {
            o.ok = true;
            o.v[0] = bload4(static_cast<const char*>(gbuf) + (int64_t)GSZ * ((int64_t)c.n * pa.HWp * ldg + gcoff), kWholeBuf, r.off + 16u * (unsigned)q,
                            (unsigned)GSZ * (unsigned)(kt * Cfg::BK));
            return o;
        }
        int yy = r.y, xx = r.x;
        o.ok = r.valid;
        if constexpr (SHIFT3) {
            const int tap = kt / (KA / Cfg::BK);
            yy = r.y + 1 - tap / 3;
            xx = r.x + 1 - tap % 3;
            o.ok = r.valid && (unsigned)yy < (unsigned)pa.H && (unsigned)xx < (unsigned)pa.W;
        }
        const int ch = a_chan0(kt) + kAE * q;                                         // first channel of staging slot q
        const int64_t pix = (int64_t)c.n * pa.HWp + (o.ok ? yy * pa.W + xx : 0);      // unconditional loads, clamped address
        o.v[0] = ld16(gbuf, (int64_t)GSZ * (pix * ldg + gcoff + ch));
        if (xbuf) o.v[1] = ld16(xbuf, (int64_t)XSZ * (pix * ldx + xcoff + ch));      // (launch-uniform)
        return o;
    }
    __device__ float4 a_xform(const Ctx& c, const ARaw& o, const KPrm&, int kt, int q, const float* sp) const {
        if constexpr (kFast) return o.v[0];            // (operand kind 3 takes the unit path: a_unit)
        if (!o.ok) return zero4();
        if (!xbuf) return o.v[0];                      // gradient already BN-corrected (bn_bwd_apply_kernel)
        return affine2(o.v[0], o.v[1], sp + a_chan(kt, q), KA);
    }
    __device__ ARaw a_fetch_d(const Ctx&, const DRow&, int, int, int) const { return ARaw{}; }
    __device__ UnitSrc b_unit_src(const Ctx& c, int kt, int piece, int k8, int r) const {
        return UnitSrc{wp, kWholeBuf, 16u * (unsigned)((piece * K8tot + k8) * ldn + r), 16u * (unsigned)(kt * Cfg::K8 * ldn + wcol0 + c.n0)};
    }
    __device__ BRaw b_unit(const Ctx& c, int kt, int piece, int k8, int r) const {
        const UnitSrc u = b_unit_src(c, kt, piece, k8, r);
        return bload_u4(u.base, u.bytes, u.voff, u.soff);
    }
    __device__ u32x4 b_unit_xform(const Ctx&, const BRaw& o, int) const { return o; }
    __device__ void epilogue(const Ctx& c, f32x16 (&acc)[Cfg::TM][Cfg::TN], float* smem, float* sp, bool active) const {
        const int t = threadIdx.x, lane = t & 63, wmn = (t >> 6) % (Cfg::WM * Cfg::WN), l31 = lane & 31, half = lane >> 5;
        const int wm0 = (wmn / Cfg::WN) * Cfg::TM * 32, wn0 = (wmn % Cfg::WN) * Cfg::TN * 32;
        const float* ep = sp + 4 * KA;
#ifdef SMG_TRACE_EPI    // dev: stamps inside the epilogue (start | stores issued | column sums reduced | atomics issued)
        unsigned long long* etr = (g_smg_trace && threadIdx.x == 0) ? g_smg_trace + 8 * (size_t)blockIdx.x : nullptr;
        if (etr) etr[0] = smg_stamp();
#endif
        float v[2][Cfg::TN];
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) v[0][j] = v[1][j] = 0.f;
        if constexpr (kOp == 3) {          // products of scaled operands: exact power-of-two correction (this wave's scale block)
            const float gi = wm0 >= kScaleBlock ? c.ginv[1] : c.ginv[0];
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] *= gi;
        }
        const int pbase = c.m0 - c.n * pa.HWp;
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) {
            const int cj = wn0 + j * 32 + l31;
            const int col = c.n0 + cj;
            const bool cok = col < N;
            const float sc = ep[cj], sh = ep[Cfg::BN + cj], mean = ep[2 * Cfg::BN + cj], invstd = ep[3 * Cfg::BN + cj];
            const float gam = ep[4 * Cfg::BN + cj];
            if (kEarly && active && c.whole) {
                // Whole tile inside the plane: no per-element predicates, operands already in registers (init_ctx), a uniform
                // tile base + running 32-bit lane offset for the stores.
                if (c.n0 + wn0 + j * 32 + 32 > N) continue;      // (wave-uniform) accumulator tile outside [0, N)
                if constexpr (kWide) {
    #pragma unroll
                    for (int i = 0; i < Cfg::TM; ++i) {
                        const int64_t ag = (int64_t)(c.m0 + wm0 + i * 32) * ldd + dcoff + c.n0 + wn0 + j * 32;
    #pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float o[4], xv4[4], go4[4] = {0.f, 0.f, 0.f, 0.f};
                            finish_acc_rows<XT>(c.xq[(kEarly && kWide) ? i : 0][(kEarly && kWide) ? j : 0][g], xv4);
                            if constexpr (EMODE == E_ACCUM) finish_acc_rows<GT>(c.gq[(kEarly && kWide && EMODE == E_ACCUM) ? i : 0][(kEarly && kWide && EMODE == E_ACCUM) ? j : 0][g], go4);
    #pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int r = 4 * g + k;
                                const float xr = xv4[k];
                                const float dy = bn1(xr, mean, sc, sh) > 0.f ? acc[i][j][r] : 0.f;
                                if constexpr (EMODE == E_STORE) o[k] = dy;
                                else o[k] = go4[k] + gam * dy;
                                v[0][j] += dy;
                                v[1][j] += dy * ((xr - mean) * invstd);
                            }
                            store_acc_rows<GT>(dst, ag, ldd, g, half, lane, o[0], o[1], o[2], o[3]);
                        }
                    }
                } else {
                    char* gb = static_cast<char*>(dst) + (int64_t)GSZ * ((int64_t)c.m0 * ldd + dcoff + c.n0);
    #pragma unroll
                    for (int i = 0; i < Cfg::TM; ++i) {
                        unsigned og = (unsigned)((wm0 + i * 32 + 4 * half) * ldd + cj);
    #pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float xr = c.xv[(kEarly && !kWide) ? i : 0][(kEarly && !kWide) ? j : 0][r];
                            const float dy = bn1(xr, mean, sc, sh) > 0.f ? acc[i][j][r] : 0.f;
                            if constexpr (EMODE == E_STORE) st1<GT>(gb, og, dy);
                            else st1<GT>(gb, og, c.gold[(kEarly && !kWide && EMODE == E_ACCUM) ? i : 0][(kEarly && !kWide && EMODE == E_ACCUM) ? j : 0][r] + gam * dy);
                            og += (r & 3) == 3 ? 5u * (unsigned)ldd : (unsigned)ldd;
                            v[0][j] += dy;
                            v[1][j] += dy * ((xr - mean) * invstd);
                        }
                    }
                }
            } else if (EMODE == E_UNPOOL && kWide && active && pbase + Cfg::BM <= pa.HW && c.n0 + Cfg::BN <= N) {
                // 16-bit storage, whole tile: every 4 x 4 block of (pooled pixel, channel) is transposed inside its quad, so a lane
                // owns ONE pooled pixel and four consecutive channels - four 8-byte mask loads and four 8-byte G' stores (its 2 x 2
                // source pixels) instead of sixteen 2-byte ones each; the per-channel sums are transposed back.
                const int cq0 = wn0 + j * 32 + 4 * (l31 >> 2);
                float pm_[4], ps_[4], ph_[4], pi_[4], pg_[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ps_[k] = ep[cq0 + k]; ph_[k] = ep[Cfg::BN + cq0 + k]; pm_[k] = ep[2 * Cfg::BN + cq0 + k];
                    pi_[k] = ep[3 * Cfg::BN + cq0 + k]; pg_[k] = ep[4 * Cfg::BN + cq0 + k];
                }
#pragma unroll
                for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float a4[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        quad_transpose4(a4[0], a4[1], a4[2], a4[3]);
                        const int p = pbase + wm0 + i * 32 + 8 * g + 4 * half + (lane & 3);
                        const int y = p / pa.W, x = p - y * pa.W;
                        const int64_t pix0 = (int64_t)c.n * pm.HWp + (2 * y) * pm.W + 2 * x;
                        float4 xq[4];
#pragma unroll
                        for (int d = 0; d < 4; ++d) xq[d] = ldq<XT>(mbuf, (pix0 + (d >> 1) * pm.W + (d & 1)) * ldm + mcoff + c.n0 + cq0);
                        float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const float xv4[4] = {xq[d].x, xq[d].y, xq[d].z, xq[d].w};
                            float o[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float dy = bn1(xv4[k], pm_[k], ps_[k], ph_[k]) > 0.f ? 0.25f * a4[k] : 0.f;
                                o[k] = pg_[k] * dy;
                                s0[k] += dy;
                                s1[k] += dy * ((xv4[k] - pm_[k]) * pi_[k]);
                            }
                            stq<GT>(dst, (pix0 + (d >> 1) * pm.W + (d & 1)) * ldd + dcoff + c.n0 + cq0, make_float4(o[0], o[1], o[2], o[3]));
                        }
                        quad_transpose4(s0[0], s0[1], s0[2], s0[3]);
                        quad_transpose4(s1[0], s1[1], s1[2], s1[3]);
                        v[0][j] += (s0[0] + s0[1]) + (s0[2] + s0[3]);
                        v[1][j] += (s1[0] + s1[1]) + (s1[2] + s1[3]);
                    }
            } else
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i) {
                // Two passes per 32x32 accumulator tile: every load first (16..64 of them in
                // flight), then the math and the stores.  Interleaving them serialises on one
                // memory round trip per element, because the stores may alias the loads.
                constexpr int NL = (EMODE == E_UNPOOL) ? 4 : 1;
                constexpr int RB = (EMODE == E_UNPOOL) ? 8 : 16;      // accumulator rows per batch (bounds live registers)
#pragma unroll
                for (int rb = 0; rb < 16; rb += RB) {
                    float xv[RB][NL], gold[RB];
                    int pix0[RB];                                     // first pixel row of the element (fits int: streams*HWp)
                    bool ok[RB];
#pragma unroll
                    for (int q = 0; q < RB; ++q) {
                        const int r = rb + q;
                        const int row = SMG_ACC_ROW(wm0, i, r, half);
                        const int p = pbase + row;
                        ok[q] = active && p < pa.HW && cok;
                        gold[q] = 0.f;
                        if constexpr (EMODE == E_UNPOOL) {
                            const int y = p / pa.W, x = p - y * pa.W;
                            pix0[q] = c.n * pm.HWp + (2 * y) * pm.W + 2 * x;
#pragma unroll
                            for (int d = 0; d < 4; ++d)
                                xv[q][d] = ok[q] ? ld1<XT>(mbuf, (int64_t)(pix0[q] + (d >> 1) * pm.W + (d & 1)) * ldm + mcoff + col) : 0.f;
                        } else {
                            pix0[q] = c.m0 + row;
                            xv[q][0] = ok[q] ? ld1<XT>(mbuf, (int64_t)pix0[q] * ldm + mcoff + col) : 0.f;
                            if constexpr (EMODE == E_ACCUM) gold[q] = ok[q] ? ld1<GT>(dst, (int64_t)pix0[q] * ldd + dcoff + col) : 0.f;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < RB; ++q) {
                        if (!ok[q]) continue;
                        const float av = acc[i][j][rb + q];
                        if constexpr (EMODE == E_UNPOOL) {
                            const float up = 0.25f * av;
#pragma unroll
                            for (int d = 0; d < 4; ++d) {
                                const float dy = bn1(xv[q][d], mean, sc, sh) > 0.f ? up : 0.f;
                                st1<GT>(dst, (int64_t)(pix0[q] + (d >> 1) * pm.W + (d & 1)) * ldd + dcoff + col, gam * dy);
                                v[0][j] += dy;
                                v[1][j] += dy * ((xv[q][d] - mean) * invstd);
                            }
                        } else {
                            const float dy = bn1(xv[q][0], mean, sc, sh) > 0.f ? av : 0.f;
                            if constexpr (EMODE == E_STORE) st1<GT>(dst, (int64_t)pix0[q] * ldd + dcoff + col, dy);
                            else st1<GT>(dst, (int64_t)pix0[q] * ldd + dcoff + col, gold[q] + gam * dy);
                            v[0][j] += dy;
                            v[1][j] += dy * ((xv[q][0] - mean) * invstd);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);    // one batch of loads in flight at a time
                }
            }
        }
        float tot[2];
#ifdef SMG_TRACE_EPI
        if (etr) etr[1] = smg_stamp();
#endif
        block_col_reduce<Cfg, 2, float>(v, smem, tot);
#ifdef SMG_TRACE_EPI
        if (etr) etr[2] = smg_stamp();
#endif
        if (t < Cfg::BN && c.n0 + t < N) {
            const int col = c.n0 + t;
            const float wgt = (EMODE == E_STORE) ? 1.f : ep[4 * Cfg::BN + t];
            const int64_t oi = (int64_t)c.n * ostride + ocoff + col;
            atomicAdd(o1 + oi + stat_rep(), (double)(wgt * tot[0]));
            atomicAdd(o2 + oi + stat_rep(), (double)(wgt * tot[1]));
            const int64_t rep = (int64_t)(blockIdx.x & 7) * rep_stride;
            atomicAdd(dbeta + rep + col, tot[0]);
            atomicAdd(dgamma + rep + col, tot[1]);
        }
#ifdef SMG_TRACE_EPI
        if (etr) { etr[3] = smg_stamp(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); etr[4] = smg_stamp(); }
#endif
    }
};

// ------------------------------------------------------------------------------------
// Layer-grouped 1x1 data gradient.
//
// Inside a dense block every layer i reads ALL earlier channels, so the backward adds
//   G'[p][c] += gamma_i[c] * relu'_i(x[p][c]) * (D2_i[p][:] . W1_i[:, c])        for c < cin_i
// once per layer: E_ACCUM above re-reads x and read-modify-writes G' over all cin_i channels for each
// layer - 26 GB of the 93 GB a training step moves.  The ReLU mask differs per layer (gamma_i, beta_i),
// so the layers cannot share one K loop, but they can share the OUTPUT TILE: for the channels below the
// lowest layer of a group (c < N = cin of that layer) this kernel keeps x and a running sum in
// registers, walks the group's layers as K segments (128 bottleneck channels each), folds the
// accumulator into the running sum at the end of every segment (mask + gamma of that layer), and
// touches G' once.  The channels produced inside the group (needed by the very next layer) are still
// done per layer by E_ACCUM on a narrow column range.
//   per-segment sums  -> dbeta_i, dgamma_i          (sum dy_i, sum dy_i * xhat)
//   running-sum sums  -> SA, SB of the block input  (sum_i gamma_i * those, = sums of the running sum)
// ------------------------------------------------------------------------------------
constexpr int GROUP_MAX = 4;
struct GroupSeg {
    const float* binv;              // operand kind 3: [streams][HWp / 64] inverse block scales of g
    const void* g;                  // finished bottleneck gradient D2_i: units (operand kind 3, see kD2K8) or [n][HWp][KA] fp32 / bf16 by mode
    const u32x4* wp; int ldn;       // conv1 weight, packed data-gradient units [piece][KA/8][cin_i]
    const float* gamma; const float* beta;
    float* dbeta; float* dgamma;
};

template <class Cfg_, int PREC = 0>
struct BwdDataGroupP {
    using Cfg = Cfg_;
    static_assert(Cfg::WK == 1 && Cfg::AT, "segment hook: no in-block split-K");
    using GT = grd_t<PREC>;      // D2 and G'
    using XT = act_t<PREC>;      // the block buffer X
    static constexpr int kOp = bwd_op(PREC), kAE = 16 / GT::size, kBE = 4, GSZ = GT::size;
    static constexpr bool kARawCopy = kAE == 8;
    static constexpr bool kAUnit = kOp == 3;            // D2 in unit form: straight copies, per-block scales
    static constexpr bool kWide = GT::size == 2;        // 16-bit storage: quad-transposed 8-byte epilogue loads / stores (see BwdDataP)
    GroupSeg seg[GROUP_MAX]; int nseg;
    int ldg; Plane pa; int KA;
    int N;                                              // output channels [0, N)
    const void* mbuf; int ldm;                          // block buffer X (mask / xhat source)
    const double* msum; const double* msq; int mstride; StatTab mtab;
    void* dst; int ldd;                                 // G'
    double* o1; double* o2; int ostride;                // SA / SB [n][C]
    int rep_stride;                                     // replicas of the dbeta / dgamma scratch (see BwdDataP)
    float eps;
    TileMap tm;
    static constexpr int kSwizzle = 1;
    // deep k-tiles with ONE tile of loads in flight (the staging registers of two half-as-deep tiles, half the barriers): mode 0
    // 128 x 64 x 32 (serialised 1.61 -> 1.47 ms per step), 16-bit modes 128 x 64 x 64 and 64 x 64 x 64 (config 3: 1.95 -> 1.79 and
    // 1.54 -> 1.35 ms, step 28.8 -> 28.3 ms)
    static constexpr int kPrefetch = ((PREC == 0 && Cfg::BK >= 32 && Cfg::TM * Cfg::TN >= 2) || (PREC != 0 && Cfg::BK >= 64)) ? kPdDgradGroup
                                     : (Cfg::TM * Cfg::TN <= 1) ? kPdDgrad : kPdDgradBig;
    static constexpr int kDmaFly = kAUnit ? kDmaFlyGroup : 0;       // unit-form D2 + packed weights: both operands by LDS-DMA (gemm_tile)
    static constexpr bool kSegmented = true;
    static constexpr bool kHasPrologue = true;
    static constexpr bool kEarlyFetch = true;
    // mode 0, 64-row tile held to 3 waves per SIMD: 176 -> 168 VGPRs (a few bytes of scratch), 121.7 -> 100.2 us per launch (blocks 3-4)
    static constexpr int kMinWaves = PREC == 0 ? 3 : 2;       // (mode 0, LDS-DMA staging: 128-row tile 186 -> 168 VGPRs + 64 bytes of scratch, three workgroups per CU; x, the running sum and the old G' of the tile live in registers)

    struct Ctx {
        int n, m0, n0; bool whole;
        // 16-bit modes: 16-register vectors like the accumulators they shadow (256 -> 206 registers for the 128-row tile); in mode 0
        // hipcc sends whole vectors to scratch instead (any of the three, 128-192 bytes), so plain arrays there - two of their 96 scalars
        // still end up in scratch at any register budget, with a vmcnt(0) in front of the fold that uses one
        typedef float farr16[16];
        using V16 = typename std::conditional<kWide, f32x16, farr16>::type;
        V16 x[Cfg::TM][Cfg::TN];                        // raw activation of this lane's accumulator elements
                                                        // (16-bit storage, whole tiles: until the first k_hook the registers hold x AS FETCHED, row segments)
        V16 gold[kWide ? 1 : Cfg::TM][kWide ? 1 : Cfg::TN];                     // fp32 storage: old G' of whole tiles, fetched with x at the start
        rawq_t<GT> gq[kWide ? Cfg::TM : 1][kWide ? Cfg::TN : 1][4];             // 16-bit storage: the same as fetched row segments
        V16 run[Cfg::TM][Cfg::TN];                      // sum_i gamma_i * dy_i
        float ls[GROUP_MAX][2][Cfg::TN];                // per-segment column partials (sum dy, sum dy*(x-mean))
    };
    struct ARow { unsigned off; };      // byte offset of the row inside the tile; rows outside the plane: kOOB (read as zero)
    struct DRow {};
    using KPrm = KPrm0;
    using KFin = KPrm0;
    __device__ KPrm k_fetch(const Ctx&, int, int) const { return KPrm{}; }
    __device__ KFin k_finish(const Ctx&, const KPrm&, int, int, const float*) const { return KFin{}; }

    // LDS parameters: mean | invstd | per segment: gamma*invstd | beta | gamma      (BN floats each)
    // operand kind 3: + per segment the inverse (gradient x weight) scale of the tile's two 64-row scale blocks
    static constexpr int kScaleAt = (2 + 3 * GROUP_MAX) * Cfg::BN;
    __host__ __device__ int param_floats() const { return kScaleAt + 2 * GROUP_MAX; }

    __device__ void d_init(const Ctx&, DRow&, int) const {}
    __device__ void d_next(const Ctx&, DRow&) const {}
    __device__ void init_params(const Ctx& c, float* sp) const {
        const double minv = 1.0 / (double)pa.HW;
        if constexpr (kOp == 3) {
            static_assert(Cfg::BM == kScaleBlock || (Cfg::BM == 2 * kScaleBlock && Cfg::WM == 2), "a wave's rows lie in one scale block");
            if (threadIdx.x >= 64 && threadIdx.x < 64 + 2 * GROUP_MAX) {      // (wave 1: wave 0 carries the parameter loads below)
                const int z = (threadIdx.x - 64) >> 1, b = (threadIdx.x - 64) & 1;
                float inv = 1.f;
                if (z < nseg) inv = seg[z].binv[(int64_t)c.n * (pa.HWp / kScaleBlock) + (c.m0 - c.n * pa.HWp) / kScaleBlock + (Cfg::BM > kScaleBlock ? b : 0)] * pack_inv_scale(seg[z].wp);
                sp[kScaleAt + 2 * z + b] = inv;
            }
        }
        for (int j = threadIdx.x; j < Cfg::BN; j += 256) {
            const int col = c.n0 + j;
            float mean = 0.f, invstd = 0.f;
            // every segment's gamma / beta first (one memory round trip; a loop that loads and stores per segment waits per load)
            float g[GROUP_MAX], be[GROUP_MAX];
#pragma unroll
            for (int s = 0; s < GROUP_MAX; ++s) {
                const bool on = s < nseg && col < N;
                g[s] = on ? seg[s].gamma[col] : 0.f;
                be[s] = on ? seg[s].beta[col] : 0.f;
            }
            if (col < N) tab_or_moments(mtab, c.n, col, msum, msq, (int64_t)c.n * mstride + col, minv, eps, mean, invstd);
            sp[j] = mean;
            sp[Cfg::BN + j] = invstd;
#pragma unroll
            for (int s = 0; s < GROUP_MAX; ++s) {
                float* q = sp + (2 + 3 * s) * Cfg::BN;
                q[j] = g[s] * invstd;
                q[Cfg::BN + j] = be[s];
                q[2 * Cfg::BN + j] = g[s];
            }
        }
    }
    __device__ bool init_ctx(Ctx& c, const VBlock& vb) const {
        int mt = vb.x, nt = vb.y;
        if (tm.nM && !tile_decode(tm, vb.x, mt, nt)) return false;
        c.m0 = sgpr(mt * Cfg::BM);          // (scalar registers for sure: the staged loads take them as scalar offsets)
        c.n0 = sgpr(nt * Cfg::BN);
        c.n = sgpr(c.m0 / pa.HWp);
        const int pbase = c.m0 - c.n * pa.HWp;
        if (pbase >= pa.HW) return false;
        c.whole = pbase + Cfg::BM <= pa.HW && c.n0 + Cfg::BN <= N;
#pragma unroll
        for (int s = 0; s < GROUP_MAX; ++s)
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) c.ls[s][0][j] = c.ls[s][1][j] = 0.f;
        return true;
    }
    __device__ void early_fetch(Ctx& c) const {
        const int pbase = c.m0 - c.n * pa.HWp;
        if constexpr (kWide) {
            // this lane's x elements (and, for whole tiles, the old G'): in flight under the first K segment.  Whole tiles fetch
            // row segments (16 / 8 bytes per lane, fetch_acc_rows); x is transposed to accumulator layout by the first segment's
            // k_hook, the old G' by the epilogue.
            const int t = threadIdx.x, lane = t & 63, wmn = (t >> 6) % (Cfg::WM * Cfg::WN), l31 = lane & 31, half = lane >> 5;
            const int wm0 = (wmn / Cfg::WN) * Cfg::TM * 32, wn0 = (wmn % Cfg::WN) * Cfg::TN * 32;
    #pragma unroll
            for (int j = 0; j < Cfg::TN; ++j)
    #pragma unroll
                for (int i = 0; i < Cfg::TM; ++i)
    #pragma unroll
                    for (int r = 0; r < 16; ++r) c.run[i][j][r] = 0.f;
            if (c.whole) {
    #pragma unroll
                for (int j = 0; j < Cfg::TN; ++j)
    #pragma unroll
                    for (int i = 0; i < Cfg::TM; ++i) {
                        const int64_t ax = (int64_t)(c.m0 + wm0 + i * 32) * ldm + c.n0 + wn0 + j * 32;
                        const int64_t ag = (int64_t)(c.m0 + wm0 + i * 32) * ldd + c.n0 + wn0 + j * 32;
    #pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const rawq_t<XT> xr = fetch_acc_rows<XT>(mbuf, ax, ldm, g, half, lane);
                            if constexpr (std::is_same<XT, e_f32>::value) { c.x[i][j][4 * g] = xr.x; c.x[i][j][4 * g + 1] = xr.y; c.x[i][j][4 * g + 2] = xr.z; c.x[i][j][4 * g + 3] = xr.w; }
                            else { c.x[i][j][4 * g] = __uint_as_float(xr.x); c.x[i][j][4 * g + 1] = __uint_as_float(xr.y); }
                            c.gq[i][j][g] = fetch_acc_rows<GT>(dst, ag, ldd, g, half, lane);
                        }
                    }
            } else {
    #pragma unroll
                for (int j = 0; j < Cfg::TN; ++j) {
                    const int col = c.n0 + wn0 + j * 32 + l31;
    #pragma unroll
                    for (int i = 0; i < Cfg::TM; ++i)
    #pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = SMG_ACC_ROW(wm0, i, r, half);
                            const bool ok = pbase + row < pa.HW && col < N;
                            // unconditional load from a clamped address (a branch around it would serialise the loads)
                            const float v = ld1<XT>(mbuf, (int64_t)(c.m0 + (ok ? row : 0)) * ldm + (ok ? col : 0));
                            c.x[i][j][r] = ok ? v : 0.f;
                        }
                }
            }
        } else {
            // this lane's x elements (and, for whole tiles, the old G'): in flight under the first K segment.  Whole tiles load
            // without a predicate - a select on the loaded value would make the workgroup wait for it here.
            const int t = threadIdx.x, lane = t & 63, wmn = (t >> 6) % (Cfg::WM * Cfg::WN), l31 = lane & 31, half = lane >> 5;
            const int wm0 = (wmn / Cfg::WN) * Cfg::TM * 32, wn0 = (wmn % Cfg::WN) * Cfg::TN * 32;
            if (c.whole) {
                const char* xb = static_cast<const char*>(mbuf) + (int64_t)XT::size * ((int64_t)c.m0 * ldm + c.n0);
                const char* gb = static_cast<const char*>(dst) + (int64_t)GSZ * ((int64_t)c.m0 * ldd + c.n0);
    #pragma unroll
                for (int j = 0; j < Cfg::TN; ++j)
    #pragma unroll
                    for (int i = 0; i < Cfg::TM; ++i) {
                        unsigned ox = (unsigned)((wm0 + i * 32 + 4 * half) * ldm + wn0 + j * 32 + l31);
                        unsigned og = (unsigned)((wm0 + i * 32 + 4 * half) * ldd + wn0 + j * 32 + l31);
    #pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            c.x[i][j][r] = ld1<XT>(xb, ox);
                            c.gold[i][j][r] = ld1<GT>(gb, og);
                            ox += (r & 3) == 3 ? 5u * (unsigned)ldm : (unsigned)ldm;
                            og += (r & 3) == 3 ? 5u * (unsigned)ldd : (unsigned)ldd;
                            c.run[i][j][r] = 0.f;
                        }
                    }
            } else {
    #pragma unroll
                for (int j = 0; j < Cfg::TN; ++j) {
                    const int col = c.n0 + wn0 + j * 32 + l31;
    #pragma unroll
                    for (int i = 0; i < Cfg::TM; ++i)
    #pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = SMG_ACC_ROW(wm0, i, r, half);
                            const bool ok = pbase + row < pa.HW && col < N;
                            // unconditional load from a clamped address (a branch around it would serialise the loads)
                            const float v = ld1<XT>(mbuf, (int64_t)(c.m0 + (ok ? row : 0)) * ldm + (ok ? col : 0));
                            c.x[i][j][r] = ok ? v : 0.f;
                            c.run[i][j][r] = 0.f;
                            c.gold[i][j][r] = 0.f;
                        }
                }
            }
        }
    }
    __device__ int kps() const { return KA / Cfg::BK; }                   // k-tiles per segment
    __device__ int ktiles(const Ctx&) const { return nseg * kps(); }
    __device__ void a_row_init(const Ctx& c, ARow& r, int line) const {
        r.off = c.m0 + line - c.n * pa.HWp < pa.HW ? (unsigned)GSZ * (unsigned)(line * ldg) : kOOB;
    }
    using ARaw = RawT<1>;
    using BRaw = u32x4;
    __device__ ARaw a_fetch(const Ctx& c, const ARow& r, int kt, int q) const {
        ARaw o;
        const int s = kt / kps(), ch0 = (kt - s * kps()) * Cfg::BK;
        o.ok = true;
        o.v[0] = bload4(static_cast<const char*>(seg[s].g) + (int64_t)GSZ * c.m0 * ldg, kWholeBuf, r.off + 16u * (unsigned)q, (unsigned)GSZ * (unsigned)ch0);
        return o;
    }
    __device__ ARaw a_quad(const ARaw& o, int h) const { ARaw r; r.ok = o.ok; r.v[0] = slot_quad<GT>(o.v[0], h); return r; }
    __device__ float4 a_xform(const Ctx&, const ARaw& o, const KPrm&, int, int, const float*) const { return o.v[0]; }
    // unit (piece, k8 of this k-tile, tile row) of the segment's D2 (operand kind 3): plane-major units, rows consecutive
    __device__ UnitSrc a_unit_src(const Ctx& c, int kt, int piece, int k8, int row) const {
        const int s = kt / kps(), k80 = (kt - s * kps()) * Cfg::K8;
        return UnitSrc{static_cast<const u32x4*>(seg[s].g) + d2_stream_units(c.n, pa.HWp), kWholeBuf,
                       16u * (unsigned)((piece * kD2K8 + k8) * pa.HWp + row), 16u * (unsigned)(k80 * pa.HWp + (c.m0 - c.n * pa.HWp))};
    }
    __device__ u32x4 a_unit(const Ctx& c, int kt, int piece, int k8, int row) const {
        const UnitSrc u = a_unit_src(c, kt, piece, k8, row);
        return bload_u4(u.base, u.bytes, u.voff, u.soff);
    }
    __device__ u32x4 a_unit_d(const Ctx&, int, int, int, int) const { return u32x4{}; }
    __device__ ARaw a_fetch_d(const Ctx&, const DRow&, int, int, int) const { return ARaw{}; }
    __device__ UnitSrc b_unit_src(const Ctx& c, int kt, int piece, int k8, int r) const {
        const int s = kt / kps(), k80 = (kt - s * kps()) * Cfg::K8;
        const unsigned ldn16 = 16u * (unsigned)seg[s].ldn;                                  // (the row pitch changes with the segment)
        return UnitSrc{seg[s].wp, kWholeBuf, __umul24((unsigned)(piece * (KA / 8) + k8), ldn16) + 16u * (unsigned)r,
                       (unsigned)k80 * ldn16 + 16u * (unsigned)c.n0};                       // rows past N: never stored
    }
    __device__ BRaw b_unit(const Ctx& c, int kt, int piece, int k8, int r) const {
        const UnitSrc u = b_unit_src(c, kt, piece, k8, r);
        return bload_u4(u.base, u.bytes, u.voff, u.soff);
    }
    __device__ u32x4 b_unit_xform(const Ctx&, const BRaw& o, int) const { return o; }

    // End of k-tile kt: at a segment boundary fold the accumulator of that layer into the running sum.
    __device__ void k_hook(Ctx& c, int kt, f32x16 (&acc)[Cfg::TM][Cfg::TN], const float* sp) const {
        const int per = kps();
        if ((kt + 1) % per) return;
        const int s = kt / per;
        const int t = threadIdx.x, lane = t & 63, wmn = (t >> 6) % (Cfg::WM * Cfg::WN), l31 = lane & 31;
        const int wn0 = (wmn % Cfg::WN) * Cfg::TN * 32;
        if (kWide && s == 0 && c.whole) {                 // (workgroup-uniform)
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
                for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v4[4];
                        rawq_t<XT> xr;
                        if constexpr (std::is_same<XT, e_f32>::value) xr = make_float4(c.x[i][j][4 * g], c.x[i][j][4 * g + 1], c.x[i][j][4 * g + 2], c.x[i][j][4 * g + 3]);
                        else xr = make_uint2(__float_as_uint(c.x[i][j][4 * g]), __float_as_uint(c.x[i][j][4 * g + 1]));
                        finish_acc_rows<XT>(xr, v4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) c.x[i][j][4 * g + k] = v4[k];
                    }
        }
        const float* q = sp + (2 + 3 * s) * Cfg::BN;
        float ginv = 1.f;                                  // operand kind 3: this segment's inverse scale, folded into gamma and the sums
        if constexpr (kOp == 3) ginv = sp[kScaleAt + 2 * s + ((t >> 6) % (Cfg::WM * Cfg::WN) / Cfg::WN * Cfg::TM * 32 >= kScaleBlock ? 1 : 0)];      // this wave's scale block
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) {
            const int cj = wn0 + j * 32 + l31;
            const float mean = sp[cj], sc = q[cj], be = q[Cfg::BN + cj], gam = q[2 * Cfg::BN + cj] * ginv;
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float xc = c.x[i][j][r] - mean;
                    const float dy = fmaf(xc, sc, be) > 0.f ? acc[i][j][r] : 0.f;     // the forward's bn1() > 0
                    c.run[i][j][r] = fmaf(gam, dy, c.run[i][j][r]);
                    v0 += dy;
                    v1 = fmaf(dy, xc, v1);
                    acc[i][j][r] = 0.f;
                }
#pragma unroll
            for (int z = 0; z < GROUP_MAX; ++z) {          // static register indices: predicated adds, no scratch
                c.ls[z][0][j] += z == s ? v0 * ginv : 0.f;
                c.ls[z][1][j] += z == s ? v1 * ginv : 0.f;
            }
        }
    }
    __device__ void epilogue(const Ctx& c, f32x16 (&)[Cfg::TM][Cfg::TN], float* smem, float* sp, bool) const {
        const int t = threadIdx.x, lane = t & 63, wmn = (t >> 6) % (Cfg::WM * Cfg::WN), l31 = lane & 31, half = lane >> 5;
        const int wm0 = (wmn / Cfg::WN) * Cfg::TM * 32, wn0 = (wmn % Cfg::WN) * Cfg::TN * 32;
        const int pbase = c.m0 - c.n * pa.HWp;
        constexpr int NQ = 2 + 2 * GROUP_MAX;
        float v[NQ][Cfg::TN];
#ifdef SMG_TRACE_EPI    // dev: stamps inside the epilogue (start | stores issued | column sums reduced | atomics issued)
        unsigned long long* etr = (g_smg_trace && threadIdx.x == 0) ? g_smg_trace + 8 * (size_t)blockIdx.x : nullptr;
        if (etr) etr[0] = smg_stamp();
#endif
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) {
            const int cj = wn0 + j * 32 + l31;
            const int col = c.n0 + cj;
            const float mean = sp[cj], invstd = sp[Cfg::BN + cj];
            float a0 = 0.f, a1 = 0.f;
            if (c.whole) {        // whole tile inside the plane: see E_ACCUM
                if constexpr (kWide) {
    #pragma unroll
                    for (int i = 0; i < Cfg::TM; ++i) {
                        const int64_t ag = (int64_t)(c.m0 + wm0 + i * 32) * ldd + c.n0 + wn0 + j * 32;
    #pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float go4[4], o[4];
                            finish_acc_rows<GT>(c.gq[kWide ? i : 0][kWide ? j : 0][g], go4);
    #pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float run = c.run[i][j][4 * g + k];
                                o[k] = go4[k] + run;
                                a0 += run;
                                a1 = fmaf(run, (c.x[i][j][4 * g + k] - mean) * invstd, a1);
                            }
                            store_acc_rows<GT>(dst, ag, ldd, g, half, lane, o[0], o[1], o[2], o[3]);
                        }
                    }
                } else {
                    char* gb = static_cast<char*>(dst) + (int64_t)GSZ * ((int64_t)c.m0 * ldd + c.n0);
    #pragma unroll
                    for (int i = 0; i < Cfg::TM; ++i) {
                        unsigned og = (unsigned)((wm0 + i * 32 + 4 * half) * ldd + cj);
    #pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float run = c.run[i][j][r];
                            st1<GT>(gb, og, c.gold[kWide ? 0 : i][kWide ? 0 : j][r] + run);
                            og += (r & 3) == 3 ? 5u * (unsigned)ldd : (unsigned)ldd;
                            a0 += run;
                            a1 = fmaf(run, (c.x[i][j][r] - mean) * invstd, a1);
                        }
                    }
                }
            } else
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i) {
                float gold[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {             // every load of the tile first (the stores alias them)
                    const int row = SMG_ACC_ROW(wm0, i, r, half);
                    const bool ok = pbase + row < pa.HW && col < N;
                    const float g = ld1<GT>(dst, (int64_t)(c.m0 + (ok ? row : 0)) * ldd + (ok ? col : 0));
                    gold[r] = ok ? g : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = SMG_ACC_ROW(wm0, i, r, half);
                    const float run = c.run[i][j][r];
                    if (pbase + row < pa.HW && col < N) st1<GT>(dst, (int64_t)(c.m0 + row) * ldd + col, gold[r] + run);
                    a0 += run;                               // run == 0 on masked elements (x, acc are zero there)
                    a1 = fmaf(run, (c.x[i][j][r] - mean) * invstd, a1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            v[0][j] = a0;
            v[1][j] = a1;
#pragma unroll
            for (int z = 0; z < GROUP_MAX; ++z) { v[2 + 2 * z][j] = c.ls[z][0][j]; v[3 + 2 * z][j] = c.ls[z][1][j] * invstd; }
        }
        float tot[NQ];
#ifdef SMG_TRACE_EPI
        if (etr) etr[1] = smg_stamp();
#endif
        block_col_reduce<Cfg, NQ, float>(v, smem, tot);
#ifdef SMG_TRACE_EPI
        if (etr) etr[2] = smg_stamp();
#endif
        if (t < Cfg::BN && c.n0 + t < N) {
            const int col = c.n0 + t;
            const int64_t oi = (int64_t)c.n * ostride + col;
            atomicAdd(o1 + oi + stat_rep(), (double)tot[0]);
            atomicAdd(o2 + oi + stat_rep(), (double)tot[1]);
            const int64_t rep = (int64_t)(blockIdx.x & 7) * rep_stride;
#pragma unroll
            for (int z = 0; z < GROUP_MAX; ++z)
                if (z < nseg) {
                    atomicAdd(seg[z].dbeta + rep + col, tot[2 + 2 * z]);
                    atomicAdd(seg[z].dgamma + rep + col, tot[3 + 2 * z]);
                }
        }
#ifdef SMG_TRACE_EPI
        if (etr) { etr[3] = smg_stamp(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); etr[4] = smg_stamp(); }
#endif
    }
};

// ------------------------------------------------------------------------------------
// Weight-gradient policy (both operands channel-major; reduction over pixels).
//   dW[i][j] += sum_p A(p, i) * B(p, j) over the pixels of one z-chunk of one stream.
//   A(p, i) = g*a[i] + x*b[i] + c[i]        (BN-backward-corrected output gradient)
//   B(p, j) = the convolution's input as the forward saw it, recomputed:
//     W_ONE BN+ReLU(bbuf) | W_THREE the same at the tap-shifted pixel |
//     W_POOL avgpool2x2(BN+ReLU(bbuf)) | W_STEM the NHWC4 image at the 7x7/stride-2 tap
//   Epilogue: fp32 atomicAdd into the gradient array in the reference's native
//   [cout][cin][kh][kw] layout (C_IDENT / C_3x3 / C_STEM index maps).
// ------------------------------------------------------------------------------------
enum { W_ONE = 0, W_THREE = 1, W_POOL = 2, W_STEM = 3, W_STEM1 = 4 };      // W_STEM1: one-channel image plane, 49 taps (see F_STEM1)
enum { C_IDENT = 0, C_3x3 = 1, C_STEM = 2, C_STEM1 = 3 };                  // C_STEM1: column = tap, written to all three input channels

// PREC: the engine's precision mode; F32IO: every buffer is fp32 whatever the mode (head conv0; the stem's image / plane).
template <class Cfg_, int BMODE, int CMAP, int PD_ = kPdWgrad, bool AFF = true, int PREC = 0, bool F32IO_ = false>
struct BwdWeightP {
    using Cfg = Cfg_;
    static_assert(!Cfg::AT, "weight-gradient form");
    static constexpr bool F32IO = F32IO_ || BMODE == 3 || BMODE == 4;      // W_STEM, W_STEM1
    using GT = typename std::conditional<F32IO, e_f32, grd_t<PREC>>::type;      // output gradient (gbuf)
    using XT = typename std::conditional<F32IO, e_f32, act_t<PREC>>::type;      // activations (xbuf, bbuf)
    static_assert(GT::size == XT::size, "one slot geometry");
    // operand kind 3: the dense layers' 1x1 weight gradient (finished gradient with a recorded maximum x BN + ReLU activation)
    static constexpr int kOp = (!AFF && BMODE == 0) ? bwd_op(PREC) : bwd_op_plain(PREC), kAE = 16 / GT::size, kBE = 16 / XT::size, GSZ = GT::size, XSZ = XT::size;
    static constexpr bool kARawCopy = !AFF && kAE == 8;      // finished bf16 gradient: copied to LDS as it is
    static constexpr bool kAUnit = !AFF && BMODE == 0 && kOp == 3;      // the gradient operand arrives in unit form (kD2K8): gbuf = units
    static constexpr bool kH = kOp == 3;      // the BN + ReLU operand B becomes fp16 (the gradient side of modes 1 / 2 is bf16): clamped (bnrelu4)
    const float* binv;              // operand kind 3: [streams][HWp / 64] inverse block scales of gbuf (bn_bwd_apply_split_kernel)
    const float* basc;              // operand kind 3: {s, 1 / s} of the BN + ReLU operand B (scale_kernel)
    const void* gbuf; int ldg; int gcoff;
    const void* xbuf; int ldx; int xcoff;
    Plane pa; int MA;
    const double* xsum; const double* xsq; int xstride;
    const double* s1; const double* s2; int sstride; int scoff;
    const float* agamma;
    const void* bbuf; int ldb; Plane pb; int NB;
    const double* bsum; const double* bsq; int bstride; StatTab btab;
    const float* bgamma; const float* bbeta;
    float eps;
    int chunk, chunks_per_stream, n_chunks;   // virtual block z = tap * n_chunks + chunk index
    float* dw; int ldw_out;
    float* part;                              // if set: partial tiles [z][gx*BM][gy*BN] (plain stores,
                                              // summed by reduce_partials_kernel) instead of fp32 atomics into dw
    int gx, gy;                               // tile grid (M tiles x N tiles) of one pixel chunk
    TileMap tm;
    static constexpr int kSwizzle = 2;
    static constexpr int kPrefetch = PD_;     // k-tiles of global loads in flight per thread
    static constexpr int kDmaFly = 0;
    static constexpr bool kSegmented = kAUnit;      // k_hook: the accumulators follow the gradient operand's per-block scale
    static constexpr bool kHasPrologue = true;
    static constexpr bool kEarlyFetch = false;
    static constexpr int kMinWaves = 1;      // (the transitions' pooling form held to 3 waves per SIMD: 92 bytes of scratch, 117 -> 209 us per launch)
    static_assert(!kAUnit || (kScaleBlock % Cfg::BK == 0 && Cfg::WK == 1 && Cfg::BM == 8 * kD2K8), "unit form: whole k-tiles per scale block, all 128 gradient channels");

    // cur_inv (operand kind 3): the inverse scale the accumulators currently carry - that of the scale block being reduced
    struct Ctx { int n, p0, m0, n0, tap, kt, z; float cur_inv; };
    using KPrm = KPrm0;
    using KFin = KPrm0;
    __device__ KPrm k_fetch(const Ctx&, int, int) const { return KPrm{}; }
    __device__ KFin k_finish(const Ctx&, const KPrm&, int, int, const float*) const { return KFin{}; }
    struct ARow { int dummy; };
    struct DRow { int p, y, x; };     // the pixel this staging slot reads, advanced BK per k-tile

    static constexpr int kInvAt = 4 * Cfg::BM + 3 * Cfg::BN;      // operand kind 3: the inverse scales of the chunk's blocks
    __host__ __device__ int param_floats() const { return kInvAt + (kAUnit ? chunk / kScaleBlock + 1 : 0); }

    __device__ void d_init(const Ctx& c, DRow& r, int kr) const {
        r.p = c.p0 + kr;
        r.y = r.p / pa.W;
        r.x = r.p - r.y * pa.W;
    }
    __device__ void d_next(const Ctx&, DRow& r) const {
        r.p += Cfg::BK;
        if constexpr (BMODE != W_ONE) {      // (y, x) only matter to the spatial modes
            // Branch-free: a data-dependent loop here makes hipcc drain vmcnt between the load groups of one
            // k-tile (three serialised memory round trips per tile).  BK <= 32 and W >= 20: two wraps at most.
            r.x += Cfg::BK;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const bool wrap = r.x >= pa.W;
                r.x -= wrap ? pa.W : 0;
                r.y += wrap ? 1 : 0;
            }
        }
    }
    __device__ bool init_ctx(Ctx& c, const VBlock& vb) const {
        int z = vb.z, tile = vb.y * gx + vb.x;
        if (tm.nM && !tile_decode(tm, vb.x, z, tile)) return false;
        c.z = sgpr(z);
        c.m0 = sgpr((tile % gx) * Cfg::BM);
        c.n0 = sgpr((tile / gx) * Cfg::BN);
        c.tap = sgpr(z / n_chunks);
        const int ci = z - c.tap * n_chunks;
        c.n = sgpr(ci / chunks_per_stream);
        c.p0 = sgpr((ci - c.n * chunks_per_stream) * chunk);
        if (c.p0 >= pa.HW) return false;
        int len = pa.HWp - c.p0;
        len = len < chunk ? len : chunk;
        c.kt = sgpr(len / Cfg::BK);
        c.cur_inv = 1.f;
        if constexpr (kAUnit) c.cur_inv = binv[(int64_t)c.n * (pa.HWp / kScaleBlock) + c.p0 / kScaleBlock];      // (workgroup-uniform: a scalar load)
        return true;
    }
    // unit (piece, channel group k8, pixel kr of k-tile kt) of the gradient operand: plane-major units, consecutive pixels consecutive
    __device__ u32x4 a_unit_d(const Ctx& c, int kt, int piece, int k8, int kr) const {
        return bload_u4(static_cast<const u32x4*>(gbuf) + d2_stream_units(c.n, pa.HWp), 16u * (unsigned)(2 * kD2K8 * pa.HWp),
                        16u * (unsigned)((piece * kD2K8 + k8) * pa.HWp + kr), 16u * (unsigned)(c.p0 + kt * Cfg::BK));
    }
    __device__ u32x4 a_unit(const Ctx&, int, int, int, int) const { return u32x4{}; }
    // End of k-tile kt: when the next k-tile starts another scale block, bring the accumulators to that block's scale (exact: a
    // power of two; a block of zeros keeps inverse scale 1).
    __device__ void k_hook(Ctx& c, int kt, f32x16 (&acc)[Cfg::TM][Cfg::TN], const float* sp) const {
        if constexpr (kAUnit) {
            const int nx = (kt + 1) * Cfg::BK;
            if (nx % kScaleBlock || kt + 1 >= c.kt) return;            // (workgroup-uniform)
            const float inv = sp[kInvAt + nx / kScaleBlock];
            if (inv != c.cur_inv) {
                const float f = c.cur_inv * __uint_as_float((254u << 23) - __float_as_uint(inv));      // cur_inv / inv
#pragma unroll
                for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] *= f;
                c.cur_inv = inv;
            }
        }
    }
    __device__ void init_params(const Ctx& c, float* sp) const {
        const double inv = 1.0 / (double)pa.HW;
        for (int k = threadIdx.x; xbuf && k < Cfg::BM; k += 256) {
            const int ch = c.m0 + k;
            float a = 0.f, q1 = 0.f, mean = 0.f, kk = 0.f;
            if (ch < MA) {
                float invstd;
                bn_moments(xsum, xsq, (int64_t)c.n * xstride + xcoff + ch, inv, eps, mean, invstd);
                const float g = agamma ? agamma[ch] : 1.f;
                q1 = (float)(stat_get(s1, (int64_t)c.n * sstride + scoff + ch) * inv);
                const float q2 = (float)(stat_get(s2, (int64_t)c.n * sstride + scoff + ch) * inv);
                a = g * invstd;
                kk = invstd * q2;
            }
            sp[k] = a;
            sp[Cfg::BM + k] = q1;
            sp[2 * Cfg::BM + k] = mean;
            sp[3 * Cfg::BM + k] = kk;
        }
        if constexpr (BMODE != W_STEM && BMODE != W_STEM1) {
            float* bp = sp + 4 * Cfg::BM;
            const double binv = 1.0 / (double)pb.HW;
            for (int j = threadIdx.x; j < Cfg::BN; j += 256) {
                const int ch = c.n0 + j;
                float mean = 0.f, invstd = 0.f, sc = 0.f, be = 0.f;
                if (ch < NB) {
                    tab_or_moments(btab, c.n, ch, bsum, bsq, (int64_t)c.n * bstride + ch, binv, eps, mean, invstd);
                    const float sa = kOp == 3 ? basc[0] : 1.f;      // the activation scale, folded into the BN + ReLU parameters
                    sc = bgamma[ch] * invstd * sa;
                    be = bbeta[ch] * sa;
                }
                bp[j] = mean;
                bp[Cfg::BN + j] = sc;
                bp[2 * Cfg::BN + j] = be;
            }
        }
        if constexpr (kAUnit) {
            const float* bi = binv + (int64_t)c.n * (pa.HWp / kScaleBlock) + c.p0 / kScaleBlock;
            for (int j = threadIdx.x; j * kScaleBlock < c.kt * Cfg::BK; j += 256) sp[kInvAt + j] = bi[j];
        }
    }
    __device__ int ktiles(const Ctx& c) const { return c.kt; }
    __device__ void a_row_init(const Ctx&, ARow&, int) const {}
    using ARaw = RawT<2>;
    struct RawTaps { float4 v[1]; bool ok; unsigned m; };      // W_STEM1: four gathered taps + which of them exist
    using BRaw = typename std::conditional<BMODE == W_STEM1, RawTaps, RawT<(BMODE == W_POOL) ? 4 : 1>>::type;
    __device__ ARaw a_fetch(const Ctx&, const ARow&, int, int) const { return ARaw{}; }
    // Unconditional loads from clamped addresses (pixel 0 / channel 0 of the stream where the slot lies outside the plane or
    // the matrix; zeroed at the LDS store): a branch around a staged load makes hipcc drain vmcnt(0) between load groups.
    // AFF = false (finished gradient): descriptor loads bounded to the HW rows of the stream - pixels past the plane and
    // channel quads past MA read as zero, nothing to mask and no address arithmetic per k-tile.
    __device__ ARaw a_fetch_d(const Ctx& c, const DRow& r, int kt, int kr, int q) const {
        ARaw o;
        const int ch = c.m0 + kAE * q;                                    // first channel of staging slot q
        if constexpr (!AFF) {
            o.ok = true;
            o.v[0] = bload4(static_cast<const char*>(gbuf) + (int64_t)GSZ * ((int64_t)c.n * pa.HWp * ldg + gcoff), (unsigned)GSZ * (unsigned)(pa.HW * ldg - gcoff),
                            ch < MA ? (unsigned)GSZ * (unsigned)(kr * ldg + ch) : kOOB, (unsigned)GSZ * (unsigned)((c.p0 + kt * Cfg::BK) * ldg));
            return o;
        }
        o.ok = r.p < pa.HW && ch < MA;
        const int64_t pix = (int64_t)c.n * pa.HWp + (o.ok ? r.p : 0);
        const int chc = o.ok ? ch : 0;
        o.v[0] = ld16(gbuf, (int64_t)GSZ * (pix * ldg + gcoff + chc));
        if (xbuf) o.v[1] = ld16(xbuf, (int64_t)XSZ * (pix * ldx + xcoff + chc));          // (launch-uniform)
        return o;
    }
    __device__ ARaw a_quad(const ARaw& o, int h) const {
        ARaw r; r.ok = o.ok;
        r.v[0] = slot_quad<GT>(o.v[0], h);
        r.v[1] = slot_quad<XT>(o.v[1], h);
        return r;
    }
    __device__ BRaw b_quad(const BRaw& o, int h) const {
        BRaw r = o;
#pragma unroll
        for (int j = 0; j < (BMODE == W_POOL ? 4 : 1); ++j) r.v[j] = slot_quad<XT>(o.v[j], h);
        return r;
    }
    __device__ float4 a_xform(const Ctx& c, const ARaw& o, const KPrm&, int, int q, const float* sp) const {
        if constexpr (!AFF) return o.v[0];             // (operand kind 3 takes the unit path: a_unit_d)
        if (!o.ok) return zero4();
        if (!xbuf) return o.v[0];
        return affine2(o.v[0], o.v[1], sp + 4 * q, Cfg::BM);
    }
    __device__ BRaw b_fetch(const Ctx& c, const DRow& r, int kt, int kr, int q) const {
        BRaw o;
        const int ch = c.n0 + kBE * q;                                    // first channel of staging slot q
        o.ok = r.p < pa.HW && ch < NB;
        if constexpr (BMODE == W_ONE) {
            // bounded to the HW rows (zero past them: the A rows there are zero, this keeps the product finite); channel quads
            // past NB read the neighbouring channels of the row - their columns are never stored
            o.ok = true;
            o.v[0] = bload4(static_cast<const char*>(bbuf) + (int64_t)XSZ * c.n * pb.HWp * ldb, (unsigned)XSZ * (unsigned)(pb.HW * ldb), (unsigned)XSZ * (unsigned)(kr * ldb + ch),
                            (unsigned)XSZ * (unsigned)((c.p0 + kt * Cfg::BK) * ldb));
        } else if constexpr (BMODE == W_THREE) {
            const int yy = r.y + c.tap / 3 - 1, xx = r.x + c.tap % 3 - 1;
            o.ok = o.ok && (unsigned)yy < (unsigned)pb.H && (unsigned)xx < (unsigned)pb.W;
            o.v[0] = ld16(bbuf, (int64_t)XSZ * (((int64_t)c.n * pb.HWp + (o.ok ? yy * pb.W + xx : 0)) * ldb + (o.ok ? ch : 0)));
        } else if constexpr (BMODE == W_POOL) {
            const int64_t b = (int64_t)XSZ * (((int64_t)c.n * pb.HWp + (o.ok ? (2 * r.y) * pb.W + 2 * r.x : 0)) * ldb + (o.ok ? ch : 0));
            o.v[0] = ld16(bbuf, b);
            o.v[1] = ld16(bbuf, b + (int64_t)XSZ * ldb);
            o.v[2] = ld16(bbuf, b + (int64_t)XSZ * pb.W * ldb);
            o.v[3] = ld16(bbuf, b + (int64_t)XSZ * (pb.W + 1) * ldb);
        } else if constexpr (BMODE == W_STEM1) {
            const float* img = static_cast<const float*>(bbuf) + (int64_t)c.n * pb.HWp;
            float tv[4];
            const bool pin = r.p < pa.HW;
            o.ok = true; o.m = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tap = ch + j;
                const int yy = 2 * r.y + tap / 7 - 3, xx = 2 * r.x + tap % 7 - 3;
                const bool in = pin && tap < 49 && (unsigned)yy < (unsigned)pb.H && (unsigned)xx < (unsigned)pb.W;
                tv[j] = img[in ? yy * pb.W + xx : 0];
                o.m |= in ? (1u << j) : 0u;
            }
            o.v[0] = make_float4(tv[0], tv[1], tv[2], tv[3]);
        } else {
            const int tap = ch >> 2;
            const int yy = 2 * r.y + tap / 7 - 3, xx = 2 * r.x + tap % 7 - 3;
            o.ok = o.ok && tap < 49 && (unsigned)yy < (unsigned)pb.H && (unsigned)xx < (unsigned)pb.W;
            o.v[0] = ld16(bbuf, 16 * ((int64_t)c.n * pb.HWp + (o.ok ? yy * pb.W + xx : 0)));
        }
        return o;
    }
    // mean | gamma*invstd | beta of the B channel quad this thread stages in every k-tile (loop-invariant: one LDS read)
    __device__ KPrm3 b_fix(const Ctx&, int q, const float* sp) const {
        KPrm3 f{};
        if constexpr (BMODE != W_STEM && BMODE != W_STEM1) {
            const float* pr = sp + 4 * Cfg::BM + 4 * q;
            f.mean = ldv4(pr);
            f.scale = ldv4(pr + Cfg::BN);
            f.beta = ldv4(pr + 2 * Cfg::BN);
        }
        return f;
    }
    __device__ float4 b_xform(const Ctx&, const BRaw& o, int, int, const float*, const KPrm3& f) const {
        if constexpr (BMODE == W_STEM1) {
            return make_float4((o.m & 1u) ? o.v[0].x : 0.f, (o.m & 2u) ? o.v[0].y : 0.f, (o.m & 4u) ? o.v[0].z : 0.f, (o.m & 8u) ? o.v[0].w : 0.f);
        } else if constexpr (BMODE == W_STEM) {
            return o.ok ? o.v[0] : zero4();
        } else {
            if constexpr (BMODE == W_ONE) return bnrelu4<kH>(o.v[0], f);      // (rows of the plane padding: zeros in, clamped out - see bnrelu4)
            if (!o.ok) return zero4();
            if constexpr (BMODE == W_POOL) {
                float4 s = bnrelu4<kH>(o.v[0], f);
                s = add4(s, bnrelu4<kH>(o.v[1], f));
                s = add4(s, bnrelu4<kH>(o.v[2], f));
                s = add4(s, bnrelu4<kH>(o.v[3], f));
                return make_float4(s.x * 0.25f, s.y * 0.25f, s.z * 0.25f, s.w * 0.25f);
            } else {
                return bnrelu4<kH>(o.v[0], f);
            }
        }
    }
    __device__ void epilogue(const Ctx& c, f32x16 (&acc)[Cfg::TM][Cfg::TN], float*, float*, bool active) const {
        const int t = threadIdx.x, lane = t & 63, wmn = (t >> 6) % (Cfg::WM * Cfg::WN), l31 = lane & 31, half = lane >> 5;
        const int wm0 = (wmn / Cfg::WN) * Cfg::TM * 32, wn0 = (wmn % Cfg::WN) * Cfg::TN * 32;
        if constexpr (kOp == 3) {          // products of scaled operands: exact power-of-two correction (the last block's scale x the activation scale)
            const float gi = c.cur_inv * basc[1];
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] *= gi;
        }
        if (CMAP == C_IDENT && c.m0 + Cfg::BM <= MA && c.n0 + Cfg::BN <= NB) {
            // whole tile inside the weight matrix: uniform base + running lane offset, no per-element predicates
            if (!active) return;
            float* ob = part ? part + ((int64_t)c.z * (gx * Cfg::BM) + c.m0) * (gy * Cfg::BN) + c.n0
                             : dw + (int64_t)c.m0 * ldw_out + c.n0;
            const unsigned ldo = part ? (unsigned)(gy * Cfg::BN) : (unsigned)ldw_out;
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j) {
                    unsigned o = (unsigned)(wm0 + i * 32 + 4 * half) * ldo + (unsigned)(wn0 + j * 32 + l31);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (part) ob[o] = acc[i][j][r]; else atomicAdd(ob + o, acc[i][j][r]);
                        o += (r & 3) == 3 ? 5u * ldo : ldo;
                    }
                }
            return;
        }
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) {
                const int col = c.n0 + wn0 + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = c.m0 + SMG_ACC_ROW(wm0, i, r, half);
                    if (active && row < MA && col < NB) {
                        if (part) {
                            part[((int64_t)c.z * (gx * Cfg::BM) + row) * (gy * Cfg::BN) + col] = acc[i][j][r];
                            continue;
                        }
                        int64_t idx;
                        if constexpr (CMAP == C_IDENT) idx = (int64_t)row * ldw_out + col;
                        else if constexpr (CMAP == C_3x3) idx = (int64_t)row * ldw_out + col * 9 + c.tap;
                        else if constexpr (CMAP == C_STEM1) {
                            if (col >= 49) continue;
                            for (int cc = 0; cc < 3; ++cc) atomicAdd(dw + (int64_t)row * ldw_out + cc * 49 + col, acc[i][j][r]);
                            continue;
                        } else {
                            const int tap = col >> 2, cc = col & 3;
                            if (cc == 3 || tap >= 49) continue;
                            idx = (int64_t)row * ldw_out + cc * 49 + tap;
                        }
                        atomicAdd(dw + idx, acc[i][j][r]);
                    }
                }
            }
    }
};

// Sum the partial weight-gradient tiles of one launch over its pixel chunks and add the
// result into the gradient array (reference layout).  Deterministic, no atomics.
//   value(tap, row, col) = sum_z part[tap*tap_stride + z*z_stride + row*ldp + col]
struct ReduceArgs {
    const float* part; int Z, taps, rows, cols, ldp; int64_t z_stride, tap_stride;
    float* dw; int ldw_out, cmap;
};
// A workgroup sums 64 elements: its four waves take every fourth partial tile each (coalesced 256-byte reads, four loads in
// flight per lane) and meet in LDS - a fixed order, so the result is reproducible.  (One thread per element over all Z partials
// was a serial chain of Z / 4 memory round trips on a grid of 144 workgroups: 10-12 us per launch, 189 launches per step.)
// One launch serves up to two reductions (a dense layer's 3x3 and 1x1 weight gradients: one launch less per layer on the side
// stream): workgroups [0, blocks_a) take `ra`, the rest `rb`; each owns exactly one group of 64 elements.
static __global__ void reduce_partials_kernel(const ReduceArgs ra, const ReduceArgs rb, const int blocks_a) {
    __shared__ float red[3][64];
    const bool second = (int)blockIdx.x >= blocks_a;
    const ReduceArgs& a = second ? rb : ra;
    const int total = a.taps * a.rows * a.cols;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {
        const int e0 = ((int)blockIdx.x - (second ? blocks_a : 0)) * 64;
        const int e = e0 + lane;
        const bool on = e < total;
        const int ec = on ? e : 0;
        const int tap = ec / (a.rows * a.cols);
        const int rc = ec - tap * a.rows * a.cols;
        const int row = rc / a.cols, col = rc - row * a.cols;
        const float* p = a.part + tap * a.tap_stride + (int64_t)row * a.ldp + col;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int z = w;
        for (; z + 12 < a.Z; z += 16) {
            s0 += p[(int64_t)z * a.z_stride];
            s1 += p[(int64_t)(z + 4) * a.z_stride];
            s2 += p[(int64_t)(z + 8) * a.z_stride];
            s3 += p[(int64_t)(z + 12) * a.z_stride];
        }
        for (; z < a.Z; z += 4) s0 += p[(int64_t)z * a.z_stride];
        float v = (s0 + s1) + (s2 + s3);
        if (w) red[w - 1][lane] = v;
        __syncthreads();
        if (w == 0 && on) {
            v = (v + red[0][lane]) + (red[1][lane] + red[2][lane]);
            int64_t idx = -1;
            if (a.cmap == C_IDENT) idx = (int64_t)row * a.ldw_out + col;
            else if (a.cmap == C_3x3) idx = (int64_t)row * a.ldw_out + col * 9 + tap;
            else if (a.cmap == C_STEM1) {         // one-channel stem: the three input channels were identical, so are their gradients
                if (col < 49)
                    for (int cc = 0; cc < 3; ++cc) a.dw[(int64_t)row * a.ldw_out + cc * 49 + col] += v;
            } else {
                const int t7 = col >> 2, cc = col & 3;
                if (cc != 3 && t7 < 49) idx = (int64_t)row * a.ldw_out + cc * 49 + t7;
            }
            if (idx >= 0) a.dw[idx] += v;
        }
        __syncthreads();
    }
}

}  // namespace smg
