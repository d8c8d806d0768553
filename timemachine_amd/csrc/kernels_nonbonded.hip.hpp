// Nonbonded device kernels for gfx950 (wave64, LDS pair queue).  Included by nonbonded.hip only.
//
// Data layout in HBM (K = number of interacting atoms, Hilbert order):
//   gathered      Real[K][8]  = {x, y, z, w, q, sig, eps, 0}   one 64 B (f64) / 32 B (f32) record per atom,
//                               cast to Real once at gather time (reference casts at load: k_nonbonded.cuh:134-151)
//   g_du_dx       u64 [3][S]   fixed-point force accumulators in Hilbert order, component-major (S = stride >= K):
//   g_du_dp       u64 [4][S]   a flush instruction's lanes share cache lines (memory-side atomics are served per line)
//   col_atoms     u32 pool     per row block a contiguous segment of interacting column atoms (CSR)
//   items         int4         work items {row_block, col_start, col_count, 0}: 32 rows x <=64 columns each
//
// Tile kernel design (one wave per work item; persistent workgroups of 8-12 waves drawing items from a per-CU pool):
//   phase 1  every lane owns one column atom in registers; 32 rounds; the row atoms sit, as two replicated halves, in
//            registers that every round reads ROTATED through the DPP control of the arithmetic itself: in round r lane l
//            meets row ((l - r) & 15) + 16 * (((l >> 5) ^ (r >> 4)) & 1) -- every row is met by exactly two lanes per round.
//            All 32x64 slots get a CHEAP, CONSERVATIVE distance filter in f32 on coordinates taken relative to the tile's
//            first row atom (so f32 resolution does not depend on how far the unwrapped coordinates have drifted), in the
//            Gram form |r|^2 - 2 r.c < cut2 - |c|^2 (5 vector instructions per round): it keeps every pair whose exact d2
//            could be below cutoff^2 (margin >> f32 rounding error).  Survivors are compacted with mask + mbcnt into an
//            LDS queue of 16-bit (round, column lane) entries, four rounds per asm statement.
//   phase 2  whenever >= 64 pairs are queued, all 64 lanes pop one pair each, redo the distance in Real precision
//            with the exact strict test `d2 < cutoff^2` (the only test that decides anything), and run the expensive
//            erfc/exp/sincos path at full lane occupancy (the reference leaves ~2/3 of the lanes idle inside
//            its `if (d2 < cutoff^2)` branch); results are converted to fixed point and added with LDS u64
//            atomics into per-tile row / column accumulators.
//   flush    one global u64 atomic per touched (atom, component) per tile.
// Integer accumulation is associative, so none of this reordering changes a single bit of the result.
#pragma once
#include <type_traits>
#include "kernels_bonded.hip.hpp"
#include "nb_pair.hip.hpp"

namespace tmamd {

// Waves per SIMD the tile kernel is register-budgeted for (512 / waves VGPRs per lane).  f64: 4 waves = 128 registers since the
// general filter paths became asm statements too (round 3: in plain C++ they took the whole kernel from 130 to 156 registers
// and 4 waves meant 60 spilled dwords; measured then: 63.2 / 68 us against 63.9 / 62 at 3 waves; now 55.9 against 58.9 on a bare
// tile launch, 57.2 against 57.8 with a step's bonded terms on board).  f32: ~100 registers -> 4 waves (5 would fit with 96,
// but only in a workgroup shape whose item pools are too small, see TileShape).
template <typename Real> struct TileWaves {
#ifndef TM_TILE_WAVES_F64
#define TM_TILE_WAVES_F64 4
#endif
#ifndef TM_TILE_WAVES_F32
#define TM_TILE_WAVES_F32 4
#endif
    static const int value = sizeof(Real) == 8 ? TM_TILE_WAVES_F64 : TM_TILE_WAVES_F32;
};
// Workgroup shape of the tile kernel.  f64: one workgroup owns a whole CU (16 waves = 4 per SIMD; the du/dp variants 12).  f32: two 8-wave
// workgroups per CU = 4 waves per SIMD -- the registers would admit 5, but 20 waves only tile a CU as five 4-wave
// groups, and pools of ~10 items for 4 waves end 29 % apart (measured: 5 x 4 waves 2740 ns/day, 2 x 8 2920, 1 x 16 2905).  Its waves draw work items from a
// pool that belongs to the workgroup through an LDS ticket counter: static, cost-sorted pools across CUs (sums over
// ~48 items are even to ~2 %), dynamic within a CU (a wave that finishes early takes the next item instead of idling
// while its three SIMD neighbours work on).  The du_dp variants need more LDS per wave: f32 runs them as one 16-wave
// workgroup per CU.  -DTM_TILE_STATIC: one wave per workgroup = the purely static deal (kept for A/B measurements).
template <typename Real, bool DU_DP> struct TileShape {
#ifdef TM_TILE_STATIC
    static const int waves = 1;
    static const int wgs_per_cu = 4 * TileWaves<Real>::value;
    static const int min_waves = TileWaves<Real>::value;
#else
    // a workgroup's waves are spread over the four SIMDs in dispatch order, so only multiples of four waves tile a CU
    // exactly (two 10-wave workgroups do not fit: 3+3+2+2 twice overfills a SIMD's register file)
#ifndef TM_TILE_F32_WG_WAVES
#define TM_TILE_F32_WG_WAVES 8
#define TM_TILE_F32_WGS 2
#endif
    // f64 du_dp variants: 3 waves per SIMD whatever the forces-only shape is (their per-wave LDS is 1.5x, their register need larger)
    static const int f64_waves_per_simd = DU_DP ? (TM_TILE_WAVES_F64 < 3 ? TM_TILE_WAVES_F64 : 3) : TM_TILE_WAVES_F64;
    static const int waves = sizeof(Real) == 8 ? 4 * f64_waves_per_simd : (DU_DP ? 16 : TM_TILE_F32_WG_WAVES);
    static const int wgs_per_cu = sizeof(Real) == 8 ? 1 : (DU_DP ? 1 : TM_TILE_F32_WGS);
    static const int min_waves = sizeof(Real) == 8 ? f64_waves_per_simd : (DU_DP ? 4 : (TM_TILE_F32_WG_WAVES * TM_TILE_F32_WGS) / 4);
#endif
    static const int waves_per_cu = waves * wgs_per_cu;
};
// Which launches take the "wide" shape (f64: 3 waves per SIMD = 168 registers; f32: one 16-wave workgroup): the du/dp variants, whose
// per-wave LDS is 1.5x; DUAL energy launches (two geometries' operands and two pair evaluations per batch: 128 registers spill);
// and, when TM_ENERGY_WIDE is set, f64 energy-only launches (their 128-register form carries 400 bytes of scratch).
#ifndef TM_ENERGY_WIDE
#define TM_ENERGY_WIDE 0
#endif
#ifndef TM_DUAL_WIDE
#define TM_DUAL_WIDE 1
#endif
template <typename Real, bool U, bool X, bool PP, bool DUAL> constexpr bool tile_wide() {
    return PP || (DUAL && TM_DUAL_WIDE && sizeof(Real) == 8) || (TM_ENERGY_WIDE && sizeof(Real) == 8 && U && !X && !PP);
}
// LDS traffic of the tile kernel is private to a wave: program order plus a compiler fence is all the synchronisation
// there is (LDS serves one wave's requests in order).  Never a workgroup barrier -- the waves of a workgroup are at
// unrelated points of unrelated items.
#ifndef TM_LDS_SINGLE_READS
#define TM_LDS_SINGLE_READS 1
#endif
// byte offset of an LDS object inside the workgroup's allocation
template <typename T> __device__ __forceinline__ unsigned int lds_offset(const T *p) {
    return static_cast<unsigned int>(reinterpret_cast<unsigned long>((__attribute__((address_space(3))) const void *)p));
}
// one ds_read_b64, issued and NOT waited for: the value is valid only after lds_wait14() on it
template <int OFFSET> __device__ __forceinline__ double lds_read_f64_async(const unsigned int byte_offset) {
    double v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(byte_offset), "n"(OFFSET));
    return v;
}
// (COL_BASE: byte distance of the column array from the row array, both addressed from the row array's base; SKIP_W: component
// 3 is neither read nor waited for)
template <int C, int ROW_STRIDE, int COL_STRIDE, int COL_BASE, bool SKIP_W = false> __device__ __forceinline__ void lds_read14(const unsigned int ra, const unsigned int ca, double (&ri)[7], double (&cj)[7]) {
    if constexpr (C < 7) {
        if constexpr (!(SKIP_W && C == 3)) {
            ri[C] = lds_read_f64_async<C * ROW_STRIDE>(ra);
            cj[C] = lds_read_f64_async<COL_BASE + C * COL_STRIDE>(ca);
        }
        lds_read14<C + 1, ROW_STRIDE, COL_STRIDE, COL_BASE, SKIP_W>(ra, ca, ri, cj);
    }
}
template <int C, int ROW_STRIDE, int COL_STRIDE, int COL_BASE, bool SKIP_W = false> __device__ __forceinline__ void lds_read14(const unsigned int, const unsigned int, float (&)[7], float (&)[7]) {}
// waits for every outstanding LDS operation of the wave; the values pass through the statement so that no use of them can
// be scheduled in front of it
template <bool SKIP_W = false> __device__ __forceinline__ void lds_wait14(double (&a)[7], double (&b)[7]) {
    if constexpr (SKIP_W) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]));
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]),
                       "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]));
    }
}
template <bool SKIP_W = false> __device__ __forceinline__ void lds_wait14(float (&)[7], float (&)[7]) {}
// experiment (TM_SPLIT_WAIT, flat items): the six coordinate reads are issued first and waited for alone -- LDS returns in order --, so
// the displacement and d^2 issue while charge, sigma and epsilon are still on their way
__device__ __forceinline__ void lds_wait_first6_of12(double (&a)[7], double (&b)[7]) {
    asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));
}
__device__ __forceinline__ void lds_wait_first6_of12(float (&)[7], float (&)[7]) {}
__device__ __forceinline__ void lds_wait_rest6(float (&)[7], float (&)[7]) {}
__device__ __forceinline__ void lds_wait_rest6(double (&a)[7], double (&b)[7]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]));
}
// branch-probability hint for the block placement of the f64 kernels (measured: +0.5 % there, -2 % on the f32 kernels)
template <bool ON> __device__ __forceinline__ bool hint(const bool c, const bool expected) {
    if constexpr (ON) {
        return __builtin_expect(c, expected);
    } else {
        return c;
    }
}
// max over lanes 0-31 of non-negative values (lanes 32-63 must hold 0), returned wave-uniform.  Five DPP row operations
// (prefix max within each 16-lane row by row_shr 1, 2, 4, 8 -- sources beyond the row read 0 -- then row 0's total into
// row 1 by row_bcast:15) and one v_readlane: VALU only.  Non-negative floats order like their bit patterns, so the max is
// an integer max (one instruction per step; an IEEE fmax costs two canonicalising moves more).  All 64 lanes must be enabled.
__device__ __forceinline__ float wave_max_low_half_nonneg(float v) {
    unsigned int x = __float_as_uint(v);
#define TM_DPP_MAX_STEP(CTRL, ROW_MASK)                                                                                \
    {                                                                                                                  \
        const unsigned int y = static_cast<unsigned int>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), CTRL, ROW_MASK, 0xf, true)); \
        x = x > y ? x : y;                                                                                             \
    }
    TM_DPP_MAX_STEP(0x111, 0xf); // row_shr:1
    TM_DPP_MAX_STEP(0x112, 0xf); // row_shr:2
    TM_DPP_MAX_STEP(0x114, 0xf); // row_shr:4
    TM_DPP_MAX_STEP(0x118, 0xf); // row_shr:8
    TM_DPP_MAX_STEP(0x142, 0xa); // row_bcast:15 into rows 1 and 3
#undef TM_DPP_MAX_STEP
    return __uint_as_float(static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(x), 31)));
}
__device__ __forceinline__ double rint_real(const double v) { return __builtin_rint(v); }
__device__ __forceinline__ float rint_real(const float v) { return __builtin_rintf(v); }
__device__ __forceinline__ double fma_real(const double a, const double b, const double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_real(const float a, const float b, const float c) { return __builtin_fmaf(a, b, c); }
// ---- phase-1 building blocks of the tile kernel (see the kernel's header) --------------------------------------------
// row_ror:N within every 16-lane row: lane l receives lane ((l & 15) - N) & 15 of its own row (scripts/microbench/dpp_row_ror.hip
// prints the mapping on the device).  All 64 lanes must be enabled.
template <int N> __device__ __forceinline__ float row_ror(const float v) {
    if constexpr (N == 0) {
        return v;
    } else {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, true));
    }
}
// Four filter rounds of a "flat" item (every w equal) in the Gram form  |r - c|^2 < cut2  <=>  |r|^2 - 2 r.c < cut2 - |c|^2:
// per round one move and three multiply-adds whose row operand arrives through the DPP rotation of the instruction itself
// (row_ror:0..3 of registers that the caller rotates by four between groups), and one compare into a scalar mask.  Written
// as one asm statement because the compiler does not fold a DPP move into v_fmac (it emits v_mov_dpp + v_fmac: 7 instead of
// 4 per round) -- and the whole kernel is bound by instruction issue, scalar instructions included.
// The leading s_nop 1 covers the one hazard the compiler cannot see inside an asm statement: a DPP read of a VGPR needs two
// wait states after the VALU write of that register (the caller's rotate-by-four moves).
__device__ __forceinline__ void filter4_gram_flat(
    const float rx, const float ry, const float rz, const float rr, const float c2x, const float c2y, const float c2z, const float thr,
    u64 &m0, u64 &m1, u64 &m2, u64 &m3) {
    float a0, a1, a2, a3;
    asm volatile(
        "s_nop 1\n\t"
        "v_mov_b32_e32 %[a0], %[rr]\n\t"
        "v_mov_b32_dpp %[a1], %[rr] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[a2], %[rr] row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[a3], %[rr] row_ror:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_e32 %[a0], %[rx], %[c2x]\n\t"
        "v_fmac_f32_dpp %[a1], %[rx], %[c2x] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a2], %[rx], %[c2x] row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a3], %[rx], %[c2x] row_ror:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_e32 %[a0], %[ry], %[c2y]\n\t"
        "v_fmac_f32_dpp %[a1], %[ry], %[c2y] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a2], %[ry], %[c2y] row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a3], %[ry], %[c2y] row_ror:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_e32 %[a0], %[rz], %[c2z]\n\t"
        "v_fmac_f32_dpp %[a1], %[rz], %[c2z] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a2], %[rz], %[c2z] row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a3], %[rz], %[c2z] row_ror:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_cmp_lt_f32_e64 %[m0], %[a0], %[thr]\n\t"
        "v_cmp_lt_f32_e64 %[m1], %[a1], %[thr]\n\t"
        "v_cmp_lt_f32_e64 %[m2], %[a2], %[thr]\n\t"
        "v_cmp_lt_f32_e64 %[m3], %[a3], %[thr]"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)
        : [rx] "v"(rx), [ry] "v"(ry), [rz] "v"(rz), [rr] "v"(rr), [c2x] "v"(c2x), [c2y] "v"(c2y), [c2z] "v"(c2z), [thr] "v"(thr));
}
// The same with the w term (items whose atoms differ in w, and the general path's Gram items): one more multiply-add per round.
__device__ __forceinline__ void filter4_gram_w(
    const float rx, const float ry, const float rz, const float rw, const float rr, const float c2x, const float c2y, const float c2z,
    const float c2w, const float thr, u64 &m0, u64 &m1, u64 &m2, u64 &m3) {
    float a0, a1, a2, a3;
    asm volatile(
        "s_nop 1\n\t"
        "v_mov_b32_e32 %[a0], %[rr]\n\t"
        "v_mov_b32_dpp %[a1], %[rr] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[a2], %[rr] row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[a3], %[rr] row_ror:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_e32 %[a0], %[rx], %[c2x]\n\t"
        "v_fmac_f32_dpp %[a1], %[rx], %[c2x] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a2], %[rx], %[c2x] row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a3], %[rx], %[c2x] row_ror:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_e32 %[a0], %[ry], %[c2y]\n\t"
        "v_fmac_f32_dpp %[a1], %[ry], %[c2y] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a2], %[ry], %[c2y] row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a3], %[ry], %[c2y] row_ror:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_e32 %[a0], %[rz], %[c2z]\n\t"
        "v_fmac_f32_dpp %[a1], %[rz], %[c2z] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a2], %[rz], %[c2z] row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a3], %[rz], %[c2z] row_ror:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_e32 %[a0], %[rw], %[c2w]\n\t"
        "v_fmac_f32_dpp %[a1], %[rw], %[c2w] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a2], %[rw], %[c2w] row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a3], %[rw], %[c2w] row_ror:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_cmp_lt_f32_e64 %[m0], %[a0], %[thr]\n\t"
        "v_cmp_lt_f32_e64 %[m1], %[a1], %[thr]\n\t"
        "v_cmp_lt_f32_e64 %[m2], %[a2], %[thr]\n\t"
        "v_cmp_lt_f32_e64 %[m3], %[a3], %[thr]"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)
        : [rx] "v"(rx), [ry] "v"(ry), [rz] "v"(rz), [rw] "v"(rw), [rr] "v"(rr), [c2x] "v"(c2x), [c2y] "v"(c2y), [c2z] "v"(c2z), [c2w] "v"(c2w),
          [thr] "v"(thr));
}
// Four rounds of the explicit form for items the Gram form does not cover (large extents, sparse boxes): differences through the
// same DPP operands, the minimum image per component (box edges and their inverses as scalars), the squared distance against
// the padded cutoff.  One round after the other on four scratch registers: these items are rare, what matters is that this
// path does not cost the kernel registers (in plain C++ the compiler interleaved the four rounds and the whole kernel went
// from 130 to 156 VGPRs).
#define TM_EXPLICIT_ROUND(CTRL_SUB, MASK)                                                                              \
    "v_sub_f32_" CTRL_SUB(dx, rx, cfx) "\n\t"                                                                          \
    "v_sub_f32_" CTRL_SUB(dy, ry, cfy) "\n\t"                                                                          \
    "v_sub_f32_" CTRL_SUB(dz, rz, cfz) "\n\t"                                                                          \
    "v_mul_f32_e32 %[t], %[ibx], %[dx]\n\t"                                                                            \
    "v_rndne_f32_e32 %[t], %[t]\n\t"                                                                                   \
    "v_fma_f32 %[dx], -%[bx], %[t], %[dx]\n\t"                                                                         \
    "v_mul_f32_e32 %[t], %[iby], %[dy]\n\t"                                                                            \
    "v_rndne_f32_e32 %[t], %[t]\n\t"                                                                                   \
    "v_fma_f32 %[dy], -%[by], %[t], %[dy]\n\t"                                                                         \
    "v_mul_f32_e32 %[t], %[ibz], %[dz]\n\t"                                                                            \
    "v_rndne_f32_e32 %[t], %[t]\n\t"                                                                                   \
    "v_fma_f32 %[dz], -%[bz], %[t], %[dz]\n\t"                                                                         \
    "v_mul_f32_e32 %[t], %[dx], %[dx]\n\t"                                                                             \
    "v_fmac_f32_e32 %[t], %[dy], %[dy]\n\t"                                                                            \
    "v_fmac_f32_e32 %[t], %[dz], %[dz]\n\t"                                                                            \
    "v_sub_f32_" CTRL_SUB(dx, rw, cfw) "\n\t"                                                                          \
    "v_fmac_f32_e32 %[t], %[dx], %[dx]\n\t"                                                                            \
    "v_cmp_lt_f32_e64 %[" MASK "], %[t], %[cut2]\n\t"
#define TM_SUB_PLAIN(D, R, C) "e32 %[" #D "], %[" #R "], %[" #C "]"
#define TM_SUB_ROR1(D, R, C) "dpp %[" #D "], %[" #R "], %[" #C "] row_ror:1 row_mask:0xf bank_mask:0xf"
#define TM_SUB_ROR2(D, R, C) "dpp %[" #D "], %[" #R "], %[" #C "] row_ror:2 row_mask:0xf bank_mask:0xf"
#define TM_SUB_ROR3(D, R, C) "dpp %[" #D "], %[" #R "], %[" #C "] row_ror:3 row_mask:0xf bank_mask:0xf"
__device__ __forceinline__ void filter4_explicit(
    const float rx, const float ry, const float rz, const float rw, const float cfx, const float cfy, const float cfz, const float cfw,
    const float bx, const float by, const float bz, const float ibx, const float iby, const float ibz, const float cut2, u64 &m0, u64 &m1,
    u64 &m2, u64 &m3) {
    float dx, dy, dz, t;
    asm volatile("s_nop 1\n\t" TM_EXPLICIT_ROUND(TM_SUB_PLAIN, "m0") TM_EXPLICIT_ROUND(TM_SUB_ROR1, "m1") TM_EXPLICIT_ROUND(TM_SUB_ROR2, "m2")
                     TM_EXPLICIT_ROUND(TM_SUB_ROR3, "m3") "s_nop 0"
                 : [dx] "=&v"(dx), [dy] "=&v"(dy), [dz] "=&v"(dz), [t] "=&v"(t), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)
                 : [rx] "v"(rx), [ry] "v"(ry), [rz] "v"(rz), [rw] "v"(rw), [cfx] "v"(cfx), [cfy] "v"(cfy), [cfz] "v"(cfz), [cfw] "v"(cfw),
                   [bx] "s"(bx), [by] "s"(by), [bz] "s"(bz), [ibx] "s"(ibx), [iby] "s"(iby), [ibz] "s"(ibz), [cut2] "v"(cut2));
}
#undef TM_EXPLICIT_ROUND
#undef TM_SUB_PLAIN
#undef TM_SUB_ROR1
#undef TM_SUB_ROR2
#undef TM_SUB_ROR3
// The row < col test of the tiles on the list's diagonal, on four rounds' masks: round r0 + k keeps lane l's hit iff the row slot
// it meets, ((l - (r0 + k)) & 15) | half_bit, lies below jrel (= the column's index minus the row block's first index).
__device__ __forceinline__ void order4(const int lane, const int r0, const unsigned int half_bit, const int jrel, u64 &m0, u64 &m1, u64 &m2, u64 &m3) {
    int t;
    u64 o;
    asm volatile(
        "v_subrev_u32_e32 %[t], %[r0], %[lane]\n\t"
        "v_and_or_b32 %[t], %[t], 15, %[hb]\n\t"
        "v_cmp_lt_i32_e64 %[o], %[t], %[jrel]\n\t"
        "s_and_b64 %[m0], %[m0], %[o]\n\t"
        "v_subrev_u32_e32 %[t], %[r1], %[lane]\n\t"
        "v_and_or_b32 %[t], %[t], 15, %[hb]\n\t"
        "v_cmp_lt_i32_e64 %[o], %[t], %[jrel]\n\t"
        "s_and_b64 %[m1], %[m1], %[o]\n\t"
        "v_subrev_u32_e32 %[t], %[r2], %[lane]\n\t"
        "v_and_or_b32 %[t], %[t], 15, %[hb]\n\t"
        "v_cmp_lt_i32_e64 %[o], %[t], %[jrel]\n\t"
        "s_and_b64 %[m2], %[m2], %[o]\n\t"
        "v_subrev_u32_e32 %[t], %[r3], %[lane]\n\t"
        "v_and_or_b32 %[t], %[t], 15, %[hb]\n\t"
        "v_cmp_lt_i32_e64 %[o], %[t], %[jrel]\n\t"
        "s_and_b64 %[m3], %[m3], %[o]"
        : [t] "=&v"(t), [o] "=&s"(o), [m0] "+s"(m0), [m1] "+s"(m1), [m2] "+s"(m2), [m3] "+s"(m3)
        : [lane] "v"(lane), [hb] "v"(half_bit), [jrel] "v"(jrel), [r0] "s"(r0), [r1] "s"(r0 + 1), [r2] "s"(r0 + 2), [r3] "s"(r0 + 3)
        : "scc");
}
// Compaction of four rounds' hits into the wave's LDS queue: the lanes set in mask k write `e0 | k << 11` to consecutive
// 16-bit slots from byte address `qaddr` on (ballot rank = v_mbcnt), and qaddr advances by two bytes per hit.  One asm
// statement: under the compiler an `if (hit)` body costs s_and_saveexec + s_cbranch_execz + s_or per round and the address
// arithmetic three more scalar instructions; here a round is 4 vector + 3 scalar instructions + the LDS write.
// All 64 lanes must be enabled on entry (EXEC is restored to all ones).
__device__ __forceinline__ void compact4(const u64 m0, const u64 m1, const u64 m2, const u64 m3, const unsigned int e0, unsigned int &qaddr) {
    unsigned int t, e, n;
    asm volatile(
        "s_mov_b64 exec, %[m0]\n\t"
        "v_mbcnt_lo_u32_b32 %[t], %[l0], 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], %[h0], %[t]\n\t"
        "v_lshl_add_u32 %[t], %[t], 1, %[q]\n\t"
        "ds_write_b16 %[t], %[e0]\n\t"
        "s_bcnt1_i32_b64 %[n], %[m0]\n\t"
        "s_lshl1_add_u32 %[q], %[n], %[q]\n\t"
        "s_mov_b64 exec, %[m1]\n\t"
        "v_mbcnt_lo_u32_b32 %[t], %[l1], 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], %[h1], %[t]\n\t"
        "v_or_b32_e32 %[e], 0x800, %[e0]\n\t"
        "v_lshl_add_u32 %[t], %[t], 1, %[q]\n\t"
        "ds_write_b16 %[t], %[e]\n\t"
        "s_bcnt1_i32_b64 %[n], %[m1]\n\t"
        "s_lshl1_add_u32 %[q], %[n], %[q]\n\t"
        "s_mov_b64 exec, %[m2]\n\t"
        "v_mbcnt_lo_u32_b32 %[t], %[l2], 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], %[h2], %[t]\n\t"
        "v_or_b32_e32 %[e], 0x1000, %[e0]\n\t"
        "v_lshl_add_u32 %[t], %[t], 1, %[q]\n\t"
        "ds_write_b16 %[t], %[e]\n\t"
        "s_bcnt1_i32_b64 %[n], %[m2]\n\t"
        "s_lshl1_add_u32 %[q], %[n], %[q]\n\t"
        "s_mov_b64 exec, %[m3]\n\t"
        "v_mbcnt_lo_u32_b32 %[t], %[l3], 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], %[h3], %[t]\n\t"
        "v_or_b32_e32 %[e], 0x1800, %[e0]\n\t"
        "v_lshl_add_u32 %[t], %[t], 1, %[q]\n\t"
        "ds_write_b16 %[t], %[e]\n\t"
        "s_bcnt1_i32_b64 %[n], %[m3]\n\t"
        "s_lshl1_add_u32 %[q], %[n], %[q]\n\t"
        "s_mov_b64 exec, -1"
        : [t] "=&v"(t), [e] "=&v"(e), [n] "=&s"(n), [q] "+s"(qaddr)
        : [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3), [e0] "v"(e0),
          [l0] "s"(static_cast<unsigned int>(m0)), [h0] "s"(static_cast<unsigned int>(m0 >> 32)), [l1] "s"(static_cast<unsigned int>(m1)), [h1] "s"(static_cast<unsigned int>(m1 >> 32)),
          [l2] "s"(static_cast<unsigned int>(m2)), [h2] "s"(static_cast<unsigned int>(m2 >> 32)), [l3] "s"(static_cast<unsigned int>(m3)), [h3] "s"(static_cast<unsigned int>(m3 >> 32))
        : "scc", "memory", "exec");
}
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
static const int NB_CHUNK = 64;        // columns per work item == wave width
#define TM_GRAM_MAX_EXTENT 4.0f        // nm: largest |row component| + |column component| the Gram-form filter is used for
// (NB_SHARDS, NB_CLASSES, NB_COUNTER_CLASS0, NB_NUM_COUNTERS -- the layout of the neighbor-list counters -- live in engine.hpp:
// the integrator's update kernel resets them too)
static const int NB_CLASS_PAIRS = 128; // bucket width; class NB_CLASSES-1 holds items with < 128 pairs
static_assert(NB_SHARDS * NB_CLASSES == 64, "one bucket per lane of the tile kernel's prefix sum");


template <typename Real> struct NbBox {
    Real x, y, z, inv_x, inv_y, inv_z;
};

template <typename Real> __device__ __forceinline__ NbBox<Real> load_box(const double *__restrict__ box) {
    NbBox<Real> b;
    b.x = static_cast<Real>(box[0]);
    b.y = static_cast<Real>(box[4]);
    b.z = static_cast<Real>(box[8]);
    b.inv_x = 1 / b.x;
    b.inv_y = 1 / b.y;
    b.inv_z = 1 / b.z;
    return b;
}

// d2 in 4D; one definition shared by every kernel (the strict cutoff test must see identical bits everywhere).
__device__ __forceinline__ double pair_d2(double dx, double dy, double dz, double dw) {
    return __builtin_fma(dw, dw, __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx)));
}
__device__ __forceinline__ float pair_d2(float dx, float dy, float dz, float dw) {
    return __builtin_fmaf(dw, dw, __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)));
}
#if defined(TM_ABLATE) && TM_ABLATE == 6 // ablation (timing only): phase 2 computes every pair but issues no LDS atomic
#define TM_LDS_ACC_GATE(v) if ((v) != 0x123456789abcull) return
#elif defined(TM_ABLATE) && TM_ABLATE == 7 // ablation (timing only): plain LDS stores instead of atomics (wrong sums, same traffic shape)
#define TM_LDS_ACC_GATE(v) do { *p = (v); return; } while (0)
#else
#define TM_LDS_ACC_GATE(v) do { } while (0)
#endif
__device__ __forceinline__ void lds_add(u64 *p, u64 v) {
    TM_LDS_ACC_GATE(v);
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_sub(u64 *p, u64 v) {
    TM_LDS_ACC_GATE(v);
    __hip_atomic_fetch_sub(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// A merged order (engine.hpp: NonbondedAllPairs as the carrier of an interaction group) pads the group's row atoms to whole 32-slot
// blocks with HOLES: perm[slot] == NB_HOLE.  A hole's record carries w = 1e18: as a row of a tile it meets the filters exactly like
// a row beyond NR (kernels below: s_rowflt[3] = 1e18 for those) and fails the exact test d2 < cutoff^2 in either precision; its
// position is never looked at by anything that decides something (the list build counts rows by the group's size, not by the block).
static const unsigned int NB_HOLE = 0xffffffffu;
template <typename Real>
__device__ __forceinline__ void write_hole_record(Real *__restrict__ gathered, u64 *__restrict__ g_du_dx, u64 *__restrict__ g_du_dp, const int acc_stride, const int idx) {
    Real *g = gathered + static_cast<size_t>(idx) * 8;
    g[0] = 0;
    g[1] = 0;
    g[2] = 0;
    g[3] = static_cast<Real>(1e18);
    g[4] = 0;
    g[5] = 0;
    g[6] = 0;
    g[7] = 0;
    if (g_du_dx) {
        g_du_dx[0 * acc_stride + idx] = 0;
        g_du_dx[1 * acc_stride + idx] = 0;
        g_du_dx[2 * acc_stride + idx] = 0;
    }
    if (g_du_dp) {
        g_du_dp[0 * acc_stride + idx] = 0;
        g_du_dp[1 * acc_stride + idx] = 0;
        g_du_dp[2 * acc_stride + idx] = 0;
        g_du_dp[3 * acc_stride + idx] = 0;
    }
}
template <typename Real>
__device__ __forceinline__ void write_second_record(Real *__restrict__ g, const Real xn, const Real yn, const Real zn, const double *__restrict__ pa) {
    g[0] = xn;
    g[1] = yn;
    g[2] = zn;
    g[3] = static_cast<Real>(pa[3]); // w
    g[4] = static_cast<Real>(pa[0]); // q
    g[5] = static_cast<Real>(pa[1]); // sig
    g[6] = static_cast<Real>(pa[2]); // eps
    g[7] = 0;
}

// ---- K1: rebuild check + gather (+ zero the Hilbert-order accumulators) -------------------------------------
// reference: k_check_rebuild_coords_and_box_gather + k_gather_coords_and_params (k_nonbonded.cuh:12-84)
template <typename Real>
__global__ void k_check_gather(
    const int K, const unsigned int *__restrict__ perm, const double *__restrict__ x, const double *__restrict__ p,
    const double *__restrict__ box, const double *__restrict__ snap_x, const double *__restrict__ snap_box,
    const double pad2_quarter, // 0.25 * padding^2
    int *__restrict__ flag_set, int *__restrict__ flag_clear, Real *__restrict__ gathered, u64 *__restrict__ g_du_dx,
    u64 *__restrict__ g_du_dp, const int acc_stride, int *__restrict__ slot_of_atom,
    // merged orders (NonbondedAllPairs as the carrier of an interaction group, engine.hpp): slots [0, guest_pad) are the group's row
    // atoms under ITS parameters p_guest (padded to whole blocks with holes, perm == NB_HOLE), the rest the all-pairs atoms under p;
    // those get a second record under p_guest at record index K + 1 + slot (what the group's items read as columns)
    const double *__restrict__ p_guest = nullptr, const int guest_pad = 0,
    // != nullptr: note whether a record the all-pairs items read changes its values (EnergyMemo, engine.hpp)
    EnergyMemo *__restrict__ memo = nullptr) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0) {
        *flag_clear = 0; // the flag the NEXT call will use; its consumers finished a call ago (stream order)
    }
    if (idx < 8) {
        gathered[static_cast<size_t>(K) * 8 + idx] = 0; // sentinel record: padded list slots point here
        if (p_guest != nullptr) {
            gathered[static_cast<size_t>(2 * K + 1) * 8 + idx] = 0; // ... and the second records' sentinel
        }
    }
    if (idx < 9) {
        if (snap_box[idx] != box[idx]) {
            *flag_set = 1;
        }
    }
    if (idx >= K) {
        return;
    }
    const unsigned int a = perm[idx];
    if (a == NB_HOLE) {
        write_hole_record(gathered, g_du_dx, g_du_dp, acc_stride, idx);
        return;
    }
    slot_of_atom[a] = idx; // inverse of perm (for consumers that pick forces up from the sorted accumulator)
    const double xd = x[a * 3 + 0], yd = x[a * 3 + 1], zd = x[a * 3 + 2];
    Real xo = static_cast<Real>(snap_x[a * 3 + 0]), yo = static_cast<Real>(snap_x[a * 3 + 1]),
         zo = static_cast<Real>(snap_x[a * 3 + 2]);
    Real xn = static_cast<Real>(xd), yn = static_cast<Real>(yd), zn = static_cast<Real>(zd);
    Real dx = xo - xn, dy = yo - yn, dz = zo - zn;
    Real d2 = dx * dx + dy * dy + dz * dz;
    if (static_cast<double>(d2) > pad2_quarter) {
        *flag_set = 1; // benign race: every writer stores the same value
    }
    Real *g = gathered + static_cast<size_t>(idx) * 8;
    const double *pa = (p_guest != nullptr && idx < guest_pad) ? p_guest : p;
    const Real wn = static_cast<Real>(pa[a * 4 + 3]), qn = static_cast<Real>(pa[a * 4 + 0]), sn = static_cast<Real>(pa[a * 4 + 1]), en = static_cast<Real>(pa[a * 4 + 2]);
    if (memo != nullptr && !(p_guest != nullptr && idx < guest_pad)) { // (guest rows' records are not operands of the all-pairs items)
        if (!(g[0] == xn && g[1] == yn && g[2] == zn && g[3] == wn && g[4] == qn && g[5] == sn && g[6] == en)) {
            memo->changed_main = 1; // benign race: every writer stores the same value.  (NaN compares unequal: changed)
        }
    }
    g[0] = xn;
    g[1] = yn;
    g[2] = zn;
    g[3] = wn;
    g[4] = qn;
    g[5] = sn;
    g[6] = en;
    g[7] = 0;
    if (p_guest != nullptr && idx >= guest_pad) {
        write_second_record(gathered + static_cast<size_t>(K + 1 + idx) * 8, xn, yn, zn, p_guest + a * 4);
    }
    // sorted accumulators are component-major (component c of slot i at [c * acc_stride + i]): see the flush of the tile kernel
    if (g_du_dx) {
        g_du_dx[0 * acc_stride + idx] = 0;
        g_du_dx[1 * acc_stride + idx] = 0;
        g_du_dx[2 * acc_stride + idx] = 0;
    }
    if (g_du_dp) {
        g_du_dp[0 * acc_stride + idx] = 0;
        g_du_dp[1 * acc_stride + idx] = 0;
        g_du_dp[2 * acc_stride + idx] = 0;
        g_du_dp[3 * acc_stride + idx] = 0;
    }
}

// The same check + gather for a potential whose box may change by small factors between list builds (a barostat in the
// Context): a changed box is not by itself a reason to rebuild.
//
// The list was built from a snapshot x_s in a box B_s and holds every pair closer than cutoff + padding there.  For ANY
// later coordinates x in a box B = rho * B_s (rho per dimension, diagonal boxes): let e_i = minimum image in B of
// x_i - rho * x_s,i.  A pair with image vector v in (x, B), |v| < cutoff, has the snapshot image vector w with
// rho * w = v - e_i + e_j, so |w| <= (cutoff + 2 max|e|) / rho_min: the list is complete as long as
//     max|e| < (rho_min (cutoff + padding) - cutoff) / 2.
// With |rho - 1| <= NB_SCALE_MAX that holds whenever max|e| < padding / 2 - NB_SCALE_MAX (cutoff + padding) / 2 =: D, the one
// threshold scale-aware potentials use everywhere (this kernel and the integrator's pre-gather test, engine.hpp).
// When the box has changed but no atom is beyond D, the snapshot is re-expressed in the new box (x_s,i <- x_i - e_i: the same
// displacement, next to the atom again) so that later tests compare like with like; snap_box[9..11] carries the accumulated
// rho since the last build (reset to 1 by the build kernels), snap_box[0..8] follows in rebase_snapshot_box().
#define NB_SCALE_MAX 0.004
template <typename Real>
__global__ void k_check_gather_scaled(
    const int K, const unsigned int *__restrict__ perm, const double *__restrict__ x, const double *__restrict__ p,
    const double *__restrict__ box, double *__restrict__ snap_x, const double *__restrict__ snap_box,
    const double threshold2, // D^2
    int *__restrict__ flag_set, int *__restrict__ flag_clear, Real *__restrict__ gathered, u64 *__restrict__ g_du_dx,
    u64 *__restrict__ g_du_dp, const int acc_stride, int *__restrict__ slot_of_atom,
    const double *__restrict__ p_guest = nullptr, const int guest_pad = 0, EnergyMemo *__restrict__ memo = nullptr) { // merged orders, memo: see k_check_gather
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0) {
        *flag_clear = 0;
    }
    if (idx < 8) {
        gathered[static_cast<size_t>(K) * 8 + idx] = 0;
        if (p_guest != nullptr) {
            gathered[static_cast<size_t>(2 * K + 1) * 8 + idx] = 0;
        }
    }
    // wave-uniform: what happened to the box since the snapshot was last expressed in it
    bool same = true, scalable = true;
    double rho[3], b[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double now = box[d * 3 + c], then = snap_box[d * 3 + c];
            same = same && now == then;
            if (c != d) {
                scalable = scalable && now == then; // off-diagonal entries (zeros) must not have changed
            }
        }
        b[d] = box[d * 4];
        rho[d] = b[d] / snap_box[d * 4];
        const double total = rho[d] * snap_box[9 + d];
        scalable = scalable && fabs(total - 1.0) <= NB_SCALE_MAX; // (false for NaN / inf: an uninitialised snapshot)
    }
    if (!same && !scalable && idx == 0) {
        *flag_set = 1;
    }
    if (idx >= K) {
        return;
    }
    const unsigned int a = perm[idx];
    if (a == NB_HOLE) {
        write_hole_record(gathered, g_du_dx, g_du_dp, acc_stride, idx);
        return;
    }
    slot_of_atom[a] = idx;
    const double xd = x[a * 3 + 0], yd = x[a * 3 + 1], zd = x[a * 3 + 2];
    double ex = xd - snap_x[a * 3 + 0], ey = yd - snap_x[a * 3 + 1], ez = zd - snap_x[a * 3 + 2];
    if (!same && scalable) {
        ex = xd - rho[0] * snap_x[a * 3 + 0];
        ey = yd - rho[1] * snap_x[a * 3 + 1];
        ez = zd - rho[2] * snap_x[a * 3 + 2];
        ex -= b[0] * rint(ex / b[0]);
        ey -= b[1] * rint(ey / b[1]);
        ez -= b[2] * rint(ez / b[2]);
        snap_x[a * 3 + 0] = xd - ex;
        snap_x[a * 3 + 1] = yd - ey;
        snap_x[a * 3 + 2] = zd - ez;
    }
    if (ex * ex + ey * ey + ez * ez > threshold2) {
        *flag_set = 1;
    }
    Real *g = gathered + static_cast<size_t>(idx) * 8;
    const double *pa = (p_guest != nullptr && idx < guest_pad) ? p_guest : p;
    {
        const Real xn = static_cast<Real>(xd), yn = static_cast<Real>(yd), zn = static_cast<Real>(zd);
        const Real wn = static_cast<Real>(pa[a * 4 + 3]), qn = static_cast<Real>(pa[a * 4 + 0]), sn = static_cast<Real>(pa[a * 4 + 1]), en = static_cast<Real>(pa[a * 4 + 2]);
        if (memo != nullptr && !(p_guest != nullptr && idx < guest_pad)) {
            if (!(g[0] == xn && g[1] == yn && g[2] == zn && g[3] == wn && g[4] == qn && g[5] == sn && g[6] == en)) {
                memo->changed_main = 1;
            }
        }
    }
    g[0] = static_cast<Real>(xd);
    g[1] = static_cast<Real>(yd);
    g[2] = static_cast<Real>(zd);
    g[3] = static_cast<Real>(pa[a * 4 + 3]); // w
    g[4] = static_cast<Real>(pa[a * 4 + 0]); // q
    g[5] = static_cast<Real>(pa[a * 4 + 1]); // sig
    g[6] = static_cast<Real>(pa[a * 4 + 2]); // eps
    g[7] = 0;
    if (p_guest != nullptr && idx >= guest_pad) {
        write_second_record(gathered + static_cast<size_t>(K + 1 + idx) * 8, g[0], g[1], g[2], p_guest + a * 4);
    }
    if (g_du_dx) {
        g_du_dx[0 * acc_stride + idx] = 0;
        g_du_dx[1 * acc_stride + idx] = 0;
        g_du_dx[2 * acc_stride + idx] = 0;
    }
    if (g_du_dp) {
        g_du_dp[0 * acc_stride + idx] = 0;
        g_du_dp[1 * acc_stride + idx] = 0;
        g_du_dp[2 * acc_stride + idx] = 0;
        g_du_dp[3 * acc_stride + idx] = 0;
    }
}

// after k_check_gather_scaled (a later launch: the check's threads read the old values): the snapshot's box becomes the current
// one, the accumulated scale takes the step.  One thread; called from the bounds kernel that follows every check on the
// non-pre-gathered path, in its "no rebuild" exit (a build resets both instead).
__device__ __forceinline__ void rebase_snapshot_box(const double *__restrict__ box, double *__restrict__ snap_box) {
    bool same = true, scalable = true;
    double total[3];
    for (int d = 0; d < 3; d++) {
        for (int c = 0; c < 3; c++) {
            same = same && box[d * 3 + c] == snap_box[d * 3 + c];
            if (c != d) {
                scalable = scalable && box[d * 3 + c] == snap_box[d * 3 + c];
            }
        }
        total[d] = box[d * 4] / snap_box[d * 4] * snap_box[9 + d];
        scalable = scalable && fabs(total[d] - 1.0) <= NB_SCALE_MAX;
    }
    if (same || !scalable) {
        return;
    }
    for (int d = 0; d < 3; d++) {
        snap_box[d * 4] = box[d * 4];
        snap_box[9 + d] = total[d];
    }
}

// ---- energy-only evaluations remembered (EnergyMemo, engine.hpp) ---------------------------------------------
// After the check + gather kernel (which noted whether an all-pairs operand changed): does the all-pairs launch of this evaluation have
// the operands, the box and the order of the remembered sum?  `trust`: the host's word that the memo describes the records (the
// previous call into the pipeline was a memo evaluation too).  Asked by every wave of that launch (which then leaves at once) and,
// with the same answer -- nothing writes the memo in between -- by k_memo_finish.
__device__ __forceinline__ bool memo_skips_main(const EnergyMemo *__restrict__ memo, const int trust, const double *__restrict__ box) {
    bool same_box = true;
    for (int k = 0; k < 9; k++) {
        same_box = same_box && memo->box[k] == box[k];
    }
    return trust && memo->valid && !memo->changed_main && same_box;
}
// ... and afterwards: the evaluation's total = (the all-pairs launch's sum, or the remembered one) + the second launch's sum; the memo
// takes the all-pairs sum over.  One workgroup.
static __global__ __launch_bounds__(256) void k_memo_finish(
    EnergyMemo *__restrict__ memo, const int trust, const double *__restrict__ box, const i128 *__restrict__ partials_main, const int n_main,
    const i128 *__restrict__ partials_second, const int n_second, i128 *__restrict__ out) {
    __shared__ i128 s_part[2][4];
    i128 a = 0, b = 0;
    const bool ran = !memo_skips_main(memo, trust, box);
    if (ran) {
        for (int i = threadIdx.x; i < n_main; i += 256) {
            a += partials_main[i];
        }
    }
    for (int i = threadIdx.x; i < n_second; i += 256) {
        b += partials_second[i];
    }
    a = wave_sum_i128(a);
    b = wave_sum_i128(b);
    if ((threadIdx.x & 63) == 0) {
        s_part[0][threadIdx.x >> 6] = a;
        s_part[1][threadIdx.x >> 6] = b;
    }
    __syncthreads(); // (every thread has read the memo by now: thread 0 may rewrite it)
    if (threadIdx.x == 0) {
        const i128 main_sum = ran ? s_part[0][0] + s_part[0][1] + s_part[0][2] + s_part[0][3] : memo->cached_main;
        out[0] = main_sum + s_part[1][0] + s_part[1][1] + s_part[1][2] + s_part[1][3];
        memo->cached_main = main_sum;
        memo->valid = 1;
        memo->changed_main = 0;
        for (int k = 0; k < 9; k++) {
            memo->box[k] = box[k];
        }
        memo->evaluations += 1;
        if (!ran) {
            memo->skipped += 1;
        }
    }
}

// ---- K5: un-permute (reference: k_scatter_accum, k_nonbonded.cuh:86-104) ------------------------------------
template <int D>
__global__ void k_scatter_accum(const int K, const unsigned int *__restrict__ perm, const u64 *__restrict__ g, const int acc_stride, u64 *__restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= K * D) {
        return;
    }
    const int d = idx / K, a = idx - d * K; // component-major source: consecutive threads read consecutive slots
    const u64 v = g[static_cast<size_t>(d) * acc_stride + a];
    if (v != 0 && perm[a] != NB_HOLE) {
        atomicAdd(out + static_cast<size_t>(perm[a]) * D + d, v);
    }
}

template <typename Real, bool NEGATED>
__device__ __forceinline__ i128 nonbonded_pair_list_term(
    const int pair, const double *__restrict__ coords, const double *__restrict__ params, const double *__restrict__ box,
    const int *__restrict__ pair_idxs, const double *__restrict__ scales, const double beta_d, const double cutoff_d,
    const double *__restrict__ es_table, u64 *__restrict__ du_dx, u64 *__restrict__ du_dp, const bool want_u,
    const ForceLayout fl = ForceLayout{3, 1});

// One 256-term block of a ForcePlan table (engine.hpp): bonded terms and pair lists, forces only.  Defined at the end of
// this header; called by k_fused_forces and by the tail of the tile kernel.
// (a real call, not inlined: inside the tile kernel its registers would be allocated together with the item loop's, and the
// f64 kernels sit at their 168-VGPR limit -- with the bonded terms inlined, any change to them moved spills into that loop)
// `window`: FORCE_WINDOW * 3 u64 of LDS private to the calling wave (ForceLayout::win), or nullptr; `window_rows`: FORCE_WINDOW
// ints of LDS private to the wave (the accumulator rows of the window's atoms, fetched while the terms compute).  All 64 lanes
// of the wave must make the call (the window is zeroed, filled and flushed by the whole wave).
typedef __attribute__((address_space(3))) int *lds_int_ptr;
template <typename Real>
__device__ __attribute__((noinline)) void fused_dispatch(
    const FusedTable *__restrict__ table, const int block, const int thread, const double *__restrict__ coords,
    const double *__restrict__ box, u64 *__restrict__ du_dx, ForceLayout fl, lds_u64_ptr window, lds_int_ptr window_rows);

// (energy-only twin, defined at the end of this header)
template <typename Real>
__device__ __forceinline__ i128 fused_dispatch_energy(
    const FusedTable *__restrict__ table, const int block, const int thread, const double *__restrict__ coords,
    const double *__restrict__ box);

// ---- K4: the tile kernel ------------------------------------------------------------------------------------
// Registers holding one work item's inputs while they are in flight from HBM/L2 (software pipeline, see below).
template <typename Real> struct TileRegs {
    int rb, ccount;
    unsigned int ja;  // this lane's column atom (K = padding)
    unsigned int ra;  // lanes 0-31: this lane's row atom (K = beyond NR)
    Real cj[7];       // column atom record
    Real rr[7];       // lanes 0-31: row atom record
    Real ox, oy, oz;  // tile origin (first row atom)
    Real cj2[3], rr2[3]; // DUAL launches: the same atoms' positions in the second geometry
};

// INSIDE_SWITCH (f64 forces-only launches): the host vouches for cutoff <= TM_ES_SWITCH_D -- every caller the reference has --
// so no pair inside the cutoff lies beyond the end of the electrostatic switch.  A template parameter and not a branch in
// the kernel: a second copy of the batch body inside the item loop cost 2 % of the launch, executed or not.
// SPLIT (1, 2 or 4; forces-only launches of SMALL systems): a work item is dealt as SPLIT tickets, each covering 32 / SPLIT of
// the item's 32 filter rounds (and the pairs they find).  With fewer items than waves, one item is one wave's whole life --
// 16 us of a lone wave's exposed latencies for a full 32 x 64 tile -- and the launch lasts as long as its heaviest item;
// split, the same work is SPLIT independent chains on different SIMDs.  (Large systems keep SPLIT = 1: several items per
// wave, and every ticket repeats the item's fetch, its LDS staging and its flush.)
// DUAL (energy-only launches; the barostat's fast path, barostat.hip): TWO geometries of the same atoms on the same list in one
// launch -- the current one (gathered, box) and a proposal (gathered2, box2: every molecule moved rigidly with a box scaled by a
// fraction of a percent).  The filter runs on the current geometry with a cutoff widened by what a pair distance can change
// (|s - 1| (|v| + 2 R), R = the largest atom-to-own-centroid distance, handed over in r2_blocks), phase 2 fetches both positions
// of a pair's atoms and evaluates each geometry's energy behind its own exact cutoff test: the same per-pair function on the same
// operands as two separate launches, hence the same two integer sums, for the filter, the staging and the queue of one.
template <typename Real, bool COMPUTE_U, bool COMPUTE_DU_DX, bool COMPUTE_DU_DP, bool INSIDE_SWITCH = false, int SPLIT = 1, bool DUAL = false>
__global__ __launch_bounds__((64 * TileShape<Real, tile_wide<Real, COMPUTE_U, COMPUTE_DU_DX, COMPUTE_DU_DP, DUAL>()>::waves), (TileShape<Real, tile_wide<Real, COMPUTE_U, COMPUTE_DU_DX, COMPUTE_DU_DP, DUAL>()>::min_waves)) void k_nonbonded_tiles(
    const int K,                               // atoms in `gathered` (record K is an all-zero sentinel)
    const int NR,                              // number of row atoms
    const int upper_triangular,                // rows == cols == all: keep only row < col
    const unsigned int *__restrict__ row_idxs, // [NR] or nullptr (identity)
    const unsigned int *__restrict__ class_counts, // [NB_SHARDS][NB_CLASSES] items per bucket (class 0 = heaviest)
    const unsigned int items_cap,                  // bucket capacity: bucket b lives at items[b * items_cap ...]
    const int4 *__restrict__ items, const unsigned int *__restrict__ col_atoms,
    const Real *__restrict__ gathered, const double *__restrict__ box, const double beta_d, const double cutoff_d,
    const double *__restrict__ es_table, // f64: the electrostatic force-factor table of beta (nb_es_table.hip.hpp); f32: unused
    u64 *__restrict__ g_du_dx, u64 *__restrict__ g_du_dp, const int acc_stride, i128 *__restrict__ u_partials,
    // piggy-backed ForcePlan table (forces-only launches; nullptr otherwise): every few waves run a 64-term slice of its
    // bonded terms / pair lists before their first tile, adding into out_du_dx (the caller's atom order)
    const FusedTable *__restrict__ fused, const int fused_blocks, const double *__restrict__ coords, u64 *__restrict__ out_du_dx,
    const int out_atom_stride, const int out_comp_stride, const int *__restrict__ out_remap, // layout of out_du_dx (ForceLayout)
    long long *__restrict__ timing, // timing: debug builds (-DTM_TIMING) only, 8 cycle counters per wave
    // DUAL launches only (nullptr / 0 otherwise): the second geometry's sorted records, box, atom-order coordinates (for the
    // piggy-backed table) and partial sums; per-block maxima of |atom - own centroid|^2 from the kernel that made the proposal
    const Real *__restrict__ gathered2 = nullptr, const double *__restrict__ box2 = nullptr, const double *__restrict__ coords2 = nullptr,
    i128 *__restrict__ u_partials2 = nullptr, const float *__restrict__ r2_blocks = nullptr, const int n_r2_blocks = 0,
    // energy-only launches of a remembered evaluation (EnergyMemo; nullptr otherwise): the launch has no work if memo_skips_main says so
    const EnergyMemo *__restrict__ gate = nullptr, const int gate_trust = 0) {
    static_assert(!DUAL || (COMPUTE_U && !COMPUTE_DU_DX && !COMPUTE_DU_DP), "DUAL is an energy-only form");
    if constexpr (COMPUTE_U && !COMPUTE_DU_DX && !COMPUTE_DU_DP && !DUAL) {
        if (gate != nullptr && memo_skips_main(gate, gate_trust, box)) {
            return; // (grid-uniform; k_memo_finish takes the remembered sum and does not read this launch's partials)
        }
    }

    constexpr int WAVES = TileShape<Real, tile_wide<Real, COMPUTE_U, COMPUTE_DU_DX, COMPUTE_DU_DP, DUAL>()>::waves;
    struct WaveLds { // one wave's private scratch
        Real row[7][TILE];
        Real col[7][NB_CHUNK];
        u64 fi[COMPUTE_DU_DX ? 3 : 1][COMPUTE_DU_DX ? TILE : 1]; // outputs that are not asked for take 8 bytes, not a row
        u64 fj[COMPUTE_DU_DX ? 3 : 1][COMPUTE_DU_DX ? NB_CHUNK : 1];
        u64 pi[COMPUTE_DU_DP ? 4 : 1][COMPUTE_DU_DP ? TILE : 1];
        u64 pj[COMPUTE_DU_DP ? 4 : 1][COMPUTE_DU_DP ? NB_CHUNK : 1];
        Real row2[DUAL ? 3 : 1][DUAL ? TILE : 1];     // DUAL: x y z of the rows / columns in the second geometry
        Real col2[DUAL ? 3 : 1][DUAL ? NB_CHUNK : 1];
        unsigned int rowatom[TILE];
        float rowflt[5][TILE]; // the rows as the f32 filter sees them: x, y, z relative to the tile origin, w, |r|^2
        unsigned short queue[2 * NB_CHUNK + 4 * NB_CHUNK]; // <= 63 left over in either queue + up to 4 rounds appended between drains
    };
    __shared__ WaveLds s_wave[WAVES];
    __shared__ unsigned int s_ticket; // next position of this workgroup's pool
    __shared__ i128 s_energy[COMPUTE_U ? WAVES : 1]; // energy launches: the waves' sums, added up by thread 0 at the end
    __shared__ i128 s_energy2[DUAL ? WAVES : 1];
    // f64: the workgroup's copy of the electrostatic force-factor table (12 KB, read-only after the barrier below)
    // (energy launches keep the energy-factor table behind it; the du/dp variants, whose per-wave LDS is the largest, read
    // that one from global memory)
    // (... and so does any shape whose waves' scratch leaves no room for a second 12 KB table in the CU's 160 KB)
    constexpr bool G_IN_LDS = sizeof(Real) == 8 && COMPUTE_U && !COMPUTE_DU_DP && WAVES * sizeof(WaveLds) + 2 * ES_TAB_DOUBLES * sizeof(double) + 64 <= 160 * 1024;
    constexpr int ES_TAB_LDS_DOUBLES = (G_IN_LDS ? 2 : 1) * ES_TAB_DOUBLES;
    __shared__ __attribute__((aligned(16))) double s_es_tab[sizeof(Real) == 8 ? ES_TAB_LDS_DOUBLES : 2];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const unsigned int global_wave = blockIdx.x * WAVES + wave, total_waves = gridDim.x * WAVES;
    WaveLds &lds = s_wave[wave];
    auto &s_row = lds.row;
    auto &s_col = lds.col;
    auto &s_fi = lds.fi;
    auto &s_fj = lds.fj;
    auto &s_pi = lds.pi;
    auto &s_pj = lds.pj;
    auto &s_rowatom = lds.rowatom;
    auto &s_queue = lds.queue;
    auto &s_rowflt = lds.rowflt;
    if (threadIdx.x == 0) {
        s_ticket = WAVES; // tickets 0 .. WAVES-1 are the waves' first items: drawn without the counter, before the barrier
    }
    // (the table copy and the workgroup's one barrier come further down, behind the request for the first item: the
    // copy's L2 round trip then runs alongside the three dependent hops of that fetch instead of in front of them)
    // how phase 2 reaches the table: f64 reads the LDS copy with three ds_read_b128 per pair
    struct EsTableLds {
        const double *tab;  // force factor F: the workgroup's LDS copy
        const double *gtab; // energy factor G: the LDS copy (energy launches) or the table in global memory (du/dp variants)
        __device__ __forceinline__ void load(const unsigned int idx, double (&c)[ES_TAB_COEFFS]) const {
            const double2 *p = reinterpret_cast<const double2 *>(tab + idx * ES_TAB_COEFFS);
            const double2 a = p[0], b = p[1], e = p[2];
            c[0] = a.x; c[1] = a.y; c[2] = b.x; c[3] = b.y; c[4] = e.x; c[5] = e.y;
        }
        __device__ __forceinline__ void load_g(const unsigned int idx, double (&c)[ES_TAB_COEFFS]) const {
            const double2 *p = reinterpret_cast<const double2 *>(gtab + idx * ES_TAB_COEFFS);
            const double2 a = p[0], b = p[1], e = p[2];
            c[0] = a.x; c[1] = a.y; c[2] = b.x; c[3] = b.y; c[4] = e.x; c[5] = e.y;
        }
    };
    using TileTab = std::conditional_t<sizeof(Real) == 8, EsTableLds, EsTableNone>;
    TileTab es_tab{};
    if constexpr (sizeof(Real) == 8) {
#ifdef TM_TABLE_L1 // experiment (round 5): the force-factor table read through the vector L1 (global loads) instead of its LDS copy -- the LDS pipe is the busiest unit of the kernel (per-batch timeline).  Measured: 55.3-55.8 us against 55.0-55.7: no difference
        es_tab.tab = (!COMPUTE_U && COMPUTE_DU_DX && !COMPUTE_DU_DP) ? es_table : s_es_tab;
#else
        es_tab.tab = s_es_tab;
#endif
        es_tab.gtab = G_IN_LDS ? s_es_tab + ES_TAB_DOUBLES : es_table + ES_TAB_DOUBLES;
    }
    constexpr bool F64 = sizeof(Real) == 8; // (hints for the f32 kernels too: re-measured at the end of round 2, +0.1 %, not applied)
    const NbBox<Real> bx = load_box<Real>(box);
    const Real cutoff = static_cast<Real>(cutoff_d);
    const Real cutoff2 = cutoff * cutoff;
    // |prefactor * 2^36| below this: every force component of a pair inside the cutoff converts on the fast path
    [[maybe_unused]] const double ps_limit = TM_FIXED_FAST_LIMIT / cutoff_d * 0.999999; // (/ 2^36: the same bound on the prefactor itself)
    const Real beta = static_cast<Real>(beta_d);
    i128 energy = 0;
    [[maybe_unused]] i128 energy2 = 0;
    [[maybe_unused]] NbBox<Real> bx2 = bx;
    double filter_cutoff = cutoff_d;
    if constexpr (DUAL) {
        bx2 = load_box<Real>(box2);
        // the cutoff the filter needs so that a pair inside the cutoff in EITHER geometry passes: |v| <= (|v'| + 2 R |s - 1|) / (1 - |s - 1|)
        float r2 = 0.0f;
        for (int k = lane; k < n_r2_blocks; k += 64) {
            r2 = fmaxf(r2, r2_blocks[k]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            r2 = fmaxf(r2, __shfl_xor(r2, o, 64));
        }
        const double ds = fmax(fabs(box2[0] / box[0] - 1.0), fmax(fabs(box2[4] / box[4] - 1.0), fabs(box2[8] / box[8] - 1.0)));
        filter_cutoff = (cutoff_d + 2.0 * sqrt(static_cast<double>(r2)) * ds * 1.001) / (1.0 - ds) + 1e-6;
    }
    // phase-1 filter in f32: box, and a cutoff^2 padded far beyond the rounding error of the filter arithmetic
    const float fbx = static_cast<float>(bx.x), fby = static_cast<float>(bx.y), fbz = static_cast<float>(bx.z);
    const float fibx = 1.0f / fbx, fiby = 1.0f / fby, fibz = 1.0f / fbz;
    const float fmaxb = fmaxf(fbx, fmaxf(fby, fbz));
    const float fcut2 = static_cast<float>(filter_cutoff * filter_cutoff) + 1e-5f * (1.0f + fmaxb) * (1.0f + static_cast<float>(filter_cutoff));
    // 1 / the largest |row component| + |column component| (relative to the tile origin) the Gram form of the filter is used for
    const float fex = 1.0f / fminf(0.49f * fbx, TM_GRAM_MAX_EXTENT), fey = 1.0f / fminf(0.49f * fby, TM_GRAM_MAX_EXTENT);
    const float fez = 1.0f / fminf(0.49f * fbz, TM_GRAM_MAX_EXTENT), few = 1.0f / TM_GRAM_MAX_EXTENT;

    // Work distribution.  Work items differ in cost by an order of magnitude (0..2048 interacting pairs) and a wave only
    // processes a handful.  The neighbor-list build files every item into bucket (shard, cost class); the cost-sorted order
    // is the concatenation of the buckets, heaviest class first.  It is dealt statically to the (persistent) workgroups
    // -- one pool per CU -- and dynamically, heaviest first, to the waves of a workgroup (TileShape).  Device-wide dynamic
    // ticket counters were measured and rejected: a returning global atomic queues behind the wave's own flush atomics at
    // the memory side and came back 10-40 us later, doubling the kernel's duration.  A purely static deal to single waves
    // (the previous scheme) left the waves ending anywhere between 55 % and 100 % of the launch, because every item also
    // carries ~17k cycles of fixed cost and a wave only gets 3-5 of them; pools took the f64 launch from 93 to 90 us.
    // (Also measured: the XCDs differ in speed by 8 % for f64 / up to 35 % for f32 tiles on a fixed frame, XCDs 3 and 5
    // slowest on three boards; shares adapted from the workgroups' measured durations did not pay in MD, where the slow
    // half of the chip changes every few hundred steps.)
    //
    // Software pipeline over items.  Fetching an item is a chain of dependent memory operations
    //   items[] -> col_atoms[] -> gathered[]
    // which costs several microseconds when paid up front.  Instead, while item k is being computed:
    //   stage A (start of k)   items[k+1] is requested (scalar load)
    //   stage B, C (end of k)  column / row atom indices of k+1, then its atom records, are requested -- after k's
    //                          heavy phase (no extra live registers there) but BEFORE k's flush atomics are issued
    // so one L2 round trip (the index load) is exposed per item instead of three dependent hops.
    const unsigned int uK = static_cast<unsigned int>(K);
    const unsigned int NO_ITEM = 0xffffffffu;
    // f64: this thread's share of the force-factor table is REQUESTED first of all (the L2 is cold at a kernel boundary: ~2.5k
    // cycles) and written to LDS behind the three dependent hops of the first item's fetch, instead of being fetched after them
    constexpr int ES_TAB_PER_THREAD = (ES_TAB_LDS_DOUBLES + WAVES * 64 - 1) / (WAVES * 64);
    [[maybe_unused]] double es_tab_mine[sizeof(Real) == 8 ? ES_TAB_PER_THREAD : 1];
    if constexpr (sizeof(Real) == 8) {
#pragma unroll
        for (int k = 0; k < ES_TAB_PER_THREAD; k++) {
            const int i = static_cast<int>(threadIdx.x) + k * WAVES * 64;
            es_tab_mine[k] = i < ES_TAB_LDS_DOUBLES ? es_table[i] : 0.0;
        }
    }
    unsigned int bucket_end; // lane l: end (inclusive prefix sum) of bucket number l in class-major order
    {
        const unsigned int cls = lane / NB_SHARDS, sh = lane % NB_SHARDS;
        // inclusive wave scan on the VALU: row_shr 1, 2, 4, 8 inside the 16-lane rows (sources beyond a row read 0), then row
        // 0's total into row 1 and row 2's into row 3 (row_bcast:15), then the lower half's total into the upper (row_bcast:31)
        // -- six DPP adds instead of six dependent LDS shuffles
        int v = static_cast<int>(class_counts[sh * NB_CLASSES + cls]);
        v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true); // row_bcast:15 into rows 1 and 3
        v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true); // row_bcast:31 into rows 2 and 3
        bucket_end = static_cast<unsigned int>(v);
    }
    const unsigned int n_items_total = static_cast<unsigned int>(__builtin_amdgcn_readlane(static_cast<int>(bucket_end), 63));
    auto next_position = [&]() -> unsigned int {
        // The pool of workgroup b: round r of the cost-sorted order contributes position r * G + (r even ? b : G-1-b)
        // (serpentine deal over the G workgroups: the one that drew the heaviest item of a round draws the lightest of
        // the next).  Rounds are handed to the waves in order, heaviest first, by an LDS ticket: a CU-local returning
        // atomic (~100 cycles), unlike a global one, which queues behind the wave's own flush atomics (10-40 us).
        unsigned int r = 0;
        if (lane == 0) {
            r = atomicAdd(&s_ticket, 1u);
        }
        r = __builtin_amdgcn_readfirstlane(r);
        const unsigned int w = (r & 1) ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
        return r * gridDim.x + w;
    };
    auto position_of_ticket = [&](const unsigned int r) -> unsigned int {
        const unsigned int w = (r & 1) ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
        return r * gridDim.x + w;
    };
    static_assert(SPLIT == 1 || SPLIT == 2 || SPLIT == 4, "SPLIT");
    constexpr int ROUNDS = TILE / SPLIT; // filter rounds per ticket
    [[maybe_unused]] int sub_drawn = 0;  // which part of its item the ticket last passed to position_to_slot covers
    auto position_to_slot = [&](unsigned int g) -> unsigned int {
        if constexpr (SPLIT > 1) {
            // positions [k * n, (k + 1) * n) are part k of the cost-sorted items: heaviest first within every pass
            int part = 0;
#pragma unroll
            for (int k = 1; k < SPLIT; k++) {
                if (g >= n_items_total) {
                    g -= n_items_total;
                    part = k;
                }
            }
            // (uniform in fact -- g and the item count are the same in every lane -- but not as far as the compiler can tell:
            // the count comes out of a shuffle)
            sub_drawn = __builtin_amdgcn_readfirstlane(part);
        }
        if (g >= n_items_total) {
            return NO_ITEM;
        }
        const unsigned int b = __popcll(__ballot(g >= bucket_end)); // first bucket whose end is beyond g
        const unsigned int first = b ? __shfl(bucket_end, b - 1, 64) : 0u;
        const unsigned int bucket = (b % NB_SHARDS) * NB_CLASSES + b / NB_SHARDS;
        return __builtin_amdgcn_readfirstlane(bucket * items_cap + (g - first));
    };
    auto load_indices = [&](const int4 it, TileRegs<Real> &r) {
        r.rb = it.x;
        // (bit 1 of `upper_triangular`: the guest rows' items of a merged order -- sign bit of their fourth word -- take no columns in
        // this launch: memo evaluations run them in a launch of their own)
        const int ncols = ((upper_triangular & 2) && it.w < 0) ? 0 : it.z;
        r.ccount = ncols;
        r.ja = lane < ncols ? col_atoms[it.y + lane] : uK;
        const int ridx = it.x * TILE + (lane & (TILE - 1));
        r.ra = uK;
        if (ridx < NR) {
            r.ra = row_idxs ? row_idxs[ridx] : static_cast<unsigned int>(ridx);
        }
    };
    // `cost`: the item's fourth word.  Its sign bit marks the items of a merged order's GUEST rows (kernels_nblist.hip.hpp): their
    // columns -- the all-pairs atoms -- are read under the group's parameters, from the second set of records K + 1 records further on
    // (a scalar offset on the base address; record 2 K + 1 is that set's zero sentinel)
    auto load_records = [&](TileRegs<Real> &r, const int cost) {
        const unsigned int ra0 = __builtin_amdgcn_readfirstlane(r.ra); // first row atom of the tile: always valid
        r.ox = gathered[static_cast<size_t>(ra0) * 8 + 0];
        r.oy = gathered[static_cast<size_t>(ra0) * 8 + 1];
        r.oz = gathered[static_cast<size_t>(ra0) * 8 + 2];
        const Real *__restrict__ col_records = gathered + (cost < 0 ? (static_cast<size_t>(uK) + 1) * 8 : 0);
#pragma unroll
        for (int c = 0; c < 7; c++) {
            r.cj[c] = col_records[static_cast<size_t>(r.ja) * 8 + c]; // record K is the zero sentinel: no branch
        }
        if (lane < TILE) {
#pragma unroll
            for (int c = 0; c < 7; c++) {
                r.rr[c] = gathered[static_cast<size_t>(r.ra) * 8 + c];
            }
        }
        if constexpr (DUAL) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                r.cj2[c] = gathered2[static_cast<size_t>(r.ja) * 8 + c];
            }
            if (lane < TILE) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    r.rr2[c] = gathered2[static_cast<size_t>(r.ra) * 8 + c];
                }
            }
        }
    };

#ifdef TM_TIMING_BATCH
    // the per-batch timeline: wall cycles of one wave between stamps inside the batch body, each stamp behind a
    // wait for the wave's own outstanding LDS operations; sums over the wave's batches (wave-uniform: scalar registers)
    long long tm_tb[7] = {0, 0, 0, 0, 0, 0, 0}, tm_tb_last = 0, tm_tb_n = 0;
#define TM_TB(k)                                                                                                       \
    {                                                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
        const long long n_ = clock64();                                                                                \
        tm_tb[k] += n_ - tm_tb_last;                                                                                   \
        tm_tb_last = n_;                                                                                               \
    }
#else
#define TM_TB(k)
#endif
#ifdef TM_TIMING
    long long tm_setup = 0, tm_p1 = 0, tm_p2 = 0, tm_flush = 0, tm_items = 0, tm_batches = 0, tm_bc = 0, tm_stage_a = 0;
    const long long tm_begin = clock64();
    const long long tm_real_begin = static_cast<long long>(__builtin_amdgcn_s_memrealtime() & 0xffffffffull);
#define TM_T(var) const long long var = clock64()
#else
#define TM_T(var)
#endif

    // prologue: first item fetched the slow way (ticket = wave index)
    unsigned int item = position_to_slot(position_of_ticket(static_cast<unsigned int>(wave)));
    [[maybe_unused]] int sub_cur = sub_drawn, sub_next = 0;
    TileRegs<Real> cur;
    if (item != NO_ITEM) {
        const int4 it_first = items[item];
        load_indices(it_first, cur);
        load_records(cur, it_first.w);
    }
    TM_T(tp_fetch);
    if constexpr (sizeof(Real) == 8) {
#pragma unroll
        for (int k = 0; k < ES_TAB_PER_THREAD; k++) {
            const int i = static_cast<int>(threadIdx.x) + k * WAVES * 64;
            if (i < ES_TAB_LDS_DOUBLES) {
                s_es_tab[i] = es_tab_mine[k];
            }
        }
    }
    TM_T(tp_copy);
    // (measured: f64 waves wait ~1.2k cycles here for the slowest wave's table loads.  Taking the barrier later -- behind the first
    // item's staging -- saves 0.4 us per launch without a ForcePlan table on board and COSTS 8 us with one: the waves that run a
    // slice of bonded terms in front of their first item then hold everybody at the barrier.)
    __syncthreads(); // the only workgroup-wide barrier: from here on the waves run independently
    TM_T(tp_barrier);

    if constexpr (!COMPUTE_U && COMPUTE_DU_DX && !COMPUTE_DU_DP) {
        if (fused) {
            // Piggy-backed bonded terms / pair lists: 64-term slices of the table's 256-term blocks, one per few waves.
            // Done here, while this wave's first tile is still on its way from memory (both are chains of dependent
            // loads with little arithmetic) and the SIMD's other waves are computing.  Measured alternatives: at the
            // end of the wave, or in extra workgroups appended / prepended to the grid -- all slower (f32: this placement
            // costs nothing, the others 8-10 us per launch).
            // slice t goes to workgroup t % G, wave (t / G) % WAVES: every CU takes the same share
            for (int t = wave * static_cast<int>(gridDim.x) + static_cast<int>(blockIdx.x); t < fused_blocks * 4; t += static_cast<int>(total_waves)) {
                // (window: this wave's s_fi + s_fj, contiguous and not in use before the first item is staged)
                static_assert(sizeof(lds.fi) + sizeof(lds.fj) >= sizeof(u64) * 3 * FORCE_WINDOW && offsetof(WaveLds, fj) == offsetof(WaveLds, fi) + sizeof(lds.fi), "window");
                static_assert(sizeof(lds.queue) >= sizeof(int) * FORCE_WINDOW, "window rows");
                fused_dispatch<Real>(fused, t >> 2, (t & 3) * 64 + lane, coords, box, out_du_dx, ForceLayout{out_atom_stride, out_comp_stride, out_remap},
                                     (lds_u64_ptr)(&s_fi[0][0]), (lds_int_ptr)(&s_queue[0]));
            }
        }
    }

#ifdef TM_TIMING_PRO
    const long long tp_loop = clock64();
#endif
    if constexpr (COMPUTE_U && !COMPUTE_DU_DX && !COMPUTE_DU_DP) {
        if (fused) {
            // the energy twin: the plan's bonded terms and pair lists add their energies to this wave's partial sum
            for (int t = wave * static_cast<int>(gridDim.x) + static_cast<int>(blockIdx.x); t < fused_blocks * 4; t += static_cast<int>(total_waves)) {
                energy += fused_dispatch_energy<Real>(fused, t >> 2, (t & 3) * 64 + lane, coords, box);
                if constexpr (DUAL) {
                    energy2 += fused_dispatch_energy<Real>(fused, t >> 2, (t & 3) * 64 + lane, coords2, box2);
                }
            }
        }
    }

    while (item != NO_ITEM) {
        TM_T(t_a);
        // ---- stage A happens late in this item (see the round loop): the later a wave draws its next item, the better
        // the pool is balanced when it runs dry; the last filter iteration still hides the items[] load
        unsigned int item_next = NO_ITEM;
        bool have_next = false;
        int4 it_next = make_int4(0, 0, 0, 0);
        TileRegs<Real> nxt;
        nxt.ja = uK;
        nxt.ra = uK;

        // ---- current item: registers -> LDS
        const int rb = cur.rb;
        const unsigned int ja = cur.ja;
        wave_lds_sync(); // previous item's flush has finished reading LDS
        TM_T(t_a2);
        // the f32 filter copy of the row atoms (displacement from the tile origin; |r|^2 for the Gram form) goes through LDS:
        // each half of the rounds reads it back in its own lane arrangement (load_half below)
        // (image(): min_image() that also notes whether anything was wrapped at all)
        bool wrapped = false;
        auto image = [&](const Real d, const Real box_d, const Real inv_d, const bool counts) -> float {
            const Real t = rint_real(d * inv_d);
            wrapped = wrapped || (counts && t != static_cast<Real>(0));
            return static_cast<float>(fma_real(-box_d, t, d)); // == min_image(d, box_d, inv_d)
        };
        float rext = 0.0f; // lanes 0-31: this lane's row atom, largest |component| in units of the extent bound
        float row_w = 0.0f;
        bool row_valid = false;
        if (lane < TILE) {
            s_rowatom[lane] = cur.ra;
#pragma unroll
            for (int c = 0; c < 7; c++) {
                s_row[c][lane] = cur.rr[c];
            }
            row_valid = cur.ra < uK;
            const float fx = image(cur.rr[0] - cur.ox, bx.x, bx.inv_x, row_valid);
            const float fy = image(cur.rr[1] - cur.oy, bx.y, bx.inv_y, row_valid);
            const float fz = image(cur.rr[2] - cur.oz, bx.z, bx.inv_z, row_valid);
            row_w = static_cast<float>(cur.rr[3]);
            s_rowflt[0][lane] = fx;
            s_rowflt[1][lane] = fy;
            s_rowflt[2][lane] = fz;
            s_rowflt[3][lane] = row_valid ? row_w : 1e18f; // an invalid row never passes the explicit filter ...
            s_rowflt[4][lane] = row_valid ? __builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx)) : 1e30f; // ... nor the Gram one
            rext = row_valid ? fmaxf(fmaxf(fabsf(fx) * fex, fabsf(fy) * fey), fmaxf(fabsf(fz) * fez, fabsf(row_w) * few)) : 0.0f;
            if constexpr (COMPUTE_DU_DX) {
                s_fi[0][lane] = 0;
                s_fi[1][lane] = 0;
                s_fi[2][lane] = 0;
            }
            if constexpr (COMPUTE_DU_DP) {
                s_pi[0][lane] = 0;
                s_pi[1][lane] = 0;
                s_pi[2][lane] = 0;
                s_pi[3][lane] = 0;
            }
        }
#pragma unroll
        for (int c = 0; c < 7; c++) {
            s_col[c][lane] = cur.cj[c];
        }
        if constexpr (DUAL) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                lds.col2[c][lane] = cur.cj2[c];
                if (lane < TILE) {
                    lds.row2[c][lane] = cur.rr2[c];
                }
            }
        }
        const bool col_live = ja < uK;
        const float cfx = image(cur.cj[0] - cur.ox, bx.x, bx.inv_x, col_live);
        const float cfy = image(cur.cj[1] - cur.oy, bx.y, bx.inv_y, col_live);
        const float cfz = image(cur.cj[2] - cur.oz, bx.z, bx.inv_z, col_live);
        const float col_w = static_cast<float>(cur.cj[3]);
        const unsigned int row_first = static_cast<unsigned int>(rb * TILE);
        // Filter specialisations, decided once per item (wave-uniform):
        //   gram         every |row - origin| + |col - origin| component stays below min(0.49 box length, 4 nm) (w: 4): row - col
        //                IS the minimum image, and the Gram form's rounding error is bounded (below).  Everything else takes the
        //                explicit form with the minimum image applied per slot.
        //   flat         every atom of the item has the same w (all of a non-alchemical system): the w term is exactly zero
        //   needs_order  only tiles whose columns reach back into (or before) the row block's own index range need row < col
        //   raw_compact  gram, and no atom of the item was wrapped on the way to the origin's image: then the UNWRAPPED
        //                coordinates that phase 2 subtracts differ by less than 0.49 box lengths too, rint(.) == 0,
        //                fma(-box, 0, delta) == delta, and phase 2 skips min_image without changing a bit.  (Atoms that have
        //                left the home cell by different numbers of box lengths make an item non-compact there although its
        //                images are close.)
        bool gram, flat, needs_order, raw_compact;
        {
            float cext = fmaxf(fmaxf(fabsf(cfx) * fex, fabsf(cfy) * fey), fmaxf(fabsf(cfz) * fez, fabsf(col_w) * few));
            cext = col_live ? cext : 0.0f;
            rext = wave_max_low_half_nonneg(rext); // VALU only; the shuffle butterfly it replaces was six dependent LDS round trips per item
            gram = __ballot(!(cext + rext < 1.0f)) == 0ull;
            raw_compact = gram && __ballot(wrapped) == 0ull;
            const float w0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, col_w))); // column 0 of an item is always live
            flat = __ballot((col_live && col_w != w0) || (row_valid && row_w != w0)) == 0ull;
            needs_order = (upper_triangular & 1) && __ballot(col_live && ja <= row_first + (TILE - 1)) != 0ull;
        }
        const bool fast = gram && flat && !needs_order; // Gram form without the w term, no order test: almost every item
        // this lane's column atom as the filters see it.  Gram form: -2c and the threshold cut2 - |c|^2 (rows bring |r|^2 along).
        // Rounding error of that form, u = 2^-24, |row components| <= R, |column components| <= C, R + C < TM_GRAM_MAX_EXTENT = 4:
        // |r|^2 and |c|^2 (four terms each) 16 u R^2 and 16 u C^2; four accumulation steps of magnitude <= 4 R^2 + 8 R C each;
        // the threshold's subtraction u (cut2 + 4 C^2): in all <= u (32 (R + C)^2 + 3) = 3.1e-5, covered three times by the 1e-4
        // added to the threshold (on top of fcut2's own padding, which covers the rounding of the inputs).
        const float c2x = -2.0f * cfx, c2y = -2.0f * cfy, c2z = -2.0f * cfz;
        const float c2w = col_live ? -2.0f * col_w : 2e18f; // explicit form: column w = -c2w / 2 = -1e18 never passes
        float thr;
        {
            float cc = __builtin_fmaf(cfz, cfz, __builtin_fmaf(cfy, cfy, cfx * cfx));
            if (!fast) {
                cc = __builtin_fmaf(col_w, col_w, cc);
            }
            thr = col_live ? (fcut2 + 1e-4f) - cc : -1e30f;
        }
        const int jrel = col_live ? static_cast<int>(ja) - static_cast<int>(row_first) : 0; // row slot i is kept iff i < jrel (ordered tiles)
        if constexpr (COMPUTE_DU_DX) {
            s_fj[0][lane] = 0;
            s_fj[1][lane] = 0;
            s_fj[2][lane] = 0;
        }
        if constexpr (COMPUTE_DU_DP) {
            s_pj[0][lane] = 0;
            s_pj[1][lane] = 0;
            s_pj[2][lane] = 0;
            s_pj[3][lane] = 0;
        }
        wave_lds_sync();

        TM_T(t_b);
#ifdef TM_TIMING
        long long tm_p2_item = 0;
#endif
        // ---- the rounds.  Lane l owns column l.  The 32 row atoms are two halves A = rows 0-15, B = rows 16-31; during rounds
        // 0-15 lanes 0-31 work through A and lanes 32-63 through B, during rounds 16-31 the other way round.  The row registers
        // hold the half replicated in every 16-lane DPP row, and round r reads them rotated by r & 15 within the row -- the
        // rotation is the DPP control of the arithmetic instruction itself (row_ror:0..3 inside a group of four rounds; one
        // move per register rotates by four between groups).  So in round r lane l meets row
        //     ((l - r) & 15) + 16 * (((l >> 5) ^ (r >> 4)) & 1):
        // every row atom is met by exactly two lanes per round (the accumulation addresses of a batch stay spread), nothing
        // moves through LDS, and a round costs 5 vector instructions + its share of the compaction.
        const int r_begin = SPLIT > 1 ? sub_cur * ROUNDS : 0;
        const int r_end = r_begin + ROUNDS;
        unsigned int qaddr = lds_offset(&s_queue[0]); // byte address of the queue's end (cnt entries of 16 bits)
        const unsigned int qbase = qaddr;
        float rx = 0.0f, ry = 0.0f, rz = 0.0f, rw = 0.0f, rq = 0.0f; // the current half's row atoms (rq = |r|^2), rotated by the rounds done
        unsigned int half_bit = 0; // 16 if this lane is working through B
        auto load_half = [&](const int r0) {
            const int h = r0 >> 4;
            half_bit = (((lane >> 5) ^ h) & 1) << 4;
            const unsigned int src = ((lane - r0) & 15) | half_bit; // already rotated by r0 & 15 (tickets of split items start mid-half)
            rx = s_rowflt[0][src];
            ry = s_rowflt[1][src];
            rz = s_rowflt[2][src];
            rw = s_rowflt[3][src];
            rq = s_rowflt[4][src];
            if (!fast) {
                rq = rw < 1e17f ? __builtin_fmaf(rw, rw, rq) : rq;
            }
        };
        // one batch: lane `lane` (if active) takes the pair queued at `slot`.
        // (Measured and dropped: queueing the pairs without a Lennard-Jones term -- 8 of 9 in water -- apart from the
        // others, so that their batches neither read sigma / epsilon nor run the LJ code: one more partial batch per item
        // and two ballots per round cost more than that saved, 82 us against 72 us per launch.  Two pairs per lane per
        // trip with branch-free code, for instruction-level parallelism: 81 us.)
        // `entry_addr`: LDS byte address of this lane's queue entry; FLAT (a tag type): the item's w are all equal, the w term is skipped
        auto pair_batch = [&](const bool active, const unsigned int entry_addr, auto flat_tag) {
            constexpr bool FLAT = decltype(flat_tag)::value;
#ifdef TM_DUMMY_SALU
            { // issue-cost probe (timing builds only): dependent scalar instructions per batch
                int d_ = 0;
#pragma unroll
                for (int i_ = 0; i_ < TM_DUMMY_SALU; i_++) {
                    asm volatile("s_add_u32 %0, %0, 1" : "+s"(d_) : : "scc");
                }
                asm volatile("" : : "s"(d_));
            }
#endif
#ifdef TM_DUMMY_VALU
            { // the same with vector instructions
                int d_ = lane;
#pragma unroll
                for (int i_ = 0; i_ < TM_DUMMY_VALU; i_++) {
                    asm volatile("v_add_u32 %0, 1, %0" : "+v"(d_));
                }
                asm volatile("" : : "v"(d_));
            }
#endif
#ifdef TM_SETPRIO
            __builtin_amdgcn_s_setprio(TM_SETPRIO); // experiment: a wave inside its batch (one long dependent chain) issues ahead of its SIMD neighbours' filter rounds
#endif
#ifdef TM_TIMING_BATCH
            tm_tb_last = clock64();
            tm_tb_n++;
#endif
            if (active) {
                // a queue entry is (round << 11) | column lane: the row slot follows from the round's lane arrangement
                const unsigned int e = *reinterpret_cast<const __attribute__((address_space(3))) unsigned short *>(static_cast<unsigned long>(entry_addr));
                const unsigned int pj = e & 0xffu;
                const unsigned int rnd = e >> 11;
                const unsigned int pi = ((pj - rnd) & 15u) | (((pj >> 1) ^ rnd) & 16u);
                TM_TB(0); // queue entry fetched and decoded
                Real ri[7], cj[7]; // x, y, z, w, q, sig, eps of the pair's row / column atom
                if constexpr (sizeof(Real) == 8 && TM_LDS_SINGLE_READS) {
                    // fourteen (flat items: twelve) single ds_read_b64 (2 LDS cycles each; row reads are conflict free: 32 rows =
                    // 64 banks).  Left to itself the compiler pairs them into ds_read2_b64, which the LDS serves at half that rate
                    // (MI355X_MICROARCH.md, LDS table: 8 cycles per wave instruction against 2 + 2).
                    // (one base register: the column array's distance from the row array rides in the instructions' offset fields)
#if defined(TM_ABLATE) && (TM_ABLATE == 10 || TM_ABLATE == 13) // ablation (timing only): operand reads at conflict-free addresses
                    const unsigned int ra = lds_offset(&s_row[0][0]) + static_cast<unsigned int>(lane & 31) * 8u, ca = lds_offset(&s_row[0][0]) + static_cast<unsigned int>(lane) * 8u;
#else
                    const unsigned int ra = lds_offset(&s_row[0][0]) + pi * 8u, ca = lds_offset(&s_row[0][0]) + pj * 8u;
#endif
                    lds_read14<0, TILE * 8, NB_CHUNK * 8, static_cast<int>(offsetof(WaveLds, col) - offsetof(WaveLds, row)), FLAT>(ra, ca, ri, cj);
#ifdef TM_SPLIT_WAIT
                    if constexpr (FLAT) {
                        lds_wait_first6_of12(ri, cj);
                    } else {
                        lds_wait14<FLAT>(ri, cj);
                    }
#else
                    lds_wait14<FLAT>(ri, cj);
#endif
                } else {
#pragma unroll
                    for (int c = 0; c < 7; c++) {
                        if (FLAT && c == 3) {
                            ri[c] = cj[c] = 0;
                            continue;
                        }
                        ri[c] = s_row[c][pi];
                        cj[c] = s_col[c][pj];
                    }
                }
                TM_TB(1); // twelve / fourteen operands fetched
                Real ddx = ri[0] - cj[0], ddy = ri[1] - cj[1], ddz = ri[2] - cj[2];
                if (hint<F64>(!raw_compact, false)) { // wave-uniform; for a compact tile the three rint / fma pairs are exact no-ops
                    ddx = min_image(ddx, bx.x, bx.inv_x);
                    ddy = min_image(ddy, bx.y, bx.inv_y);
                    ddz = min_image(ddz, bx.z, bx.inv_z);
                }
                // (flat items: w_i - w_j == +0 and fma(0, 0, s) == s, so leaving the term out changes no bit)
                const Real ddw = FLAT ? static_cast<Real>(0) : ri[3] - cj[3];
                const Real dd2 = FLAT ? fma_real(ddz, ddz, fma_real(ddy, ddy, ddx * ddx)) : pair_d2(ddx, ddy, ddz, ddw);
                [[maybe_unused]] Real dd2b = cutoff2; // DUAL: the pair's squared distance in the second geometry
                if constexpr (DUAL) {
                    const Real ex = min_image(lds.row2[0][pi] - lds.col2[0][pj], bx2.x, bx2.inv_x);
                    const Real ey = min_image(lds.row2[1][pi] - lds.col2[1][pj], bx2.y, bx2.inv_y);
                    const Real ez = min_image(lds.row2[2][pi] - lds.col2[2][pj], bx2.z, bx2.inv_z);
                    dd2b = FLAT ? fma_real(ez, ez, fma_real(ey, ey, ex * ex)) : pair_d2(ex, ey, ez, ddw);
                    if (dd2b < cutoff2) {
                        PairOut<Real> o2;
                        nb_pair<true, false>(static_cast<Real>(1), static_cast<Real>(1), ri[4], cj[4], ri[5], cj[5], ri[6], cj[6], dd2b, beta, o2, es_tab);
                        energy2 += float_to_fixed_energy_hot<Real>(o2.u);
                    }
                }
#ifdef TM_SPLIT_WAIT
                if constexpr (sizeof(Real) == 8 && TM_LDS_SINGLE_READS && FLAT) {
                    lds_wait_rest6(ri, cj);
                }
#endif
#ifdef TM_TIMING_BATCH
                asm volatile("" : "+v"(ddx), "+v"(ddy), "+v"(ddz)); // (the stamp must not be hoisted over the arithmetic)
                {
                    Real pin = dd2;
                    asm volatile("" : "+v"(pin));
                    TM_TB(2); // displacement, d^2
                }
#endif
                if (dd2 < cutoff2) { // the exact, strict test: atoms with w == cutoff never interact
                const Real qi = ri[4], qj = cj[4];
                const Real sig_i = ri[5], sig_j = cj[5], eps_i = ri[6], eps_j = cj[6];
                if constexpr (sizeof(Real) == 8 && !COMPUTE_U && COMPUTE_DU_DX && !COMPUTE_DU_DP) {
                    // MD: the prefactor only, and ONE wave-uniform escape for both rare cases -- d2 under the table
                    // (clashing atoms: analytic electrostatics) and a product beyond the fast conversion's range
                    bool below, big;
                    double prefactor = nb_pair_prefactor_deferred<INSIDE_SWITCH>(1.0, 1.0, qi, qj, sig_i, sig_j, eps_i, eps_j, dd2, es_tab, below);
#ifdef TM_TIMING_BATCH
                    asm volatile("" : "+v"(prefactor));
                    TM_TB(3); // table index, table fetch, polynomial, Lennard-Jones, prefactor
#endif
                    u64 fx, fy, fz;
                    pair_force_fixed_fast_bounded(prefactor, ddx, ddy, ddz, ps_limit, fx, fy, fz, big);
#ifdef TM_TIMING_BATCH
                    asm volatile("" : "+v"(fx), "+v"(fy), "+v"(fz));
                    TM_TB(4); // three products, magic-add conversions
#endif
                    const bool rare = below || big;
                    if (__builtin_expect(__ballot(rare) != 0ull, 0)) {
                        if (rare) {
                            const double p = below ? nb_pair_prefactor_below_table(1.0, 1.0, qi, qj, sig_i, sig_j, eps_i, eps_j, dd2, beta) : prefactor;
                            pair_force_fixed_slow(p, ddx, ddy, ddz, fx, fy, fz);
                        }
                    }
#if defined(TM_ABLATE) && (TM_ABLATE == 12 || TM_ABLATE == 13) // ablation (timing only): the six accumulations at conflict-free addresses
                    const unsigned int api = static_cast<unsigned int>(lane & 31), apj = static_cast<unsigned int>(lane);
#else
                    const unsigned int api = pi, apj = pj;
#endif
                    lds_add(&s_fi[0][api], fx);
                    lds_add(&s_fi[1][api], fy);
                    lds_add(&s_fi[2][api], fz);
                    lds_sub(&s_fj[0][apj], fx); // FIX(-p d) == -FIX(p d)
                    lds_sub(&s_fj[1][apj], fy);
                    lds_sub(&s_fj[2][apj], fz);
                    TM_TB(5); // six LDS atomics issued and acknowledged
                } else {
                PairOut<Real> o;
                nb_pair<COMPUTE_U || COMPUTE_DU_DP, COMPUTE_DU_DX || COMPUTE_DU_DP>(static_cast<Real>(1), static_cast<Real>(1), qi, qj, sig_i, sig_j, eps_i, eps_j, dd2, beta, o, es_tab);
#ifdef TM_TIMING_BATCH
                asm volatile("" : "+v"(o.prefactor));
                TM_TB(3); // (f32: analytic erfc / exp / switch, Lennard-Jones, prefactor)
#endif
                if constexpr (COMPUTE_DU_DX) {
                    u64 fx, fy, fz;
                    // (f32: one range test on the prefactor instead of three on the products: 3204 -> 3241 ns/day; same bits)
                    pair_force_fixed_bounded(o.prefactor, ddx, ddy, ddz, static_cast<Real>(ps_limit * (1.0 / 68719476736.0)), fx, fy, fz);
#ifdef TM_TIMING_BATCH
                    asm volatile("" : "+v"(fx), "+v"(fy), "+v"(fz));
                    TM_TB(4);
#endif
                    lds_add(&s_fi[0][pi], fx);
                    lds_add(&s_fi[1][pi], fy);
                    lds_add(&s_fi[2][pi], fz);
                    lds_sub(&s_fj[0][pj], fx); // FIX(-p d) == -FIX(p d)
                    lds_sub(&s_fj[1][pj], fy);
                    lds_sub(&s_fj[2][pj], fz);
                    TM_TB(5);
                }
                if constexpr (COMPUTE_DU_DP) {
                    // The integers of float_to_fixed_exp() value by value (fixed_point.hip.hpp: scale in Real, widen, round half even),
                    // with ONE range test and one wave-uniform escape for all of them instead of one per value (round 6: six ballots
                    // and branch pairs per batch fewer); lanes without a Lennard-Jones term carry zeros through it.
                    const double x_qi = static_cast<double>((qj * o.inv_dij * o.ebd) * static_cast<Real>(TM_FIXED_EXPONENT_DU_DCHARGE));
                    const double x_qj = static_cast<double>((qi * o.inv_dij * o.ebd) * static_cast<Real>(TM_FIXED_EXPONENT_DU_DCHARGE));
                    const double x_sg = o.has_lj ? static_cast<double>(o.sig_grad * static_cast<Real>(TM_FIXED_EXPONENT_DU_DSIG)) : 0.0;
                    const double x_ei = o.has_lj ? static_cast<double>((o.eps_grad * eps_j) * static_cast<Real>(TM_FIXED_EXPONENT_DU_DEPS)) : 0.0;
                    const double x_ej = o.has_lj ? static_cast<double>((o.eps_grad * eps_i) * static_cast<Real>(TM_FIXED_EXPONENT_DU_DEPS)) : 0.0;
                    // flat items: w_i - w_j == +0, so the w gradient is an exact zero for every finite prefactor (and adding zero is
                    // no operation) -- only a non-finite prefactor (0 * inf) keeps the general form, behind the same escape
                    const double x_w = FLAT ? (static_cast<double>(o.prefactor) - static_cast<double>(o.prefactor)) // 0, or NaN for inf / NaN
                                            : static_cast<double>((o.prefactor * ddw) * static_cast<Real>(TM_FIXED_EXPONENT_DU_DW));
                    long long r_qi = real_to_int64_fast(x_qi), r_qj = real_to_int64_fast(x_qj), r_sg = real_to_int64_fast(x_sg);
                    long long r_ei = real_to_int64_fast(x_ei), r_ej = real_to_int64_fast(x_ej), r_w = real_to_int64_fast(x_w);
                    const double lim = TM_FIXED_FAST_LIMIT;
                    const bool big = !(__builtin_fabs(x_qi) < lim && __builtin_fabs(x_qj) < lim && __builtin_fabs(x_sg) < lim && __builtin_fabs(x_ei) < lim &&
                                       __builtin_fabs(x_ej) < lim && __builtin_fabs(x_w) < lim); // (true for NaN)
                    bool w_general = !FLAT;
                    if (__builtin_expect(__ballot(big) != 0ull, 0)) {
                        if (big) {
                            r_qi = llrint(x_qi);
                            r_qj = llrint(x_qj);
                            r_sg = llrint(x_sg);
                            r_ei = llrint(x_ei);
                            r_ej = llrint(x_ej);
                            r_w = tm_llrint_odd(FLAT ? static_cast<double>((o.prefactor * ddw) * static_cast<Real>(TM_FIXED_EXPONENT_DU_DW)) : x_w);
                        } else if (FLAT) {
                            r_w = 0;
                        }
                        w_general = true;
                    }
                    lds_add(&s_pi[0][pi], static_cast<u64>(r_qi));
                    lds_add(&s_pj[0][pj], static_cast<u64>(r_qj));
                    if (o.has_lj) {
                        lds_add(&s_pi[1][pi], static_cast<u64>(r_sg));
                        lds_add(&s_pj[1][pj], static_cast<u64>(r_sg));
                        lds_add(&s_pi[2][pi], static_cast<u64>(r_ei));
                        lds_add(&s_pj[2][pj], static_cast<u64>(r_ej));
                    }
                    if (w_general) { // (wave-uniform: not a flat item, or the rare escape)
                        lds_add(&s_pi[3][pi], static_cast<u64>(r_w));
                        lds_sub(&s_pj[3][pj], static_cast<u64>(r_w));
                    }
                }
                if constexpr (COMPUTE_U) {
                    energy += float_to_fixed_energy_hot<Real>(o.u);
                }
                } // forces only / everything else
                } // exact cutoff test
            }
#ifdef TM_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        };
        // ---- one group: four rounds of phase 1, then phase 2 on every full batch of 64 queued pairs (LAST: on everything left).
        // FAST: flat Gram items off the diagonal -- almost every item of a production system.
        auto group = [&](const int r0, auto fast_tag, auto last_tag) {
            constexpr bool FAST = decltype(fast_tag)::value, LAST = decltype(last_tag)::value;
            const unsigned int e0 = (static_cast<unsigned int>(r0) << 11) | static_cast<unsigned int>(lane);
            if constexpr (FAST) {
                u64 m0, m1, m2, m3;
                filter4_gram_flat(rx, ry, rz, rq, c2x, c2y, c2z, thr, m0, m1, m2, m3);
                compact4(m0, m1, m2, m3, e0, qaddr);
            } else {
                // the general path: Gram form with the w term, or the explicit form where the Gram form's bounds do not hold;
                // then the row < col test on diagonal tiles (all asm statements for the registers' sake, see above; four
                // straight-line combinations: masks that merge behind a branch are no longer scalar as far as the compiler
                // can tell)
                u64 m0, m1, m2, m3;
                if (gram) {
                    if (needs_order) {
                        filter4_gram_w(rx, ry, rz, rw, rq, c2x, c2y, c2z, c2w, thr, m0, m1, m2, m3);
                        order4(lane, r0, half_bit, jrel, m0, m1, m2, m3);
                        compact4(m0, m1, m2, m3, e0, qaddr);
                    } else {
                        filter4_gram_w(rx, ry, rz, rw, rq, c2x, c2y, c2z, c2w, thr, m0, m1, m2, m3);
                        compact4(m0, m1, m2, m3, e0, qaddr);
                    }
                } else {
                    const auto uni = [](const float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
                    if (needs_order) {
                        filter4_explicit(rx, ry, rz, rw, -0.5f * c2x, -0.5f * c2y, -0.5f * c2z, -0.5f * c2w, uni(fbx), uni(fby), uni(fbz), uni(fibx), uni(fiby), uni(fibz), fcut2, m0, m1, m2, m3);
                        order4(lane, r0, half_bit, jrel, m0, m1, m2, m3);
                        compact4(m0, m1, m2, m3, e0, qaddr);
                    } else {
                        filter4_explicit(rx, ry, rz, rw, -0.5f * c2x, -0.5f * c2y, -0.5f * c2z, -0.5f * c2w, uni(fbx), uni(fby), uni(fbz), uni(fibx), uni(fiby), uni(fibz), fcut2, m0, m1, m2, m3);
                        compact4(m0, m1, m2, m3, e0, qaddr);
                    }
                }
                // (an asm result that merges behind a branch counts as divergent: say that it is not)
                qaddr = static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(qaddr)));
            }
            // the next group reads the rows rotated by four more
            rx = row_ror<4>(rx);
            ry = row_ror<4>(ry);
            rz = row_ror<4>(rz);
            rq = row_ror<4>(rq);
            if constexpr (!FAST) {
                rw = row_ror<4>(rw);
            }
#if defined(TM_ABLATE) && TM_ABLATE == 1
            qaddr = qbase; // ablation: no phase 2 at all
#endif
            // ---- phase 2: drain full batches (and everything after the last rounds).  A batch is the LAST 64 entries of the
            // queue; only the last group's drain meets batches that are not full (lanes masked).
            while (LAST ? qaddr > qbase : qaddr >= qbase + 2 * NB_CHUNK) {
                TM_T(t_h0);
                unsigned int qhead = qaddr - 2 * NB_CHUNK;
                bool active = true;
                if constexpr (LAST) {
                    // scalar on purpose: the compiler's own form of this is a VALU clamp + readfirstlane per batch
                    // (the queue sits kilobytes into the wave's LDS block: qaddr - 128 cannot wrap)
                    asm("s_max_u32 %0, %1, %2" : "=s"(qhead) : "s"(qaddr - 2 * NB_CHUNK), "s"(qbase) : "scc");
                    active = static_cast<unsigned int>(lane) < ((qaddr - qhead) >> 1);
                }
                wave_lds_sync();
                pair_batch(active, qhead + 2 * static_cast<unsigned int>(lane), std::integral_constant<bool, FAST>{});
                qaddr = qhead;
#ifdef TM_TIMING
                wave_lds_sync();
                tm_p2_item += clock64() - t_h0;
                tm_batches++;
#endif
            }
        };
        // A ticket's rounds [r_begin, r_end) in segments that stay inside one half; the ticket's last group is peeled off: the
        // next item is drawn in front of it (the later a wave draws, the better the pool is balanced when it runs dry; the last
        // group still hides the descriptor load -- measured in round 2, f64 / f32 ns/day: drawn at round 0: 2145 / 2900;
        // 16: 2175 / 2905; 24: 2190 / 2897; 28: 2207 / 2925), and its drain loop is the one that empties the queue.
        auto run_rounds = [&](auto fast_tag) {
            for (int seg = r_begin; seg < r_end;) {
                const int half_end = (seg & ~15) + 16;
                const int seg_end = r_end < half_end ? r_end : half_end;
                load_half(seg);
                const int loop_end = seg_end == r_end ? r_end - 4 : seg_end;
                for (int r0 = seg; r0 < loop_end; r0 += 4) {
                    group(r0, fast_tag, std::false_type{});
                }
                seg = seg_end;
            }
            // ---- stage A: draw the next item, request its descriptor
            item_next = position_to_slot(next_position());
            sub_next = sub_drawn;
            have_next = item_next != NO_ITEM;
            if (have_next) {
                it_next = items[item_next];
            }
            group(r_end - 4, fast_tag, std::true_type{});
        };
        if (hint<F64>(fast, true)) {
            run_rounds(std::true_type{});
        } else {
            run_rounds(std::false_type{});
        }
        // ---- stages B + C: the next item's indices, then its atom records, enter the memory queue ahead of this item's
        // flush atomics (returns are in order per wave: a load issued behind the atomics could not be observed before
        // every one of them has been acknowledged by the memory side)
        TM_T(t_bc);
        if (have_next) {
            load_indices(it_next, nxt);
            load_records(nxt, it_next.w);
        }
        wave_lds_sync();
        TM_T(t_c);

        // ---- flush: one global atomic per touched (atom, component)
#if defined(TM_ABLATE) && TM_ABLATE == 3
        if (false) // ablation: no global flush
#endif
        // The accumulators are component-major: a flush instruction's 64 lanes then touch 8 cache lines (8 consecutive u64
        // each) instead of 24 with (x, y, z) interleaved per atom.  Global atomics are executed at the memory side one
        // cache-line request at a time -- 3.5 M of them per launch took 66 us back to back interleaved, 28 us component-major
        // (scripts/microbench/atomic_scope.hip); scattered atoms: 165 us.
        if constexpr (COMPUTE_DU_DX) {
            for (int t = lane; t < TILE * 3; t += 64) {
                const int c = t / TILE, a = t - c * TILE;
                const u64 v = s_fi[c][a];
                const unsigned int ra = s_rowatom[a];
                if (v != 0 && ra < uK) {
                    atomicAdd(g_du_dx + static_cast<size_t>(c) * acc_stride + ra, v);
                }
            }
            if (ja < uK) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const u64 v = s_fj[c][lane];
                    if (v != 0) {
                        atomicAdd(g_du_dx + static_cast<size_t>(c) * acc_stride + ja, v);
                    }
                }
            }
        }
        if constexpr (COMPUTE_DU_DP) {
            for (int t = lane; t < TILE * 4; t += 64) {
                const int c = t / TILE, a = t - c * TILE;
                const u64 v = s_pi[c][a];
                const unsigned int ra = s_rowatom[a];
                if (v != 0 && ra < uK) {
                    atomicAdd(g_du_dp + static_cast<size_t>(c) * acc_stride + ra, v);
                }
            }
            if (ja < uK) {
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const u64 v = s_pj[c][lane];
                    if (v != 0) {
                        atomicAdd(g_du_dp + static_cast<size_t>(c) * acc_stride + ja, v);
                    }
                }
            }
        }
#ifdef TM_TIMING
        {
            const long long t_d = clock64();
            tm_setup += t_b - t_a;
            tm_p2 += tm_p2_item;
            tm_p1 += (t_bc - t_b) - tm_p2_item;
            tm_flush += t_d - t_c;
            tm_bc += t_c - t_bc;
            tm_stage_a += t_a2 - t_a;
            tm_items++;
        }
#endif
        cur = nxt;
        item = item_next;
        sub_cur = sub_next;
    }
#ifdef TM_TIMING_BATCH
    if (lane == 0 && timing) { // (a TM_TIMING_BATCH build reports the batch timeline INSTEAD of the phase counters)
        long long *t = timing + static_cast<size_t>(global_wave) * 8;
        for (int k = 0; k < 6; k++) {
            t[k] = tm_tb[k];
        }
        t[6] = 1; // (marks the record as filled: the readers test column 6)
        t[7] = tm_tb_n;
    }
#elif defined(TM_TIMING)
    if (lane == 0 && timing) {
        long long *t = timing + static_cast<size_t>(global_wave) * 8;
#ifdef TM_TIMING_PRO
        // prologue view: cycles to [first fetch issued | table copy issued | barrier passed | item loop entered]
        t[0] = (tp_fetch - tm_begin) | ((tp_copy - tm_begin) << 16) | ((tp_barrier - tm_begin) << 32) | ((tp_loop - tm_begin) << 48);
#else
        t[0] = tm_setup;
#endif
        t[1] = tm_p1;
        t[2] = tm_p2;
        t[3] = tm_flush | (static_cast<long long>(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf) << 56) | (static_cast<long long>(__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffff) << 40);
        t[4] = tm_items | (tm_bc << 20);      // low 20 bits: items; the rest: cycles in stages B + C
        t[5] = tm_batches | (tm_stage_a << 20); // low 20 bits: batches; the rest: cycles in stage A (up to the barrier)
        t[6] = clock64() - tm_begin;
        t[7] = tm_real_begin | (static_cast<long long>(__builtin_amdgcn_s_memrealtime()) << 32); // 100 MHz, device-wide
    }
#endif

    if constexpr (COMPUTE_U) {
        // one partial sum per WORKGROUP (round 5; one per wave before): whoever adds them up -- k_reduce_i128[_sources], the
        // barostat's decision kernel, which every one of its workgroups does for itself -- reads 256-512 values, not 4096.
        // (every wave of the workgroup gets here: the item loop has no early exit)
        const i128 total = wave_sum_i128(energy);
        if (lane == 0) {
            s_energy[wave] = total;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            i128 sum = 0;
#pragma unroll
            for (int w = 0; w < WAVES; w++) {
                sum += s_energy[w];
            }
            u_partials[blockIdx.x] = sum;
        }
        if constexpr (DUAL) {
            const i128 total2 = wave_sum_i128(energy2);
            if (lane == 0) {
                s_energy2[wave] = total2;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                i128 sum = 0;
#pragma unroll
                for (int w = 0; w < WAVES; w++) {
                    sum += s_energy2[w];
                }
                u_partials2[blockIdx.x] = sum;
            }
        }
    }
}

// ---- pair-list kernel (NonbondedPairList / NonbondedExclusions) ----------------------------------------------
// reference: k_nonbonded_pair_list (k_nonbonded_pair_list.cuh:18-190).  One thread per listed pair, same nb_pair().
// one listed pair; returns its (signed) energy in fixed point (0 unless want_u)
template <typename Real, bool NEGATED>
__device__ __forceinline__ i128 nonbonded_pair_list_term(
    const int pair, const double *__restrict__ coords, const double *__restrict__ params, const double *__restrict__ box,
    const int *__restrict__ pair_idxs, const double *__restrict__ scales, const double beta_d, const double cutoff_d,
    const double *__restrict__ es_table, u64 *__restrict__ du_dx, u64 *__restrict__ du_dp, const bool want_u, const ForceLayout fl) {
    i128 energy = 0;
    const NbBox<Real> bx = load_box<Real>(box);
    const int ia = pair_idxs[pair * 2 + 0], ja = pair_idxs[pair * 2 + 1];
    const Real cutoff = static_cast<Real>(cutoff_d);
    const Real cutoff2 = cutoff * cutoff;
    const Real dx = min_image(static_cast<Real>(coords[ia * 3 + 0]) - static_cast<Real>(coords[ja * 3 + 0]), bx.x, bx.inv_x);
    const Real dy = min_image(static_cast<Real>(coords[ia * 3 + 1]) - static_cast<Real>(coords[ja * 3 + 1]), bx.y, bx.inv_y);
    const Real dz = min_image(static_cast<Real>(coords[ia * 3 + 2]) - static_cast<Real>(coords[ja * 3 + 2]), bx.z, bx.inv_z);
    const Real qi = static_cast<Real>(params[ia * 4 + 0]), qj = static_cast<Real>(params[ja * 4 + 0]);
    const Real sig_i = static_cast<Real>(params[ia * 4 + 1]), sig_j = static_cast<Real>(params[ja * 4 + 1]);
    const Real eps_i = static_cast<Real>(params[ia * 4 + 2]), eps_j = static_cast<Real>(params[ja * 4 + 2]);
    const Real dw = static_cast<Real>(params[ia * 4 + 3]) - static_cast<Real>(params[ja * 4 + 3]);
    const Real d2 = pair_d2(dx, dy, dz, dw);
    if (d2 < cutoff2) {
        const Real charge_scale = static_cast<Real>(scales[pair * 2 + 0]);
        const Real lj_scale = static_cast<Real>(scales[pair * 2 + 1]);
        PairOut<Real> o;
        // the pair function of the tile kernel, fed from the table in global memory (12 KB: L1 / L2 resident)
        using Tab = std::conditional_t<sizeof(Real) == 8, EsTableGlobal, EsTableNone>;
        Tab tab{};
        if constexpr (sizeof(Real) == 8) {
            tab.tab = es_table;
        }
        if (du_dp != nullptr || want_u) { // uniform across the launch
            nb_pair<true>(charge_scale, lj_scale, qi, qj, sig_i, sig_j, eps_i, eps_j, d2, static_cast<Real>(beta_d), o, tab);
        } else {
            nb_pair<false>(charge_scale, lj_scale, qi, qj, sig_i, sig_j, eps_i, eps_j, d2, static_cast<Real>(beta_d), o, tab);
        }
#if defined(TM_ABLATE) && TM_ABLATE == 5
#define TM_ACC_GATE(v) ((v) == 0x123456789abcull) // ablation (timing only): terms computed, atomics not issued
#else
#define TM_ACC_GATE(v) true
#endif
#define TM_ACC(ptr, val)                                                                                               \
do {                                                                                                               \
    const u64 v_ = (val);                                                                                          \
    if (TM_ACC_GATE(v_)) {                                                                                         \
        atomicAdd((ptr), NEGATED ? (0ull - v_) : v_);                                                              \
    }                                                                                                              \
} while (0)
        if (du_dx) {
            u64 fx, fy, fz;
            pair_force_fixed(o.prefactor, dx, dy, dz, fx, fy, fz);
            // (through force_add: inside a fused slice the pair's atoms usually sit in the wave's LDS window)
            force_add(du_dx, fl, ia, 0, NEGATED ? 0ull - fx : fx);
            force_add(du_dx, fl, ia, 1, NEGATED ? 0ull - fy : fy);
            force_add(du_dx, fl, ia, 2, NEGATED ? 0ull - fz : fz);
            force_add(du_dx, fl, ja, 0, NEGATED ? fx : 0ull - fx);
            force_add(du_dx, fl, ja, 1, NEGATED ? fy : 0ull - fy);
            force_add(du_dx, fl, ja, 2, NEGATED ? fz : 0ull - fz);
        }
        if (du_dp) {
            TM_ACC(du_dp + ia * 4 + 0, (float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DCHARGE>(charge_scale * qj * o.inv_dij * o.ebd)));
            TM_ACC(du_dp + ja * 4 + 0, (float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DCHARGE>(charge_scale * qi * o.inv_dij * o.ebd)));
            if (o.has_lj) {
                const u64 sg = float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DSIG>(o.sig_grad);
                TM_ACC(du_dp + ia * 4 + 1, sg);
                TM_ACC(du_dp + ja * 4 + 1, sg);
                TM_ACC(du_dp + ia * 4 + 2, (float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DEPS>(o.eps_grad * eps_j)));
                TM_ACC(du_dp + ja * 4 + 2, (float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DEPS>(o.eps_grad * eps_i)));
            }
            const u64 gw = float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DW>(o.prefactor * dw);
            TM_ACC(du_dp + ia * 4 + 3, gw);
            TM_ACC(du_dp + ja * 4 + 3, 0ull - gw);
        }
#undef TM_ACC
        if (want_u) {
            // negate the fixed-point value, not the float (k_nonbonded_pair_list.cuh:185-188)
            const i128 e = float_to_fixed_energy<Real>(o.u);
            energy = NEGATED ? -e : e;
        }
    }

    return energy;
}

template <typename Real, bool NEGATED>
__global__ __launch_bounds__(256) void k_nonbonded_pair_list(
    const int M, const double *__restrict__ coords, const double *__restrict__ params, const double *__restrict__ box,
    const int *__restrict__ pair_idxs, const double *__restrict__ scales, const double beta_d, const double cutoff_d,
    const double *__restrict__ es_table, u64 *__restrict__ du_dx, u64 *__restrict__ du_dp, i128 *__restrict__ u_partials) {
    const int pair = blockIdx.x * blockDim.x + threadIdx.x;
    i128 energy = 0;
    if (pair < M) {
        energy = nonbonded_pair_list_term<Real, NEGATED>(
            pair, coords, params, box, pair_idxs, scales, beta_d, cutoff_d, es_table, du_dx, du_dp, u_partials != nullptr);
    }
    if (u_partials) {
        // per-wave partial sums instead of one 16-byte store per pair
        const i128 total = wave_sum_i128(energy);
        if ((threadIdx.x & 63) == 0) {
            u_partials[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = total;
        }
    }
}

// ---- pair list with precomputed pair parameters (NonbondedPairListPrecomputed) ---------------------------------
// reference: k_nonbonded_precomputed (k_nonbonded_precomputed.cuh:11-186); JAX: nonbonded_on_precomputed_pairs
// (potentials/nonbonded.py:403-446).  params[pair] = (q_ij, sig_ij, eps_ij, w_offset_ij): combining rules and scale
// factors already applied.  Note: the reference kernel only issues its force / du_dp atomics inside the LJ branch,
// silently dropping pairs with eps_ij == 0 or sig_ij == 0; this follows the JAX definition instead (every pair counts).
template <typename Real>
__device__ __forceinline__ i128 nonbonded_precomputed_term(
    const int pair, const double *__restrict__ coords, const double *__restrict__ params, const double *__restrict__ box,
    const int *__restrict__ pair_idxs, const double beta_d, const double cutoff_d, u64 *__restrict__ du_dx,
    u64 *__restrict__ du_dp, const bool want_u, const ForceLayout fl = ForceLayout{3, 1}) {
    i128 energy = 0;
    const NbBox<Real> bx = load_box<Real>(box);
    const int ia = pair_idxs[pair * 2 + 0], ja = pair_idxs[pair * 2 + 1];
    const Real cutoff = static_cast<Real>(cutoff_d);
    const Real q_ij = static_cast<Real>(params[pair * 4 + 0]);
    const Real sig_ij = static_cast<Real>(params[pair * 4 + 1]);
    const Real eps_ij = static_cast<Real>(params[pair * 4 + 2]);
    const Real dw = static_cast<Real>(params[pair * 4 + 3]);
    const Real dx = min_image(static_cast<Real>(coords[ia * 3 + 0]) - static_cast<Real>(coords[ja * 3 + 0]), bx.x, bx.inv_x);
    const Real dy = min_image(static_cast<Real>(coords[ia * 3 + 1]) - static_cast<Real>(coords[ja * 3 + 1]), bx.y, bx.inv_y);
    const Real dz = min_image(static_cast<Real>(coords[ia * 3 + 2]) - static_cast<Real>(coords[ja * 3 + 2]), bx.z, bx.inv_z);
    const Real d2 = pair_d2(dx, dy, dz, dw);
    if (d2 < cutoff * cutoff) {
        const Real inv = tm_rsqrt(d2);
        const Real d = d2 * inv;
        const Real inv2 = inv * inv;
        Real prefactor = 0, u = 0, g_q = 0, g_sig = 0, g_eps = 0;
        if (q_ij != 0) {
            Real damping;
            const Real es_factor = real_es_factor(static_cast<Real>(beta_d), d, inv, inv2, damping);
            u += q_ij * inv * damping;
            prefactor += q_ij * inv * es_factor;
            g_q = damping * inv;
        }
        if (eps_ij != 0 && sig_ij != 0) {
            const Real s = sig_ij * inv;
            const Real s2 = s * s;
            const Real s6 = s2 * s2 * s2;
            g_eps = 4 * (s6 - 1) * s6;
            u += eps_ij * g_eps;
            const Real w24 = 24 * eps_ij * s6 * (2 * s6 - 1);
            prefactor -= w24 * inv2;
            g_sig = w24 / sig_ij;
        }
        if (du_dx) {
            u64 fx, fy, fz;
            pair_force_fixed(prefactor, dx, dy, dz, fx, fy, fz);
            force_add(du_dx, fl, ia, 0, fx);
            force_add(du_dx, fl, ia, 1, fy);
            force_add(du_dx, fl, ia, 2, fz);
            force_add(du_dx, fl, ja, 0, 0ull - fx);
            force_add(du_dx, fl, ja, 1, 0ull - fy);
            force_add(du_dx, fl, ja, 2, 0ull - fz);
        }
        if (du_dp) {
            atomicAdd(du_dp + pair * 4 + 0, (float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DCHARGE>(g_q)));
            atomicAdd(du_dp + pair * 4 + 1, (float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DSIG>(g_sig)));
            atomicAdd(du_dp + pair * 4 + 2, (float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DEPS>(g_eps)));
            atomicAdd(du_dp + pair * 4 + 3, (float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DW>(prefactor * dw)));
        }
        if (want_u) {
            energy = float_to_fixed_energy<Real>(u);
        }
    }
    return energy;
}

template <typename Real>
__global__ __launch_bounds__(256) void k_nonbonded_precomputed(
    const int M, const double *__restrict__ coords, const double *__restrict__ params, const double *__restrict__ box,
    const int *__restrict__ pair_idxs, const double beta_d, const double cutoff_d, u64 *__restrict__ du_dx,
    u64 *__restrict__ du_dp, i128 *__restrict__ u_partials) {
    const int pair = blockIdx.x * blockDim.x + threadIdx.x;
    i128 energy = 0;
    if (pair < M) {
        energy = nonbonded_precomputed_term<Real>(pair, coords, params, box, pair_idxs, beta_d, cutoff_d, du_dx, du_dp, u_partials != nullptr);
    }
    if (u_partials) {
        const i128 total = wave_sum_i128(energy);
        if ((threadIdx.x & 63) == 0) {
            u_partials[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = total;
        }
    }
}

template <typename Real>
__device__ __attribute__((noinline)) void fused_dispatch(
    const FusedTable *__restrict__ table, const int block, const int thread, const double *__restrict__ coords,
    const double *__restrict__ box, u64 *__restrict__ du_dx, ForceLayout fl, lds_u64_ptr window, lds_int_ptr window_rows) {
    const int n = table->n;
    int s = 0, first = 0;
    for (int k = 0; k + 1 < n; k++) { // wave-uniform: scalar loads
        const int end = table->block_end[k];
        if (block >= end) {
            s = k + 1;
            first = end;
        }
    }
    const FusedSegment seg = table->seg[s];
    const int idx = (block - first) * 256 + thread;
    const int lane = thread & 63;
    const int idx0 = idx - lane; // the slice's first term (wave-uniform)
    if (idx0 >= seg.count) {
        return; // the whole wave: nothing left of this segment
    }
    // The slice's LDS window starts a few atoms below the first atom of its first term: term lists come in atom order
    // (molecule by molecule), so the atoms of 64 consecutive terms nearly always fall inside; whatever does not takes the
    // global atomic as before.
    if (table->window == 0 || seg.window == 0) {
        window = nullptr; // sparse term lists (ForcePlan::add_segment), or the A/B switch TM_AMD_NO_FUSED_WINDOW: straight to the global accumulator
    }
    if (window != nullptr) {
        const int width = seg.kind == FUSED_ANGLE ? 3 : ((seg.kind == FUSED_TORSION || seg.kind == FUSED_CHIRAL_ATOM || seg.kind == FUSED_CHIRAL_BOND) ? 4 : 2);
        const int a0 = __builtin_amdgcn_readfirstlane(seg.idxs[static_cast<size_t>(idx0) * width]);
        fl.win = window;
        fl.win_base = a0 > 8 ? a0 - 8 : 0;
        for (int t = lane; t < 3 * FORCE_WINDOW; t += 64) {
            window[t] = 0;
        }
        // the window atoms' accumulator rows (a dependent global load when the accumulator is remapped): requested now, needed
        // only by the flush -- the terms compute underneath.  (Fetched in the flush itself, the window cost small systems 2 us
        // per step: a lone wave's second dependent load stage.)  Atoms past the end of the system never receive a force.
        if (fl.remap != nullptr) {
            for (int t = lane; t < FORCE_WINDOW; t += 64) {
                const int a = fl.win_base + t;
                window_rows[t] = a < table->num_atoms ? fl.remap[a] : 0;
            }
        }
        wave_lds_sync();
    }
    if (idx < seg.count) {
    switch (seg.kind) {
    case FUSED_BOND: harmonic_bond_term<Real>(idx, coords, seg.params, seg.idxs, du_dx, nullptr, false, fl); break;
    case FUSED_ANGLE: harmonic_angle_term<Real>(idx, coords, seg.params, seg.idxs, du_dx, nullptr, false, fl); break;
    case FUSED_TORSION: periodic_torsion_term<Real>(idx, coords, seg.params, seg.idxs, du_dx, nullptr, false, fl); break;
    case FUSED_PAIR_LIST:
        nonbonded_pair_list_term<Real, false>(idx, coords, seg.params, box, seg.idxs, seg.scales, seg.beta, seg.cutoff, seg.es_table, du_dx, nullptr, false, fl);
        break;
    case FUSED_PAIR_LIST_NEGATED:
        nonbonded_pair_list_term<Real, true>(idx, coords, seg.params, box, seg.idxs, seg.scales, seg.beta, seg.cutoff, seg.es_table, du_dx, nullptr, false, fl);
        break;
    case FUSED_PAIR_LIST_PRECOMPUTED:
        nonbonded_precomputed_term<Real>(idx, coords, seg.params, box, seg.idxs, seg.beta, seg.cutoff, du_dx, nullptr, false, fl);
        break;
    case FUSED_CHIRAL_ATOM: chiral_atom_term<Real>(idx, coords, seg.params, seg.idxs, du_dx, nullptr, false, fl); break;
    case FUSED_CHIRAL_BOND: chiral_bond_term<Real>(idx, coords, seg.params, seg.idxs, seg.aux, du_dx, nullptr, false, fl); break;
    case FUSED_FLAT_BOTTOM_BOND: flat_bottom_bond_term<Real, false>(idx, coords, box, seg.params, seg.idxs, seg.beta, du_dx, nullptr, false, fl); break;
    case FUSED_LOG_FLAT_BOTTOM_BOND: flat_bottom_bond_term<Real, true>(idx, coords, box, seg.params, seg.idxs, seg.beta, du_dx, nullptr, false, fl); break;
    default: break;
    }
    }
    if (window != nullptr) {
        // flush: one global atomic per touched (atom, component) of the slice
        wave_lds_sync();
        for (int t = lane; t < 3 * FORCE_WINDOW; t += 64) {
            const u64 v = window[t];
            if (v != 0) {
                const int d = t / FORCE_WINDOW, off = t - d * FORCE_WINDOW;
                const size_t row = fl.remap != nullptr ? static_cast<size_t>(window_rows[off]) * fl.atom : static_cast<size_t>(fl.win_base + off) * fl.atom;
                atomicAdd(du_dx + row + static_cast<size_t>(d) * fl.comp, v);
            }
        }
        wave_lds_sync();
    }
}

// The energy-only twin of fused_dispatch: same table, same per-term device functions with no force outputs asked for;
// returns the term's fixed-point energy (0 for threads past a segment's end).
template <typename Real>
__device__ __forceinline__ i128 fused_dispatch_energy(
    const FusedTable *__restrict__ table, const int block, const int thread, const double *__restrict__ coords,
    const double *__restrict__ box) {
    const int n = table->n;
    int s = 0, first = 0;
    for (int k = 0; k + 1 < n; k++) { // wave-uniform: scalar loads
        const int end = table->block_end[k];
        if (block >= end) {
            s = k + 1;
            first = end;
        }
    }
    const FusedSegment seg = table->seg[s];
    const int idx = (block - first) * 256 + thread;
    if (idx >= seg.count) {
        return 0;
    }
    switch (seg.kind) {
    case FUSED_BOND: return harmonic_bond_term<Real>(idx, coords, seg.params, seg.idxs, nullptr, nullptr, true);
    case FUSED_ANGLE: return harmonic_angle_term<Real>(idx, coords, seg.params, seg.idxs, nullptr, nullptr, true);
    case FUSED_TORSION: return periodic_torsion_term<Real>(idx, coords, seg.params, seg.idxs, nullptr, nullptr, true);
    case FUSED_PAIR_LIST:
        return nonbonded_pair_list_term<Real, false>(idx, coords, seg.params, box, seg.idxs, seg.scales, seg.beta, seg.cutoff, seg.es_table, nullptr, nullptr, true);
    case FUSED_PAIR_LIST_NEGATED:
        return nonbonded_pair_list_term<Real, true>(idx, coords, seg.params, box, seg.idxs, seg.scales, seg.beta, seg.cutoff, seg.es_table, nullptr, nullptr, true);
    case FUSED_PAIR_LIST_PRECOMPUTED:
        return nonbonded_precomputed_term<Real>(idx, coords, seg.params, box, seg.idxs, seg.beta, seg.cutoff, nullptr, nullptr, true);
    case FUSED_CHIRAL_ATOM: return chiral_atom_term<Real>(idx, coords, seg.params, seg.idxs, nullptr, nullptr, true);
    case FUSED_CHIRAL_BOND: return chiral_bond_term<Real>(idx, coords, seg.params, seg.idxs, seg.aux, nullptr, nullptr, true);
    case FUSED_FLAT_BOTTOM_BOND: return flat_bottom_bond_term<Real, false>(idx, coords, box, seg.params, seg.idxs, seg.beta, nullptr, nullptr, true);
    case FUSED_LOG_FLAT_BOTTOM_BOND: return flat_bottom_bond_term<Real, true>(idx, coords, box, seg.params, seg.idxs, seg.beta, nullptr, nullptr, true);
    default: return 0;
    }
}


} // namespace tmamd
