// The MD form of the nonbonded tile kernel (forces only): one WORKGROUP per "unit" = (32-row block, up to RB_UCAP of its listed
// columns), lane-owned columns, no pair queue.  Included by nonbonded.hip only; every other call form (energies, du/dp, small
// systems) keeps k_nonbonded_tiles (kernels_nonbonded.hip.hpp), and integer accumulation makes the two interchangeable bit for
// bit.  reference being replaced: cpp/src/kernels/k_nonbonded.cuh:109-327 (k_nonbonded_unified).
//
// Why a second decomposition (costed on the CPU first, scripts/decomp_stats.py; EXPERIMENTS.md round 4): the wave-per-item
// kernel spends about as many instructions on FINDING pairs (32 filter rounds + queue compaction per 2048 slots, of which 30 %
// hit) and on moving operands (14 LDS reads + 6 LDS atomics per pair) as on the pair function.  Here
//   phase 1  every lane owns one listed column atom; the unit's 32 row atoms sit in two register quadruples replicated in every
//            16-lane DPP row and round k reads row k through `row_newbcast:k` of the arithmetic itself -- ALL 64 lanes test the
//            same row, the hit bit goes straight into the lane's 32-bit row mask through the carry of v_addc (m = 2 m + hit):
//            6 vector instructions per 64 slots, no compaction, no queue;
//   sort     lane-owned columns only pay if the lanes of a wave hold similar numbers of hits (in list order mean / max hits per
//            64 columns is 0.40; sorted by hit count over ~1000 columns 0.9): a counting sort by popcount through LDS
//            (33 bins, one returning LDS atomic per column) regroups the unit's columns into "virtual items" of 64 columns
//            with nearly equal hit counts; columns without a hit (27 % of a padded list) drop out here;
//   phase 2  a wave takes a virtual item: the column's record and its three force sums stay in registers, every trip pops one
//            set bit per lane (scan start rotated by the lane number, so that the 64 row addresses of a trip stay spread),
//            fetches the ROW operands (6 single ds_read_b64, conflict free: 32 rows x 8 B = 64 banks) and runs the very
//            pair function of the item kernel (same operand order => same bits); row forces go to the unit's LDS accumulator
//            (3 ds_add_u64 per pair instead of 6), the column's sums to an LDS slot at the column's LIST position;
//   flush    per unit, in list order: the column atomics of an instruction still share ~8 cache lines (the memory side serves
//            atomics per 64-byte line), and the rows flush once per unit instead of once per 64 columns.
// Units are dealt statically: position k of workgroup b's list is unit k * G + (k even ? b : G - 1 - b) of the row-block order
// (the upper-triangular list makes that order roughly largest first).  Static, so that the NEXT unit is known while the current
// one computes: its column indices are requested a unit ahead, its row atoms are staged into a second row buffer underneath the
// current unit's pops (a unit fetched from scratch is a chain of five dependent memory hops), and a device-wide ticket would
// come back microseconds late on this memory system anyway (EXPERIMENTS.md, round 3: device-wide tickets).
#pragma once
#include "kernels_nonbonded.hip.hpp"

namespace tmamd {

#ifndef TM_RB_WAVES
#define TM_RB_WAVES 8 // waves per workgroup
#endif
#ifndef TM_RB_WGS_PER_CU
#define TM_RB_WGS_PER_CU 2
#endif
#ifndef TM_RB_UCAP
#define TM_RB_UCAP 1024 // listed columns per unit (the scope of the popcount sort)
#endif
#ifndef TM_RB_SPLIT_POP
#define TM_RB_SPLIT_POP 16 // virtual items whose heaviest column has more hits than this are dealt as two half-row pieces
#endif
static const int RB_WAVES = TM_RB_WAVES;
static const int RB_THREADS = 64 * RB_WAVES;
static const int RB_WGS_PER_CU = TM_RB_WGS_PER_CU;
static const int RB_UCAP = TM_RB_UCAP;
static const int RB_SLICES = RB_UCAP / 64;
static const int RB_SLICES_PER_WAVE = (RB_SLICES + RB_WAVES - 1) / RB_WAVES;
#ifndef TM_RB_MAX_ROW_BLOCKS
#define TM_RB_MAX_ROW_BLOCKS 2048
#endif
static const int RB_MAX_ROW_BLOCKS = TM_RB_MAX_ROW_BLOCKS; // the unit table (one prefix entry per row block) lives in LDS: K <= 65 536 atoms
static_assert(RB_UCAP % 64 == 0 && RB_UCAP <= 65536, "unit capacity");
static const int RB_FUSED_WAVES = RB_WAVES < 8 ? RB_WAVES : 8; // waves of a workgroup that run slices of a piggy-backed ForcePlan table
static_assert(RB_FUSED_WAVES * FORCE_WINDOW * 3 <= 3 * RB_UCAP && RB_FUSED_WAVES * FORCE_WINDOW <= RB_UCAP, "the fused slices' windows are carved out of the unit arrays");

// Four filter rounds on rows K3 > K2 > K1 > K0 of one register quadruple (rows 0-15 or 16-31, replicated in every 16-lane DPP
// row): a = |r_k|^2 + r_k . (-2 c), hit iff a < thr (the Gram form of the item kernel, same operation order, same error bound);
// the four hit bits enter the mask from the right, K3 first: after the eight statements of a slice bit k of m is row k.
template <int K0>
__device__ __forceinline__ void rb_filter4(
    const float rx, const float ry, const float rz, const float rr, const float c2x, const float c2y, const float c2z, const float thr,
    unsigned int &m) {
    float a0, a1, a2, a3;
    u64 s0, s1, s2, s3;
    asm volatile(
        "s_nop 1\n\t"
        "v_mov_b32_dpp %[a3], %[rr] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[a2], %[rr] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[a1], %[rr] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %[a0], %[rr] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a3], %[rx], %[c2x] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a2], %[rx], %[c2x] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a1], %[rx], %[c2x] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a0], %[rx], %[c2x] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a3], %[ry], %[c2y] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a2], %[ry], %[c2y] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a1], %[ry], %[c2y] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a0], %[ry], %[c2y] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a3], %[rz], %[c2z] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a2], %[rz], %[c2z] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a1], %[rz], %[c2z] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a0], %[rz], %[c2z] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t"
        "v_cmp_lt_f32_e64 %[s3], %[a3], %[thr]\n\t"
        "v_cmp_lt_f32_e64 %[s2], %[a2], %[thr]\n\t"
        "v_cmp_lt_f32_e64 %[s1], %[a1], %[thr]\n\t"
        "v_cmp_lt_f32_e64 %[s0], %[a0], %[thr]\n\t"
        "v_addc_co_u32_e64 %[m], %[s3], %[m], %[m], %[s3]\n\t"
        "v_addc_co_u32_e64 %[m], %[s2], %[m], %[m], %[s2]\n\t"
        "v_addc_co_u32_e64 %[m], %[s1], %[m], %[m], %[s1]\n\t"
        "v_addc_co_u32_e64 %[m], %[s0], %[m], %[m], %[s0]"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3), [m] "+v"(m)
        : [rx] "v"(rx), [ry] "v"(ry), [rz] "v"(rz), [rr] "v"(rr), [c2x] "v"(c2x), [c2y] "v"(c2y), [c2z] "v"(c2z), [thr] "v"(thr),
          [k0] "n"(K0), [k1] "n"(K0 + 1), [k2] "n"(K0 + 2), [k3] "n"(K0 + 3));
}
// lane k of the own 16-lane DPP row, in every lane (row_newbcast:K).  All 64 lanes must be enabled.
template <int K> __device__ __forceinline__ float rb_bcast(const float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + K, 0xf, 0xf, true));
}
// The general filter (columns that differ in w, extents the Gram form does not cover, sparse or small boxes): rows R .. 0 of
// the explicit form, one after the other -- differences against the broadcast row, the minimum image per component in f32, the
// squared 4-D distance against the padded cutoff.  Rare; plain C++.
struct RbGeneral {
    float cfx, cfy, cfz, cfw, bx, by, bz, ibx, iby, ibz, cut2;
};
template <int R>
__device__ __forceinline__ void rb_general_rounds(
    const float (&ra)[5], const float (&rb)[5], const RbGeneral &g, unsigned int &m) {
    constexpr int K = R & 15;
    const float rx = rb_bcast<K>(R < 16 ? ra[0] : rb[0]), ry = rb_bcast<K>(R < 16 ? ra[1] : rb[1]), rz = rb_bcast<K>(R < 16 ? ra[2] : rb[2]);
    const float rw = rb_bcast<K>(R < 16 ? ra[3] : rb[3]);
    float dx = rx - g.cfx, dy = ry - g.cfy, dz = rz - g.cfz;
    dx = __builtin_fmaf(-g.bx, __builtin_rintf(dx * g.ibx), dx);
    dy = __builtin_fmaf(-g.by, __builtin_rintf(dy * g.iby), dy);
    dz = __builtin_fmaf(-g.bz, __builtin_rintf(dz * g.ibz), dz);
    const float dw = rw - g.cfw; // invalid rows carry w = 1e18, dead columns -1e18: never inside
    const float d2 = __builtin_fmaf(dw, dw, __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)));
    m = (m << 1) | (d2 < g.cut2 ? 1u : 0u);
    if constexpr (R > 0) {
        rb_general_rounds<R - 1>(ra, rb, g, m);
    }
}

struct RbEsTable { // the workgroup's LDS copy of the force-factor table (nb_es_table.hip.hpp)
    const double *tab;
    __device__ __forceinline__ void load(const unsigned int idx, double (&c)[ES_TAB_COEFFS]) const {
        const double2 *p = reinterpret_cast<const double2 *>(tab + idx * ES_TAB_COEFFS);
        const double2 a = p[0], b = p[1], e = p[2];
        c[0] = a.x; c[1] = a.y; c[2] = b.x; c[3] = b.y; c[4] = e.x; c[5] = e.y;
    }
};

// row operands of one pair: single ds_read_b64 issued from asm (the compiler would pair them into ds_read2_b64, which the LDS
// serves at half rate), component stride 32 rows * 8 bytes
template <int C, bool SKIP_W> __device__ __forceinline__ void rb_read_row(const unsigned int addr, double (&ri)[7]) {
    if constexpr (C < 7) {
        if constexpr (!(SKIP_W && C == 3)) {
            ri[C] = lds_read_f64_async<C * TILE * 8>(addr);
        }
        rb_read_row<C + 1, SKIP_W>(addr, ri);
    }
}
template <bool SKIP_W> __device__ __forceinline__ void rb_wait_row(double (&a)[7]) {
    if constexpr (SKIP_W) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]));
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]));
    }
}

template <typename Real, bool INSIDE_SWITCH>
__global__ __launch_bounds__(RB_THREADS, (RB_WAVES * RB_WGS_PER_CU) / 4) void k_nonbonded_rowblocks(
    const int K,                               // atoms in `gathered` (record K is an all-zero sentinel)
    const int NR,                              // number of row atoms
    const int upper_triangular,                // rows == cols == all: keep only row < col
    const unsigned int *__restrict__ row_idxs, // [NR] or nullptr (identity)
    const int n_row_blocks, const int2 *__restrict__ row_segments, // per row block {start in col_atoms, listed columns}
    const unsigned int *__restrict__ col_atoms, const Real *__restrict__ gathered, const double *__restrict__ box,
    const double beta_d, const double cutoff_d, const double *__restrict__ es_table, u64 *__restrict__ g_du_dx, const int acc_stride,
    // piggy-backed ForcePlan table, as in k_nonbonded_tiles
    const FusedTable *__restrict__ fused, const int fused_blocks, const double *__restrict__ coords, u64 *__restrict__ out_du_dx,
    const int out_atom_stride, const int out_comp_stride, const int *__restrict__ out_remap,
    long long *__restrict__ timing) { // debug builds (-DTM_TIMING) only: 8 counters per workgroup
    constexpr bool F64 = sizeof(Real) == 8;
#ifdef TM_TIMING
    long long tm_acc[6] = {0, 0, 0, 0, 0, 0}, tm_units = 0, tm_trips = 0, tm_pops = 0, tm_loop = 0, tm_piece = 0, tm_wait3 = 0, tm_npieces = 0;
    __shared__ unsigned long long s_tm_counts[6];
    if (threadIdx.x < 6) {
        s_tm_counts[threadIdx.x] = 0;
    }
    const long long tm_begin = clock64();
    const long long tm_real_begin = static_cast<long long>(__builtin_amdgcn_s_memrealtime() & 0xffffffffull);
    long long tm_last = tm_begin;
#define TM_RB_STAMP(k)                                                                                                 \
    {                                                                                                                  \
        const long long now_ = clock64();                                                                              \
        tm_acc[k] += now_ - tm_last;                                                                                   \
        tm_last = now_;                                                                                                \
    }
#else
#define TM_RB_STAMP(k)
#endif
    struct RowBuf {
        Real row[7][TILE];     // row atom records: x y z w q sig eps
        float rowflt[5][TILE]; // as the filter sees them: x y z relative to the first row atom, w, |r|^2
        unsigned int rowatom[TILE];
        float rext, w0;        // largest |component| of a row in units of the Gram extent bound; the first row's w
        unsigned int flags;    // 1: something wrapped on the way to the origin's image or beyond the Gram extents; 2: w differ
        unsigned int pad_;
    };
    struct Lds {
        u64 colacc[3][RB_UCAP]; // column force sums at the columns' list positions
        u64 rowacc[3][TILE];
        RowBuf rows[2];         // the current unit's row atoms and, staged underneath its pops, the next unit's
        unsigned int sorted_ja[RB_UCAP];   // columns regrouped by hit count: atom,
        unsigned int sorted_mask[RB_UCAP]; // row mask,
        unsigned short sorted_lp[RB_UCAP]; // list position within the unit
        unsigned int hist[64], cnt[64];    // columns per hit count; rank counters of the scatter
        unsigned int prefix[RB_MAX_ROW_BLOCKS]; // inclusive prefix sum of units per row block
        unsigned int wave_tot[RB_WAVES];
        unsigned int piece_ticket;
    };
    __shared__ Lds lds;
    __shared__ __attribute__((aligned(16))) double s_es_tab[F64 ? ES_TAB_DOUBLES : 2];
    // two workgroups per CU share its 160 KB; a workgroup's static allocation is what the launch asks for (TM_TIMING adds 48 B)
    static_assert(sizeof(Lds) + (F64 ? ES_TAB_DOUBLES : 2) * sizeof(double) + 64 <= 64 * 1024, "row-block kernel: static LDS beyond 64 KB (RB_UCAP / RB_MAX_ROW_BLOCKS)");

    const int tid = static_cast<int>(threadIdx.x);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned int uK = static_cast<unsigned int>(K);
    // f64: this thread's share of the table is requested first of all and stored behind the unit table's scan
    constexpr int ES_PER_THREAD = (ES_TAB_DOUBLES + RB_THREADS - 1) / RB_THREADS;
    [[maybe_unused]] double es_mine[F64 ? ES_PER_THREAD : 1];
    if constexpr (F64) {
#pragma unroll
        for (int k = 0; k < ES_PER_THREAD; k++) {
            const int i = tid + k * RB_THREADS;
            es_mine[k] = i < ES_TAB_DOUBLES ? es_table[i] : 0.0;
        }
    }
    // ---- unit table: units per row block = ceil(listed columns / RB_UCAP), inclusive prefix sum over the row blocks
    {
        const int per = (n_row_blocks + RB_THREADS - 1) / RB_THREADS; // <= RB_MAX_ROW_BLOCKS / RB_THREADS
        const int b0 = tid * per;
        unsigned int mine = 0;
        for (int b = b0; b < b0 + per && b < n_row_blocks; b++) {
            mine += (static_cast<unsigned int>(row_segments[b].y) + RB_UCAP - 1) / RB_UCAP;
        }
        int v = static_cast<int>(mine); // inclusive wave scan on the VALU (see k_nonbonded_tiles)
        v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);
        if (lane == 63) {
            lds.wave_tot[wave] = static_cast<unsigned int>(v);
        }
        __syncthreads();
        unsigned int before = 0;
        for (int w = 0; w < wave; w++) {
            before += lds.wave_tot[w];
        }
        unsigned int run = before + static_cast<unsigned int>(v) - mine;
        for (int b = b0; b < b0 + per && b < n_row_blocks; b++) {
            run += (static_cast<unsigned int>(row_segments[b].y) + RB_UCAP - 1) / RB_UCAP;
            lds.prefix[b] = run;
        }
    }
    if constexpr (F64) {
#pragma unroll
        for (int k = 0; k < ES_PER_THREAD; k++) {
            const int i = tid + k * RB_THREADS;
            if (i < ES_TAB_DOUBLES) {
                s_es_tab[i] = es_mine[k];
            }
        }
    }
    __syncthreads();
    const unsigned int total_units = __builtin_amdgcn_readfirstlane(n_row_blocks > 0 ? lds.prefix[n_row_blocks - 1] : 0u);

    using Tab = std::conditional_t<F64, RbEsTable, EsTableNone>;
    Tab es_tab{};
    if constexpr (F64) {
        es_tab.tab = s_es_tab;
    }
    const NbBox<Real> bx = load_box<Real>(box);
    const Real cutoff = static_cast<Real>(cutoff_d);
    const Real cutoff2 = cutoff * cutoff;
    [[maybe_unused]] const double ps_limit = TM_FIXED_FAST_LIMIT / cutoff_d * 0.999999;
    [[maybe_unused]] const Real beta = static_cast<Real>(beta_d);
    const float fbx = static_cast<float>(bx.x), fby = static_cast<float>(bx.y), fbz = static_cast<float>(bx.z);
    const float fibx = 1.0f / fbx, fiby = 1.0f / fby, fibz = 1.0f / fbz;
    const float fmaxb = fmaxf(fbx, fmaxf(fby, fbz));
    const float fcut2 = static_cast<float>(cutoff_d * cutoff_d) + 1e-5f * (1.0f + fmaxb) * (1.0f + static_cast<float>(cutoff_d));
    const float fex = 1.0f / fminf(0.49f * fbx, TM_GRAM_MAX_EXTENT), fey = 1.0f / fminf(0.49f * fby, TM_GRAM_MAX_EXTENT);
    const float fez = 1.0f / fminf(0.49f * fbz, TM_GRAM_MAX_EXTENT), few = 1.0f / TM_GRAM_MAX_EXTENT;

    // ---- piggy-backed bonded terms / pair lists: 64-term slices, spread over the workgroups (as in k_nonbonded_tiles); their
    // accumulation windows are carved out of the unit arrays, which the first unit does not touch before its first barrier
    if (fused) {
        const int total_waves = static_cast<int>(gridDim.x) * RB_FUSED_WAVES;
        for (int t = wave * static_cast<int>(gridDim.x) + static_cast<int>(blockIdx.x); wave < RB_FUSED_WAVES && t < fused_blocks * 4; t += total_waves) {
            fused_dispatch<Real>(fused, t >> 2, (t & 3) * 64 + lane, coords, box, out_du_dx, ForceLayout{out_atom_stride, out_comp_stride, out_remap},
                                 (lds_u64_ptr)(&lds.colacc[0][0] + wave * (3 * FORCE_WINDOW)), (lds_int_ptr)(&lds.sorted_ja[0] + wave * FORCE_WINDOW));
        }
        __syncthreads();
    }

    // ---- units of this workgroup: position k of its list is unit k * G + (k even ? b : G - 1 - b) of the row-block order (the
    // upper-triangular list makes that order roughly largest first: the serpentine deal pairs a large unit with a small one).
    // Static on purpose: the next unit is known while the current one is computed, so its column indices, its records and its
    // row atoms are requested a phase ahead (a unit is a chain of five dependent memory hops otherwise: 11k of a unit's 52k
    // cycles), and a device-wide ticket would come back microseconds late on this memory system anyway.
    struct UnitDesc {
        int rb;
        unsigned int ucount, col_start, row_first;
    };
    auto unit_of = [&](const unsigned int k) -> unsigned int { return k * gridDim.x + ((k & 1u) ? gridDim.x - 1u - blockIdx.x : blockIdx.x); };
    auto decode = [&](const unsigned int unit) -> UnitDesc {
        int lo = 0, hi = n_row_blocks - 1; // first row block whose prefix is beyond the unit (uniform binary search in LDS)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (lds.prefix[mid] > unit) {
                hi = mid;
            } else {
                lo = mid + 1;
            }
        }
        UnitDesc d;
        d.rb = __builtin_amdgcn_readfirstlane(lo);
        const unsigned int part = __builtin_amdgcn_readfirstlane(unit - (d.rb > 0 ? lds.prefix[d.rb - 1] : 0u));
        const int2 seg = row_segments[d.rb];
        const unsigned int count = static_cast<unsigned int>(seg.y);
        const unsigned int parts = (count + RB_UCAP - 1) / RB_UCAP;
        const unsigned int usize = (((count + parts - 1) / parts) + 63u) & ~63u; // columns per unit of this row block: <= RB_UCAP
        const unsigned int c0 = part * usize;
        d.ucount = __builtin_amdgcn_readfirstlane(count - c0 < usize ? count - c0 : usize);
        d.col_start = static_cast<unsigned int>(seg.x) + c0;
        d.row_first = static_cast<unsigned int>(d.rb) * TILE;
        return d;
    };
    auto request_indices = [&](const UnitDesc &d, unsigned int (&out)[RB_SLICES_PER_WAVE]) {
#pragma unroll
        for (int s = 0; s < RB_SLICES_PER_WAVE; s++) {
            const unsigned int lp = static_cast<unsigned int>((wave + s * RB_WAVES) * 64 + lane);
            out[s] = lp < d.ucount ? col_atoms[d.col_start + lp] : uK;
        }
    };
    // the row atoms of a unit into one of the two row buffers (ONE wave, all 64 lanes enabled)
    auto stage_rows = [&](const UnitDesc &d, RowBuf &buf) {
        const unsigned int ridx = d.row_first + static_cast<unsigned int>(lane);
        unsigned int ra = uK;
        if (lane < TILE && ridx < static_cast<unsigned int>(NR)) {
            ra = row_idxs ? row_idxs[ridx] : ridx;
        }
        Real rr[7];
#pragma unroll
        for (int c = 0; c < 7; c++) {
            rr[c] = gathered[static_cast<size_t>(ra) * 8 + c]; // record K is the zero sentinel: no branch
        }
        if (lane < TILE) {
            buf.rowatom[lane] = ra;
#pragma unroll
            for (int c = 0; c < 7; c++) {
                buf.row[c][lane] = rr[c];
            }
        }
        wave_lds_sync();
        const Real ox = buf.row[0][0], oy = buf.row[1][0], oz = buf.row[2][0]; // the first row atom of a block is always valid
        const float w0 = static_cast<float>(buf.row[3][0]);
        const bool row_valid = lane < TILE && ra < uK;
        const Real tx = rint_real((rr[0] - ox) * bx.inv_x), ty = rint_real((rr[1] - oy) * bx.inv_y), tz = rint_real((rr[2] - oz) * bx.inv_z);
        const float fx = static_cast<float>(fma_real(-bx.x, tx, rr[0] - ox));
        const float fy = static_cast<float>(fma_real(-bx.y, ty, rr[1] - oy));
        const float fz = static_cast<float>(fma_real(-bx.z, tz, rr[2] - oz));
        const float rw = static_cast<float>(rr[3]);
        const bool wrapped = row_valid && (tx != 0 || ty != 0 || tz != 0);
        if (lane < TILE) {
            buf.rowflt[0][lane] = fx;
            buf.rowflt[1][lane] = fy;
            buf.rowflt[2][lane] = fz;
            buf.rowflt[3][lane] = row_valid ? rw : 1e18f;                                                             // an invalid row never passes the explicit filter ...
            buf.rowflt[4][lane] = row_valid ? __builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx)) : 1e30f; // ... nor the Gram one
        }
        float rext = row_valid ? fmaxf(fmaxf(fabsf(fx) * fex, fabsf(fy) * fey), fmaxf(fabsf(fz) * fez, fabsf(rw) * few)) : 0.0f;
        rext = wave_max_low_half_nonneg(rext);
        const bool any_wrapped = __ballot(wrapped) != 0ull;
        const bool not_flat = __ballot(row_valid && rw != w0) != 0ull;
        if (lane == 0) {
            buf.rext = rext;
            buf.w0 = w0;
            buf.flags = (any_wrapped ? 1u : 0u) | (not_flat ? 2u : 0u);
        }
    };

    unsigned int k_unit = 0;
    if (unit_of(0) >= total_units) {
        return; // (whole workgroup)
    }
    UnitDesc cur = decode(unit_of(0));
    unsigned int ja_next[RB_SLICES_PER_WAVE];
    request_indices(cur, ja_next);
    if (wave == 0) {
        stage_rows(cur, lds.rows[0]);
    }
    // accumulators and counters start at zero; every unit leaves them at zero again (the flush clears what it reads)
    for (int i = tid; i < 3 * RB_UCAP; i += RB_THREADS) {
        (&lds.colacc[0][0])[i] = 0;
    }
    if (tid < 3 * TILE) {
        (&lds.rowacc[0][0])[tid] = 0;
    }
    if (tid < 64) {
        lds.hist[tid] = 0;
        lds.cnt[tid] = 0;
    }
    if (tid == 0) {
        lds.piece_ticket = RB_WAVES; // pieces 0 .. RB_WAVES-1 are the waves' first ones
    }
    __syncthreads();
    TM_RB_STAMP(0); // prologue: unit table, force-factor table, fused slices, first unit's rows
    while (true) {
        RowBuf &rows = lds.rows[k_unit & 1u];
        const unsigned int ucount = cur.ucount, col_start = cur.col_start, row_first = cur.row_first;
        // ---- phase 1, requests: the records of this wave's column slices (their indices were requested a unit ago)
        unsigned int ja[RB_SLICES_PER_WAVE];
        Real cx[RB_SLICES_PER_WAVE], cy[RB_SLICES_PER_WAVE], cz[RB_SLICES_PER_WAVE], cw[RB_SLICES_PER_WAVE];
#pragma unroll
        for (int s = 0; s < RB_SLICES_PER_WAVE; s++) {
            ja[s] = ja_next[s];
            const Real *g = gathered + static_cast<size_t>(ja[s]) * 8; // record K is the zero sentinel: no branch
            cx[s] = g[0];
            cy[s] = g[1];
            cz[s] = g[2];
            cw[s] = g[3];
        }
        // ---- the NEXT unit: decoded now, its column indices requested now
        const unsigned int unit_next = unit_of(k_unit + 1u);
        const bool have_next = unit_next < total_units;
        UnitDesc nxt = cur;
        if (have_next) {
            nxt = decode(unit_next);
            request_indices(nxt, ja_next);
        }
        TM_RB_STAMP(1);

        // ---- phase 1: the row masks of this wave's slices
        const Real ox = rows.row[0][0], oy = rows.row[1][0], oz = rows.row[2][0];
        const float rext = rows.rext, w0 = rows.w0;
        // (read while faster waves of this unit may already be OR-ing their slices' bits in, below: which filter form a wave takes
        // can therefore depend on timing -- but never a result: the flat and the general Gram form are both conservative supersets
        // of the exact `d2 < cutoff^2` test that phase 2 makes, and a flag only ever switches a wave to the MORE general form)
        const bool rows_flat = (rows.flags & 2u) == 0u;
        float ra_[5], rb_[5]; // rows (lane & 15) and 16 + (lane & 15): x y z w |r|^2
#pragma unroll
        for (int c = 0; c < 5; c++) {
            ra_[c] = rows.rowflt[c][lane & 15];
            rb_[c] = rows.rowflt[c][16 + (lane & 15)];
        }
        unsigned int mask[RB_SLICES_PER_WAVE];
        unsigned int my_flags = 0;
#pragma unroll
        for (int s = 0; s < RB_SLICES_PER_WAVE; s++) {
            mask[s] = 0;
            if ((wave + s * RB_WAVES) * 64 >= static_cast<int>(ucount)) { // wave-uniform: nothing listed here
                continue;
            }
            const bool live = ja[s] < uK;
            const Real tx = rint_real((cx[s] - ox) * bx.inv_x), ty = rint_real((cy[s] - oy) * bx.inv_y), tz = rint_real((cz[s] - oz) * bx.inv_z);
            const float cfx = static_cast<float>(fma_real(-bx.x, tx, cx[s] - ox));
            const float cfy = static_cast<float>(fma_real(-bx.y, ty, cy[s] - oy));
            const float cfz = static_cast<float>(fma_real(-bx.z, tz, cz[s] - oz));
            const float col_w = static_cast<float>(cw[s]);
            const bool wrapped = live && (tx != 0 || ty != 0 || tz != 0);
            float cext = fmaxf(fmaxf(fabsf(cfx) * fex, fabsf(cfy) * fey), fmaxf(fabsf(cfz) * fez, fabsf(col_w) * few));
            cext = live ? cext : 0.0f;
            // (the same specialisations as the item kernel decides per item: gram = the difference of the origin-relative images IS
            // the minimum image and the Gram form's rounding error is bounded; flat = every w equal)
            const bool gram = __ballot(!(cext + rext < 1.0f)) == 0ull;
            const bool flat = rows_flat && __ballot(live && col_w != w0) == 0ull;
            my_flags |= ((!gram || __ballot(wrapped) != 0ull) ? 1u : 0u) | (flat ? 0u : 2u);
            unsigned int m = 0;
            if (hint<F64>(gram && flat, true)) {
                const float c2x = -2.0f * cfx, c2y = -2.0f * cfy, c2z = -2.0f * cfz;
                const float cc = __builtin_fmaf(cfz, cfz, __builtin_fmaf(cfy, cfy, cfx * cfx));
                const float thr = live ? (fcut2 + 1e-4f) - cc : -1e30f;
                rb_filter4<12>(rb_[0], rb_[1], rb_[2], rb_[4], c2x, c2y, c2z, thr, m);
                rb_filter4<8>(rb_[0], rb_[1], rb_[2], rb_[4], c2x, c2y, c2z, thr, m);
                rb_filter4<4>(rb_[0], rb_[1], rb_[2], rb_[4], c2x, c2y, c2z, thr, m);
                rb_filter4<0>(rb_[0], rb_[1], rb_[2], rb_[4], c2x, c2y, c2z, thr, m);
                rb_filter4<12>(ra_[0], ra_[1], ra_[2], ra_[4], c2x, c2y, c2z, thr, m);
                rb_filter4<8>(ra_[0], ra_[1], ra_[2], ra_[4], c2x, c2y, c2z, thr, m);
                rb_filter4<4>(ra_[0], ra_[1], ra_[2], ra_[4], c2x, c2y, c2z, thr, m);
                rb_filter4<0>(ra_[0], ra_[1], ra_[2], ra_[4], c2x, c2y, c2z, thr, m);
            } else {
                const RbGeneral g{cfx, cfy, cfz, live ? col_w : -1e18f, fbx, fby, fbz, fibx, fiby, fibz, fcut2};
                rb_general_rounds<31>(ra_, rb_, g, m);
            }
            if (upper_triangular) { // row slot i is kept iff i < column index - first row index
                const int jrel = static_cast<int>(ja[s]) - static_cast<int>(row_first);
                m &= jrel >= TILE ? 0xffffffffu : (jrel <= 0 ? 0u : ((1u << jrel) - 1u));
            }
            m = live ? m : 0u;
            mask[s] = m;
            if (live) {
                atomicAdd(&lds.hist[__popc(m)], 1u);
            }
        }
        if (lane == 0 && my_flags != 0) {
            atomicOr(&rows.flags, my_flags);
        }
        __syncthreads(); // B1: histogram complete
        TM_RB_STAMP(2);

        // ---- sort: every wave scans the histogram for itself (bin 32 first), then ranks its columns with one returning LDS
        // atomic each and writes them to their sorted places
        unsigned int nnz, n_heavy;
        {
            int h = lane <= 32 ? static_cast<int>(lds.hist[32 - lane]) : 0; // lane q: columns with 32 - q hits
            int v = h;
            v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
            v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
            v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
            v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
            v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);
            v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);
            const int start = v - h; // first sorted position of this lane's bin
            nnz = static_cast<unsigned int>(__builtin_amdgcn_readlane(v, 31));                       // columns with >= 1 hit
            n_heavy = static_cast<unsigned int>(__builtin_amdgcn_readlane(v, 31 - TM_RB_SPLIT_POP)); // ... with > TM_RB_SPLIT_POP hits
#pragma unroll
            for (int s = 0; s < RB_SLICES_PER_WAVE; s++) {
                const unsigned int p = static_cast<unsigned int>(__popc(mask[s]));
                const int bin_start = __shfl(start, 32 - static_cast<int>(p), 64);
                if (p != 0) {
                    const unsigned int pos = static_cast<unsigned int>(bin_start) + atomicAdd(&lds.cnt[p], 1u);
                    lds.sorted_ja[pos] = ja[s];
                    lds.sorted_mask[pos] = mask[s];
                    lds.sorted_lp[pos] = static_cast<unsigned short>((wave + s * RB_WAVES) * 64 + lane);
                }
            }
        }
        __syncthreads(); // B2: sorted
        TM_RB_STAMP(3);
        const unsigned int unit_flags = rows.flags;
        const bool compact = (unit_flags & 1u) == 0u; // nothing wrapped, everything within the Gram extents: row - col is its own minimum image
        const bool unit_flat = (unit_flags & 2u) == 0u;

        // ---- the next unit's row atoms go into the other row buffer underneath this unit's pops (the last wave: its first
        // piece is the lightest of the first RB_WAVES)
        if (have_next && wave == RB_WAVES - 1) {
            stage_rows(nxt, lds.rows[(k_unit + 1u) & 1u]);
        }

        // ---- phase 2: virtual items (64 sorted columns), heaviest first; the first n_split of them as two half-row pieces
        const unsigned int n_virtual = (nnz + 63u) >> 6, n_split = (n_heavy + 63u) >> 6;
        const unsigned int n_pieces = n_virtual + n_split;
        auto pieces = [&](auto flat_tag) {
            constexpr bool FLAT = decltype(flat_tag)::value;
            unsigned int piece = static_cast<unsigned int>(wave);
            while (piece < n_pieces) {
#ifdef TM_TIMING
                const long long tm_p0 = clock64();
                tm_npieces++;
#endif
                const bool split = piece < 2u * n_split; // a heavy item's rows 0-15 or 16-31 only
                const unsigned int v = split ? piece >> 1 : piece - n_split;
                const unsigned int row_base = split ? (piece & 1u) << 4 : 0u;
                const unsigned int row_wrap = split ? 15u : 31u;
                const unsigned int idx = v * 64u + static_cast<unsigned int>(lane);
                const bool valid = idx < nnz;
                const unsigned int cja = valid ? lds.sorted_ja[idx] : uK;
                unsigned int m = valid ? lds.sorted_mask[idx] : 0u;
                const unsigned int lp = valid ? lds.sorted_lp[idx] : 0u;
                Real cj[7];
#pragma unroll
                for (int c = 0; c < 7; c++) {
                    cj[c] = (FLAT && c == 3) ? static_cast<Real>(0) : gathered[static_cast<size_t>(cja) * 8 + c];
                }
                // the next piece is drawn while the records are on their way
                unsigned int piece_next = 0;
                if (lane == 0) {
                    piece_next = atomicAdd(&lds.piece_ticket, 1u);
                }
                piece_next = __builtin_amdgcn_readfirstlane(piece_next);
                // scan start rotated by the lane number -- within the piece's rows: bit p of the rotated mask is row
                // row_base + ((p + lane) & row_wrap) --, so that the lanes of a trip (neighbours in hit count, often with the same
                // bits set) do not all pop the same row: a trip's row atomics then meet ~2 (split pieces: ~4) lanes per address.
                // (Rotating a half piece's 16 bits within 32 sent half the lanes to row_base first: 34 of 104 us per launch.)
                if (split) {
                    m = (m >> row_base) & 0xffffu;
                    m |= m << 16; // the 16 rows twice: a 32-bit rotation by lane & 15 is then a 16-bit one
                }
                m = __builtin_amdgcn_alignbit(m, m, static_cast<unsigned int>(lane) & row_wrap);
                m = split ? m & 0xffffu : m;
                u64 ax = 0, ay = 0, az = 0; // sum of the ROW-side values of this column's pairs (the column receives the negative)
#ifdef TM_TIMING
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (timing builds: the records' latency counts as the piece's, not the loop's)
                const long long tm_l0 = clock64();
#endif
                // every trip: all lanes pop (a lane whose mask is empty computes on some row and is discarded by the one branch
                // of the trip, together with the pairs the exact test rejects)
                if (__ballot(m != 0u) != 0ull) do {
                    const bool act = m != 0u;
#ifdef TM_TIMING
                    tm_trips++;
                    tm_pops += __popcll(__ballot(act));
#endif
                    unsigned int p;
                    asm("v_ffbl_b32 %0, %1" : "=v"(p) : "v"(m)); // lowest set bit (0xffffffff for an empty mask)
                    m &= m - 1u;
                    const unsigned int pi = row_base + ((p + static_cast<unsigned int>(lane)) & row_wrap);
                    Real ri[7];
                    if constexpr (F64) {
                        const unsigned int a = lds_offset(&rows.row[0][0]) + pi * 8u;
                        rb_read_row<0, FLAT>(a, ri);
                        rb_wait_row<FLAT>(ri);
                    } else {
#pragma unroll
                        for (int c = 0; c < 7; c++) {
                            ri[c] = (FLAT && c == 3) ? static_cast<Real>(0) : rows.row[c][pi];
                        }
                    }
                    Real ddx = ri[0] - cj[0], ddy = ri[1] - cj[1], ddz = ri[2] - cj[2];
                    if (hint<F64>(!compact, false)) { // wave-uniform; for a compact unit the three rint / fma pairs are exact no-ops
                        ddx = min_image(ddx, bx.x, bx.inv_x);
                        ddy = min_image(ddy, bx.y, bx.inv_y);
                        ddz = min_image(ddz, bx.z, bx.inv_z);
                    }
                    const Real ddw = FLAT ? static_cast<Real>(0) : ri[3] - cj[3];
                    const Real dd2 = FLAT ? fma_real(ddz, ddz, fma_real(ddy, ddy, ddx * ddx)) : pair_d2(ddx, ddy, ddz, ddw);
                    if (act && dd2 < cutoff2) { // the exact, strict test -- the only one that decides anything
                        const Real qi = ri[4], qj = cj[4];
                        const Real sig_i = ri[5], sig_j = cj[5], eps_i = ri[6], eps_j = cj[6];
                        u64 fx, fy, fz;
                        if constexpr (F64) {
                            bool below, big;
                            const double prefactor = nb_pair_prefactor_deferred<INSIDE_SWITCH>(1.0, 1.0, qi, qj, sig_i, sig_j, eps_i, eps_j, dd2, es_tab, below);
                            pair_force_fixed_fast_bounded(prefactor, ddx, ddy, ddz, ps_limit, fx, fy, fz, big);
                            const bool rare = below || big;
                            if (__builtin_expect(__ballot(rare) != 0ull, 0)) {
                                if (rare) {
                                    const double pf = below ? nb_pair_prefactor_below_table(1.0, 1.0, qi, qj, sig_i, sig_j, eps_i, eps_j, dd2, beta) : prefactor;
                                    pair_force_fixed_slow(pf, ddx, ddy, ddz, fx, fy, fz);
                                }
                            }
                        } else {
                            PairOut<Real> o;
                            nb_pair<false>(static_cast<Real>(1), static_cast<Real>(1), qi, qj, sig_i, sig_j, eps_i, eps_j, dd2, beta, o, es_tab);
                            pair_force_fixed_bounded(o.prefactor, ddx, ddy, ddz, static_cast<Real>(ps_limit * (1.0 / 68719476736.0)), fx, fy, fz);
                        }
#if defined(TM_RB_ABLATE) && TM_RB_ABLATE == 1 // timing only: no row accumulation
                        asm volatile("" : : "v"(fx), "v"(fy), "v"(fz));
#elif defined(TM_RB_ABLATE) && TM_RB_ABLATE == 2 // timing only: row accumulation at conflict-free addresses
                        lds_add(&lds.colacc[0][lane + 64 * wave], fx);
                        lds_add(&lds.colacc[1][lane + 64 * wave], fy);
                        lds_add(&lds.colacc[2][lane + 64 * wave], fz);
#else
                        lds_add(&lds.rowacc[0][pi], fx);
                        lds_add(&lds.rowacc[1][pi], fy);
                        lds_add(&lds.rowacc[2][pi], fz);
#endif
                        // (in place, as asm: left to the compiler the three sums change registers across the branch -- six 64-bit moves per trip)
                        asm("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(ax) : "v"(fx));
                        asm("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(ay) : "v"(fy));
                        asm("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(az) : "v"(fz));
                    }
                } while (__ballot(m != 0u) != 0ull);
#ifdef TM_TIMING
                tm_loop += clock64() - tm_l0;
#endif
                if (valid) { // FIX(-p d) == -FIX(p d): the column's sums are the negatives (a split item's two pieces meet here)
                    lds_sub(&lds.colacc[0][lp], ax);
                    lds_sub(&lds.colacc[1][lp], ay);
                    lds_sub(&lds.colacc[2][lp], az);
                }
                piece = piece_next;
#ifdef TM_TIMING
                tm_piece += clock64() - tm_p0;
#endif
            }
        };
        if (hint<F64>(unit_flat, true)) {
            pieces(std::true_type{});
        } else {
            pieces(std::false_type{});
        }
#ifdef TM_TIMING
        const long long tm_w0 = clock64();
#endif
        __syncthreads(); // B3: every pair of the unit accumulated
#ifdef TM_TIMING
        tm_wait3 += clock64() - tm_w0;
#endif
        TM_RB_STAMP(4);

        // ---- flush, in list order: one global atomic per touched (atom, component); what is read is cleared for the next unit
        for (unsigned int i = static_cast<unsigned int>(tid); i < ucount; i += RB_THREADS) {
            const unsigned int a = col_atoms[col_start + i];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const u64 v = lds.colacc[c][i];
                if (v != 0) {
                    atomicAdd(g_du_dx + static_cast<size_t>(c) * acc_stride + a, v);
                    lds.colacc[c][i] = 0;
                }
            }
        }
        if (tid >= RB_THREADS - 3 * TILE) { // the last waves: they are the least likely to have column work left
            const int t = tid - (RB_THREADS - 3 * TILE);
            const int c = t / TILE, a = t - c * TILE;
            const u64 v = lds.rowacc[c][a];
            const unsigned int ra = rows.rowatom[a];
            if (v != 0 && ra < uK) {
                atomicAdd(g_du_dx + static_cast<size_t>(c) * acc_stride + ra, v);
            }
            lds.rowacc[c][a] = 0;
        }
        if (tid < 64) {
            lds.hist[tid] = 0;
            lds.cnt[tid] = 0;
        }
        if (tid == 64) {
            lds.piece_ticket = RB_WAVES;
        }
        __syncthreads(); // B4: the unit's LDS is free
        TM_RB_STAMP(5);
#ifdef TM_TIMING
        tm_units++;
#endif
        if (!have_next) {
            break;
        }
        cur = nxt;
        k_unit++;
    }
#ifdef TM_TIMING
    if (lane == 0) {
        atomicAdd(&s_tm_counts[0], static_cast<unsigned long long>(tm_trips));
        atomicAdd(&s_tm_counts[1], static_cast<unsigned long long>(tm_pops));
        atomicAdd(&s_tm_counts[2], static_cast<unsigned long long>(tm_loop));
        atomicAdd(&s_tm_counts[3], static_cast<unsigned long long>(tm_piece));
        atomicAdd(&s_tm_counts[4], static_cast<unsigned long long>(tm_wait3));
        atomicAdd(&s_tm_counts[5], static_cast<unsigned long long>(tm_npieces));
    }
    __syncthreads();
    tm_acc[0] = static_cast<long long>(s_tm_counts[0] | (s_tm_counts[1] << 32)); // trips | lane-pops of the whole workgroup (replaces the prologue's cycles)
    if (tid == 0 && timing) { // cycles of wave 0 up to each barrier, summed over the workgroup's units
        long long *t = timing + static_cast<size_t>(blockIdx.x) * 8;
        for (int k = 0; k < 6; k++) {
            t[k] = tm_acc[k];
        }
        t[6] = (clock64() - tm_begin) | (tm_units << 48);
        // (the phase slots 1 and 3 -- requests, sort -- give way to the pops' inner accounting, summed over the workgroup's waves)
        t[1] = static_cast<long long>(s_tm_counts[2] | (s_tm_counts[5] << 40)); // cycles inside the trip loops | pieces
        t[3] = static_cast<long long>(s_tm_counts[3] | (s_tm_counts[4] << 32)); // cycles inside pieces (loads + loop + store) | cycles waiting at B3
        t[7] = tm_real_begin | (static_cast<long long>(__builtin_amdgcn_s_memrealtime()) << 32); // 100 MHz, device-wide
    }
#endif
}

} // namespace tmamd
