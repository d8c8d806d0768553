// Neighbor-list construction kernels for gfx950.  Included by nonbonded.hip only.
//
// What is built (all on device, no host round trip):
//   block bounds   Real[nblocks][3] centre / extent per 32-atom block (PBC-aware; the reported boxes follow the
//                  running re-imaging of reference k_find_block_bounds, cpp/src/kernels/k_neighborlist.cuh:11-116)
//   CSR lists      per row block a contiguous segment in `col_atoms` holding every column atom that is
//                  within `cutoff` (the list cutoff = cutoff + padding) of at least one row atom -- the same
//                  set the reference compacts into 32-wide `ixn_atoms` tiles (k_neighborlist.cuh:199-458),
//                  which is what Neighborlist.get_nblist must report (tests/test_nblist.py:180-186)
//   work items     {row_block, col_start, col_count<=64} for the tile kernel
//
// Differences from the reference's CUDA structure, on purpose: one 1024-thread workgroup (16 waves) per row block
// scans ALL column blocks (the reference launches a (row, col/32) grid of 32-thread blocks and needs a second
// "trim compaction" kernel); 64-wide ballots; every row block owns a fixed, worst-case-sized segment of the pool, so
// its columns are contiguous and neither per-tile atomics nor a counting pass are needed.
#pragma once
#include "kernels_nonbonded.hip.hpp"

namespace tmamd {

// A value every lane of the wave holds alike, moved to scalar registers (the compiler keeps the result of a vector instruction
// -- an LDS read, a division, a shuffle -- in vector registers even when it is uniform; the list kernel's budget is 64).
__device__ __forceinline__ float wave_uniform(const float v) {
    return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v)));
}
__device__ __forceinline__ double wave_uniform(const double v) {
    const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(v));
    const unsigned int lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned int>(b));
    const unsigned int hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned int>(b >> 32));
    return __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo));
}

// K2: block bounds (one wave per 32-atom block), coordinate snapshot and counter reset.
// REFERENCE_FOLD = true reproduces the reference's boxes (what Neighborlist.compute_block_bounds reports): its running
// min/max fold is inherently sequential (each atom is re-imaged around the CURRENT centre), so every lane of the wave
// runs the same fold on shuffled-in positions: lanes 0-31 load one atom each (coalesced), no LDS, no divergence.
// REFERENCE_FOLD = false is what the list build uses: every atom is imaged next to the block's first atom and the
// min / max are butterfly reductions -- 5 shuffle steps instead of a 33-step dependent chain (14 us -> launch latency).
// Any box that contains an image of each of its atoms yields a valid (superset) list, and integer accumulation makes
// the forces independent of which superset is used, so the two folds are interchangeable for everything but the
// reported boxes themselves.
template <typename Real, bool REFERENCE_FOLD>
__global__ __launch_bounds__(256) void k_block_bounds(
    const int n_col_blocks, const int NC, const unsigned int *__restrict__ col_idxs, // nullptr => identity
    const int n_row_blocks, const int NR, const unsigned int *__restrict__ row_idxs, // only used when rows != cols
    const int rows_equal_cols, const Real *__restrict__ gathered, const double *__restrict__ box,
    Real *__restrict__ col_ctr, Real *__restrict__ col_ext, Real *__restrict__ row_ctr, Real *__restrict__ row_ext,
    unsigned int *__restrict__ counters, // [0]=unused [1]=n_items [2]=tile count [3]=builds so far [4..4+NB_SHARDS*NB_CLASSES)=items per (shard, cost class) bucket
    const int n_snap, const double *__restrict__ x, double *__restrict__ snap_x, double *__restrict__ snap_box,
    const int *__restrict__ flag, const int force,
    double *__restrict__ rebase_snap_box, // != nullptr: scale-aware potentials (k_check_gather_scaled ran in front of this launch)
    // merged orders (engine.hpp): the first guest_blocks blocks hold guest_rows atoms followed by holes
    const int guest_rows = 0, const int guest_blocks = 0) {
    if (!force && *flag == 0) {
        if (rebase_snap_box != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
            rebase_snapshot_box(box, rebase_snap_box);
        }
        return;
    }
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nthreads = gridDim.x * blockDim.x;
    if (tid == 0) {
        counters[0] = 0;
        counters[1] = 0;
        counters[2] = 0;
        counters[3] += 1; // list builds since construction (diagnostic: rebuild period of an MD run)
    }
    if (tid < NB_SHARDS * NB_CLASSES) {
        counters[NB_COUNTER_CLASS0 + tid] = 0;
    }
    if (tid == 0) {
        counters[NB_COUNTER_GUEST] = 0;
    }
    if (snap_x) {
        for (int t = tid; t < n_snap; t += nthreads) {
            snap_x[t] = x[t];
        }
        if (tid < 9) {
            snap_box[tid] = box[tid];
        }
        if (tid < 3) {
            snap_box[9 + tid] = 1.0; // the snapshot is in this box's own scale again (k_check_gather_scaled)
        }
    }
    const int total_blocks = n_col_blocks + (rows_equal_cols ? 0 : n_row_blocks);
    const int lane = threadIdx.x & 63;
    const NbBox<Real> bx = load_box<Real>(box);
    const Real half = static_cast<Real>(0.5);
    const Real b[3] = {bx.x, bx.y, bx.z};
    const Real ib[3] = {bx.inv_x, bx.inv_y, bx.inv_z};
    for (int wblk = tid >> 6; wblk < total_blocks; wblk += nthreads >> 6) {
        const bool is_row = wblk >= n_col_blocks;
        const int blk = is_row ? wblk - n_col_blocks : wblk;
        const unsigned int *idxs = is_row ? row_idxs : col_idxs;
        const int count = (!is_row && blk < guest_blocks) ? guest_rows : (is_row ? NR : NC);
        const int first = blk * TILE;
        const int n = (count - first) < TILE ? (count - first) : TILE;
        Real p[3] = {0, 0, 0};
        if (lane < n) {
            const unsigned int a = idxs ? idxs[first + lane] : static_cast<unsigned int>(first + lane);
            for (int d = 0; d < 3; d++) {
                p[d] = gathered[static_cast<size_t>(a) * 8 + d];
            }
        }
        Real lo[3], hi[3];
        for (int d = 0; d < 3; d++) {
            lo[d] = hi[d] = __shfl(p[d], 0, 64);
        }
        if constexpr (REFERENCE_FOLD) {
            // visiting order of the reference's lane rotation: atoms 1, 2, ..., n-1, then atom 0 again
            for (int k = 1; k <= n; k++) {
                const int kk = k == n ? 0 : k;
                for (int d = 0; d < 3; d++) {
                    const Real pd = __shfl(p[d], kk, 64);
                    const Real img = pd - b[d] * nearbyint((pd - half * (hi[d] + lo[d])) * ib[d]);
                    lo[d] = min(lo[d], img);
                    hi[d] = max(hi[d], img);
                }
            }
        } else {
            for (int d = 0; d < 3; d++) {
                const Real p0 = lo[d];
                const Real img = lane < n ? p[d] - b[d] * nearbyint((p[d] - p0) * ib[d]) : p0;
                Real l = img, h = img;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) {
                    l = min(l, __shfl_xor(l, o, 64));
                    h = max(h, __shfl_xor(h, o, 64));
                }
                // lanes 32-63 hold p0 only; every lane ends with the reduction of its own half, use the lower one
                lo[d] = __shfl(l, 0, 64);
                hi[d] = __shfl(h, 0, 64);
            }
        }
        if (lane < 3) {
            Real *ctr = is_row ? row_ctr : col_ctr;
            Real *ext = is_row ? row_ext : col_ext;
            // every lane holds the same lo/hi; lane d writes component d
            const Real l = lane == 0 ? lo[0] : (lane == 1 ? lo[1] : lo[2]);
            const Real h = lane == 0 ? hi[0] : (lane == 1 ? hi[1] : hi[2]);
            ctr[blk * 3 + lane] = half * (h + l);
            ext[blk * 3 + lane] = half * (h - l);
        }
    }
}

// K3: per row block, find interacting column atoms.  One NBL_THREADS-thread workgroup (16 waves) per row block.
//   pass 1  per chunk of NBL_CHUNK column blocks: compact the passing block ids into an LDS list (ballot + popcount)
//   pass 2  the waves stride over the list two column blocks at a time (lanes 0-31 / 32-63 = one column atom
//           each); every lane tests its atom against all 32 rows without a branch (f32, origin-relative, Gram form) and
//           counts the rows inside the cost cutoff on the way; see the comment at the loop.
#ifndef TM_NBL_COST_STRIDE
#define TM_NBL_COST_STRIDE 4
#endif
static const int NBL_COST_STRIDE = TM_NBL_COST_STRIDE; // cost estimates count every 4th row
static const int NBL_CHUNK = 2048;   // column blocks per LDS list chunk (8 KB of LDS)
#ifndef TM_NBL_CAND_CAP
#define TM_NBL_CAND_CAP 8192
#endif
static const int NBL_CAND_CAP = TM_NBL_CAND_CAP; // accepted column atoms per row block whose pair counts are staged in LDS (16 KB; a 1.3 nm list at water density holds ~2100)
#ifndef TM_NBL_THREADS
#define TM_NBL_THREADS 1024
#endif
static const int NBL_THREADS = TM_NBL_THREADS; // 16 waves per row block: the per-row-block critical path is what bounds this kernel
#ifndef TM_NBL_WAVES_PER_SIMD
#define TM_NBL_WAVES_PER_SIMD 8
#endif

template <typename Real, bool UPPER_TRIANGULAR>
__global__ __launch_bounds__(NBL_THREADS, TM_NBL_WAVES_PER_SIMD) void k_find_ixns( // 8 waves per SIMD = 2 workgroups per CU: the register budget is 64
    const int K, const int NC, const int NR, const unsigned int *__restrict__ col_idxs, const unsigned int *__restrict__ row_idxs,
    const Real *__restrict__ col_ctr, const Real *__restrict__ col_ext, const Real *__restrict__ row_ctr,
    const Real *__restrict__ row_ext, const Real *__restrict__ gathered, const double *__restrict__ box, const double cutoff_d,
    const double cost_cutoff_d, // pairs closer than this are what an item's cost estimate counts
    unsigned int *__restrict__ counters, unsigned int *__restrict__ col_atoms, int4 *__restrict__ items,
    const unsigned int items_cap,    // capacity of each of the NB_SHARDS * NB_CLASSES item buckets
    int2 *__restrict__ row_segments, // per row block {start, count}
    const int *__restrict__ flag, const int force,
    // snap_x != nullptr: no bounds kernel ran in front of this launch (its boxes came with the sorted hand-over, see
    // engine.hpp: PregatherTarget); its other duties on a rebuild fall to this kernel: the coordinate / box snapshot and the
    // build count.  (The counter reset was made by whoever raised the flag.)
    const int n_snap, const double *__restrict__ x, double *__restrict__ snap_x, double *__restrict__ snap_box,
    // Merged orders (UPPER_TRIANGULAR lists only; engine.hpp: NonbondedAllPairs as the carrier of an interaction group): the first
    // guest_blocks row blocks are the GROUP's row atoms (guest_rows of them, then holes up to the block boundary).  They see every
    // column block from guest_blocks on -- the all-pairs atoms -- and none of their own kind; their items carry the sign bit in
    // their fourth word, which tells the tile kernel to read the columns' records under the group's parameters.
    const int guest_rows = 0, const int guest_blocks = 0,
    int4 *__restrict__ guest_items = nullptr) { // ... and go once more into this compact list (counted in counters[NB_COUNTER_GUEST])
    if (!force && *flag == 0) {
        return;
    }
    if (snap_x != nullptr) {
        for (int t = blockIdx.x * NBL_THREADS + threadIdx.x; t < n_snap; t += gridDim.x * NBL_THREADS) {
            snap_x[t] = x[t];
        }
        if (blockIdx.x == 0 && threadIdx.x < 9) {
            snap_box[threadIdx.x] = box[threadIdx.x];
        }
        if (blockIdx.x == 0 && threadIdx.x < 3) {
            snap_box[9 + threadIdx.x] = 1.0;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            counters[3] += 1;
        }
    }
    __shared__ int s_list[NBL_CHUNK];
    __shared__ Real s_rx[TILE], s_ry[TILE], s_rz[TILE];
    __shared__ unsigned int s_nlist, s_count;
    __shared__ unsigned int s_hist[NB_CLASSES], s_base[NB_CLASSES];
    __shared__ float s_rf[3][TILE];
    // the rows once more in the form the fine pass multiplies with: (-2 x, -2 y, -2 z, |r|^2) of the origin-relative f32
    // position, so that |r - c|^2 - |c|^2 is three FMAs on a broadcast 16-byte read; rows past the block's end hold |r|^2 = 1e30
    __shared__ float4 s_rg[TILE];
    // per accepted column atom, in list order: how many of the block's rows it has inside the cost cutoff (the fine pass
    // counts them while it is there); the items' cost estimates are sums over 64 consecutive entries
    __shared__ unsigned short s_cnt[NBL_CAND_CAP];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int rb = blockIdx.x;
#ifdef TM_NBL_TIMING
    unsigned long long tm_t[8];
    int tm_n = 0;
#define TM_NBL_STAMP() do { __syncthreads(); tm_t[tm_n++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TM_NBL_STAMP() do { } while (0)
#endif
    TM_NBL_STAMP();
    const int n_col_blocks = (NC + TILE - 1) / TILE;
    const bool guest = UPPER_TRIANGULAR && rb < guest_blocks;
    const int row_limit = guest ? guest_rows : NR;
    const int nrow = (row_limit - rb * TILE) < TILE ? (row_limit - rb * TILE) : TILE;
    const int guest_bit = guest ? static_cast<int>(0x80000000u) : 0;
    const int cb_first = UPPER_TRIANGULAR ? (guest ? guest_blocks : rb) : 0;

    // Everything the kernel needs from memory before its first barrier is requested in one go -- this thread's first column
    // box, the row block's own box, the row atoms: a dependent global load costs 1-3 us right behind a kernel boundary (the
    // inputs were written by another XCD), and the workgroup's life is a chain of such hops.
    Real pc[6] = {0, 0, 0, 0, 0, 0};
    const int cb_pre = cb_first + tid;
    if (cb_pre < n_col_blocks) {
#pragma unroll
        for (int d = 0; d < 3; d++) {
            pc[d] = col_ctr[cb_pre * 3 + d];
            pc[3 + d] = col_ext[cb_pre * 3 + d];
        }
    }
    const Real rcx = row_ctr[rb * 3 + 0], rcy = row_ctr[rb * 3 + 1], rcz = row_ctr[rb * 3 + 2];
    const Real rex = row_ext[rb * 3 + 0], rey = row_ext[rb * 3 + 1], rez = row_ext[rb * 3 + 2];
    NbBox<Real> bx = load_box<Real>(box);
    bx.inv_x = wave_uniform(bx.inv_x); // (the divisions ran on the vector unit)
    bx.inv_y = wave_uniform(bx.inv_y);
    bx.inv_z = wave_uniform(bx.inv_z);
    const Real cutoff = static_cast<Real>(cutoff_d);
    const Real cutoff2 = cutoff * cutoff;

    if (tid == 0) {
        s_count = 0;
        s_nlist = 0;
    }
    if (tid < TILE) {
        const int ridx = rb * TILE + tid;
        if (ridx < row_limit) {
            const unsigned int a = row_idxs ? row_idxs[ridx] : static_cast<unsigned int>(ridx);
            s_rx[tid] = gathered[static_cast<size_t>(a) * 8 + 0];
            s_ry[tid] = gathered[static_cast<size_t>(a) * 8 + 1];
            s_rz[tid] = gathered[static_cast<size_t>(a) * 8 + 2];
        } else {
            s_rx[tid] = s_ry[tid] = s_rz[tid] = 0;
        }
    }
    __syncthreads();

    // row bbox vs column bbox (k_neighborlist.cuh:296-329)
    auto coarse_box = [&](Real ccx, Real ccy, Real ccz, Real cex, Real cey, Real cez) -> bool {
        Real ddx = min_image(rcx - ccx, bx.x, bx.inv_x);
        Real ddy = min_image(rcy - ccy, bx.y, bx.inv_y);
        Real ddz = min_image(rcz - ccz, bx.z, bx.inv_z);
        ddx = max(static_cast<Real>(0), fabs(ddx) - rex - cex);
        ddy = max(static_cast<Real>(0), fabs(ddy) - rey - cey);
        ddz = max(static_cast<Real>(0), fabs(ddz) - rez - cez);
        return (ddx * ddx + ddy * ddy + ddz * ddz) < cutoff2;
    };

    // The row block's segment of the pool has a fixed place and worst-case room: an upper-triangular list only ever sees
    // columns from its own block onwards (triangular layout, exactly the pool's size); a row/column-subset list gets
    // NC entries per row block.  No counting pass and no claim (a returning global atomic) before the real work.
    const unsigned int ncp = static_cast<unsigned int>(n_col_blocks) * TILE;
    const unsigned int seg_start =
        UPPER_TRIANGULAR ? static_cast<unsigned int>(rb) * ncp - static_cast<unsigned int>(TILE) * (static_cast<unsigned int>(rb) * (rb - 1) / 2)
                         : static_cast<unsigned int>(rb) * ncp;

    // f32 copies of the row atoms relative to the first one (resolution independent of coordinate drift): the fine pass
    // and the cost estimate run on these; only distances within rounding reach of the cutoff are re-tested in Real
    const float fbx = static_cast<float>(bx.x), fby = static_cast<float>(bx.y), fbz = static_cast<float>(bx.z);
    const float fibx = wave_uniform(1.0f / fbx), fiby = wave_uniform(1.0f / fby), fibz = wave_uniform(1.0f / fbz);
    const Real ox = wave_uniform(s_rx[0]), oy = wave_uniform(s_ry[0]), oz = wave_uniform(s_rz[0]);
    if (tid < TILE) {
        const bool valid = tid < nrow;
        const float fx = valid ? static_cast<float>(min_image(s_rx[tid] - ox, bx.x, bx.inv_x)) : 0.0f;
        const float fy = valid ? static_cast<float>(min_image(s_ry[tid] - oy, bx.y, bx.inv_y)) : 0.0f;
        const float fz = valid ? static_cast<float>(min_image(s_rz[tid] - oz, bx.z, bx.inv_z)) : 0.0f;
        s_rf[0][tid] = fx;
        s_rf[1][tid] = fy;
        s_rf[2][tid] = fz;
        s_rg[tid] = valid ? make_float4(-2.0f * fx, -2.0f * fy, -2.0f * fz, __builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx)))
                          : make_float4(0.0f, 0.0f, 0.0f, 1e30f);
    }
    __syncthreads();
    // largest |row - origin| per dimension (every wave computes it for itself)
    float rmx = lane < TILE ? fabsf(s_rf[0][lane]) : 0.0f, rmy = lane < TILE ? fabsf(s_rf[1][lane]) : 0.0f,
          rmz = lane < TILE ? fabsf(s_rf[2][lane]) : 0.0f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        rmx = fmaxf(rmx, __shfl_xor(rmx, o, 64));
        rmy = fmaxf(rmy, __shfl_xor(rmy, o, 64));
        rmz = fmaxf(rmz, __shfl_xor(rmz, o, 64));
    }
    rmx = wave_uniform(rmx);
    rmy = wave_uniform(rmy);
    rmz = wave_uniform(rmz);
    const float rm2 = wave_uniform(__builtin_fmaf(rmz, rmz, __builtin_fmaf(rmy, rmy, rmx * rmx)));
    const float fcut = static_cast<float>(cutoff_d);
    const float fcut2 = static_cast<float>(cutoff_d * cutoff_d);
    const float fmargin = 2e-5f * (1.0f + fcut2); // >> the f32 rounding error of a squared distance of this size
    const float cost_cutoff2 = static_cast<float>(cost_cutoff_d * cost_cutoff_d);
    const Real cost_cutoff2_real = static_cast<Real>(cost_cutoff_d * cost_cutoff_d);
    TM_NBL_STAMP(); // rows loaded
    for (int chunk0 = cb_first; chunk0 < n_col_blocks; chunk0 += NBL_CHUNK) {
        // ---- pass 1: compact the passing column blocks of this chunk into s_list (s_nlist is zero here)
        const int chunk_end = (chunk0 + NBL_CHUNK) < n_col_blocks ? (chunk0 + NBL_CHUNK) : n_col_blocks;
        for (int cb0 = chunk0; cb0 < chunk_end; cb0 += NBL_THREADS) {
            const int cb = cb0 + tid;
            bool pass = false;
            if (cb < chunk_end) {
                if (cb0 == cb_first) { // the box requested at the top
                    pass = coarse_box(pc[0], pc[1], pc[2], pc[3], pc[4], pc[5]);
                } else {
                    pass = coarse_box(col_ctr[cb * 3 + 0], col_ctr[cb * 3 + 1], col_ctr[cb * 3 + 2], col_ext[cb * 3 + 0], col_ext[cb * 3 + 1], col_ext[cb * 3 + 2]);
                }
            }
            const u64 m = __ballot(pass);
            if (m) {
                unsigned int base = 0;
                if (lane == 0) {
                    base = atomicAdd(&s_nlist, static_cast<unsigned int>(__popcll(m)));
                }
                base = __shfl(base, 0, 64);
                if (pass) {
                    s_list[base + __popcll(m & ((1ull << lane) - 1ull))] = cb;
                }
            }
        }
        __syncthreads();
        const int nlist = s_nlist;

        TM_NBL_STAMP(); // coarse list ready
        // ---- pass 2: two column blocks per wave iteration (lanes 0-31 / 32-63 = one column atom each).  Every lane tests its
        // atom against ALL rows, branch-free, in f32 on origin-relative coordinates: per row one broadcast read of s_rg and
        // three FMAs give |r - c|^2 - |c|^2; the minimum decides, the (sampled) count inside the cost cutoff is the atom's share
        // of its item's cost estimate.  (The reference filters rows against the column block's box first, k_neighborlist.cuh:349-365,
        // and walks the survivors; on a wave that walk is a chain of ballots and branches that costs more than the 32 rows.)
        // Only distances within rounding reach of the cutoff, and atoms so far from the origin that row - column could need
        // re-imaging, take the exact test in Real -- so the listed set equals the brute-force Real set (tests/test_nblist.py:180-186).
        const int half_id = lane >> 5; // 0: lanes 0-31, 1: lanes 32-63
        const int sub = lane & 31;
        const int kstride = 2 * (NBL_THREADS / 64);
        auto fetch = [&](const int k, int &cb, unsigned int &ja, Real &xj, Real &yj, Real &zj) {
            const int my_entry = k + half_id;
            cb = my_entry < nlist ? s_list[my_entry] : -1;
            ja = K;
            xj = yj = zj = 0;
            if (cb >= 0) {
                const int jpos = cb * TILE + sub;
                if (jpos < NC) {
                    ja = col_idxs ? col_idxs[jpos] : static_cast<unsigned int>(jpos);
                    xj = gathered[static_cast<size_t>(ja) * 8 + 0];
                    yj = gathered[static_cast<size_t>(ja) * 8 + 1];
                    zj = gathered[static_cast<size_t>(ja) * 8 + 2];
                }
            }
        };
        int cb_n = -1;
        unsigned int ja_n = K;
        Real xn = 0, yn = 0, zn = 0;
        if (wave * 2 < nlist) {
            fetch(wave * 2, cb_n, ja_n, xn, yn, zn);
        }
        for (int k = wave * 2; k < nlist; k += kstride) {
            const int cb = cb_n;
            const unsigned int ja = ja_n;
            const bool valid = ja < static_cast<unsigned int>(K);
            const float cfx = static_cast<float>(min_image(xn - ox, bx.x, bx.inv_x));
            const float cfy = static_cast<float>(min_image(yn - oy, bx.y, bx.inv_y));
            const float cfz = static_cast<float>(min_image(zn - oz, bx.z, bx.inv_z));
            if (k + kstride < nlist) { // the next trip's atoms are on their way while this trip computes
                fetch(k + kstride, cb_n, ja_n, xn, yn, zn);
            }
            const float acx = fabsf(cfx), acy = fabsf(cfy), acz = fabsf(cfz);
            const bool no_wrap = (acx + rmx) * fibx < 0.49f && (acy + rmy) * fiby < 0.49f && (acz + rmz) * fibz < 0.49f;
            // farther than the cutoff from every row in one dimension alone (true under any re-imaging: |c| <= L/2)
            const bool far = fmaxf(fmaxf(acx - rmx, acy - rmy), acz - rmz) > fcut + 1e-4f;
            const float c2 = __builtin_fmaf(cfz, cfz, __builtin_fmaf(cfy, cfy, cfx * cfx));
            const float thr_cost = cost_cutoff2 - c2;
            float gmin = 1e30f;
            unsigned int cnt = 0;
            // (the rows are the same on every trip: without an address the compiler cannot see through it reads all 32 once,
            // in front of the loop, and spills them)
            int opaque0 = 0;
            asm volatile("" : "+s"(opaque0));
            const float4 *rg = s_rg + opaque0;
#pragma unroll 8
            for (int r = 0; r < TILE; r++) { // eight rows' reads in flight: 32 registers (the budget is 64)
                const float4 g = rg[r];
                const float acc = __builtin_fmaf(cfz, g.z, __builtin_fmaf(cfy, g.y, __builtin_fmaf(cfx, g.x, g.w)));
                gmin = fminf(gmin, acc);
                if ((r % NBL_COST_STRIDE) == 0) { // the count samples every fourth row (three instructions per counted row)
                    cnt += acc < thr_cost ? static_cast<unsigned int>(NBL_COST_STRIDE) : 0u;
                }
            }
            const float dmin = c2 + gmin;
            const float marg = __builtin_fmaf(2e-6f, c2 + rm2, fmargin); // + the Gram form's own rounding
            bool interacts = valid && !far && no_wrap && dmin < fcut2 - marg;
            const bool exact_needed = valid && !far && !interacts && (!no_wrap || dmin < fcut2 + marg);
            if (__ballot(exact_needed)) {
                if (exact_needed) {
                    // (the atom's Real position is not kept across the row loop for this: registers)
                    const Real xj = gathered[static_cast<size_t>(ja) * 8 + 0], yj = gathered[static_cast<size_t>(ja) * 8 + 1],
                               zj = gathered[static_cast<size_t>(ja) * 8 + 2];
                    unsigned int c = 0; // (every row: this path is rare)
                    for (int r = 0; r < nrow; r++) {
                        const Real dx = min_image(s_rx[r] - xj, bx.x, bx.inv_x);
                        const Real dy = min_image(s_ry[r] - yj, bx.y, bx.inv_y);
                        const Real dz = min_image(s_rz[r] - zj, bx.z, bx.inv_z);
                        const Real d2 = dx * dx + dy * dy + dz * dz;
                        interacts = interacts || d2 < cutoff2;
                        c += d2 < cost_cutoff2_real ? 1u : 0u;
                    }
                    cnt = c;
                }
            }
            if (UPPER_TRIANGULAR && cb == rb) {
                cnt >>= 1; // the diagonal tile evaluates row < column only
            }
            const u64 hits = __ballot(interacts);
            if (hits) {
                unsigned int base = 0;
                if (lane == 0) {
                    base = atomicAdd(&s_count, static_cast<unsigned int>(__popcll(hits)));
                }
                base = __shfl(base, 0, 64);
                if (interacts) {
                    const unsigned int pos = base + __popcll(hits & ((1ull << lane) - 1ull));
                    col_atoms[seg_start + pos] = ja;
                    if (pos < static_cast<unsigned int>(NBL_CAND_CAP)) {
                        s_cnt[pos] = static_cast<unsigned short>(cnt);
                    }
                }
            }
        }
        __syncthreads();
        if (chunk0 + NBL_CHUNK < n_col_blocks) { // (uniform) another chunk follows
            if (tid == 0) {
                s_nlist = 0;
            }
            __syncthreads();
        }
    }

    TM_NBL_STAMP(); // pass 2 done
    // ---- publish the segment and its work items
    // Every item gets a cost estimate -- the number of (row, column) pairs inside `cost_cutoff` -- and is filed into
    // bucket (shard = row block % NB_SHARDS, cost class), class 0 = heaviest.  The tile kernel deals the buckets to its
    // waves in that order (longest processing time first): items differ in cost by more than an order of magnitude and
    // a wave only processes a handful, so an arbitrary order leaves most waves idle while the unluckiest one finishes a
    // heavy item it drew last.  Bucket space is claimed per workgroup (one returning atomic per non-empty class --
    // returning global atomics are slow) and spread over NB_SHARDS cursors per class.
    const unsigned int count = s_count;
    const unsigned int n_chunks = (count + NB_CHUNK - 1) / NB_CHUNK;
    const unsigned int n_staged = n_chunks < static_cast<unsigned int>(NBL_CHUNK) ? n_chunks : NBL_CHUNK; // staged in s_list
    const unsigned int shard = static_cast<unsigned int>(rb) & (NB_SHARDS - 1);
    if (tid == 0) {
        row_segments[rb] = make_int2(static_cast<int>(seg_start), static_cast<int>(count));
        atomicAdd(&counters[2], (count + TILE - 1) / TILE);
        if (n_chunks) {
            atomicAdd(&counters[1], n_chunks);
        }
    }
    if (tid < NB_CLASSES) {
        s_hist[tid] = 0;
        s_base[tid] = 0;
    }
    // The histogram must be zero before ANY wave files its first chunk below.  Without this barrier a fast wave could
    // count into s_hist before wave 0 had zeroed it; the count was then lost, the class looked empty, its s_base stayed
    // uninitialised LDS, and the item went to a wild address (intermittent GPU memory faults once the cost estimate
    // stopped waiting on global loads).
    __syncthreads();
    for (unsigned int c = wave; c < n_chunks; c += NBL_THREADS / 64) {
        const unsigned int off = c * NB_CHUNK;
        const unsigned int len = (count - off) < NB_CHUNK ? (count - off) : NB_CHUNK;
        const bool staged = off + len <= static_cast<unsigned int>(NBL_CAND_CAP); // wave-uniform
        unsigned int mine = 0;
        if (staged) {
            if (static_cast<unsigned int>(lane) < len) {
                mine = s_cnt[off + lane];
            }
        } else if (static_cast<unsigned int>(lane) < len) {
            // more accepted atoms than the staging area holds: a sampled count (every NBL_COST_STRIDE-th row, lane-staggered)
            // on the atom re-read from memory -- written by other waves of this workgroup a barrier ago: read past the
            // (non-coherent) vector L1
            const unsigned int ja = __hip_atomic_load(col_atoms + seg_start + off + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float xj = static_cast<float>(min_image(gathered[static_cast<size_t>(ja) * 8 + 0] - ox, bx.x, bx.inv_x));
            const float yj = static_cast<float>(min_image(gathered[static_cast<size_t>(ja) * 8 + 1] - oy, bx.y, bx.inv_y));
            const float zj = static_cast<float>(min_image(gathered[static_cast<size_t>(ja) * 8 + 2] - oz, bx.z, bx.inv_z));
            for (int i = (lane & (NBL_COST_STRIDE - 1)); i < nrow; i += NBL_COST_STRIDE) {
                float dx = s_rf[0][i] - xj, dy = s_rf[1][i] - yj, dz = s_rf[2][i] - zj;
                dx = __builtin_fmaf(-fbx, __builtin_rintf(dx * fibx), dx);
                dy = __builtin_fmaf(-fby, __builtin_rintf(dy * fiby), dy);
                dz = __builtin_fmaf(-fbz, __builtin_rintf(dz * fibz), dz);
                const bool order_ok = !UPPER_TRIANGULAR || static_cast<unsigned int>(rb * TILE + i) < ja;
                mine += (order_ok && __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)) < cost_cutoff2) ? NBL_COST_STRIDE : 0u;
            }
        }
        unsigned int total = mine;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            total += __shfl_xor(total, o, 64);
        }
        total = total < 2048u ? total : 2048u;
        if (lane == 0) {
            const unsigned int heavy = total / NB_CLASS_PAIRS;
            const unsigned int cls = NB_CLASSES - 1 - (heavy < NB_CLASSES - 1 ? heavy : NB_CLASSES - 1);
            if (c < n_staged) {
                const unsigned int pos = atomicAdd(&s_hist[cls], 1u); // LDS
                s_list[c] = static_cast<int>(total | (pos << 12) | (cls << 23)); // total <= 2048, pos < NBL_CHUNK = 2048
            } else { // more chunks than the staging area holds (N > 131k): claim one by one
                const unsigned int bucket = shard * NB_CLASSES + cls;
                const unsigned int pos = atomicAdd(&counters[NB_COUNTER_CLASS0 + bucket], 1u);
                const int4 item = make_int4(rb, static_cast<int>(seg_start + off), static_cast<int>(len), static_cast<int>(total) | guest_bit);
                items[static_cast<size_t>(bucket) * items_cap + pos] = item;
                if (guest && guest_items != nullptr) {
                    guest_items[atomicAdd(&counters[NB_COUNTER_GUEST], 1u)] = item;
                }
            }
        }
    }
    TM_NBL_STAMP(); // costs done
    __syncthreads();
    if (tid < NB_CLASSES && s_hist[tid] != 0) {
        s_base[tid] = atomicAdd(&counters[NB_COUNTER_CLASS0 + shard * NB_CLASSES + tid], s_hist[tid]);
    }
    __syncthreads();
    for (unsigned int c = tid; c < n_staged; c += NBL_THREADS) {
        const unsigned int packed = static_cast<unsigned int>(s_list[c]);
        const unsigned int total = packed & 0xfffu, pos = (packed >> 12) & 0x7ffu, cls = packed >> 23;
        const unsigned int off = c * NB_CHUNK;
        const unsigned int len = (count - off) < NB_CHUNK ? (count - off) : NB_CHUNK;
        const int4 item = make_int4(rb, static_cast<int>(seg_start + off), static_cast<int>(len), static_cast<int>(total) | guest_bit);
        items[static_cast<size_t>(shard * NB_CLASSES + cls) * items_cap + s_base[cls] + pos] = item;
        if (guest && guest_items != nullptr) {
            guest_items[atomicAdd(&counters[NB_COUNTER_GUEST], 1u)] = item;
        }
    }
#ifdef TM_NBL_TIMING
    TM_NBL_STAMP();
    if (tid == 0 && (rb % 97) == 0) {
        printf("rb %d nlist %d count %u: rows %llu coarse %llu pass2 %llu cost %llu publish %llu (10 ns ticks), start %llu\n", rb, static_cast<int>(s_nlist), count,
               tm_t[1] - tm_t[0], tm_t[2] - tm_t[1], tm_t[3] - tm_t[2], tm_t[4] - tm_t[3], tm_t[5] - tm_t[4], tm_t[0] % 1000000ull);
    }
#endif
}

} // namespace tmamd
