// Neighbor-list construction kernels for gfx950.  Included by nonbonded.hip only.
//
// What is built (all on device, no host round trip):
//   block bounds   Real[nblocks][3] centre / extent per 32-atom block (PBC-aware running re-imaging, the
//                  algorithm of reference k_find_block_bounds, cpp/src/kernels/k_neighborlist.cuh:11-116)
//   CSR lists      per row block a contiguous segment in `col_atoms` holding every column atom that is
//                  within `cutoff` (the list cutoff = cutoff + padding) of at least one row atom -- the same
//                  set the reference compacts into 32-wide `ixn_atoms` tiles (k_neighborlist.cuh:199-458),
//                  which is what Neighborlist.get_nblist must report (tests/test_nblist.py:180-186)
//   work items     {row_block, col_start, col_count<=64} for the tile kernel
//
// Differences from the reference's CUDA structure, on purpose: one 256-thread workgroup (4 waves) per row block
// scans ALL column blocks (the reference launches a (row, col/32) grid of 32-thread blocks and needs a second
// "trim compaction" kernel); 64-wide ballots; segments are claimed from the pool with one atomic per row block
// after the coarse pass, so a row block's columns are contiguous and no per-tile atomics are issued.
#pragma once
#include "kernels_nonbonded.cuh"

namespace tmamd {

// K2: block bounds (thread per block: the fold is inherently sequential), coordinate snapshot and counter reset.
template <typename Real>
__global__ void k_block_bounds(
    const int n_col_blocks, const int NC, const unsigned int *__restrict__ col_idxs, // nullptr => identity
    const int n_row_blocks, const int NR, const unsigned int *__restrict__ row_idxs, // only used when rows != cols
    const int rows_equal_cols, const Real *__restrict__ gathered, const double *__restrict__ box,
    Real *__restrict__ col_ctr, Real *__restrict__ col_ext, Real *__restrict__ row_ctr, Real *__restrict__ row_ext,
    unsigned int *__restrict__ counters, // [0]=pool cursor [1]=n_items [2]=tile count
    const int n_snap, const double *__restrict__ x, double *__restrict__ snap_x, double *__restrict__ snap_box,
    const int *__restrict__ flag, const int force) {
    if (!force && *flag == 0) {
        return;
    }
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nthreads = gridDim.x * blockDim.x;
    if (tid == 0) {
        counters[0] = 0;
        counters[1] = 0;
        counters[2] = 0;
    }
    if (snap_x) {
        for (int t = tid; t < n_snap; t += nthreads) {
            snap_x[t] = x[t];
        }
        if (tid < 9) {
            snap_box[tid] = box[tid];
        }
    }
    const int total_blocks = n_col_blocks + (rows_equal_cols ? 0 : n_row_blocks);
    if (tid >= total_blocks) {
        return;
    }
    const bool is_row = tid >= n_col_blocks;
    const int blk = is_row ? tid - n_col_blocks : tid;
    const unsigned int *idxs = is_row ? row_idxs : col_idxs;
    const int count = is_row ? NR : NC;
    const NbBox<Real> bx = load_box<Real>(box);
    const Real half = static_cast<Real>(0.5);

    const int first = blk * TILE;
    const int n = (count - first) < TILE ? (count - first) : TILE;
    Real lo[3], hi[3];
    {
        const unsigned int a = idxs ? idxs[first] : static_cast<unsigned int>(first);
        for (int d = 0; d < 3; d++) {
            lo[d] = hi[d] = gathered[static_cast<size_t>(a) * 8 + d];
        }
    }
    // visiting order of the reference's lane rotation: atoms 1, 2, ..., n-1, then atom 0 again
    for (int k = 1; k <= n; k++) {
        const int kk = k == n ? 0 : k;
        const unsigned int a = idxs ? idxs[first + kk] : static_cast<unsigned int>(first + kk);
        const Real p[3] = {gathered[static_cast<size_t>(a) * 8 + 0], gathered[static_cast<size_t>(a) * 8 + 1], gathered[static_cast<size_t>(a) * 8 + 2]};
        const Real b[3] = {bx.x, bx.y, bx.z};
        const Real ib[3] = {bx.inv_x, bx.inv_y, bx.inv_z};
        for (int d = 0; d < 3; d++) {
            const Real img = p[d] - b[d] * nearbyint((p[d] - half * (hi[d] + lo[d])) * ib[d]);
            lo[d] = min(lo[d], img);
            hi[d] = max(hi[d], img);
        }
    }
    Real *ctr = is_row ? row_ctr : col_ctr;
    Real *ext = is_row ? row_ext : col_ext;
    for (int d = 0; d < 3; d++) {
        ctr[blk * 3 + d] = half * (hi[d] + lo[d]);
        ext[blk * 3 + d] = half * (hi[d] - lo[d]);
    }
}

// K3: per row block, find interacting column atoms.
//   dynamic LDS: ceil(n_col_blocks / 64) 64-bit words holding the coarse (bbox-bbox) pass bitmap.
template <typename Real, bool UPPER_TRIANGULAR>
__global__ __launch_bounds__(256) void k_find_ixns(
    const int K, const int NC, const int NR, const unsigned int *__restrict__ col_idxs, const unsigned int *__restrict__ row_idxs,
    const Real *__restrict__ col_ctr, const Real *__restrict__ col_ext, const Real *__restrict__ row_ctr,
    const Real *__restrict__ row_ext, const Real *__restrict__ gathered, const double *__restrict__ box, const double cutoff_d,
    unsigned int *__restrict__ counters, unsigned int *__restrict__ col_atoms, int4 *__restrict__ items,
    int2 *__restrict__ row_segments, // per row block {start, count}
    const int *__restrict__ flag, const int force) {
    if (!force && *flag == 0) {
        return;
    }
    extern __shared__ u64 s_bitmap[];
    __shared__ Real s_rx[TILE], s_ry[TILE], s_rz[TILE];
    __shared__ unsigned int s_npass, s_count, s_seg_start, s_item_base;
    __shared__ int s_nrow;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int rb = blockIdx.x;
    const int n_col_blocks = (NC + TILE - 1) / TILE;
    const int n_words = (n_col_blocks + 63) / 64;
    const NbBox<Real> bx = load_box<Real>(box);
    const Real cutoff = static_cast<Real>(cutoff_d);
    const Real cutoff2 = cutoff * cutoff;

    if (tid == 0) {
        s_npass = 0;
        s_count = 0;
        const int rem = NR - rb * TILE;
        s_nrow = rem < TILE ? rem : TILE;
    }
    if (tid < TILE) {
        const int ridx = rb * TILE + tid;
        if (ridx < NR) {
            const unsigned int a = row_idxs ? row_idxs[ridx] : static_cast<unsigned int>(ridx);
            s_rx[tid] = gathered[static_cast<size_t>(a) * 8 + 0];
            s_ry[tid] = gathered[static_cast<size_t>(a) * 8 + 1];
            s_rz[tid] = gathered[static_cast<size_t>(a) * 8 + 2];
        } else {
            s_rx[tid] = s_ry[tid] = s_rz[tid] = 0;
        }
    }
    __syncthreads();

    // ---- coarse pass: row bbox vs every column bbox (k_neighborlist.cuh:296-329)
    const Real rcx = row_ctr[rb * 3 + 0], rcy = row_ctr[rb * 3 + 1], rcz = row_ctr[rb * 3 + 2];
    const Real rex = row_ext[rb * 3 + 0], rey = row_ext[rb * 3 + 1], rez = row_ext[rb * 3 + 2];
    unsigned int my_pass = 0;
    for (int w = wave; w < n_words; w += 4) {
        const int cb = w * 64 + lane;
        bool pass = cb < n_col_blocks && (!UPPER_TRIANGULAR || cb >= rb);
        if (pass) {
            Real ddx = min_image(rcx - col_ctr[cb * 3 + 0], bx.x, bx.inv_x);
            Real ddy = min_image(rcy - col_ctr[cb * 3 + 1], bx.y, bx.inv_y);
            Real ddz = min_image(rcz - col_ctr[cb * 3 + 2], bx.z, bx.inv_z);
            ddx = max(static_cast<Real>(0), fabs(ddx) - rex - col_ext[cb * 3 + 0]);
            ddy = max(static_cast<Real>(0), fabs(ddy) - rey - col_ext[cb * 3 + 1]);
            ddz = max(static_cast<Real>(0), fabs(ddz) - rez - col_ext[cb * 3 + 2]);
            pass = (ddx * ddx + ddy * ddy + ddz * ddz) < cutoff2;
        }
        const u64 m = __ballot(pass);
        if (lane == 0) {
            s_bitmap[w] = m;
            my_pass += __popcll(m);
        }
    }
    if (lane == 0 && my_pass) {
        atomicAdd(&s_npass, my_pass);
    }
    __syncthreads();
    if (tid == 0) {
        // one pool claim per row block, sized by the coarse upper bound (never more than 32 * passing blocks)
        s_seg_start = atomicAdd(&counters[0], s_npass * TILE);
    }
    __syncthreads();
    const unsigned int seg_start = s_seg_start;
    const int nrow = s_nrow;

    // ---- fine pass: two column blocks per wave iteration (lanes 0-31 / 32-63)
    for (int w = wave; w < n_words; w += 4) {
        u64 m = s_bitmap[w];
        while (m) {
            const int a0 = __builtin_ctzll(m);
            m &= m - 1;
            int a1 = -1;
            if (m) {
                a1 = __builtin_ctzll(m);
                m &= m - 1;
            }
            const int sel = lane < 32 ? a0 : a1;
            unsigned int ja = K;
            if (sel >= 0) {
                const int jpos = (w * 64 + sel) * TILE + (lane & 31);
                if (jpos < NC) {
                    ja = col_idxs ? col_idxs[jpos] : static_cast<unsigned int>(jpos);
                }
            }
            bool interacts = false;
            if (ja < static_cast<unsigned int>(K)) {
                const Real xj = gathered[static_cast<size_t>(ja) * 8 + 0];
                const Real yj = gathered[static_cast<size_t>(ja) * 8 + 1];
                const Real zj = gathered[static_cast<size_t>(ja) * 8 + 2];
                for (int i = 0; i < nrow && !interacts; i++) {
                    const Real dx = min_image(s_rx[i] - xj, bx.x, bx.inv_x);
                    const Real dy = min_image(s_ry[i] - yj, bx.y, bx.inv_y);
                    const Real dz = min_image(s_rz[i] - zj, bx.z, bx.inv_z);
                    interacts = (dx * dx + dy * dy + dz * dz) < cutoff2;
                }
            }
            const u64 hits = __ballot(interacts);
            if (hits) {
                unsigned int base = 0;
                if (lane == 0) {
                    base = atomicAdd(&s_count, static_cast<unsigned int>(__popcll(hits)));
                }
                base = __shfl(base, 0, 64);
                if (interacts) {
                    col_atoms[seg_start + base + __popcll(hits & ((1ull << lane) - 1ull))] = ja;
                }
            }
        }
    }
    __syncthreads();

    // ---- publish the segment and its work items
    const unsigned int count = s_count;
    const unsigned int n_chunks = (count + NB_CHUNK - 1) / NB_CHUNK;
    if (tid == 0) {
        row_segments[rb] = make_int2(static_cast<int>(seg_start), static_cast<int>(count));
        atomicAdd(&counters[2], (count + TILE - 1) / TILE);
        s_item_base = n_chunks ? atomicAdd(&counters[1], n_chunks) : 0;
    }
    __syncthreads();
    const unsigned int item_base = s_item_base;
    for (unsigned int c = tid; c < n_chunks; c += blockDim.x) {
        const unsigned int off = c * NB_CHUNK;
        const unsigned int len = (count - off) < NB_CHUNK ? (count - off) : NB_CHUNK;
        items[item_base + c] = make_int4(rb, static_cast<int>(seg_start + off), static_cast<int>(len), 0);
    }
}

} // namespace tmamd
