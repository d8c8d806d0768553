// Nonbonded device kernels for gfx950 (wave64, LDS pair queue).  Included by nonbonded.hip only.
//
// Data layout in HBM (K = number of interacting atoms, Hilbert order):
//   gathered      Real[K][8]  = {x, y, z, w, q, sig, eps, 0}   one 64 B (f64) / 32 B (f32) record per atom,
//                               cast to Real once at gather time (reference casts at load: k_nonbonded.cuh:134-151)
//   g_du_dx       u64 [K][3]   fixed-point force accumulators in Hilbert order (contiguous per-tile flushes)
//   g_du_dp       u64 [K][4]
//   col_atoms     u32 pool     per row block a contiguous segment of interacting column atoms (CSR)
//   items         int4         work items {row_block, col_start, col_count, 0}: 32 rows x <=64 columns each
//
// Tile kernel design (one wave = one 64-thread workgroup per work item, grid-stride):
//   phase 1  every lane owns one column atom in registers; 32 rounds, lane l meets row (round + l) & 31
//            (rotation => the rows, and the columns, hit in one round are all distinct); the cheap distance test
//            (`d2 < cutoff^2`, strict) is evaluated for all 32x64 slots; survivors are compacted with
//            ballot + popcount into an LDS queue of (row, col) byte pairs.
//   phase 2  whenever >= 64 pairs are queued, all 64 lanes pop one pair each and run the expensive
//            erfc/exp/sincos path at full lane occupancy (the reference leaves ~2/3 of the lanes idle inside
//            its `if (d2 < cutoff^2)` branch); results are converted to fixed point and added with LDS u64
//            atomics into per-tile row / column accumulators.
//   flush    one global u64 atomic per touched (atom, component) per tile.
// Integer accumulation is associative, so none of this reordering changes a single bit of the result.
#pragma once
#include "nb_pair.cuh"

namespace tmamd {

static const int NB_CHUNK = 64; // columns per work item == wave width

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
template <typename Real> __device__ __forceinline__ Real pair_d2(Real dx, Real dy, Real dz, Real dw) {
    return dx * dx + dy * dy + dz * dz + dw * dw;
}

// ---- K1: rebuild check + gather (+ zero the Hilbert-order accumulators) -------------------------------------
// reference: k_check_rebuild_coords_and_box_gather + k_gather_coords_and_params (k_nonbonded.cuh:12-84)
template <typename Real>
__global__ void k_check_gather(
    const int K, const unsigned int *__restrict__ perm, const double *__restrict__ x, const double *__restrict__ p,
    const double *__restrict__ box, const double *__restrict__ snap_x, const double *__restrict__ snap_box,
    const double pad2_quarter, // 0.25 * padding^2
    int *__restrict__ flag_set, int *__restrict__ flag_clear, Real *__restrict__ gathered, u64 *__restrict__ g_du_dx,
    u64 *__restrict__ g_du_dp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0) {
        *flag_clear = 0; // the flag the NEXT call will use; its consumers finished a call ago (stream order)
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
    g[0] = xn;
    g[1] = yn;
    g[2] = zn;
    g[3] = static_cast<Real>(p[a * 4 + 3]); // w
    g[4] = static_cast<Real>(p[a * 4 + 0]); // q
    g[5] = static_cast<Real>(p[a * 4 + 1]); // sig
    g[6] = static_cast<Real>(p[a * 4 + 2]); // eps
    g[7] = 0;
    if (g_du_dx) {
        g_du_dx[idx * 3 + 0] = 0;
        g_du_dx[idx * 3 + 1] = 0;
        g_du_dx[idx * 3 + 2] = 0;
    }
    if (g_du_dp) {
        g_du_dp[idx * 4 + 0] = 0;
        g_du_dp[idx * 4 + 1] = 0;
        g_du_dp[idx * 4 + 2] = 0;
        g_du_dp[idx * 4 + 3] = 0;
    }
}

// ---- K5: un-permute (reference: k_scatter_accum, k_nonbonded.cuh:86-104) ------------------------------------
template <int D>
__global__ void k_scatter_accum(const int K, const unsigned int *__restrict__ perm, const u64 *__restrict__ g, u64 *__restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= K * D) {
        return;
    }
    const int a = idx / D, d = idx - a * D;
    const u64 v = g[idx];
    if (v != 0) {
        atomicAdd(out + static_cast<size_t>(perm[a]) * D + d, v);
    }
}

// ---- K4: the tile kernel ------------------------------------------------------------------------------------
template <typename Real, bool COMPUTE_U, bool COMPUTE_DU_DX, bool COMPUTE_DU_DP>
__global__ __launch_bounds__(64, 2) void k_nonbonded_tiles(
    const int K,                               // atoms in `gathered` (sentinel index == K)
    const int NR,                              // number of row atoms
    const int upper_triangular,                // rows == cols == all: keep only row < col
    const unsigned int *__restrict__ row_idxs, // [NR] or nullptr (identity)
    const unsigned int *__restrict__ n_items_ptr, const int4 *__restrict__ items, const unsigned int *__restrict__ col_atoms,
    const Real *__restrict__ gathered, const double *__restrict__ box, const double beta_d, const double cutoff_d,
    u64 *__restrict__ g_du_dx, u64 *__restrict__ g_du_dp, i128 *__restrict__ u_partials) {

    __shared__ Real s_row[7][TILE];
    __shared__ Real s_col[7][NB_CHUNK];
    __shared__ unsigned int s_rowatom[TILE];
    __shared__ u64 s_fi[COMPUTE_DU_DX ? 3 : 1][TILE];
    __shared__ u64 s_fj[COMPUTE_DU_DX ? 3 : 1][NB_CHUNK];
    __shared__ u64 s_pi[COMPUTE_DU_DP ? 4 : 1][TILE];
    __shared__ u64 s_pj[COMPUTE_DU_DP ? 4 : 1][NB_CHUNK];
    __shared__ unsigned short s_queue[2 * NB_CHUNK];

    const int lane = threadIdx.x;
    const NbBox<Real> bx = load_box<Real>(box);
    const Real cutoff = static_cast<Real>(cutoff_d);
    const Real cutoff2 = cutoff * cutoff;
    const Real beta = static_cast<Real>(beta_d);
    i128 energy = 0;

    const unsigned int n_items = *n_items_ptr;
    for (unsigned int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int4 it = items[item];
        const int rb = it.x, cstart = it.y, ccount = it.z;

        __syncthreads(); // previous item's flush has finished reading LDS
        if (lane < TILE) {
            const int ridx = rb * TILE + lane;
            unsigned int ra = K;
            if (ridx < NR) {
                ra = row_idxs ? row_idxs[ridx] : static_cast<unsigned int>(ridx);
            }
            s_rowatom[lane] = ra;
#pragma unroll
            for (int c = 0; c < 7; c++) {
                s_row[c][lane] = ra < static_cast<unsigned int>(K) ? gathered[static_cast<size_t>(ra) * 8 + c] : static_cast<Real>(0);
            }
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
        const unsigned int ja = lane < ccount ? col_atoms[cstart + lane] : static_cast<unsigned int>(K);
        Real cj[7];
#pragma unroll
        for (int c = 0; c < 7; c++) {
            cj[c] = ja < static_cast<unsigned int>(K) ? gathered[static_cast<size_t>(ja) * 8 + c] : static_cast<Real>(0);
            s_col[c][lane] = cj[c];
        }
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
        __syncthreads();

        int cnt = 0; // wave-uniform number of queued pairs
        for (int round = 0; round < TILE; round++) {
            // ---- phase 1: one distance test per lane
            const int i = (round + lane) & (TILE - 1);
            const unsigned int ia = s_rowatom[i];
            const Real dx = min_image(s_row[0][i] - cj[0], bx.x, bx.inv_x);
            const Real dy = min_image(s_row[1][i] - cj[1], bx.y, bx.inv_y);
            const Real dz = min_image(s_row[2][i] - cj[2], bx.z, bx.inv_z);
            const Real dw = s_row[3][i] - cj[3];
            const Real d2 = pair_d2(dx, dy, dz, dw);
            const bool valid = ia < static_cast<unsigned int>(K) && ja < static_cast<unsigned int>(K) && (!upper_triangular || ia < ja);
            const bool hit = valid && d2 < cutoff2; // strict: atoms with w == cutoff never interact
            const u64 mask = __ballot(hit);
            if (hit) {
                const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
                s_queue[pos] = static_cast<unsigned short>((i << 8) | lane);
            }
            cnt += __popcll(mask);

            // ---- phase 2: drain full batches (and everything on the last round)
            const bool last = round == TILE - 1;
            while (cnt >= NB_CHUNK || (last && cnt > 0)) {
                const int n = cnt < NB_CHUNK ? cnt : NB_CHUNK;
                const int base = cnt - n;
                __syncthreads();
                if (lane < n) {
                    const unsigned int e = s_queue[base + lane];
                    const int pi = e >> 8, pj = e & 0xff;
                    const Real ddx = min_image(s_row[0][pi] - s_col[0][pj], bx.x, bx.inv_x);
                    const Real ddy = min_image(s_row[1][pi] - s_col[1][pj], bx.y, bx.inv_y);
                    const Real ddz = min_image(s_row[2][pi] - s_col[2][pj], bx.z, bx.inv_z);
                    const Real ddw = s_row[3][pi] - s_col[3][pj];
                    const Real dd2 = pair_d2(ddx, ddy, ddz, ddw);
                    const Real qi = s_row[4][pi], qj = s_col[4][pj];
                    const Real eps_i = s_row[6][pi], eps_j = s_col[6][pj];
                    PairOut<Real> o;
                    nb_pair<Real>(1, 1, qi, qj, s_row[5][pi], s_col[5][pj], eps_i, eps_j, dd2, beta, o);
                    if constexpr (COMPUTE_DU_DX) {
                        atomicAdd(&s_fi[0][pi], float_to_fixed<Real>(o.prefactor * ddx));
                        atomicAdd(&s_fi[1][pi], float_to_fixed<Real>(o.prefactor * ddy));
                        atomicAdd(&s_fi[2][pi], float_to_fixed<Real>(o.prefactor * ddz));
                        atomicAdd(&s_fj[0][pj], float_to_fixed<Real>(-o.prefactor * ddx));
                        atomicAdd(&s_fj[1][pj], float_to_fixed<Real>(-o.prefactor * ddy));
                        atomicAdd(&s_fj[2][pj], float_to_fixed<Real>(-o.prefactor * ddz));
                    }
                    if constexpr (COMPUTE_DU_DP) {
                        atomicAdd(&s_pi[0][pi], float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DCHARGE>(qj * o.inv_dij * o.ebd));
                        atomicAdd(&s_pj[0][pj], float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DCHARGE>(qi * o.inv_dij * o.ebd));
                        if (o.has_lj) {
                            const u64 sg = float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DSIG>(o.sig_grad);
                            atomicAdd(&s_pi[1][pi], sg);
                            atomicAdd(&s_pj[1][pj], sg);
                            atomicAdd(&s_pi[2][pi], float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DEPS>(o.eps_grad * eps_j));
                            atomicAdd(&s_pj[2][pj], float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DEPS>(o.eps_grad * eps_i));
                        }
                        atomicAdd(&s_pi[3][pi], float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DW>(o.prefactor * ddw));
                        atomicAdd(&s_pj[3][pj], float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DW>(-o.prefactor * ddw));
                    }
                    if constexpr (COMPUTE_U) {
                        energy += float_to_fixed_energy<Real>(o.u);
                    }
                }
                cnt = base;
            }
        }
        __syncthreads();

        // ---- flush: one global atomic per touched (atom, component)
        if constexpr (COMPUTE_DU_DX) {
            for (int t = lane; t < TILE * 3; t += 64) {
                const int a = t / 3, c = t - a * 3;
                const u64 v = s_fi[c][a];
                const unsigned int ra = s_rowatom[a];
                if (v != 0 && ra < static_cast<unsigned int>(K)) {
                    atomicAdd(g_du_dx + static_cast<size_t>(ra) * 3 + c, v);
                }
            }
            if (ja < static_cast<unsigned int>(K)) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const u64 v = s_fj[c][lane];
                    if (v != 0) {
                        atomicAdd(g_du_dx + static_cast<size_t>(ja) * 3 + c, v);
                    }
                }
            }
        }
        if constexpr (COMPUTE_DU_DP) {
            for (int t = lane; t < TILE * 4; t += 64) {
                const int a = t >> 2, c = t & 3;
                const u64 v = s_pi[c][a];
                const unsigned int ra = s_rowatom[a];
                if (v != 0 && ra < static_cast<unsigned int>(K)) {
                    atomicAdd(g_du_dp + static_cast<size_t>(ra) * 4 + c, v);
                }
            }
            if (ja < static_cast<unsigned int>(K)) {
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const u64 v = s_pj[c][lane];
                    if (v != 0) {
                        atomicAdd(g_du_dp + static_cast<size_t>(ja) * 4 + c, v);
                    }
                }
            }
        }
    }

    if constexpr (COMPUTE_U) {
        const i128 total = wave_sum_i128(energy);
        if (lane == 0) {
            u_partials[blockIdx.x] = total;
        }
    }
}

// ---- pair-list kernel (NonbondedPairList / NonbondedExclusions) ----------------------------------------------
// reference: k_nonbonded_pair_list (k_nonbonded_pair_list.cuh:18-190).  One thread per listed pair, same nb_pair().
template <typename Real, bool NEGATED>
__global__ __launch_bounds__(256) void k_nonbonded_pair_list(
    const int M, const double *__restrict__ coords, const double *__restrict__ params, const double *__restrict__ box,
    const int *__restrict__ pair_idxs, const double *__restrict__ scales, const double beta_d, const double cutoff_d,
    u64 *__restrict__ du_dx, u64 *__restrict__ du_dp, i128 *__restrict__ u_partials) {
    const int pair = blockIdx.x * blockDim.x + threadIdx.x;
    i128 energy = 0;
    if (pair < M) {
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
            nb_pair<Real>(charge_scale, lj_scale, qi, qj, sig_i, sig_j, eps_i, eps_j, d2, static_cast<Real>(beta_d), o);
#define TM_ACC(ptr, val)                                                                                               \
    do {                                                                                                               \
        const u64 v_ = (val);                                                                                          \
        atomicAdd((ptr), NEGATED ? (0ull - v_) : v_);                                                                  \
    } while (0)
            if (du_dx) {
                TM_ACC(du_dx + ia * 3 + 0, float_to_fixed<Real>(o.prefactor * dx));
                TM_ACC(du_dx + ia * 3 + 1, float_to_fixed<Real>(o.prefactor * dy));
                TM_ACC(du_dx + ia * 3 + 2, float_to_fixed<Real>(o.prefactor * dz));
                TM_ACC(du_dx + ja * 3 + 0, float_to_fixed<Real>(-o.prefactor * dx));
                TM_ACC(du_dx + ja * 3 + 1, float_to_fixed<Real>(-o.prefactor * dy));
                TM_ACC(du_dx + ja * 3 + 2, float_to_fixed<Real>(-o.prefactor * dz));
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
                TM_ACC(du_dp + ia * 4 + 3, (float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DW>(o.prefactor * dw)));
                TM_ACC(du_dp + ja * 4 + 3, (float_to_fixed_exp<Real, TM_FIXED_EXPONENT_DU_DW>(-o.prefactor * dw)));
            }
#undef TM_ACC
            if (u_partials) {
                // negate the fixed-point value, not the float (k_nonbonded_pair_list.cuh:185-188)
                const i128 e = float_to_fixed_energy<Real>(o.u);
                energy = NEGATED ? -e : e;
            }
        }
    }
    if (u_partials) {
        // per-wave partial sums instead of one 16-byte store per pair
        const i128 total = wave_sum_i128(energy);
        if ((threadIdx.x & 63) == 0) {
            u_partials[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = total;
        }
    }
}

} // namespace tmamd
