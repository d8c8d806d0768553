// The rebuild test of a neighbor list whose box may have changed by small factors since the list was built (a barostat at work),
// as ONE device function: used by the potential's own check (k_check_gather_scaled states the same rule with a re-basing of the
// snapshot on top), by the integrator's update kernel when it leaves the potential's next gather done (PregatherTarget), and by
// the barostat's fast path for the geometry it proposes and for the one it commits (barostat.hip).
//
// The list was built from a snapshot x_s in a box B_s and holds every pair closer than cutoff + padding there.  For ANY later
// coordinates x in a box B = rho * B_s (rho per dimension, diagonal boxes) let e_i = minimum image in B of x_i - rho * x_s,i.  A pair
// with image vector v in (x, B), |v| < cutoff, has the snapshot image vector w with rho * w = v - e_i + e_j, so
// |w| <= (cutoff + 2 max|e|) / rho_min: the list is complete as long as max|e| < (rho_min (cutoff + padding) - cutoff) / 2, which
// with |rho - 1| <= NB_SCALE_MAX holds whenever max|e| < padding / 2 - NB_SCALE_MAX (cutoff + padding) / 2 =: D
// (NonbondedAllPairs::rebuild_threshold2() = D^2).  snap_box[0..8] is the box the stored snapshot is expressed in, snap_box[9..11]
// the scale accumulated between the build and that box (k_check_gather_scaled re-bases; nothing here does).
#pragma once
#include "common.hpp"

namespace tmamd {

#ifndef NB_SCALE_MAX
#define NB_SCALE_MAX 0.004
#endif

// true iff atom position (x, y, z) in `box` calls for a rebuild of the list whose snapshot of this atom is snap[0..2]
__device__ __forceinline__ bool snapshot_calls_for_rebuild(
    const double x, const double y, const double z, const double *__restrict__ snap, const double *__restrict__ box,
    const double *__restrict__ snap_box, const double threshold2) {
    bool same = true, scalable = true;
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
    }
    double ex = x - snap[0], ey = y - snap[1], ez = z - snap[2];
    if (!same) { // (wave-uniform: the box is the launch's)
        // hardware reciprocals (1 ulp) instead of six f64 divisions per atom: this test decides WHEN a list is rebuilt, never a
        // result, and both the scale bound and the threshold carry orders of magnitude more slack than 1e-16 relative
        double e[3];
        const double xyz[3] = {x, y, z};
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const double b = box[d * 4];
            const double rho = b * __builtin_amdgcn_rcp(snap_box[d * 4]);
            scalable = scalable && fabs(rho * snap_box[9 + d] - 1.0) <= NB_SCALE_MAX; // (false for NaN / inf: an uninitialised snapshot)
            e[d] = xyz[d] - rho * snap[d];
            e[d] -= b * rint(e[d] * __builtin_amdgcn_rcp(b));
        }
        if (!scalable) {
            return true;
        }
        ex = e[0];
        ey = e[1];
        ez = e[2];
    }
    return ex * ex + ey * ey + ez * ez > threshold2;
}

} // namespace tmamd
