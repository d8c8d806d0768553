// ForcePlan: all bonded terms and pair lists of a step in one kernel launch per precision (see engine.hpp).
// The per-term device functions are the ones the stand-alone kernels call (kernels_bonded.hip.hpp, kernels_nonbonded.hip.hpp),
// so a fused launch produces the same bits as separate launches.
#include <cstring>

#include "kernels_bonded.hip.hpp"
#include "kernels_nonbonded.hip.hpp"
#include "profiler.hpp"

namespace tmamd {

template <typename Real>
__global__ __launch_bounds__(256) void k_fused_forces(
    const FusedTable *__restrict__ table, const double *__restrict__ coords, const double *__restrict__ box, u64 *__restrict__ du_dx,
    const ForceLayout fl) {
    fused_dispatch<Real>(table, static_cast<int>(blockIdx.x), static_cast<int>(threadIdx.x), coords, box, du_dx, fl);
}

void ForcePlan::clear() {
    host_[0].n = 0;
    host_[1].n = 0;
    rest_.clear();
}

void ForcePlan::add_segment(const int precision_bytes, const FusedSegment &seg, Potential *owner, const int P, const double *d_p) {
    FusedTable &t = host_[precision_bytes == 8 ? 1 : 0];
    if (t.n >= FUSED_MAX_SEGMENTS) {
        rest_.push_back({owner, P, d_p}); // table full: this one runs on its own
        return;
    }
    const int blocks = ceil_divide(seg.count, 256);
    t.block_end[t.n] = (t.n ? t.block_end[t.n - 1] : 0) + blocks;
    t.seg[t.n] = seg;
    t.n++;
}

bool ForcePlan::run(
    const int N, const double *d_x, const double *d_box, u64 *d_du_dx, hipStream_t stream, std::vector<DeferredForces> *deferred,
    const int max_deferred, u64 *d_du_dx_cm, const int cm_stride) {
    bool wrote_du_dx = false; // did anything add to the [N, 3] array?
    // the table's terms go to the caller's component-major accumulator when there is one (lanes working on neighbouring
    // atoms then share cache lines: fewer line requests for the memory-side atomics), to the [N, 3] array otherwise
    u64 *table_acc = d_du_dx_cm ? d_du_dx_cm : d_du_dx;
    const ForceLayout table_fl = d_du_dx_cm ? ForceLayout{1, cm_stride} : ForceLayout{3, 1};
    // 1. tables to the device (only when they changed since the last step)
    bool pending[2] = {false, false};
    bool table_went_to_acc = false; // a table's forces were (or will be) added to table_acc
    for (int prec = 0; prec < 2; prec++) {
        FusedTable &t = host_[prec];
        if (t.n == 0) {
            continue;
        }
        pending[prec] = true;
        // unused tail of the table: keep it deterministic so the "unchanged since the last upload" test is exact
        for (int k = t.n; k < FUSED_MAX_SEGMENTS; k++) {
            t.block_end[k] = 0;
            t.seg[k] = FusedSegment{0, 0, nullptr, nullptr, nullptr, 0.0, 0.0};
        }
        if (!uploaded_valid_[prec] || std::memcmp(&uploaded_[prec], &t, sizeof(FusedTable)) != 0) {
            d_table_[prec].reserve(1);
            // uploads happen only when the table changed (potentials added / re-bound), never on a plain MD step: copy
            // from the snapshot, in stream order, and wait -- host_ is rewritten by the next clear()/add_segment, and
            // whether an async copy from pageable memory has been staged by the time the call returns is not guaranteed
            std::memcpy(&uploaded_[prec], &t, sizeof(FusedTable));
            HIP_CHECK(hipMemcpyAsync(d_table_[prec].data, &uploaded_[prec], sizeof(FusedTable), hipMemcpyHostToDevice, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            uploaded_valid_[prec] = true;
        }
    }
    // 2. a long-running force kernel of the same precision may take a table along (its early-finishing waves run it)
    for (const Rest &r : rest_) {
        for (int prec = 0; prec < 2; prec++) {
            if (pending[prec] &&
                r.pot->piggyback_forces(d_table_[prec].data, host_[prec].block_end[host_[prec].n - 1], prec ? 8 : 4, table_acc, table_fl.atom, table_fl.comp)) {
                pending[prec] = false;
                table_went_to_acc = table_went_to_acc || !r.pot->piggyback_lands_in_own_accumulator();
            }
        }
    }
    // 3. the potentials that launch their own kernels
    for (const Rest &r : rest_) {
        DeferredForces df;
        // a potential bound more than once is never deferred: its other call would zero and reuse the accumulator this
        // one hands over before the consumer has read it
        bool shared = false;
        for (const Rest &q : rest_) {
            shared = shared || (&q != &r && q.pot == r.pot);
        }
        if (!shared && deferred != nullptr && static_cast<int>(deferred->size()) < max_deferred &&
            r.pot->execute_forces_deferred(N, r.P, d_x, r.d_p, d_box, d_du_dx, stream, df)) {
            deferred->push_back(df);
        } else {
            r.pot->execute_device(N, r.P, d_x, r.d_p, d_box, d_du_dx, nullptr, nullptr, stream);
            wrote_du_dx = true;
        }
    }
    // 4. tables nobody took
    for (int prec = 0; prec < 2; prec++) {
        if (!pending[prec]) {
            continue;
        }
        table_went_to_acc = true;
        const int blocks = host_[prec].block_end[host_[prec].n - 1];
        const int prof = Profiler::get().begin("fused_forces", stream);
        if (prec == 1) {
            k_fused_forces<double><<<blocks, 256, 0, stream>>>(d_table_[prec].data, d_x, d_box, table_acc, table_fl);
        } else {
            k_fused_forces<float><<<blocks, 256, 0, stream>>>(d_table_[prec].data, d_x, d_box, table_acc, table_fl);
        }
        HIP_CHECK(hipGetLastError());
        Profiler::get().end("fused_forces", prof, stream);
    }
    // the table's terms (piggy-backed or launched above) went to table_acc -- unless a potential took them into its own
    // accumulator
    cm_written_ = table_went_to_acc && table_acc == d_du_dx_cm && d_du_dx_cm != nullptr;
    return wrote_du_dx || (table_went_to_acc && table_acc == d_du_dx);
}

} // namespace tmamd
