// ForcePlan: all bonded terms and pair lists of a step in one kernel launch per precision (see engine.hpp).
// The per-term device functions are the ones the stand-alone kernels call (kernels_bonded.hip.hpp, kernels_nonbonded.hip.hpp),
// so a fused launch produces the same bits as separate launches.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "kernels_bonded.hip.hpp"
#include "kernels_nonbonded.hip.hpp"
#include "profiler.hpp"

namespace tmamd {

template <typename Real>
__global__ __launch_bounds__(256) void k_fused_forces(
    const FusedTable *__restrict__ table, const double *__restrict__ coords, const double *__restrict__ box, u64 *__restrict__ du_dx,
    const ForceLayout fl) {
    __shared__ u64 s_window[4][3 * FORCE_WINDOW]; // one accumulation window per wave (kernels_bonded.hip.hpp: ForceLayout::win)
    __shared__ int s_window_rows[4][FORCE_WINDOW];
    fused_dispatch<Real>(table, static_cast<int>(blockIdx.x), static_cast<int>(threadIdx.x), coords, box, du_dx, fl, (lds_u64_ptr)(&s_window[threadIdx.x >> 6][0]),
                         (lds_int_ptr)(&s_window_rows[threadIdx.x >> 6][0]));
}

// every wave of every block writes its partial sum (zero included): the buffer needs no clearing
template <typename Real>
__global__ __launch_bounds__(256) void k_fused_energy(
    const FusedTable *__restrict__ table, const double *__restrict__ coords, const double *__restrict__ box, i128 *__restrict__ partials,
    i128 *__restrict__ zero_slots, const int n_zero_slots) {
    if (blockIdx.x == 0 && static_cast<int>(threadIdx.x) < n_zero_slots) {
        zero_slots[threadIdx.x] = 0; // slots of potentials that run after this launch and may leave theirs untouched
    }
    const i128 e = fused_dispatch_energy<Real>(table, static_cast<int>(blockIdx.x), static_cast<int>(threadIdx.x), coords, box);
    store_wave_energy<Real>(e, partials);
}

// final reduction over up to ENERGY_MAX_SOURCES arrays of partial sums (one workgroup: a few thousand values at most)
static const int ENERGY_MAX_SOURCES = 12;
struct EnergySources {
    int n;
    const i128 *p[ENERGY_MAX_SOURCES];
    int count[ENERGY_MAX_SOURCES];
};
__global__ __launch_bounds__(256) void k_reduce_i128_sources(const EnergySources src, i128 *__restrict__ out) {
    __shared__ i128 s_part[4];
    i128 acc = 0;
    for (int k = 0; k < src.n; k++) {
        const i128 *__restrict__ in = src.p[k];
        for (int i = threadIdx.x; i < src.count[k]; i += 256) {
            acc += in[i];
        }
    }
    acc = wave_sum_i128(acc);
    if ((threadIdx.x & 63) == 0) {
        s_part[threadIdx.x >> 6] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    }
}

ForcePlan::~ForcePlan() {
    if (h_ring_ != nullptr) {
        for (int k = 0; k < TABLE_RING; k++) {
            (void)hipEventDestroy(ring_ev_[k]);
        }
        (void)hipHostFree(h_ring_);
    }
}

void ForcePlan::clear() {
    static const int window = std::getenv("TM_AMD_NO_FUSED_WINDOW") == nullptr ? 1 : 0;
    host_[0].n = 0;
    host_[1].n = 0;
    host_[0].window = window;
    host_[1].window = window;
    host_[0].num_atoms = 0; // set by run() / run_energy(), which know N
    host_[1].num_atoms = 0;
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
    t.seg[t.n].window = (owner != nullptr && owner->max_atom_incidence() < FUSED_WINDOW_MIN_INCIDENCE) ? 0 : 1;
    t.n++;
}

void ForcePlan::upload_tables(bool pending[2], hipStream_t stream) {
    for (int prec = 0; prec < 2; prec++) {
        FusedTable &t = host_[prec];
        pending[prec] = false;
        if (t.n == 0) {
            continue;
        }
        pending[prec] = true;
        // unused tail of the table: keep it deterministic so the "unchanged since the last upload" test is exact
        for (int k = t.n; k < FUSED_MAX_SEGMENTS; k++) {
            t.block_end[k] = 0;
            t.seg[k] = FusedSegment{0, 0, nullptr, nullptr, nullptr, 0.0, 0.0};
            t.seg[k].window = 0;
        }
        if (!uploaded_valid_[prec] || std::memcmp(&uploaded_[prec], &t, sizeof(FusedTable)) != 0) {
            d_table_[prec].reserve(1);
            // never on a plain MD step (the table is the last step's); on every evaluation of a batch over parameter sets.  The copy
            // leaves from a pinned slot of its own, in stream order, and nobody waits: host_ is rewritten by the next
            // clear()/add_segment, the slot only when the ring comes round (its event says when the copy has been made)
            if (h_ring_ == nullptr) {
                HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&h_ring_), TABLE_RING * sizeof(FusedTable), hipHostMallocDefault));
                for (int k = 0; k < TABLE_RING; k++) {
                    HIP_CHECK(hipEventCreateWithFlags(&ring_ev_[k], hipEventDisableTiming));
                    ring_used_[k] = false;
                }
            }
            const int slot = ring_pos_;
            ring_pos_ = (ring_pos_ + 1) % TABLE_RING;
            if (ring_used_[slot]) {
                HIP_CHECK(hipEventSynchronize(ring_ev_[slot]));
            }
            std::memcpy(&uploaded_[prec], &t, sizeof(FusedTable));
            std::memcpy(&h_ring_[slot], &t, sizeof(FusedTable));
            HIP_CHECK(hipMemcpyAsync(d_table_[prec].data, &h_ring_[slot], sizeof(FusedTable), hipMemcpyHostToDevice, stream));
            HIP_CHECK(hipEventRecord(ring_ev_[slot], stream));
            ring_used_[slot] = true;
            uploaded_valid_[prec] = true;
        }
    }
}

void ForcePlan::prepare_tables(const int N, hipStream_t stream, const FusedTable *d_tables[2], int blocks[2]) {
    host_[0].num_atoms = host_[1].num_atoms = N;
    bool pending[2];
    this->upload_tables(pending, stream);
    for (int prec = 0; prec < 2; prec++) {
        d_tables[prec] = pending[prec] ? d_table_[prec].data : nullptr;
        blocks[prec] = pending[prec] ? host_[prec].block_end[host_[prec].n - 1] : 0;
    }
}

void ForcePlan::merge_producers() {
    // at most one pairing per plan (the reference's states hold one all-pairs potential and one interaction group); potentials
    // planned more than once stay as they are (their second call would meet the carrier's state half-way)
    auto planned_once = [&](const Potential *p) {
        int n = 0;
        for (const Rest &q : rest_) {
            n += q.pot == p ? 1 : 0;
        }
        return n == 1;
    };
    for (size_t i = 0; i < rest_.size(); i++) {
        NonbondedAllPairsBase *host = dynamic_cast<NonbondedAllPairsBase *>(rest_[i].pot);
        if (host == nullptr || host->is_interaction_group() || !planned_once(rest_[i].pot)) {
            continue;
        }
        for (size_t j = 0; j < rest_.size(); j++) {
            NonbondedAllPairsBase *group = j == i ? nullptr : dynamic_cast<NonbondedAllPairsBase *>(rest_[j].pot);
            if (group == nullptr || !group->is_interaction_group() || !planned_once(rest_[j].pot)) {
                continue;
            }
            Potential *carrier = host->merged_carrier(group, rest_[j].P, rest_[j].d_p);
            if (carrier != nullptr) {
                rest_[i].pot = carrier; // evaluated with the all-pairs potential's own (P, d_p); the group's are bound inside
                rest_.erase(rest_.begin() + static_cast<long>(j));
                return;
            }
        }
    }
}

void ForcePlan::run_energy(const int N, const double *d_x, const double *d_box, i128 *d_u, hipStream_t stream) {
    this->merge_producers();
    host_[0].num_atoms = host_[1].num_atoms = N;
    bool pending[2];
    this->upload_tables(pending, stream);
    EnergySources src;
    src.n = 0;
    // potentials that reduce for themselves get a slot each; every slot is one value of one source
    const int n_rest = static_cast<int>(rest_.size());
    d_e_slots_.reserve(std::max(n_rest, 1));
    int n_slots = 0;
    bool slots_zeroed = false;
    // 0. a potential whose own energy launch is long (the nonbonded tile kernel) may take a table of its precision along.
    // A carrier MUST be evaluated through execute_energy_partials in step 2, so its source slot is reserved here: room is kept
    // for both tables (step 1), the carriers accepted so far and the slot array (at the end).
    std::vector<Potential *> carriers;
    for (const Rest &r : rest_) {
        bool shared = false;
        for (const Rest &q : rest_) {
            shared = shared || (&q != &r && q.pot == r.pot);
        }
        for (int prec = 0; prec < 2 && !shared && 2 + static_cast<int>(carriers.size()) + 1 < ENERGY_MAX_SOURCES - 1; prec++) {
            if (pending[prec] && r.pot->piggyback_energy(d_table_[prec].data, host_[prec].block_end[host_[prec].n - 1], prec ? 8 : 4)) {
                pending[prec] = false;
                if (std::find(carriers.begin(), carriers.end(), r.pot) == carriers.end()) {
                    carriers.push_back(r.pot);
                }
            }
        }
    }
    int carriers_left = static_cast<int>(carriers.size());
    try {
    // 1. the tables nobody took: one launch per precision, per-wave partial sums
    for (int prec = 0; prec < 2; prec++) {
        if (!pending[prec]) {
            continue;
        }
        const int blocks = host_[prec].block_end[host_[prec].n - 1];
        d_e_partials_[prec].reserve(static_cast<size_t>(blocks) * 4);
        const int zero_now = slots_zeroed ? 0 : std::min(n_rest, 256);
        if (prec == 1) {
            k_fused_energy<double><<<blocks, 256, 0, stream>>>(d_table_[prec].data, d_x, d_box, d_e_partials_[prec].data, d_e_slots_.data, zero_now);
        } else {
            k_fused_energy<float><<<blocks, 256, 0, stream>>>(d_table_[prec].data, d_x, d_box, d_e_partials_[prec].data, d_e_slots_.data, zero_now);
        }
        HIP_CHECK(hipGetLastError());
        slots_zeroed = slots_zeroed || n_rest <= 256;
        src.p[src.n] = d_e_partials_[prec].data;
        src.count[src.n] = blocks * 4;
        src.n++;
    }
    // 2. the potentials that launch their own kernels
    for (int i = 0; i < n_rest; i++) {
        const Rest &r = rest_[i];
        bool shared = false; // bound more than once: its partial buffer would be overwritten before the final reduction
        for (int j = 0; j < n_rest; j++) {
            shared = shared || (j != i && rest_[j].pot == r.pot);
        }
        const i128 *partials = nullptr;
        int count = 0;
        const bool carrier = std::find(carriers.begin(), carriers.end(), r.pot) != carriers.end();
        // (a carrier's slot was reserved in step 0; anybody else leaves the carriers still to come their room)
        const bool room = carrier || src.n + carriers_left < ENERGY_MAX_SOURCES - 1;
        if (carrier) {
            carriers_left--;
        }
        // the only contributor (every table rode along with it): it may leave the total in d_u itself
        i128 *d_final = (n_rest == 1 && src.n == 0 && !pending[0] && !pending[1]) ? d_u : nullptr;
        if (!shared && room && r.pot->execute_energy_partials(N, r.P, d_x, r.d_p, d_box, stream, partials, count, d_final)) {
            if (count == -1) {
                return; // (d_u[0] is written: nothing to add up)
            }
            if (count > 0) {
                src.p[src.n] = partials;
                src.count[src.n] = count;
                src.n++;
            }
            continue;
        }
        if (!slots_zeroed) { // (a potential may leave its d_u as the caller set it: interaction groups without interactions)
            d_e_slots_.zero_async(stream, std::max(n_rest, 1));
            slots_zeroed = true;
        }
        r.pot->execute_device(N, r.P, d_x, r.d_p, d_box, nullptr, nullptr, d_e_slots_.data + n_slots, stream);
        n_slots++;
    }
    if (n_slots > 0) {
        src.p[src.n] = d_e_slots_.data;
        src.count[src.n] = n_slots;
        src.n++;
    }
    // 3. one reduction over everything
    k_reduce_i128_sources<<<1, 256, 0, stream>>>(src, d_u);
    HIP_CHECK(hipGetLastError());
    } catch (...) {
        for (Potential *c : carriers) { // a table offered above may never have met its call
            c->drop_piggybacks();
        }
        throw;
    }
}

bool ForcePlan::run(
    const int N, const double *d_x, const double *d_box, u64 *d_du_dx, hipStream_t stream, std::vector<DeferredForces> *deferred,
    const int max_deferred, u64 *d_du_dx_cm, const int cm_stride) {
    this->merge_producers();
    bool wrote_du_dx = false; // did anything add to the [N, 3] array?
    // the table's terms go to the caller's component-major accumulator when there is one (lanes working on neighbouring
    // atoms then share cache lines: fewer line requests for the memory-side atomics), to the [N, 3] array otherwise
    u64 *table_acc = d_du_dx_cm ? d_du_dx_cm : d_du_dx;
    const ForceLayout table_fl = d_du_dx_cm ? ForceLayout{1, cm_stride} : ForceLayout{3, 1};
    // 1. tables to the device (only when they changed since the last step)
    host_[0].num_atoms = host_[1].num_atoms = N;
    bool pending[2] = {false, false};
    this->upload_tables(pending, stream);
    bool table_went_to_acc = false; // a table's forces were (or will be) added to table_acc
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
    try {
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
    } catch (...) {
        for (const Rest &r : rest_) { // a table offered in step 2 may never have met its call
            r.pot->drop_piggybacks();
        }
        throw;
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
