// Host-side object model of the hot path.  Mirrors the class surface the reference binds in
// cpp/src/wrap_kernels.cpp (Potential / BoundPotential / Summed / Fanout / NonbondedAllPairs / NonbondedPairList /
// HarmonicBond / HarmonicAngle / PeriodicTorsion / Neighborlist / HilbertSort / LangevinIntegrator / Context) so the
// C ABI in include/timemachine_amd.h is a 1:1 door onto it.  Everything device-side is HIP for gfx950.
#pragma once
#include "common.hpp"
#include "nb_es_table.hip.hpp"

#include <algorithm>
#include <limits>
#include <memory>
#include <optional>

namespace tmamd {

// Sums n signed 128-bit values on `stream` into *d_out (overwrites).  Integer => order independent.
void reduce_i128_device(const i128 *d_in, int n, i128 *d_out, hipStream_t stream);

int device_cu_count();

// ------------------------------------------------------------------------------------------------------------
// reference: cpp/src/potential.hpp:7-96, cpp/src/potential.cu
// ---- fused force evaluation (the MD path: du_dx only) ---------------------------------------------------------
// The bonded terms and the exclusion / pair lists are each a few microseconds of work -- shorter than a kernel launch
// is worth.  When only forces are requested, potentials describe themselves to a ForcePlan instead of launching; the
// plan runs every listed term in ONE kernel per precision and executes whatever cannot be fused (the neighbor-list
// potentials) the normal way.  Integer accumulation makes the result independent of this regrouping, bit for bit.
enum FusedKind : int {
    FUSED_BOND = 0,
    FUSED_ANGLE = 1,
    FUSED_TORSION = 2,
    FUSED_PAIR_LIST = 3,
    FUSED_PAIR_LIST_NEGATED = 4,
    FUSED_PAIR_LIST_PRECOMPUTED = 5,
    FUSED_CHIRAL_ATOM = 6,
    FUSED_CHIRAL_BOND = 7,
    FUSED_FLAT_BOTTOM_BOND = 8,
    FUSED_LOG_FLAT_BOTTOM_BOND = 9
};
static const int FUSED_MAX_SEGMENTS = 16;
// Neighbor-list counters (one array of NB_NUM_COUNTERS u32 per list): [0..2] list totals, [3] builds since construction,
// [NB_COUNTER_CLASS0 + shard * NB_CLASSES + class] work items per (shard, cost class) bucket.  Written by the list build
// (kernels_nblist.hip.hpp), read by the tile kernel (kernels_nonbonded.hip.hpp), reset by whoever raises the rebuild flag
// (the bounds kernel, or the integrator's update kernel on the sorted hand-over path).
static const int NB_SHARDS = 4;         // item buckets per cost class (row block % 4): spreads the build's bucket-cursor atomics
static const int NB_CLASSES = 16;       // work items are bucketed by cost (estimated interacting pairs), heaviest first
static const int NB_COUNTER_CLASS0 = 4;
static const int NB_COUNTER_GUEST = NB_COUNTER_CLASS0 + NB_SHARDS * NB_CLASSES; // merged orders: items of the guest rows, once more in a list of their own
static const int NB_NUM_COUNTERS = NB_COUNTER_GUEST + 1;
// What a nonbonded pipeline remembers about its last ENERGY-ONLY evaluation (device side; NonbondedAllPairs::run_pipeline, "memo"):
// the check + gather kernel compares every record it is about to overwrite with what it writes, so the device knows whether the
// all-pairs part of the coming evaluation has the operands of the last one -- then its launch gets an empty item list and the sum is
// taken from here.  Exact: equal operands (as the kernels read them, in Real), equal box, same order => the same integer sum.
struct EnergyMemo {
    int changed_main; // a record that the all-pairs items read was rewritten with other values since the cached sum was made
    int valid;        // cached_main belongs to the records as they are (but for changed_main)
    int reserved_[2];
    double box[9];    // the box of the cached sum
    long long evaluations, skipped; // diagnostics: memo evaluations since construction, of which the all-pairs launch was empty
    i128 cached_main;
};
// words of zeros kept behind a list's counters: the second launch of a memo evaluation reads its 64 "bucket counts" from
// counters + NB_COUNTER_GUEST on (bucket 0 = the guest rows' list, every other bucket empty)
static const int NB_COUNTERS_ALLOC = NB_NUM_COUNTERS + NB_SHARDS * NB_CLASSES;
struct FusedSegment {
    int kind;
    int count;            // terms
    const int *idxs;      // [count][2|3|4]
    const double *params; // bonded: [count][2|3]; pair lists: the [N][4] nonbonded parameters
    const double *scales; // pair lists: [count][2]
    double beta, cutoff;  // pair lists
    const int *aux;       // chiral bond restraints: signs [count]
    const double *es_table = nullptr; // pair lists, f64 kernels: the electrostatic force-factor table of `beta` (nb_es_table.hip.hpp)
    int window = 1; // accumulate through the per-wave LDS window (set by ForcePlan::add_segment from the owner's max_atom_incidence)
    int reserved_ = 0; // (no padding bytes: tables are compared with memcmp to decide on re-uploads)
};
static const int FUSED_WINDOW_MIN_INCIDENCE = 8;
struct FusedTable {
    int n;
    int window; // != 0: slices accumulate in the calling wave's LDS window before they touch the global accumulator (ForceLayout::win)
    int num_atoms; // atoms of the system (bounds the window's row prefetch)
    int reserved_ = 0; // (no padding bytes: tables are compared with memcmp)
    int block_end[FUSED_MAX_SEGMENTS]; // exclusive prefix sum of 256-thread blocks per segment
    FusedSegment seg[FUSED_MAX_SEGMENTS];
};
// A force contribution left in a potential's own (sorted) accumulator instead of being scattered into du_dx: the
// consumer adds g_du_dx[d * stride + slot_of_atom[a]] for every atom a with slot_of_atom[a] >= 0.  Lets the integrator's
// update kernel pick the nonbonded forces up directly (one launch and one pass over du_dx less per step).
// What the consumer needs to leave the producer's NEXT gather already done while it moves the atoms (the integrator's
// update kernel touches every atom anyway): the new position goes straight into the producer's sorted record, the
// displacement-vs-snapshot test that decides on a neighbor-list rebuild is made on the spot, and the consumed accumulator
// slot is zeroed.  The producer then skips its own check + gather launch on the next call (one launch and one pass over
// the atoms less per MD step).  Plain data: passed to kernels by value.
struct PregatherTarget {
    void *gathered = nullptr; // Real[K + 1][8] records (x, y, z, w, q, sig, eps, 0); only x, y, z are rewritten
    int real_bytes = 0;       // sizeof(Real) of the producer: 4 or 8
    const double *snap_x = nullptr; // coordinates at the last list build, atom order
    double pad2_quarter = 0;        // (padding / 2)^2
    // != nullptr: the producer follows small box changes without rebuilding (a barostat is at work): the test is the scale-aware
    // one (nb_snapshot_test.hip.hpp) of the new position in `cur_box` against the snapshot in `snap_box`, and pad2_quarter is its D^2
    const double *snap_box = nullptr, *cur_box = nullptr;
    int *flag_set = nullptr;        // rebuild flag of the producer's next call
    int *flag_clear = nullptr;      // flag of the call that has just been consumed
    u64 *g_du_dx = nullptr;         // the accumulator handed over in DeferredForces (to be zeroed slot by slot)
    int stride = 0;                 // its component stride
    // "Sorted" hand-over (sorted_n > 0: every one of the sorted_n atoms has a slot, the list is the plain upper-triangular
    // one): the consumer may walk the SLOTS in order (atom = perm[slot]; accumulator reads and record writes coalesced) and
    // then also leaves behind what the producer's block-bounds kernel would have computed on a rebuild step -- the 32-atom
    // blocks' bounding boxes, every step (two per wave, a handful of shuffles) -- and resets the neighbor-list counters
    // whenever it raises the rebuild flag.  The producer then launches no bounds kernel at all on MD steps.
    const unsigned int *perm = nullptr;
    int sorted_n = 0; // SLOTS of the order: a merged producer's (below) has up to 31 holes, perm[slot] == 0xffffffff
    int covers_atoms = 0; // atoms that have a slot (the consumer walks the slots iff this is every atom it moves)
    void *blk_ctr = nullptr, *blk_ext = nullptr; // Real[ceil(sorted_n / 32)][3]
    unsigned int *nbl_counters = nullptr;        // kernels_nonbonded.hip.hpp: NB_NUM_COUNTERS words
    // != 0: a MERGED producer (an all-pairs potential carrying an interaction group's pairs in its own pipeline): its all-pairs atoms
    // have a second record, under the group's parameters, at record index second_records + slot -- positions go there as well
    int second_records = 0;
};
class Potential;
struct DeferredForces {
    const u64 *g_du_dx = nullptr;       // component-major: component d of slot s at [d * stride + s]
    int stride = 0;
    const int *slot_of_atom = nullptr;
    PregatherTarget next;       // gathered == nullptr: the producer does not take pre-gathered positions
    Potential *owner = nullptr; // to be told (pregather_committed) once a kernel filling `next` has been enqueued
    // this call consumed a SORTED hand-over: inputs, parameters and the order (perm) are exactly what the consumer that
    // filled it last left behind -- a consumer that keeps per-slot state of its own may trust it for this step
    bool consumed_sorted_pregather = false;
};
class ForcePlan {
public:
    ForcePlan() {}
    ~ForcePlan();
    ForcePlan(const ForcePlan &) = delete;
    ForcePlan &operator=(const ForcePlan &) = delete;
    struct Rest {
        Potential *pot;
        int P;
        const double *d_p;
    };
    void clear();
    void add_segment(const int precision_bytes, const FusedSegment &seg, Potential *owner, const int P, const double *d_p);
    void add_rest(Potential *pot, const int P, const double *d_p) { rest_.push_back({pot, P, d_p}); }
    // launches everything; accumulates into d_du_dx.  With `deferred` != nullptr, up to `max_deferred` contributions may
    // be handed back un-scattered instead (see DeferredForces).
    // d_du_dx_cm != nullptr: a second, component-major accumulator (component d of atom a at [d * cm_stride + a]) that
    // receives the table's terms; potentials that launch their own kernels keep adding to the [N, 3] array d_du_dx.
    // Returns whether anything was (or may have been) added to d_du_dx -- a consumer that owns both arrays can skip
    // reading and re-zeroing an untouched [N, 3] array.
    bool run(const int N, const double *d_x, const double *d_box, u64 *d_du_dx, hipStream_t stream,
             std::vector<DeferredForces> *deferred = nullptr, const int max_deferred = 0, u64 *d_du_dx_cm = nullptr,
             const int cm_stride = 0);
    // Energy only: the total of everything planned, as one signed 128-bit fixed-point sum in d_u[0].  The table's terms
    // run in one launch that leaves per-wave partial sums; potentials that launch their own kernels leave theirs in their own
    // buffers (Potential::execute_energy_partials) or, failing that, reduce into a slot; ONE final reduction adds it all up.
    // Integer sums: the same bits as summing child by child (SummedPotential / FanoutSummedPotential / the barostat).
    void run_energy(const int N, const double *d_x, const double *d_box, i128 *d_u, hipStream_t stream);
    // after run(): did anything go to d_du_dx_cm?  (false when every table rode on a potential that took it into its own
    // accumulator: the consumer can skip reading and re-zeroing the array)
    bool cm_written() const { return cm_written_; }
    // what was planned (a mover's fast path inspects it): the potentials that launch their own kernels, and -- after
    // prepare_tables() -- the uploaded table of each precision (nullptr: none planned)
    const std::vector<Rest> &rest() const { return rest_; }
    void prepare_tables(const int N, hipStream_t stream, const FusedTable *d_tables[2], int blocks[2]);
    // Two planned tile producers that read the same coordinates -- an all-pairs potential over the atom set H and an interaction
    // group (rows R, columns == H): the reference's HostGuestSystem, fe/system.py:133-146 -- become ONE: the all-pairs potential's
    // merged carrier (NonbondedAllPairsBase::merged_carrier), which lists, evaluates and hands over both pair sets in one
    // pipeline.  Idempotent; run() and run_energy() call it, a mover that inspects rest() calls it first.  Same bits either way
    // (the same pair function on the same operands, integer sums).
    void merge_producers();

private:
    FusedTable host_[2];                  // [0] f32 kernels, [1] f64 kernels
    FusedTable uploaded_[2];              // what d_table_ currently holds
    bool uploaded_valid_[2] = {false, false};
    DeviceBuffer<FusedTable> d_table_[2];
    // Uploads leave the host through a ring of pinned copies (one per upload, an event behind each): a table changes whenever a
    // caller walks parameter sets (execute_batch: every set is another d_p), and an upload that waited for the stream drained the
    // launch queue once per evaluation (round 6: 63 -> see EXPERIMENTS.md us per further parameter set)
    static const int TABLE_RING = 32;
    FusedTable *h_ring_ = nullptr;
    hipEvent_t ring_ev_[TABLE_RING];
    bool ring_used_[TABLE_RING];
    int ring_pos_ = 0;
    std::vector<Rest> rest_;
    bool cm_written_ = true;
    DeviceBuffer<i128> d_e_partials_[2]; // run_energy: per-wave sums of the table launches
    DeviceBuffer<i128> d_e_slots_;       // run_energy: totals of potentials that reduce for themselves
    void upload_tables(bool pending[2], hipStream_t stream);
};

class Potential {
public:
    virtual ~Potential();
    // How many terms of this potential's list the busiest atom takes part in (set by the constructors of the term-list
    // potentials; "unknown" = very many).  A ForcePlan uses it to decide whether the potential's slices accumulate through the
    // per-wave LDS window (ForceLayout::win): it pays where many terms meet on the same atoms -- a protein's angles, torsions and
    // exclusions -- and costs a lone wave 1-2 us where they do not (water: 2 bonds, 1 angle, 2 exclusions per atom).
    int max_atom_incidence() const { return max_atom_incidence_; }
    void note_term_atoms(const std::vector<int> &idxs) {
        // counted on a sorted copy: nothing here is sized by the caller's indices (constructors call this before or after
        // their range checks; a garbage index must cost nothing but its own entry)
        std::vector<int> sorted(idxs);
        std::sort(sorted.begin(), sorted.end());
        int most = 0, run = 0;
        for (size_t i = 0; i < sorted.size(); i++) {
            run = (i > 0 && sorted[i] == sorted[i - 1]) ? run + 1 : 1;
            if (sorted[i] >= 0 && run > most) {
                most = run;
            }
        }
        max_atom_incidence_ = most;
    }

protected:
    int max_atom_incidence_ = 1 << 20;

public:
    static const int D = 3;

    // Forces-only planning hook (see ForcePlan).  Default: not fusable, executed through execute_device.
    virtual void plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) { plan.add_rest(this, P, d_p); }
    // Offer of a ForcePlan table to run inside this potential's own (long) force kernel during its NEXT forces-only
    // execute_device call.  true = accepted (the plan then skips its own launch).
    // The table's forces go to `acc` (component d of atom a at acc[a * atom_stride + d * comp_stride]).
    virtual bool piggyback_forces(
        const FusedTable *d_table, const int blocks, const int precision_bytes, u64 *acc, const int atom_stride, const int comp_stride) {
        return false;
    }
    // after an accepted offer: will the table's forces land in this potential's own accumulator (delivered with its own
    // forces) instead of `acc`?  The plan then reports `acc` as untouched.
    virtual bool piggyback_lands_in_own_accumulator() const { return false; }
    // Forces-only evaluation that leaves the result in the potential's own accumulator (see DeferredForces) instead of
    // adding it to a du_dx array.  A piggy-backed table still accumulates into d_du_dx.  false = not supported.
    virtual bool execute_forces_deferred(
        const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, hipStream_t stream,
        DeferredForces &out) {
        return false;
    }

    // Offer of a ForcePlan table to run inside this potential's NEXT energy-only partial-sum evaluation (execute_energy_partials):
    // the table's energies then arrive in that call's partial sums.  true = accepted (the plan skips its own energy launch).
    virtual bool piggyback_energy(const FusedTable *d_table, const int blocks, const int precision_bytes) { return false; }
    // forget a table accepted through piggyback_forces / piggyback_energy that the plan can no longer deliver a call for (an
    // exception between the offer and the call): without this the next call of any other form would refuse to run
    virtual void drop_piggybacks() {}
    // Energy-only evaluation that leaves per-wave partial sums (to be added up by the caller) in a buffer of the potential's
    // own instead of reducing them into a d_u -- saves a launch per evaluation.  false = not supported (nothing was run).
    // d_final != nullptr: the caller has nothing else to add -- an implementation that ends in a launch of its own anyway may put the
    // TOTAL straight into d_final[0] and report count == -1 (no partials, no reduction by the caller).
    virtual bool execute_energy_partials(
        const int N, const int P, const double *d_x, const double *d_p, const double *d_box, hipStream_t stream, const i128 *&partials,
        int &count, i128 *d_final = nullptr) {
        return false;
    }
    // The caller vouches that the coordinates and the box behind the pointers of the NEXT call hold what they held at the call numbered
    // `prev_call` (execute_batch walking the parameter sets of one frame; calls are numbered by g_eval_serial): a stateful child that
    // last ran its pipeline IN that call may skip what only coordinates can invalidate.  A child that did not run in it -- an
    // interaction group whose work the all-pairs potential's merged carrier did, or the reverse -- must not: its own "last call" was
    // another frame behind what may well be the same pointers (found by tests/test_gpu_interleavings.py: a stale list, pairs near the
    // cutoff missing).  One call only: the giver withdraws the hint (on = false) behind the call.
    virtual void hint_same_frame(const bool on = true, const long long prev_call = 0) {}

    // The consumer of DeferredForces has enqueued (on the same stream) a kernel that filled `next` for the coordinates
    // in d_x / d_box: the following execute_forces_deferred call with the same pointers may skip its gather.
    virtual void pregather_committed(const double *d_x, const double *d_box, const bool sorted_bounds_done) {}
    // Anything a potential remembers about its inputs between calls (pre-gathered positions) is dropped.  Called when
    // coordinates, box or parameters change behind an unchanged pointer (set_params, Context::set_x_t, movers).
    virtual void invalidate_cached_inputs() {}
    // A barostat is going to evaluate and move this potential's system: the box changes by fractions of a percent between
    // calls.  Potentials with a neighbor list then follow small box changes without rebuilding it (k_check_gather_scaled).
    virtual void expect_box_scaling() {}

    // Accumulates into d_du_dx / d_du_dp (caller zeroes them), overwrites d_u.  Any output may be nullptr.
    virtual void execute_device(
        const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp,
        i128 *d_u, hipStream_t stream) = 0;

    virtual void du_dp_fixed_to_float(const int N, const int P, const u64 *du_dp, double *du_dp_float);
    // the same conversion as a layout the device can apply (execute_host_f64): spans of the parameter vector whose derivatives carry
    // the nonbonded per-column exponents (2^36, 2^37, 2^38, 2^36 by index % 4) -- everything else is 2^36
    struct DuDpSpan {
        int offset, count;
    };
    virtual void du_dp_nonbonded_spans(const int N, const int P, const int offset, std::vector<DuDpSpan> &out) const {}

    void execute_host(
        const int N, const int P, const double *h_x, const double *h_p, const double *h_box, u64 *h_du_dx, u64 *h_du_dp, i128 *h_u);

    void execute_batch_host(
        const int coord_batch_size, const int N, const int param_batch_size, const int P, const double *h_x, const double *h_p,
        const double *h_box, u64 *h_du_dx, u64 *h_du_dp, i128 *h_u);

    void execute_batch_sparse_host(
        const int coords_size, const int N, const int params_size, const int P, const int batch_size,
        const unsigned int *coords_batch_idxs, const unsigned int *params_batch_idxs, const double *h_x, const double *h_p,
        const double *h_box, u64 *h_du_dx, u64 *h_du_dp, i128 *h_u);

    void execute_batch_device(
        const int coord_batch_size, const int N, const int param_batch_size, const int P, const double *d_x, const double *d_p,
        const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u, hipStream_t stream);

    // The host entry points as the reference's BINDING delivers them (wrap_kernels.cpp:731-1131: doubles, NaN for an overflowed
    // energy): one packed host-to-device copy from pinned staging, one memset, the evaluations, the fixed-point -> double conversion
    // ON THE DEVICE, the results copied back, one synchronisation.  (The u64 forms above return the raw accumulators and convert
    // on the host, 3.3 M elements per 20-evaluation batch of a 23.5k-atom system: more time than the kernels take.)
    // batch_size < 0: the dense coords x params matrix (execute / execute_batch); else the listed (coords, params) entries.
    // d_bound_p != nullptr: a BoundPotential's device-resident parameters (params_size == 1, h_p unused).
    void execute_host_f64(
        const int coords_size, const int N, const int params_size, const int P, const int batch_size, const unsigned int *coords_batch_idxs,
        const unsigned int *params_batch_idxs, const double *h_x, const double *h_p, const double *h_box, double *h_du_dx, double *h_du_dp,
        double *h_u, const double *d_bound_p = nullptr);

    void execute_batch_sparse_device(
        const int N, const int P, const int batch_size, const unsigned int *coords_batch_idxs,
        const unsigned int *params_batch_idxs, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx,
        u64 *d_du_dp, i128 *d_u, hipStream_t stream);

private:
    // grow-only scratch for the host entry points (the reference mallocs and frees on every call)
    DeviceBuffer<double> hs_x_, hs_p_, hs_box_;
    DeviceBuffer<u64> hs_du_dx_, hs_du_dp_;
    DeviceBuffer<i128> hs_u_;
    // execute_host_f64: one device block (inputs | fixed-point outputs | double outputs) and its pinned host staging, grow-only
    DeviceBuffer<char> hf_block_;
    void *hf_pinned_ = nullptr;
    size_t hf_pinned_bytes_ = 0;
};

// reference: cpp/src/bound_potential.{hpp,cu}
class BoundPotential {
public:
    BoundPotential(std::shared_ptr<Potential> potential, const std::vector<double> &params);
    int size;
    DeviceBuffer<double> d_p;
    std::shared_ptr<Potential> potential;

    void set_params(const std::vector<double> &params);
    void set_params_device(const int size, const double *d_p, hipStream_t stream);
    // the first params.size() values of the buffer become the parameters (size <= the constructor's): local MD's restraints
    void set_params_prefix(const std::vector<double> &params);
    void execute_device(const int N, const double *d_x, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u, hipStream_t stream);
    void execute_host(const int N, const double *h_x, const double *h_box, u64 *h_du_dx, i128 *h_u);
    void execute_batch_host(const int coord_batch_size, const int N, const double *h_x, const double *h_box, u64 *h_du_dx, i128 *h_u);
    // BoundPotential.execute / execute_batch as the binding delivers them (doubles; Potential::execute_host_f64)
    void execute_host_f64(const int coord_batch_size, const int N, const double *h_x, const double *h_box, double *h_du_dx, double *h_u) {
        potential->execute_host_f64(coord_batch_size, N, 1, size, -1, nullptr, nullptr, h_x, nullptr, h_box, h_du_dx, nullptr, h_u, d_p.data);
    }

private:
    DeviceBuffer<double> hs_x_, hs_box_;
    DeviceBuffer<u64> hs_du_dx_;
    DeviceBuffer<i128> hs_u_;
};


// reference: cpp/src/summed_potential.cu:33-97
// `parallel` (reference: children on per-child streams joined by events, stream_manager.cu:18-56) is kept as a
// constructor argument and otherwise ignored: results never depend on it (integer accumulation), forces-only calls go
// through one fused ForcePlan anyway, and forked streams measured ~30 us per MD step slower than children in sequence.
class SummedPotential : public Potential {
public:
    SummedPotential(const std::vector<std::shared_ptr<Potential>> potentials, const std::vector<int> params_sizes, const bool parallel);
    const std::vector<std::shared_ptr<Potential>> &get_potentials() { return potentials_; }
    const std::vector<int> &get_parameter_sizes() { return params_sizes_; }
    bool get_parallel() const { return parallel_; }
    void plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) override;
    void invalidate_cached_inputs() override {
        for (auto &pot : potentials_) {
            pot->invalidate_cached_inputs();
        }
    }
    void expect_box_scaling() override {
        for (auto &pot : potentials_) {
            pot->expect_box_scaling();
        }
    }
    void hint_same_frame(const bool on = true, const long long prev_call = 0) override {
        for (auto &pot : potentials_) {
            pot->hint_same_frame(on, prev_call);
        }
    }
    void execute_device(const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u, hipStream_t stream) override;
    void du_dp_fixed_to_float(const int N, const int P, const u64 *du_dp, double *du_dp_float) override;
    void du_dp_nonbonded_spans(const int N, const int P, const int offset, std::vector<DuDpSpan> &out) const override {
        int off = offset;
        for (size_t i = 0; i < potentials_.size(); i++) {
            potentials_[i]->du_dp_nonbonded_spans(N, params_sizes_[i], off, out);
            off += params_sizes_[i];
        }
    }

private:
    std::vector<std::shared_ptr<Potential>> potentials_;
    std::vector<int> params_sizes_;
    int P_;
    bool parallel_; // accepted, not used: see SummedPotential
    DeviceBuffer<i128> d_u_buffer_;
    ForcePlan plan_;
};

// reference: cpp/src/fanout_summed_potential.cu:23-68
class FanoutSummedPotential : public Potential {
public:
    FanoutSummedPotential(const std::vector<std::shared_ptr<Potential>> potentials, const bool parallel);
    const std::vector<std::shared_ptr<Potential>> &get_potentials() { return potentials_; }
    bool get_parallel() const { return parallel_; }
    void plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) override;
    void invalidate_cached_inputs() override {
        for (auto &pot : potentials_) {
            pot->invalidate_cached_inputs();
        }
    }
    void expect_box_scaling() override {
        for (auto &pot : potentials_) {
            pot->expect_box_scaling();
        }
    }
    void hint_same_frame(const bool on = true, const long long prev_call = 0) override {
        for (auto &pot : potentials_) {
            pot->hint_same_frame(on, prev_call);
        }
    }
    void execute_device(const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u, hipStream_t stream) override;
    void du_dp_fixed_to_float(const int N, const int P, const u64 *du_dp, double *du_dp_float) override;
    void du_dp_nonbonded_spans(const int N, const int P, const int offset, std::vector<DuDpSpan> &out) const override {
        if (!potentials_.empty()) {
            potentials_[0]->du_dp_nonbonded_spans(N, P, offset, out);
        }
    }

private:
    std::vector<std::shared_ptr<Potential>> potentials_;
    bool parallel_; // accepted, not used: see SummedPotential
    DeviceBuffer<i128> d_u_buffer_;
    ForcePlan plan_;
};

// ------------------------------------------------------------------------------------------------------------
// Bonded terms.  kind: 0 = HarmonicBond (2 idx, 2 params), 1 = HarmonicAngle (3, 3), 2 = PeriodicTorsion (4, 3)
// reference: cpp/src/harmonic_bond.cu, harmonic_angle.cu, periodic_torsion.cu
template <typename Real> class HarmonicBond : public Potential {
public:
    explicit HarmonicBond(const std::vector<int> &bond_idxs);
    void plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) override;
    void execute_device(const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u, hipStream_t stream) override;
private:
    int B_;
    DeviceBuffer<int> d_idxs_;
    DeviceBuffer<i128> d_u_partials_;
};

template <typename Real> class HarmonicAngle : public Potential {
public:
    explicit HarmonicAngle(const std::vector<int> &angle_idxs);
    void plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) override;
    void execute_device(const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u, hipStream_t stream) override;
private:
    int A_;
    DeviceBuffer<int> d_idxs_;
    DeviceBuffer<i128> d_u_partials_;
};

template <typename Real> class PeriodicTorsion : public Potential {
public:
    explicit PeriodicTorsion(const std::vector<int> &torsion_idxs);
    void plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) override;
    void execute_device(const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u, hipStream_t stream) override;
private:
    int T_;
    DeviceBuffer<int> d_idxs_;
    DeviceBuffer<i128> d_u_partials_;
};

// reference: cpp/src/flat_bottom_bond.{hpp,cu} (Log == false), log_flat_bottom_bond.{hpp,cu} (Log == true)
template <typename Real, bool Log> class FlatBottomBond : public Potential {
public:
    FlatBottomBond(const std::vector<int> &bond_idxs, const double beta);
    // local MD re-targets its restraints on every call (reference: set_bonds_device, flat_bottom_bond.cu); no src != dst
    // re-validation here either -- the caller builds the pairs
    void set_bonds(const std::vector<int> &bond_idxs);
    int num_bonds() const { return B_; }
    void plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) override;
    void execute_device(const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u, hipStream_t stream) override;
private:
    int B_;
    double beta_;
    DeviceBuffer<int> d_idxs_;
    DeviceBuffer<i128> d_u_partials_;
    void check_size(const int P) const;
};

// reference: cpp/src/centroid_restraint.{hpp,cu}
template <typename Real> class CentroidRestraint : public Potential {
public:
    CentroidRestraint(const std::vector<int> &group_a_idxs, const std::vector<int> &group_b_idxs, const double kb, const double b0);
    void execute_device(const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u, hipStream_t stream) override;
private:
    int NA_, NB_;
    double kb_, b0_;
    DeviceBuffer<int> d_a_, d_b_;
    DeviceBuffer<u64> d_sums_;
};

// reference: cpp/src/chiral_atom_restraint.{hpp,cu}, chiral_bond_restraint.{hpp,cu}
template <typename Real> class ChiralAtomRestraint : public Potential {
public:
    explicit ChiralAtomRestraint(const std::vector<int> &idxs);
    void plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) override;
    void execute_device(const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u, hipStream_t stream) override;
private:
    int R_;
    DeviceBuffer<int> d_idxs_;
    DeviceBuffer<i128> d_u_partials_;
    void check_size(const int P) const;
};

template <typename Real> class ChiralBondRestraint : public Potential {
public:
    ChiralBondRestraint(const std::vector<int> &idxs, const std::vector<int> &signs);
    void plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) override;
    void execute_device(const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u, hipStream_t stream) override;
private:
    int R_;
    DeviceBuffer<int> d_idxs_, d_signs_;
    DeviceBuffer<i128> d_u_partials_;
    void check_size(const int P) const;
};

// ------------------------------------------------------------------------------------------------------------
// reference: cpp/src/hilbert_sort.{hpp,cu}
class HilbertSort {
public:
    explicit HilbertSort(const int N);
    ~HilbertSort();
    void sort_device(const int N, const unsigned int *d_atom_idxs, const double *d_coords, const double *d_box, unsigned int *d_output_perm, hipStream_t stream);
    std::vector<unsigned int> sort_host(const int N, const double *h_coords, const double *h_box);
    static const std::vector<unsigned int> &lut(); // bin (i,j,k) -> Hilbert index, 128^3 entries
private:
    int N_;
    DeviceBuffer<unsigned int> d_bin_to_idx_, d_keys_in_, d_keys_out_, d_vals_in_;
    void *d_sort_storage_;
    size_t sort_storage_bytes_;
};

// reference: cpp/src/neighborlist.{hpp,cu}
template <typename Real> class Neighborlist {
public:
    explicit Neighborlist(const int N);
    ~Neighborlist() {}

    void set_row_idxs(std::vector<unsigned int> idxs);
    void reset_row_idxs();
    void resize(const int size);
    void set_idxs_device(const int NC, const int NR, const unsigned int *d_col_idxs, const unsigned int *d_row_idxs, hipStream_t stream);

    unsigned int num_tile_ixns();
    unsigned int num_builds(); // list builds since construction (diagnostic)
    std::vector<std::vector<int>> get_nblist_host(const int N, const double *h_coords, const double *h_box, const double cutoff);
    void compute_block_bounds_host(const int N, const double *h_coords, const double *h_box, double *h_bb_ctrs, double *h_bb_exts);

    // `d_gathered` holds Real[K][8] records (x, y, z, ...).  If d_flag != nullptr the kernels return immediately
    // unless force != 0 or *d_flag != 0 (device-side rebuild decision: no host synchronisation).
    // When d_snap_x != nullptr the n_snap doubles of d_x (and the box) are copied into the snapshot as part of the build.
    // `cost_cutoff` (<= cutoff) is the distance the work items' cost estimates count pairs within.
    void build_device(
        const Real *d_gathered, const double *d_box, const double cutoff, const double cost_cutoff, const int *d_flag,
        const int force, const int n_snap, const double *d_x, double *d_snap_x, double *d_snap_box, hipStream_t stream,
        // bounds_done: the column-block bounds are current and the counters have been reset with the flag by somebody else
        // (PregatherTarget's sorted hand-over): no bounds kernel is launched, the list kernel takes the snapshot itself
        const bool bounds_done = false,
        // rebase_box: the caller's rebuild test was the scale-aware one (k_check_gather_scaled): when no rebuild follows, the
        // bounds kernel re-expresses the snapshot's box in the current one on its way out
        const bool rebase_box = false);

    // merged orders (NonbondedAllPairs as a carrier): the first `blocks` blocks are `rows` guest atoms padded with holes
    void set_guest(const int rows, const int blocks) {
        guest_rows_ = rows;
        guest_blocks_ = blocks;
        d_guest_items_.realloc(static_cast<size_t>(std::max(blocks, 1)) * (ceil_divide(max_size_, 64) + 1));
    }
    int get_num_row_idxs() const { return NR_; }
    int num_row_blocks() const { return ceil_divide(NR_, TILE); }
    int num_column_blocks() const { return ceil_divide(NC_, TILE); }
    int max_ixn_count() const;
    bool upper_triangular() const { return NR_ == N_ && NC_ == N_; }
    int num_atoms() const { return N_; }

    const unsigned int *row_idxs_or_null() const { return upper_triangular() ? nullptr : d_row_idxs_.data; }
    const unsigned int *d_counters() const { return d_counters_.data; }
    unsigned int *d_counters_rw() { return d_counters_.data; }
    Real *d_col_ctr() { return d_col_ctr_.data; }
    Real *d_col_ext() { return d_col_ext_.data; }
    const int4 *d_items() const { return d_items_.data; }
    unsigned int items_cap() const { return static_cast<unsigned int>(items_cap_); }
    const unsigned int *d_col_atoms() const { return d_col_atoms_.data; }
    const int2 *d_row_segments() const { return d_row_segments_.data; } // per row block {start in d_col_atoms, listed columns}
    const int4 *d_guest_items() const { return d_guest_items_.data; }   // merged orders: the guest rows' items once more, compact
    unsigned int guest_items_cap() const { return static_cast<unsigned int>(d_guest_items_.length); }

private:
    const int max_size_;
    int N_, NC_, NR_;
    int guest_rows_ = 0, guest_blocks_ = 0;
    DeviceBuffer<Real> d_col_ctr_, d_col_ext_, d_row_ctr_, d_row_ext_;
    DeviceBuffer<unsigned int> d_row_idxs_, d_col_idxs_;
    DeviceBuffer<unsigned int> d_counters_;  // [0] pool cursor, [1] work items, [2] 32-wide tile count, [4..12) items per cost class
    DeviceBuffer<unsigned int> d_col_atoms_; // CSR pool
    DeviceBuffer<int4> d_items_;             // NB_CLASSES buckets of items_cap_ work items, heaviest class first
    DeviceBuffer<int4> d_guest_items_;       // merged orders (set_guest): the guest rows' items again, counted in counters[NB_COUNTER_GUEST]
    size_t items_cap_ = 0;
    DeviceBuffer<int2> d_row_segments_;
    DeviceBuffer<Real> d_scratch_gathered_; // host entry points only
    void gather_host_coords(const int N, const double *h_coords, const double *h_box, DeviceBuffer<double> &d_box);
};

void verify_atom_idxs(const int N, const std::vector<int> &atom_idxs, const bool allow_empty = false);

extern double g_last_host_call_device_ms; // potential.hip: device time of the evaluations of the last execute_host_f64 call
extern bool g_same_frame_hint; // the batch entry points' hint_same_frame() is honoured (tm_debug_set_same_frame_hint)
// Numbers of the batch entries (Potential::execute_batch[_sparse]_device): drawn from one process-wide counter, so that no two entries
// -- of whatever thread -- share one; g_eval_serial is the number of the entry being evaluated on THIS thread (children run inside
// their caller's call), which a pipeline notes when it runs.
long long next_eval_serial();
extern thread_local long long g_eval_serial;
extern bool g_energy_memo;     // energy-only evaluations are remembered on the device (EnergyMemo; tm_debug_set_energy_memo)
extern bool g_merge_producers; // ForcePlan::merge_producers runs all-pairs + interaction group as one pipeline (tm_debug_set_merge_producers)
extern bool g_barostat_fast_path; // MonteCarloBarostat attempts run on the potential's current list when its state allows (tm_debug_set_barostat_fast_path)
extern bool g_box_scaling_reuse; // process-wide switch of the scale-aware rebuild test (tm_debug_set_box_scaling_reuse)
extern int g_rowblock_min_k;     // forces-only launches over at least this many atoms run the row-block kernel (tm_debug_set_rowblock_min_k)
extern const bool g_rowblock_built; // ... which only libraries built with -DTM_ROWBLOCK carry (the variant library of the parity tests)
extern int g_static_list_max_k;  // potentials over at most this many atoms keep a static, complete list (tm_debug_set_static_list_max_k)

// What a mover that proposes a SECOND geometry (the barostat: x', box') needs from the nonbonded potential whose sorted
// pre-gathered state describes the current one (x, box), so that both energies can be evaluated on the potential's CURRENT list
// without giving that state up, and an accepted proposal can be committed in place (MonteCarloBarostat's fast path,
// barostat.hip).  Plain device pointers: passed to kernels by value.
struct ProbeTarget {
    void *gathered = nullptr;  // Real[K + 1][8] sorted records of the current geometry (left by the integrator's update kernel)
    void *gathered2 = nullptr; // the same for the proposal: filled by the mover (x y z; w q sig eps copied from `gathered`)
    int real_bytes = 0;        // sizeof(Real) of the potential
    int n = 0;                 // SLOTS of the potential's order (every atom has one; a merged order also has holes, perm == 0xffffffff)
    const int *slot_of_atom = nullptr;
    const unsigned int *perm = nullptr;
    double *snap_x = nullptr;   // coordinates at the last list build, atom order (re-based by the commit)
    double *snap_box = nullptr; // [0..8] the box the snapshot is expressed in, [9..11] accumulated scale since the build
    double threshold2 = 0;      // squared displacement (against the scaled snapshot) beyond which the list is rebuilt
    int scale_aware = 0;        // the potential follows small box changes without rebuilding (else: any box change rebuilds)
    int *flag_probe = nullptr;  // the rebuild flag the probe's list launch reads (raised when the PROPOSAL fails the test)
    int *flag_next = nullptr;   // the flag of the force call after the probe (raised by the commit's own test)
    unsigned int *nbl_counters = nullptr; // reset by whoever raises a flag (sorted hand-over: no bounds kernel does it)
    void *blk_ctr = nullptr, *blk_ext = nullptr; // Real[ceil(n / 32)][3]: block bounds, recomputed by the commit
    int second_records = 0; // merged producers (PregatherTarget::second_records): a commit writes the positions there as well
    // how far the CURRENT list reaches even when it has just been rebuilt (cutoff + padding; 0: a static complete list -- every pair),
    // and the cutoff: a proposal whose pairs inside the cutoff can lie beyond that reach in the current geometry is one the list
    // cannot vouch for however fresh it is (the mover checks: k_barostat_propose_probe)
    double list_reach = 0, cutoff = 0;
};

// reference: cpp/src/nonbonded_all_pairs.{hpp,cu}
class NonbondedAllPairsBase : public Potential {
public:
    // ---- probing a second geometry on the current list (see ProbeTarget) ----
    // true iff the sorted pre-gathered state the last MD step left describes exactly these inputs and nothing (re-sort, forced
    // rebuild, atom subset, interaction group) stands in the way
    virtual bool probe_ready(const int N, const int P, const double *d_x, const double *d_p, const double *d_box) { return false; }
    // begins a probe: the call counts as one evaluation (launch parity, sort cadence); the pre-gathered state stays valid
    virtual ProbeTarget probe_begin() { return ProbeTarget{}; }
    // energy-only tile launch on geometry `which` (0: gathered / d_box, preceded by the flag-driven list launch; 1: gathered2 /
    // d_box2) that leaves per-workgroup partial sums; `table` (may be nullptr): a ForcePlan table of this precision whose
    // energies, evaluated on `coords`, ride along
    virtual void probe_energy(const int which, const double *d_box_which, const FusedTable *table, const int table_blocks, const double *coords,
                              hipStream_t stream, const i128 *&partials, int &count) {}
    // both geometries in ONE tile launch (k_nonbonded_tiles<..., DUAL>), preceded by the list launch: `r2_blocks` = n_r2 per-block
    // maxima of |atom - own molecule's centroid|^2 (what bounds the change of a pair distance between the two geometries)
    virtual void probe_energy_dual(const double *d_box2, const FusedTable *table, const int table_blocks, const double *coords, const double *coords2,
                                   const float *r2_blocks, const int n_r2, hipStream_t stream, const i128 *&partials, const i128 *&partials2, int &count) {}
    virtual double get_cutoff() const = 0;
    virtual double get_nblist_padding() const = 0;
    virtual double get_beta() const = 0;
    virtual int precision_bytes() const = 0;
    virtual bool is_interaction_group() const { return false; } // NonbondedInteractionGroup shares the all-pairs pipeline
    // ---- merging with an interaction group (ForcePlan::merge_producers) ----
    // all-pairs side: the potential (owned by this one) that evaluates this potential's pairs AND `group`'s in one pipeline, bound
    // to the group's parameters for the coming call -- or nullptr when the two do not fit (different precision / beta / cutoff,
    // the group's columns are not exactly this potential's atoms, an empty side, du/dp wanted ...)
    virtual Potential *merged_carrier(NonbondedAllPairsBase *group, const int P_group, const double *d_p_group) { return nullptr; }
    // The carrier is about to evaluate in this potential's place: whatever this potential remembers about its own inputs -- positions
    // an update kernel pre-gathered for it while it was a producer of its own -- describes a call it will not make.  Without this a
    // window that went  separate producers -> merged (restore of the group's atom set) -> separate again  found its all-pairs potential
    // holding a "valid" pre-gather from before the merged stretch behind the same pointers, and stepped on coordinates that many steps
    // old (tests/test_gpu_interleavings.py at config-5 size; every reference-shaped barostat attempt happened to clear it, attempts on
    // the current list do not).  The carrier's own state is not touched.
    virtual void carrier_took_over() {}
    // what the all-pairs side needs to know about a candidate group
    virtual const std::vector<unsigned int> &host_atom_idxs() const = 0; // group: rows (ascending) then columns; all-pairs: its atoms (ascending)
    virtual int num_group_rows() const { return 0; }
    virtual unsigned int idxs_version() const = 0;   // bumped by every change of the atom sets
    virtual unsigned int inputs_epoch() const = 0;   // bumped by invalidate_cached_inputs
    virtual bool hilbert_disabled() const = 0;
    virtual bool expects_box_scaling() const = 0;
    virtual bool is_empty_group() const { return false; }
    // local MD narrows an all-pairs potential to the free atoms for the length of a call and widens it again afterwards
    virtual void narrow_to(const std::vector<int> &atom_idxs) = 0;
    virtual std::vector<int> current_atom_idxs() = 0;
};

template <typename Real> class NonbondedAllPairs : public NonbondedAllPairsBase {
public:
    NonbondedAllPairs(const int N, const double beta, const double cutoff, const std::optional<std::vector<int>> &atom_idxs, const bool disable_hilbert_sort, const double nblist_padding);
    ~NonbondedAllPairs() {}

    void set_atom_idxs(const std::vector<int> &atom_idxs);
    std::vector<int> get_atom_idxs();
    void narrow_to(const std::vector<int> &atom_idxs) override { this->set_atom_idxs(atom_idxs); }
    std::vector<int> current_atom_idxs() override { return this->get_atom_idxs(); }
    int get_num_atom_idxs() const { return K_; }
    double get_beta() const override { return beta_; }
    int precision_bytes() const override { return static_cast<int>(sizeof(Real)); }
    bool piggyback_forces(const FusedTable *d_table, const int blocks, const int precision_bytes, u64 *acc, const int atom_stride, const int comp_stride) override;
    bool piggyback_lands_in_own_accumulator() const override;
    bool piggyback_energy(const FusedTable *d_table, const int blocks, const int precision_bytes) override;
    void drop_piggybacks() override {
        piggyback_table_ = nullptr;
        piggyback_blocks_ = 0;
        piggyback_energy_table_ = nullptr;
        piggyback_energy_blocks_ = 0;
    }
    bool execute_forces_deferred(const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, hipStream_t stream, DeferredForces &out) override;
    bool execute_energy_partials(const int N, const int P, const double *d_x, const double *d_p, const double *d_box, hipStream_t stream, const i128 *&partials, int &count, i128 *d_final = nullptr) override;
    void hint_same_frame(const bool on = true, const long long prev_call = 0) override {
        same_frame_hint_ = on;
        hint_call_ = prev_call;
        if (!on && merged_) {
            merged_->hint_same_frame(false);
        }
    }
    void pregather_committed(const double *d_x, const double *d_box, const bool sorted_bounds_done) override;
    bool probe_ready(const int N, const int P, const double *d_x, const double *d_p, const double *d_box) override;
    ProbeTarget probe_begin() override;
    void probe_energy(const int which, const double *d_box_which, const FusedTable *table, const int table_blocks, const double *coords, hipStream_t stream, const i128 *&partials, int &count) override;
    void probe_energy_dual(const double *d_box2, const FusedTable *table, const int table_blocks, const double *coords, const double *coords2, const float *r2_blocks, const int n_r2, hipStream_t stream, const i128 *&partials, const i128 *&partials2, int &count) override;
    void probe_list_launch(hipStream_t stream);
    void invalidate_cached_inputs() override {
        pre_valid_ = false;
        last_x_ = last_box_ = nullptr;
        inputs_epoch_++;
        if (merged_) {
            merged_->invalidate_cached_inputs();
        }
    }
    void expect_box_scaling() override { box_scales_ = true; }
    void carrier_took_over() override {
        pre_valid_ = false;
        last_x_ = last_box_ = nullptr;
    }
    Potential *merged_carrier(NonbondedAllPairsBase *group, const int P_group, const double *d_p_group) override;
    const std::vector<unsigned int> &host_atom_idxs() const override { return h_atom_idxs_; }
    int num_group_rows() const override { return group_rows_; }
    unsigned int idxs_version() const override { return idxs_version_; }
    unsigned int inputs_epoch() const override { return inputs_epoch_; }
    bool hilbert_disabled() const override { return disable_hilbert_; }
    bool expects_box_scaling() const override { return box_scales_; }
    bool is_empty_group() const override { return empty_; }
    // diagnostics (tests assert which path ran): force / energy evaluations this potential made as a merged carrier's host
    // diagnostic: {memo evaluations made, of which the device skipped the all-pairs launch} -- reads the device
    void memo_stats(long long *evaluations, long long *skipped);
    long long same_frame_skips() const { return same_frame_skips_ + (merged_ ? merged_->same_frame_skips_ : 0); }
    void merged_stats(long long *calls, unsigned int *tiles, unsigned int *builds) {
        *calls = merged_ ? merged_->pipeline_calls_ : 0;
        *tiles = merged_ ? merged_->num_tile_ixns() : 0;
        *builds = merged_ ? merged_->num_builds() : 0;
    }
    double get_cutoff() const override { return cutoff_; }
    double get_nblist_padding() const override { return nblist_padding_; }
    unsigned int num_tile_ixns() { return nblist_.num_tile_ixns(); }
    unsigned int num_builds() { return nblist_.num_builds(); }
    std::vector<long long> debug_timing(); // [grid][8] raw counters of the last tile-kernel launch (TM_TIMING builds)

    void execute_device(const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u, hipStream_t stream) override;
    void du_dp_fixed_to_float(const int N, const int P, const u64 *du_dp, double *du_dp_float) override;
    void du_dp_nonbonded_spans(const int N, const int P, const int offset, std::vector<DuDpSpan> &out) const override { out.push_back({offset, P}); }

protected:
    // shared with NonbondedInteractionGroup: same pipeline over K_ = (rows | columns) atoms with a row/column list
    struct GroupTag {};
    NonbondedAllPairs(const int N, const double beta, const double cutoff, const bool disable_hilbert_sort, const double nblist_padding, GroupTag);
    void allocate();
    // ---- merged carrier (ForcePlan::merge_producers; reference composition: fe/system.py:133-146) ----
    // A carrier is a NonbondedAllPairs of its own (own list, order, records, accumulators) over the slots
    //     [ group rows (guest_rows_) | holes up to a block boundary (guest_pad_) | all-pairs atoms ]
    // The list is the plain upper-triangular one except that the guest row blocks start their columns at block guest_pad_ / 32
    // (no guest x guest pairs) -- so host x host pairs and guest x host pairs come out of ONE list build, ONE tile launch and ONE
    // sorted hand-over.  Records: slot s under the parameters of its own potential (guest slots: the group's); the all-pairs
    // atoms once more under the group's parameters at record K_ + 1 + s, which the guest rows' items read as their columns
    // (kernels_nonbonded.hip.hpp: load_records).  Forces-only and energy-only evaluation; du/dp stays with the separate potentials.
    struct MergedTag {};
    NonbondedAllPairs(const int N, const double beta, const double cutoff, const bool disable_hilbert_sort, const double nblist_padding, MergedTag,
                      const std::vector<unsigned int> &host_idxs, const std::vector<unsigned int> &guest_idxs);
    int guest_rows_ = 0, guest_pad_ = 0; // merged carrier: L and L rounded up to 32 (0: not a carrier)
    int n_atoms_ = 0;                    // atoms that have a slot (K_ counts slots: == n_atoms_ unless merged)
    const double *guest_p_ = nullptr;    // the group's parameters for the coming call (merged_carrier binds them)
    bool covers_all() const { return n_atoms_ == N_ && group_rows_ == 0; }
    int slot_capacity() const { return merged_mode_ ? N_ + TILE : N_; }
    const bool merged_mode_ = false;
    std::unique_ptr<NonbondedAllPairs<Real>> merged_; // this potential's carrier, built on first use
    NonbondedAllPairsBase *merged_group_ = nullptr;   // ... for this group, at these versions of the two atom sets
    unsigned int merged_versions_[2] = {0, 0}, merged_group_epoch_ = 0;
    bool merged_refused_ = false;                     // the pair (merged_group_, versions) does not fit: do not ask again
    std::vector<unsigned int> h_atom_idxs_;           // host copy of d_atom_idxs_ (group: rows then columns)
    unsigned int idxs_version_ = 1, inputs_epoch_ = 1;
    long long pipeline_calls_ = 0;
    // ---- energy-only evaluations remembered (EnergyMemo) ----
    DeviceBuffer<EnergyMemo> d_memo_;
    DeviceBuffer<i128> d_u_partials_b_; // the second launch of a memo evaluation (guest rows' items + the plan's table)
    bool memo_chain_ = false;           // the last pipeline call was a memo evaluation: the device's memo describes the records
    i128 *memo_final_ = nullptr;        // execute_energy_partials: where the coming memo evaluation may leave its total
    long long hint_call_ = 0, hint_call_now_ = 0; // ... the call the hint speaks of (g_eval_serial at the time)
    long long last_call_ = -1;          // the call this pipeline last ran in
    bool same_frame_hint_ = false;      // hint_same_frame(): taken by the next evaluation entry point ...
    bool same_frame_now_ = false;       // ... for its run_pipeline
    long long same_frame_skips_ = 0;    // diagnostic: evaluations that launched no list kernel on the strength of the hint
    const double *last_x_ = nullptr, *last_box_ = nullptr; // the last run_pipeline's coordinate / box pointers (nullptr: its state was touched since)
    long long memo_skips_ = 0;          // (host-side count of memo evaluations; what the device decided is in d_memo_)
    const char *name_ = "NonbondedAllPairs"; // class name used in error messages
    int steps_per_sort_;
    int group_rows_ = 0;  // > 0: the first group_rows_ entries of d_atom_idxs_ are the row group (sorted separately)
    bool empty_ = false;  // interaction group with an empty side: execute_device does nothing
    const int N_;
    int K_;
    const double beta_, cutoff_, nblist_padding_;
    const double *d_es_table_ = nullptr; // f64 kernels: electrostatic force-factor table of beta_ (shared per device and beta)
    const bool disable_hilbert_;
    int calls_since_sort_;
    int parity_;
    bool force_rebuild_;
    int grid_;
    Neighborlist<Real> nblist_;
    std::unique_ptr<HilbertSort> hilbert_;
    DeviceBuffer<unsigned int> d_atom_idxs_, d_perm_;
    DeviceBuffer<Real> d_gathered_;
    DeviceBuffer<u64> d_g_du_dx_, d_g_du_dp_; // sorted accumulators, component-major with stride acc_stride_
    int acc_stride_ = 0;
    DeviceBuffer<double> d_snap_x_, d_snap_box_;
    DeviceBuffer<int> d_flags_;
    DeviceBuffer<i128> d_u_partials_;
    DeviceBuffer<Real> d_gathered2_;    // a probe's second geometry (ProbeTarget::gathered2), allocated on first use
    DeviceBuffer<i128> d_u_partials2_;  // ... and its partial sums
    const double *probe_d_box_ = nullptr; // the box pointer of the probe in flight (geometry 0)
    DeviceBuffer<long long> d_timing_; // per-wave cycle counters, filled only by -DTM_TIMING builds
    DeviceBuffer<int> d_slot_of_atom_;            // [N]: position of each atom in the sorted order, -1 = not one of ours
    void check_sizes(const int N, const int P) const;
    void run_pipeline(const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u, const bool scatter_du_dx, hipStream_t stream, const bool pregathered = false);
    // positions pre-gathered by the consumer of the last deferred call (see PregatherTarget): valid for exactly these
    // input pointers, dropped by any other call into the pipeline
    bool pre_valid_ = false;
    bool pre_sorted_ = false; // the consumer also left the block bounds done and resets the list counters with the flag
    const double *pre_x_ = nullptr, *pre_p_ = nullptr, *pre_box_ = nullptr, *offer_p_ = nullptr, *pre_guest_p_ = nullptr, *offer_guest_p_ = nullptr;
    const FusedTable *piggyback_table_ = nullptr; // consumed by the next forces-only call
    int piggyback_blocks_ = 0;
    const FusedTable *piggyback_energy_table_ = nullptr; // consumed by the next energy-only partial-sum call
    int piggyback_energy_blocks_ = 0;
    bool box_scales_ = false; // a barostat works on this potential (expect_box_scaling)
    // scale-aware rebuild test (kernels_nonbonded.hip.hpp: k_check_gather_scaled): on when a mover is at work and the padding
    // leaves room for the scale allowance
    bool scale_aware() const { return box_scales_ && g_box_scaling_reuse && 0.5 * list_padding() - 0.5 * 0.004 * (cutoff_ + list_padding()) > 0.25 * list_padding(); }
    // squared displacement (against the list build's snapshot) beyond which the list is rebuilt
    double rebuild_threshold2() const {
        if (static_list()) {
            // never: no displacement invalidates a complete list (and whoever raised the flag would also reset list counters
            // that no build refills -- an exploding system would lose its forces instead of being reported unstable)
            return std::numeric_limits<double>::infinity();
        }
        const double d = scale_aware() ? 0.5 * list_padding() - 0.5 * 0.004 * (cutoff_ + list_padding()) : 0.5 * list_padding();
        return d * d;
    }
    // SMALL systems keep a STATIC, complete list: with K <= static_list_max_k() interacting atoms every column block is listed for
    // every row block (a padding far beyond any box), so no displacement can invalidate the list, the rebuild flag never goes
    // up, and no list kernel is launched on MD steps at all -- at this size a step is three kernel launches' worth of latency,
    // not work, and the launch that only reads the flag is one of them.  The phase-2 test `d2 < cutoff^2` decides alone, as
    // always: same bits.  nblist_padding_ stays what the caller asked for (get_nblist_padding); this is what the list is built with.
    // (latched per potential by run_pipeline when the limit or the atom set changed -- the switch forces a rebuild --, so that the
    // rebuild threshold, the padding and the list that exists always belong to the same mode: tm_debug_set_static_list_max_k
    // may be called at any time)
    bool static_list() const { return static_mode_; }
    bool wants_static_list() const { return K_ <= static_list_max_k() && group_rows_ == 0; }
    bool static_mode_ = false;
    void sync_list_mode();
    double list_padding() const { return static_list() ? 1.0e3 : nblist_padding_; }
    static int static_list_max_k();
    bool static_list_built_ = false; // the complete list of the current order exists
    bool defer_u_reduce_ = false; // run_pipeline leaves the tile kernel's energy partials un-reduced (execute_energy_partials)
    int u_partials_count_ = 0;
    u64 *piggyback_acc_ = nullptr; // where the piggy-backed table's forces go, and its layout
    int piggyback_atom_stride_ = 3, piggyback_comp_stride_ = 1;
    bool piggyback_redirect_ = false; // the pending table accumulates into g_du_dx through slot_of_atom
};

// reference: cpp/src/nonbonded_interaction_group.{hpp,cu}.  Row atoms x column atoms (disjoint sets); both groups are
// Hilbert-sorted independently and laid out as [rows | columns] in the sorted order the tile pipeline works on.
template <typename Real> class NonbondedInteractionGroup : public NonbondedAllPairs<Real> {
public:
    NonbondedInteractionGroup(const int N, const std::vector<int> &row_atom_idxs, const std::vector<int> &col_atom_idxs, const double beta, const double cutoff, const bool disable_hilbert_sort, const double nblist_padding);
    void set_atom_idxs(const std::vector<int> &row_atom_idxs, const std::vector<int> &col_atom_idxs);
    bool is_interaction_group() const override { return true; }
    int get_num_row_idxs() const { return this->group_rows_; }
    int get_num_col_idxs() const { return this->empty_ ? n_cols_ : this->K_ - this->group_rows_; }
private:
    int n_cols_ = 0;
    static void validate_idxs(const int N, const std::vector<int> &row_atom_idxs, const std::vector<int> &col_atom_idxs, const bool allow_empty);
};

void nb_du_dp_fixed_to_float(const int N, const u64 *du_dp, double *out);
// debug: the device fixed-point conversions applied to host-supplied values (nonbonded.hip)
void debug_float_to_fixed(const int precision_bytes, const int kind, const int n, const double *h_in, u64 *h_out);
void debug_float_to_fixed_energy(const int precision_bytes, const int n, const double *h_in, i128 *h_out);

// reference: cpp/src/nonbonded_pair_list.{hpp,cu}; Negated == true is bound as NonbondedExclusions_*
template <typename Real, bool Negated> class NonbondedPairList : public Potential {
public:
    NonbondedPairList(const std::vector<int> &pair_idxs, const std::vector<double> &scales, const double beta, const double cutoff);
    void plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) override;
    void execute_device(const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u, hipStream_t stream) override;
    void du_dp_fixed_to_float(const int N, const int P, const u64 *du_dp, double *du_dp_float) override;
    void du_dp_nonbonded_spans(const int N, const int P, const int offset, std::vector<DuDpSpan> &out) const override { out.push_back({offset, P}); }
private:
    int M_;
    double beta_, cutoff_;
    const double *d_es_table_ = nullptr; // f64 kernels: electrostatic force-factor table of beta_
    DeviceBuffer<int> d_pair_idxs_;
    DeviceBuffer<double> d_scales_;
    DeviceBuffer<i128> d_u_partials_;
};

// reference: cpp/src/nonbonded_precomputed.{hpp,cu}
template <typename Real> class NonbondedPairListPrecomputed : public Potential {
public:
    NonbondedPairListPrecomputed(const std::vector<int> &pair_idxs, const double beta, const double cutoff);
    void plan_forces(const int N, const int P, const double *d_p, ForcePlan &plan) override;
    void execute_device(const int N, const int P, const double *d_x, const double *d_p, const double *d_box, u64 *d_du_dx, u64 *d_du_dp, i128 *d_u, hipStream_t stream) override;
    void du_dp_fixed_to_float(const int N, const int P, const u64 *du_dp, double *du_dp_float) override;
    void du_dp_nonbonded_spans(const int N, const int P, const int offset, std::vector<DuDpSpan> &out) const override { out.push_back({offset, P}); }
private:
    int B_;
    double beta_, cutoff_;
    DeviceBuffer<int> d_pair_idxs_;
    DeviceBuffer<i128> d_u_partials_;
    void check_size(const int P) const;
};

// ------------------------------------------------------------------------------------------------------------
// reference: cpp/src/integrator.hpp, cpp/src/langevin_integrator.{hpp,cu}
class Integrator {
public:
    virtual ~Integrator() {}
    virtual void step_fwd(std::vector<std::shared_ptr<BoundPotential>> &bps, double *d_x_t, double *d_v_t, double *d_box_t, unsigned int *d_idxs, hipStream_t stream) = 0;
    virtual void initialize(std::vector<std::shared_ptr<BoundPotential>> &bps, double *d_x_t, double *d_v_t, double *d_box_t, unsigned int *d_idxs, hipStream_t stream) = 0;
    virtual void finalize(std::vector<std::shared_ptr<BoundPotential>> &bps, double *d_x_t, double *d_v_t, double *d_box_t, unsigned int *d_idxs, hipStream_t stream) = 0;
    // coordinates or velocities were changed behind the integrator's back (Context setters, movers): drop derived state
    virtual void invalidate_state_cache() {}
    // Optional progress word: a 32-bit counter in pinned host memory in which the update kernel of every step leaves the number of
    // steps this integrator has enqueued so far (progress_enqueued) -- from one thread, as the kernel starts.  The stepping loops
    // read it to stay a bounded number of steps ahead of the device without a runtime call (integrator.hip, RunAhead).
    virtual const volatile unsigned int *progress_word() const { return nullptr; }
    virtual unsigned int progress_enqueued() const { return 0; }
};

template <typename Real> class LangevinIntegrator : public Integrator {
public:
    LangevinIntegrator(const int N, const double *masses, const double temperature, const double dt, const double friction, const int seed);
    ~LangevinIntegrator() override;
    double get_temperature() const { return temperature_; }
    const volatile unsigned int *progress_word() const override { return h_progress_; }
    unsigned int progress_enqueued() const override { return enqueued_; }
    void step_fwd(std::vector<std::shared_ptr<BoundPotential>> &bps, double *d_x_t, double *d_v_t, double *d_box_t, unsigned int *d_idxs, hipStream_t stream) override;
    void initialize(std::vector<std::shared_ptr<BoundPotential>> &, double *, double *, double *, unsigned int *, hipStream_t) override {}
    void finalize(std::vector<std::shared_ptr<BoundPotential>> &, double *, double *, double *, unsigned int *, hipStream_t) override {}
    void invalidate_state_cache() override { state_cache_valid_ = false; }
private:
    const int N_;
    const double temperature_;
    const Real dt_;
    const double friction_;
    Real ca_;
    unsigned long long seed_;
    unsigned long long step_;
    DeviceBuffer<Real> d_cbs_, d_ccs_;
    // x, v, cb, cc once more in the SLOT order of the nonbonded producer whose sorted hand-over the update kernel walks
    // (k_update_forward_baoab_sorted): written every sorted step, read instead of the atom-order arrays -- one dependent
    // memory hop less per atom -- whenever the producer confirms that nothing changed since (DeferredForces::
    // consumed_sorted_pregather) and nobody touched x / v behind the integrator's back
    DeviceBuffer<double> d_xs_, d_vs_;
    DeviceBuffer<Real> d_cbs_s_, d_ccs_s_;
    bool state_cache_valid_ = false;
    const void *cache_owner_ = nullptr;
    const double *cache_x_ = nullptr, *cache_v_ = nullptr;
    DeviceBuffer<u64> d_du_dx_;    // [N, 3]: what potentials that launch their own kernels add to
    int cm_stride_;
    DeviceBuffer<u64> d_du_dx_cm_; // component-major [3][cm_stride_]: what the fused table's terms add to
    ForcePlan plan_;
    std::vector<DeferredForces> deferred_;
    unsigned int *h_progress_ = nullptr; // pinned host memory (progress_word)
    unsigned int enqueued_ = 0;          // step_fwd calls so far, modulo 2^32
};

// reference: cpp/src/verlet_integrator.{hpp,cu}; kernels/k_integrator.cuh:64-130.  Arithmetic in double, as the reference
// instantiates its kernels.  cbs = -dt / mass (the Python dataclass flips the sign: lib/__init__.py:24-37).
class VelocityVerletIntegrator : public Integrator {
public:
    VelocityVerletIntegrator(const int N, const double dt, const double *h_cbs);
    void step_fwd(std::vector<std::shared_ptr<BoundPotential>> &bps, double *d_x_t, double *d_v_t, double *d_box_t, unsigned int *d_idxs, hipStream_t stream) override;
    void initialize(std::vector<std::shared_ptr<BoundPotential>> &bps, double *d_x_t, double *d_v_t, double *d_box_t, unsigned int *d_idxs, hipStream_t stream) override;
    void finalize(std::vector<std::shared_ptr<BoundPotential>> &bps, double *d_x_t, double *d_v_t, double *d_box_t, unsigned int *d_idxs, hipStream_t stream) override;
private:
    const int N_;
    const double dt_;
    bool initialized_;
    DeviceBuffer<double> d_cbs_;
    DeviceBuffer<u64> d_du_dx_;    // [N, 3]: what potentials that launch their own kernels add to
    int cm_stride_;
    DeviceBuffer<u64> d_du_dx_cm_; // component-major [3][cm_stride_]: what the fused table's terms add to
    ForcePlan plan_;
    std::vector<DeferredForces> deferred_;
    // mode 0: v += cb F, x += dt v;  1: v += cb/2 F, x += dt v;  2: v += cb/2 F
    void forces_then_update(const int mode, std::vector<std::shared_ptr<BoundPotential>> &bps, double *d_x_t, double *d_v_t, double *d_box_t, unsigned int *d_idxs, hipStream_t stream);
};

// reference: cpp/src/mover.{hpp,cu}
class Mover {
public:
    virtual ~Mover() {}
    virtual void move(const int N, double *d_x, double *d_box, hipStream_t stream) = 0;
    void move_host(const int N, const double *h_x, const double *h_box, double *h_x_out, double *h_box_out);
    void set_interval(const int interval) {
        if (interval <= 0) {
            throw std::runtime_error("interval must be greater than 0");
        }
        interval_ = interval;
        step_ = 0; // in `interval` steps from now the mover triggers
    }
    int get_interval() const { return interval_; }
    void set_step(const int step) {
        if (step < 0) {
            throw std::runtime_error("step must be at least 0");
        }
        step_ = step;
    }
    // false only when the last move() call provably left coordinates and box untouched (off-interval call); the
    // Context drops the potentials' pre-gathered inputs otherwise
    bool acted_last_call() const { return acted_; }
    // true when the last move() call changed coordinates / box (or not) but left the potentials' pre-gathered inputs describing the
    // result (the barostat's fast path commits an accepted proposal into them): the Context then only drops the integrator's
    // slot-ordered copies of x / v
    bool kept_potential_inputs() const { return kept_inputs_; }
    // called by the Context after it has waited for the stream its movers work on (end of a stepping call): a mover that leaves
    // itself notes from the device in host-visible memory checks them here -- and throws if the device reported a failure
    virtual void after_wait() {}
    // the bound potentials this mover evaluates on its own (a barostat's energy evaluations run in THEIR neighbor lists and
    // accumulators): Context::multiple_steps_group counts them as device state of the mover's context
    virtual std::vector<std::shared_ptr<BoundPotential>> held_potentials() const { return {}; }
protected:
    explicit Mover(const int interval) : interval_(interval), step_(0) {}
    int interval_;
    int step_;
    bool acted_ = true;
    bool kept_inputs_ = false;
};

// reference: cpp/src/barostat.{hpp,cu}, kernels/k_barostat.cuh.  Molecular-scaling Monte Carlo barostat: every
// `interval`-th call proposes a volume change, moves every molecule's centroid with the box, evaluates the bound
// potentials' energy before / after (two energy-only force-field evaluations) and accepts by Metropolis.
// Everything stays on the device; the two uniforms of an attempt come from Philox4x32-10 keyed on (seed; attempt).
template <typename Real> class MonteCarloBarostat : public Mover {
public:
    MonteCarloBarostat(const int N, const double pressure, const double temperature, const std::vector<std::vector<int>> &group_idxs, const int interval, const std::vector<std::shared_ptr<BoundPotential>> &bps, const int seed, const bool adaptive_scaling_enabled, const double initial_volume_scale_factor);
    void move(const int N, double *d_x, double *d_box, hipStream_t stream) override;
    std::vector<std::shared_ptr<BoundPotential>> held_potentials() const override { return bps_; }
    double get_volume_scale_factor();
    void set_volume_scale_factor(const double volume_scale_factor);
    bool get_adaptive_scaling() const { return adaptive_; }
    void set_adaptive_scaling(const bool enabled) { adaptive_ = enabled; }
    void set_pressure(const double pressure);
    // diagnostics used by the parity tests: (accepted, attempted) counters and the uniforms of attempt k
    void get_counters(int *accepted, int *attempted);
    // diagnostic: attempts since construction, and how many of them ran on the potential's current list (the fast path)
    void get_attempt_paths(long long *attempts, long long *fast) const {
        *attempts = static_cast<long long>(attempt_);
        *fast = fast_attempts_;
    }
    ~MonteCarloBarostat();
    void after_wait() override;
private:
    const int N_;
    bool adaptive_;
    std::vector<std::shared_ptr<BoundPotential>> bps_;
    Real pressure_, temperature_;
    const unsigned long long seed_;
    int num_mols_, num_grouped_atoms_;
    unsigned long long attempt_;
    DeviceBuffer<double> d_x_proposed_, d_box_proposed_, d_volume_scale_;
    DeviceBuffer<Real> d_move_; // {volume, volume_delta, length_scale, u2}
    DeviceBuffer<i128> d_u_buffer_, d_u_init_, d_u_final_;
    DeviceBuffer<u64> d_centroids_;
    DeviceBuffer<int> d_atom_idxs_, d_mol_idxs_, d_mol_offsets_, d_counters_;
    ForcePlan plan_; // the two energy evaluations of an attempt
    void reset_counters();
    // ---- fast path (barostat.hip: move_on_current_list): both energies on the nonbonded potential's current list ----
    DeviceBuffer<int4> d_mol_of_atom_;  // [N]: {molecule of the atom (-1: not grouped), its first entry in atom_idxs, its size, its first atom if consecutive else -1}
    DeviceBuffer<float> d_r2_blocks_;   // per block of the proposal kernel: max |atom - own centroid|^2 (the DUAL tile launch's filter margin)
    int max_mol_size_ = 0;
    long long fast_attempts_ = 0;
    bool centroids_clean_ = false; // d_centroids_ is all zero (left so by the last fast-path attempt)
    // pinned host word: 1 + the number of the first fast-path attempt whose proposal lay beyond the list's reach (0: none so far)
    unsigned int *h_overreach_ = nullptr;
    bool move_on_current_list(double *d_x, double *d_box, hipStream_t stream);
};

// reference: cpp/src/local_md_potentials.{hpp,cu}, local_md_utils.cu, kernels/k_local_md.cuh.
// Local MD: for the length of one multiple_steps_local* call only the "free" atoms move -- those selected around a
// reference atom -- held near it by a flat-bottom restraint, while the rest of the system stays frozen.  The context's
// own potentials are kept; the one NonbondedAllPairs among them is narrowed to free x free, a NonbondedInteractionGroup of
// the same parameters adds free x frozen (frozen x frozen forces move nobody), and the restraints are appended.
// Unlike the reference (a chain of index kernels + two device partitions + a D2H count), the selection flags are computed
// by one kernel, read back once, and the row / column / bond lists are laid out on the host: it is a per-call setup
// (hundreds of steps follow), the lists have to be known to the host anyway (sizes, validation, neighbor-list resize),
// and every list comes out in ascending atom order -- deterministic, where the reference's partition order is not.
class LocalMDPotentials {
public:
    LocalMDPotentials(const int N, const std::vector<std::shared_ptr<BoundPotential>> &bps, const bool freeze_reference, const double temperature);
    // reference atom = local_idxs[mt19937(seed) draw]; atom i is free with probability exp(-U_flat_bottom(r_i) / kT)
    void setup_from_idxs(const double *d_x_t, const double *d_box_t, const std::vector<int> &local_idxs, const int seed, const double radius, const double k, hipStream_t stream);
    // the caller chose the free atoms
    void setup_from_selection(const int reference_idx, const std::vector<int> &selection_idxs, const double radius, const double k, hipStream_t stream);
    std::vector<std::shared_ptr<BoundPotential>> &get_potentials() { return all_potentials_; }
    unsigned int *get_free_idxs() { return d_free_idxs_.data; } // [N]: i if atom i is free, N otherwise
    void reset_potentials(); // the all-pairs potential gets its own atom set back
    const bool freeze_reference;
    const double temperature;
    // what the last setup decided (tests / diagnostics)
    int last_reference_idx() const { return last_reference_; }
    const std::vector<unsigned int> &last_free_idxs() const { return h_free_; }

private:
    const int N_;
    std::vector<std::shared_ptr<BoundPotential>> all_potentials_;
    std::shared_ptr<NonbondedAllPairsBase> all_pairs_;
    std::vector<int> all_pairs_idxs_;      // its atom set outside local MD
    std::vector<char> in_all_pairs_;       // [N] membership
    std::shared_ptr<NonbondedAllPairsBase> ixn_group_;
    std::shared_ptr<FlatBottomBond<float, false>> free_restraint_;
    std::shared_ptr<BoundPotential> bound_free_restraint_;
    std::shared_ptr<FlatBottomBond<float, true>> frozen_restraint_;
    std::shared_ptr<BoundPotential> bound_frozen_restraint_;
    DeviceBuffer<unsigned int> d_free_idxs_;
    std::vector<unsigned int> h_free_;
    int last_reference_ = -1;
    void setup_given_free_flags(const int reference_idx, const double radius, const double k, hipStream_t stream);
};

void verify_local_md_parameters(const double radius, const double k);

// reference: cpp/src/context.{hpp,cu}
class Context {
public:
    Context(int N, const double *x_0, const double *v_0, const double *box_0, std::shared_ptr<Integrator> intg, std::vector<std::shared_ptr<BoundPotential>> &bps, std::vector<std::shared_ptr<Mover>> &movers);
    ~Context();

    void step();
    void initialize();
    void finalize();
    void multiple_steps(const int n_steps, const int n_samples, double *h_x, double *h_box);
    // n_steps of SEVERAL contexts, interleaved step by step on streams of their own (fed by one or two enqueueing host threads,
    // TM_AMD_GROUP_THREADS): the device then runs one context's list / update kernels and kernel boundaries underneath another's force kernel.  Two
    // DHFR-sized replicas step at 59.5 us each this way against 69.6 us alone (free-energy windows and HREX replicas that share a
    // GPU).  Same trajectories as n_steps of multiple_steps on each: the contexts share nothing.
    // No frames are stored (as multiple_steps with store_x_interval = 0).  The contexts must be distinct objects.
    static void multiple_steps_group(const std::vector<Context *> &ctxts, const int n_steps);
    // reference: context.cu:90-213.  Movers do not run during local MD (context.cu:268).
    void setup_local_md(const double temperature, const bool freeze_reference);
    void multiple_steps_local(const int n_steps, const std::vector<int> &local_idxs, const int n_samples, const double radius, const double k, const int seed, double *h_x, double *h_box);
    void multiple_steps_local_selection(const int n_steps, const int reference_idx, const std::vector<int> &selection_idxs, const int n_samples, const double radius, const double k, double *h_x, double *h_box);
    // diagnostics of the last local-MD setup: the reference atom and the [N] free-index array the integrator was given
    int local_md_last_reference() const;
    std::vector<unsigned int> local_md_last_free_idxs() const;
    int num_atoms() const { return N_; }
    void set_x_t(const double *in);
    void set_v_t(const double *in);
    void set_box(const double *in);
    void get_x_t(double *out) const;
    void get_v_t(double *out) const;
    void get_box(double *out) const;
    std::shared_ptr<Integrator> get_integrator() const { return intg_; }
    std::vector<std::shared_ptr<BoundPotential>> get_potentials() const { return bps_; }
    std::vector<std::shared_ptr<Mover>> get_movers() const { return movers_; }
    hipStream_t stream() const { return stream_; }
    // device time of the steps of the last multiple_steps call: HIP events recorded on the context's stream right
    // before the first step and right after the last one (the final frame's device-to-host copy comes after)
    double last_multiple_steps_ms();

private:
    hipEvent_t ev_start_ = nullptr, ev_stop_ = nullptr;
    bool ev_valid_ = false;
    int N_;
    std::vector<std::shared_ptr<Mover>> movers_;
    int step_;
    DeviceBuffer<double> d_x_t_, d_v_t_, d_box_t_;
    std::shared_ptr<Integrator> intg_;
    std::vector<std::shared_ptr<BoundPotential>> bps_;
    std::vector<double> nb_cutoffs_with_padding_;
    hipStream_t stream_;
    std::unique_ptr<LocalMDPotentials> local_md_pots_;
    void _step(hipStream_t stream);
    double _get_temperature() const;
    void _ensure_local_md_initialized();
    void _run_local_steps(const int n_steps, const int n_samples, double *h_x, double *h_box);
    void invalidate_potential_inputs();
    void _verify_coords_and_box(const double *coords, const double *box, hipStream_t stream);
};

// recursive search used by Context for the box-size check (reference: cpp/src/nonbonded_common.cpp:77-124)
void collect_nonbonded_cutoffs(const std::shared_ptr<Potential> &pot, std::vector<double> &out);

} // namespace tmamd
