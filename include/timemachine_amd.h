/* timemachine_amd.h -- C ABI of the MI355X-native timemachine hot path (libtimemachine_amd.so).
 *
 * This is the drop-in boundary.  Every entry point is the plain-C door onto one method that the reference binds
 * with pybind11 in timemachine/cpp/src/wrap_kernels.cpp (the `timemachine.lib.custom_ops` module); the citation
 * next to each declaration is the reference interface it replaces (paths relative to the reference checkout).
 * Host arrays in, host arrays out, exactly like the pybind layer: coordinates / parameters / boxes are C-contiguous
 * f64, index arrays are int32 (uint32 where the reference uses uint32), forces and parameter derivatives come
 * back as the raw 64-bit fixed-point accumulators (scale 2^36; see tm_fixed_to_float / tm_potential_du_dp_fixed_to_float),
 * energies as signed 128-bit fixed point (tm_int128; "overflowed" means |u| reaches the int64 range => NaN).
 *
 * Conventions
 *   - every function returns TM_OK (0) or an error code; tm_last_error() returns the message of the last failure
 *     on the calling thread.  Messages are the reference's std::runtime_error texts verbatim.
 *   - handles are opaque and reference counted internally: a BoundPotential keeps its Potential alive, a
 *     Summed/Fanout potential keeps its children alive, a Context keeps integrator and potentials alive
 *     (reference: all bound classes are held by std::shared_ptr).  Destroy every handle you were given exactly once.
 *   - Threading: objects are stateful and NOT thread-safe (reference: cpp/src/potential.hpp:7).  Every entry point runs under the
 *     recursive lock of the calling thread's CURRENT DEVICE (hipGetDevice), held for the whole call (a tm_context_multiple_steps
 *     call included): two threads never enter objects of one device at once, and threads that drive DIFFERENT GPUs from one
 *     process -- each with its own device current (tm_set_device) and its own objects -- run concurrently (one process-wide lock
 *     up to round 5).  One process per GPU, as bench.py and the reference's parallel/client.py do, remains the tested layout.  A
 *     binding that keeps the Python GIL while it waits for the lock stalls the interpreter; the compiled binding releases the GIL
 *     around every call that reaches the device.  Process-wide debug switches (tm_debug_set_*) are plain flags: set them before
 *     threads start.
 *   - loading the library exports GPU_MAX_HW_QUEUES=8 to the process environment unless the variable is already set (see
 *     tm_context_multiple_steps_group); nothing else in the environment is touched.
 *   - a long tm_context_multiple_steps[_group] call does not spin on the host while the device works: the enqueueing thread stays
 *     at most 64 steps ahead of the device (TM_AMD_RUN_AHEAD_STEPS) and sleeps in between (it reads a progress word the integrator's kernel leaves in
 *     pinned host memory), and the end of the call is awaited by polling an event with short sleeps: ~0.2-0.3 CPUs busy per
 *     process instead of 1.0-1.6.  Results and device time do not depend on it; TM_AMD_SPIN_WAIT=1 in the environment restores
 *     unbounded enqueueing and the runtime's spinning waits (the call then returns ~30 us sooner).
 *   - "precision" selects the arithmetic type of the kernels (the reference's *_f32 / *_f64 classes); storage of
 *     coordinates, velocities, box and parameters is always f64.
 */
#ifndef TIMEMACHINE_AMD_H
#define TIMEMACHINE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TM_OK 0
#define TM_ERR_RUNTIME 1          /* -> Python RuntimeError (reference: std::runtime_error) */
#define TM_ERR_INVALID_HARDWARE 2 /* -> custom_ops.InvalidHardware (reference: cpp/src/exceptions.hpp, gpu_utils.cuh:27-53) */

#define TM_F32 0
#define TM_F64 1

#define TM_FIXED_EXPONENT_VALUE 0x1000000000ULL /* custom_ops.FIXED_EXPONENT, wrap_kernels.cpp:2144; fixed_point.hpp:5 */

typedef struct tm_int128 { /* little-endian two's-complement, layout-identical to __int128 */
    uint64_t lo;
    int64_t hi;
} tm_int128;

typedef struct tm_potential_s *tm_potential_t;
typedef struct tm_bound_potential_s *tm_bound_potential_t;
typedef struct tm_integrator_s *tm_integrator_t;
typedef struct tm_mover_s *tm_mover_t;
typedef struct tm_context_s *tm_context_t;
typedef struct tm_neighborlist_s *tm_neighborlist_t;
typedef struct tm_hilbert_sort_s *tm_hilbert_sort_t;

/* ---- library / device ------------------------------------------------------------------------------------- */
const char *tm_last_error(void);
const char *tm_version(void);
int tm_device_count(int *count);
int tm_set_device(int device);     /* one process per GPU: call with LOCAL_RANK before creating objects */
int tm_device_synchronize(void);
int tm_device_reset(void);         /* custom_ops.cuda_device_reset(), wrap_kernels.cpp:2222-2225 */
int tm_device_name(char *buf, size_t cap);

/* fixed point helpers (host).  cpp/src/fixed_point.hpp:13-34, wrap_kernels.cpp:83-89 */
double tm_fixed_to_float(uint64_t v);
int tm_energy_overflowed(const tm_int128 *u);            /* fixed_point_overflow */
double tm_energy_to_float(const tm_int128 *u);           /* convert_energy_to_fp: NaN when overflowed */

/* ---- potentials: construction ----------------------------------------------------------------------------- */
/* HarmonicBond_f32/_f64(bond_idxs int32[B,2])                      wrap_kernels.cpp:1311-1322; cpp/src/harmonic_bond.cu:11-34 */
int tm_harmonic_bond_create(int precision, const int32_t *bond_idxs, int num_bonds, tm_potential_t *out);
/* HarmonicAngle_*(angle_idxs int32[A,3])                            wrap_kernels.cpp:1396-1408; harmonic_angle.cu:11-36 */
int tm_harmonic_angle_create(int precision, const int32_t *angle_idxs, int num_angles, tm_potential_t *out);
/* PeriodicTorsion_*(angle_idxs int32[T,4])                          wrap_kernels.cpp:1432-1444; periodic_torsion.cu:11-38 */
int tm_periodic_torsion_create(int precision, const int32_t *torsion_idxs, int num_torsions, tm_potential_t *out);
/* NonbondedAllPairs_*(num_atoms, beta, cutoff, atom_idxs_i=None, disable_hilbert_sort=False, nblist_padding=0.1)
 *                                                                    wrap_kernels.cpp:1446-1478; nonbonded_all_pairs.cu:21-86
 * atom_idxs == NULL means "all atoms"; duplicates are removed as the binding does (unique_idxs). */
int tm_nonbonded_all_pairs_create(int precision, int num_atoms, double beta, double cutoff, const int32_t *atom_idxs,
                                  int num_atom_idxs, int disable_hilbert_sort, double nblist_padding, tm_potential_t *out);
/* NonbondedPairList_* (negated=0) / NonbondedExclusions_* (negated=1)(pair_idxs_i int32[M,2], scales_i f64[M,2], beta, cutoff)
 *                                                                    wrap_kernels.cpp:1563-1589; nonbonded_pair_list.cu:12-50 */
int tm_nonbonded_pair_list_create(int precision, int negated, const int32_t *pair_idxs, int num_pairs, const double *scales,
                                  int num_scales, double beta, double cutoff, tm_potential_t *out);
/* NonbondedInteractionGroup_*(num_atoms, row_atom_idxs_i, beta, cutoff, col_atom_idxs_i=None, disable_hilbert_sort=False,
 * nblist_padding=0.1)                                                wrap_kernels.cpp:1481-1561; nonbonded_interaction_group.cu:21-101
 * col_atom_idxs == NULL means "every atom that is not a row atom". */
int tm_nonbonded_interaction_group_create(int precision, int num_atoms, const int32_t *row_atom_idxs, int num_rows,
                                          const int32_t *col_atom_idxs, int num_cols, double beta, double cutoff,
                                          int disable_hilbert_sort, double nblist_padding, tm_potential_t *out);
/* NonbondedInteractionGroup_*.set_atom_idxs(row_atom_idxs, col_atom_idxs)   wrap_kernels.cpp:1485-1504; nonbonded_interaction_group.cu:262-334 */
int tm_nonbonded_interaction_group_set_atom_idxs(tm_potential_t pot, const int32_t *row_atom_idxs, int num_rows,
                                                 const int32_t *col_atom_idxs, int num_cols);
/* NonbondedPairListPrecomputed_*(pair_idxs int32[B,2], beta, cutoff); params are [B,4] = (q_ij, sig_ij, eps_ij, w_offset_ij)
 *                                                                    wrap_kernels.cpp:1353-1364; nonbonded_precomputed.cu:12-87 */
int tm_nonbonded_pair_list_precomputed_create(int precision, const int32_t *pair_idxs, int num_pairs, double beta, double cutoff,
                                              tm_potential_t *out);
/* FlatBottomBond_*(bond_idxs int32[B,2]) (log_form = 0) / LogFlatBottomBond_*(bond_idxs, beta) (log_form = 1); params [B,3] =
 * (k, r_min, r_max)                                 wrap_kernels.cpp:1324-1351; flat_bottom_bond.cu:12-89, log_flat_bottom_bond.cu:12-93 */
int tm_flat_bottom_bond_create(int precision, int log_form, const int32_t *bond_idxs, int num_bonds, double beta, tm_potential_t *out);
/* CentroidRestraint_*(group_a_idxs, group_b_idxs, kb, b0); no parameters                wrap_kernels.cpp:1410-1430; centroid_restraint.cu:8-76 */
int tm_centroid_restraint_create(int precision, const int32_t *group_a_idxs, int num_a, const int32_t *group_b_idxs, int num_b,
                                 double kb, double b0, tm_potential_t *out);
/* ChiralAtomRestraint_*(idxs int32[R,4]); params [R]                 wrap_kernels.cpp:1366-1378; chiral_atom_restraint.cu:10-68 */
int tm_chiral_atom_restraint_create(int precision, const int32_t *idxs, int num_restraints, tm_potential_t *out);
/* ChiralBondRestraint_*(idxs int32[R,4], signs int32[R]); params [R] wrap_kernels.cpp:1380-1394; chiral_bond_restraint.cu:10-82 */
int tm_chiral_bond_restraint_create(int precision, const int32_t *idxs, int num_restraints, const int32_t *signs, int num_signs,
                                    tm_potential_t *out);
/* SummedPotential(potentials, params_sizes, parallel=True)          wrap_kernels.cpp:1661-1676; summed_potential.cu:13-26
   `parallel` is accepted for interface parity and ignored: children run in sequence, results cannot depend on it. */
int tm_summed_potential_create(const tm_potential_t *potentials, int num_potentials, const int32_t *params_sizes,
                               int num_params_sizes, int parallel, tm_potential_t *out);
/* FanoutSummedPotential(potentials, parallel=True)                  wrap_kernels.cpp:1678-1691; fanout_summed_potential.cu:9-16 */
int tm_fanout_summed_potential_create(const tm_potential_t *potentials, int num_potentials, int parallel, tm_potential_t *out);
int tm_potential_destroy(tm_potential_t pot);

/* .get_potentials() of Summed / Fanout: fills up to cap NEW handles (destroy each); *count = number of children */
int tm_potential_get_children(tm_potential_t pot, tm_potential_t *out, int cap, int *count);
/* NonbondedAllPairs.set_atom_idxs / get_atom_idxs / get_num_atom_idxs   wrap_kernels.cpp:1452-1454 */
int tm_nonbonded_all_pairs_set_atom_idxs(tm_potential_t pot, const int32_t *atom_idxs, int num_atom_idxs);
int tm_nonbonded_all_pairs_get_num_atom_idxs(tm_potential_t pot, int *count);
int tm_nonbonded_all_pairs_get_atom_idxs(tm_potential_t pot, int32_t *out, int cap);
/* tiles (32 rows x 32 columns) in the current interaction list; diagnostic used by bench.py */
int tm_nonbonded_all_pairs_get_tile_count(tm_potential_t pot, unsigned int *count);
/* diagnostic: neighbor-list builds since construction (rebuild period of an MD run = calls / builds) */
int tm_nonbonded_all_pairs_get_build_count(tm_potential_t pot, unsigned int *count);
/* diagnostic: an all-pairs potential planned next to an interaction group on exactly its atoms (the reference's HostGuestSystem,
 * fe/system.py:133-146: Nonbonded(atom_idxs=host) + NonbondedInteractionGroup(ligand, host)) evaluates BOTH pair sets in one pipeline
 * of its own -- one list, one tile launch, one sorted hand-over to the integrator (csrc/engine.hpp: merged carrier;
 * tm_debug_set_merge_producers).  *calls = force / energy evaluations made that way since construction (0: never merged);
 * *tiles / *builds = tile count and list builds of that pipeline's list (as the two entry points above report for the potential's own) */
int tm_nonbonded_all_pairs_get_merged_stats(tm_potential_t pot, long long *calls, unsigned int *tiles, unsigned int *builds);
/* diagnostic: energy-only evaluations that gather for themselves (batches over stored frames / parameter sets) are remembered on the
 * device: when the check + gather kernel finds every operand of the all-pairs items unchanged (values as the kernels read them, the
 * box, the order), the all-pairs launch gets an empty item list and its sum is the remembered one -- the same integer, the cost of a
 * kernel prologue (csrc/engine.hpp: EnergyMemo; tm_debug_set_energy_memo).  *evaluations = such evaluations since construction (the
 * potential's own and its merged carrier's), *skipped = those whose all-pairs launch was empty. */
int tm_nonbonded_all_pairs_get_memo_stats(tm_potential_t pot, long long *evaluations, long long *skipped);
/* diagnostic: evaluations (of the potential and of its merged carrier) that launched no neighbor-list kernel because the batch entry
 * point vouched that coordinates and box were the previous evaluation's (tm_potential_execute_batch* walking the parameter sets of one
 * frame; the reference orders its loops the same way, wrap_kernels.cpp:997-1001) -- csrc/engine.hpp: Potential::hint_same_frame;
 * tm_debug_set_same_frame_hint is the A/B switch. */
int tm_nonbonded_all_pairs_get_same_frame_skips(tm_potential_t pot, long long *skips);
/* per-wave cycle counters of the last tile-kernel launch: [waves][8] = {setup, phase1, phase2, flush, items, batches, total, 0};
 * all zero unless the library was built with -DTM_TIMING (development aid, see scripts/ablate.py) */
int tm_nonbonded_all_pairs_debug_timing(tm_potential_t pot, long long *out, int cap, int *n);

/* ---- potentials: evaluation --------------------------------------------------------------------------------
 * Potential.execute(coords[N,3], params[P], box[3,3], compute_du_dx, compute_du_dp, compute_u)
 *                                                                    wrap_kernels.cpp:1039-1105; potential.cu:224-292
 * Pass NULL for an output that is not requested.  du_dx: uint64[N*3]; du_dp: uint64[P]; u: tm_int128[1]. */
int tm_potential_execute(tm_potential_t pot, int N, int P, const double *coords, const double *params, const double *box,
                         uint64_t *du_dx, uint64_t *du_dp, tm_int128 *u);
/* Potential.execute_batch(coords[C,N,3], params[Pb,P], boxes[C,3,3], ...) -> du_dx[C,Pb,N,3], du_dp[C,Pb,P], u[C,Pb]
 *                                                                    wrap_kernels.cpp:731-862; potential.cu:70-145 */
int tm_potential_execute_batch(tm_potential_t pot, int coord_batch_size, int N, int param_batch_size, int P,
                               const double *coords, const double *params, const double *boxes, uint64_t *du_dx,
                               uint64_t *du_dp, tm_int128 *u);
/* Potential.execute_batch_sparse(coords, params, boxes, coords_batch_idxs u32[B], params_batch_idxs u32[B], ...)
 *                                                                    wrap_kernels.cpp:863-1038; potential.cu:147-222 */
int tm_potential_execute_batch_sparse(tm_potential_t pot, int coords_size, int N, int params_size, int P, int batch_size,
                                      const uint32_t *coords_batch_idxs, const uint32_t *params_batch_idxs,
                                      const double *coords, const double *params, const double *boxes, uint64_t *du_dx,
                                      uint64_t *du_dp, tm_int128 *u);
/* The same three calls as the reference's BINDING delivers them (wrap_kernels.cpp:1066-1101, 820-860, 960-990): du_dx / du_dp / u as
 * doubles (FIXED_TO_FLOAT, the virtual du_dp_fixed_to_float with its per-column exponents, convert_energy_to_fp: NaN for an
 * overflowed energy) -- converted ON THE DEVICE, inputs staged through pinned memory in one copy, one synchronisation per call.
 * Values equal the u64 forms converted on the host bit for bit (tests/test_gpu_potentials_surface.py); the u64 forms stay for
 * callers that want the integers.  The compiled pybind11 module calls these; its ctypes twin keeps calling the u64 forms and
 * converting on the host, which makes every test that runs on both bindings a comparison of the two. */
int tm_potential_execute_f64(tm_potential_t pot, int N, int P, const double *coords, const double *params, const double *box,
                             double *du_dx, double *du_dp, double *u);
int tm_potential_execute_batch_f64(tm_potential_t pot, int coord_batch_size, int N, int param_batch_size, int P,
                                   const double *coords, const double *params, const double *boxes, double *du_dx, double *du_dp, double *u);
int tm_potential_execute_batch_sparse_f64(tm_potential_t pot, int coords_size, int N, int params_size, int P, int batch_size,
                                          const uint32_t *coords_batch_idxs, const uint32_t *params_batch_idxs, const double *coords,
                                          const double *params, const double *boxes, double *du_dx, double *du_dp, double *u);
/* virtual Potential::du_dp_fixed_to_float (per-column exponents for nonbonded terms, slices for Summed)
 *                                                                    potential.cu:322-326; nonbonded_all_pairs.cu:292-308 */
int tm_potential_du_dp_fixed_to_float(tm_potential_t pot, int N, int P, const uint64_t *du_dp, double *out);
/* Potential::execute_device: everything already resident in HBM (device pointers, hipStream_t as void*).
 * Accumulates into d_du_dx / d_du_dp, overwrites d_u.             cpp/src/potential.hpp:87-96 */
int tm_potential_execute_device(tm_potential_t pot, int N, int P, const double *d_x, const double *d_p, const double *d_box,
                                uint64_t *d_du_dx, uint64_t *d_du_dp, tm_int128 *d_u, void *hip_stream);

/* ---- BoundPotential(potential, params)                            wrap_kernels.cpp:1133-1309; bound_potential.cu ---- */
int tm_bound_potential_create(tm_potential_t pot, const double *params, int P, tm_bound_potential_t *out);
int tm_bound_potential_destroy(tm_bound_potential_t bp);
int tm_bound_potential_set_params(tm_bound_potential_t bp, const double *params, int P); /* size must match: RuntimeError */
int tm_bound_potential_size(tm_bound_potential_t bp, int *size);
int tm_bound_potential_get_potential(tm_bound_potential_t bp, tm_potential_t *out); /* NEW handle */
int tm_bound_potential_execute(tm_bound_potential_t bp, int N, const double *coords, const double *box, uint64_t *du_dx, tm_int128 *u);
int tm_bound_potential_execute_batch(tm_bound_potential_t bp, int coord_batch_size, int N, const double *coords,
                                     const double *boxes, uint64_t *du_dx, tm_int128 *u);
/* ... and as the binding delivers them (doubles; see tm_potential_execute_f64)                   wrap_kernels.cpp:1186-1290 */
int tm_bound_potential_execute_f64(tm_bound_potential_t bp, int N, const double *coords, const double *box, double *du_dx, double *u);
int tm_bound_potential_execute_batch_f64(tm_bound_potential_t bp, int coord_batch_size, int N, const double *coords, const double *boxes,
                                         double *du_dx, double *u);

/* ---- LangevinIntegrator(masses f64[N], temperature, dt, friction, seed)   wrap_kernels.cpp:691-715; langevin_integrator.cu:14-43
 * Bound as <float> only, like the reference (wrap_kernels.cpp:700). */
int tm_langevin_integrator_create(const double *masses, int N, double temperature, double dt, double friction, int seed,
                                  tm_integrator_t *out);
/* VelocityVerletIntegrator(dt, cbs f64[N]) with cbs = -dt / mass      wrap_kernels.cpp:717-729; verlet_integrator.cu:10-111 */
int tm_velocity_verlet_integrator_create(double dt, const double *cbs, int N, tm_integrator_t *out);
int tm_integrator_destroy(tm_integrator_t intg);

/* ---- Context(x0, v0, box, integrator, bps, movers=None)           wrap_kernels.cpp:296-689; context.cu ---------- */
int tm_context_create(const double *x0, const double *v0, const double *box, int N, tm_integrator_t intg,
                      const tm_bound_potential_t *bps, int num_bps, tm_context_t *out);
/* Context(x0, v0, box, integrator, bps, movers)                          wrap_kernels.cpp:296-335; context.cu:28-50,262-277 */
int tm_context_create_with_movers(const double *x0, const double *v0, const double *box, int N, tm_integrator_t intg,
                                  const tm_bound_potential_t *bps, int num_bps, const tm_mover_t *movers, int num_movers,
                                  tm_context_t *out);
/* MonteCarloBarostat(N, pressure [bar], temperature [K], group_idxs, interval, bps, seed, adaptive_scaling_enabled,
 * initial_volume_scale_factor)  -- <float> arithmetic as bound by the reference       wrap_kernels.cpp:1619-1659; barostat.cu:19-259
 * group_idxs arrives flattened: atoms of group g are group_atom_idxs[group_offsets[g] .. group_offsets[g+1]). */
int tm_monte_carlo_barostat_create(int N, double pressure, double temperature, const int32_t *group_atom_idxs,
                                   const int32_t *group_offsets, int num_groups, int interval, const tm_bound_potential_t *bps,
                                   int num_bps, int seed, int adaptive_scaling_enabled, double initial_volume_scale_factor,
                                   tm_mover_t *out);
/* Mover.set_interval / get_interval / set_step / move(coords, box)     wrap_kernels.cpp:1591-1617; mover.hpp:12-46, mover.cu:7-23 */
int tm_mover_destroy(tm_mover_t mover);
int tm_mover_set_interval(tm_mover_t mover, int interval);
int tm_mover_get_interval(tm_mover_t mover, int *interval);
int tm_mover_set_step(tm_mover_t mover, int step);
int tm_mover_move(tm_mover_t mover, int N, const double *x, const double *box, double *x_out, double *box_out);
/* MonteCarloBarostat accessors                                          wrap_kernels.cpp:1654-1658 */
int tm_barostat_set_volume_scale_factor(tm_mover_t mover, double volume_scale_factor);
int tm_barostat_get_volume_scale_factor(tm_mover_t mover, double *volume_scale_factor);
int tm_barostat_set_adaptive_scaling(tm_mover_t mover, int enabled);
int tm_barostat_get_adaptive_scaling(tm_mover_t mover, int *enabled);
int tm_barostat_set_pressure(tm_mover_t mover, double pressure);
/* diagnostic (not in the reference surface): acceptance counters since the last adaptive reset */
int tm_barostat_get_counters(tm_mover_t mover, int *accepted, int *attempted);
/* diagnostic (not in the reference surface): attempts since construction, and how many of them ran on the nonbonded potential's
 * current list (csrc/barostat.hip, "the fast path"; tm_debug_set_barostat_fast_path) */
int tm_barostat_get_attempt_paths(tm_mover_t mover, long long *attempts, long long *fast);
int tm_context_destroy(tm_context_t ctxt);
int tm_context_num_atoms(tm_context_t ctxt, int *N);
int tm_context_step(tm_context_t ctxt);
int tm_context_initialize(tm_context_t ctxt);
int tm_context_finalize(tm_context_t ctxt);
/* multiple_steps(n_steps, store_x_interval): the caller computes n_samples = n_steps / (store_x_interval or n_steps)
 * exactly as the binding does (wrap_kernels.cpp:347-369) and passes xs[n_samples,N,3], boxes[n_samples,3,3]. */
int tm_context_multiple_steps(tm_context_t ctxt, int n_steps, int n_samples, double *xs, double *boxes);
/* n_steps of several DISTINCT contexts, interleaved step by step on streams of their own (no frames are stored: as
 * tm_context_multiple_steps with n_samples = 0 on each).  The device runs one context's list / update kernels and kernel
 * boundaries underneath another's force kernel: the way free-energy windows or HREX replicas that share a GPU should be stepped.
 * Trajectories are exactly those of separate calls (the contexts share no state; contexts that do -- a potential bound twice, a
 * shared integrator or mover, a mover built on another context's bound potentials -- are refused).  The contexts are dealt to
 * TM_AMD_GROUP_THREADS enqueueing host threads inside the call (default: 2 while every context has at most 5000 atoms -- there the
 * 6-10 us a host thread needs per context-step is the limit --, 1 above); the calling thread returns when all have finished.
 * Every stream needs a hardware queue of its own: the library exports GPU_MAX_HW_QUEUES=8 (unless the variable is already set)
 * when it is loaded, which the HIP runtime reads when it first touches the device -- a process that initialised HIP before loading
 * this library should export the variable itself (with the runtime's default of 4, four replicas step ~10 % slower).
 * No counterpart in wrap_kernels.cpp: the reference steps the contexts of a device one after the other
 * (fe/free_energy.py:1537-1551 loops over windows). */
int tm_context_multiple_steps_group(const tm_context_t *ctxts, int n_ctxts, int n_steps);
/* measurement aid (bench.py): device time, in ms, of the steps of the last tm_context_multiple_steps[_group] call -- HIP events
 * on the stream the context was stepped on around the first .. last step (the final frame's device-to-host copy is outside) */
int tm_context_last_multiple_steps_ms(tm_context_t ctxt, double *ms);
/* ---- local MD                                   wrap_kernels.cpp:399-631; context.cu:90-213; local_md_potentials.cu ----
 * Context.setup_local_md(temperature, freeze_reference): idempotent for equal arguments, "local md configured with
 * different parameters, ..." otherwise. */
int tm_context_setup_local_md(tm_context_t ctxt, double temperature, int freeze_reference);
/* Context.multiple_steps_local(n_steps, local_idxs, store_x_interval=0, radius=1.2, k=10000.0, seed=2022).
 * Validation and messages of the binding (wrap_kernels.cpp:408-420): "local steps must be at least one",
 * "store_x_interval must be greater than or equal to zero", verify_local_md_parameters, verify_atom_idxs.
 * xs[n_samples,N,3], boxes[n_samples,3,3] with n_samples = n_steps / (store_x_interval ? store_x_interval : n_steps). */
int tm_context_multiple_steps_local(tm_context_t ctxt, int n_steps, const int *local_idxs, int num_local_idxs,
                                    int store_x_interval, double radius, double k, int seed, double *xs, double *boxes);
/* Context.multiple_steps_local_selection(n_steps, reference_idx, selection_idxs, store_x_interval=0, radius=1.2,
 * k=10000.0)                                                                          wrap_kernels.cpp:502-556 */
int tm_context_multiple_steps_local_selection(tm_context_t ctxt, int n_steps, int reference_idx, const int *selection_idxs,
                                              int num_selection_idxs, int store_x_interval, double radius, double k,
                                              double *xs, double *boxes);
/* diagnostic (not in the reference surface): the reference atom the last local-MD setup picked and the [N] index array it
 * handed the integrator (i = atom i moves, N = frozen); reference_idx = -1 before the first local-MD call */
int tm_context_local_md_last_selection(tm_context_t ctxt, int *reference_idx, unsigned int *free_idxs);
int tm_context_get_x_t(tm_context_t ctxt, double *out);
int tm_context_get_v_t(tm_context_t ctxt, double *out);
int tm_context_get_box(tm_context_t ctxt, double *out);
int tm_context_set_x_t(tm_context_t ctxt, const double *in);
int tm_context_set_v_t(tm_context_t ctxt, const double *in);
int tm_context_set_box(tm_context_t ctxt, const double *in);

/* ---- Neighborlist_f32/_f64(N)                                     wrap_kernels.cpp:113-172; neighborlist.cu ------- */
int tm_neighborlist_create(int precision, int N, tm_neighborlist_t *out);
int tm_neighborlist_destroy(tm_neighborlist_t nb);
/* get_nblist(coords, box, cutoff) -> list per 32-row block.  Builds, then reports sizes; fetch with ..._copy_nblist:
 * offsets int32[num_row_blocks+1], atoms int32[offsets[num_row_blocks]]. */
int tm_neighborlist_get_nblist(tm_neighborlist_t nb, int N, const double *coords, const double *box, double cutoff,
                               int *num_row_blocks, int *total_atoms);
int tm_neighborlist_copy_nblist(tm_neighborlist_t nb, int32_t *offsets, int32_t *atoms);
/* compute_block_bounds(coords, box, block_size) -> (ctrs[B,3], exts[B,3]); block_size must be 32 (wrap_kernels.cpp:125-128) */
int tm_neighborlist_compute_block_bounds(tm_neighborlist_t nb, int N, const double *coords, const double *box, int block_size,
                                         double *ctrs, double *exts);
int tm_neighborlist_set_row_idxs(tm_neighborlist_t nb, const uint32_t *idxs, int count);
int tm_neighborlist_reset_row_idxs(tm_neighborlist_t nb);
int tm_neighborlist_resize(tm_neighborlist_t nb, int size);
int tm_neighborlist_get_tile_ixn_count(tm_neighborlist_t nb, unsigned int *count);
int tm_neighborlist_get_max_ixn_count(tm_neighborlist_t nb, int *count);
int tm_neighborlist_get_num_row_idxs(tm_neighborlist_t nb, int *count);

/* ---- HilbertSort(size).sort(coords, box) -> uint32[N]             wrap_kernels.cpp:174-194; hilbert_sort.cu ------- */
int tm_hilbert_sort_create(int size, tm_hilbert_sort_t *out);
int tm_hilbert_sort_destroy(tm_hilbert_sort_t hs);
int tm_hilbert_sort_sort(tm_hilbert_sort_t hs, int N, const double *coords, const double *box, uint32_t *perm);
/* host-only: the 128^3 bin -> curve-index table the sort uses (hilbert_sort.cu:18-31); out uint32[128*128*128] */
int tm_hilbert_lut(uint32_t *out);

/* ---- kernel timing (HIP events on the launch stream of the tile kernel; used by bench.py's roofline leg) ---- */
int tm_profile_set_enabled(int enabled);
int tm_profile_read(const char *kernel_name, double *total_ms, long long *launches); /* "nonbonded_tiles" */
int tm_profile_reset(void);
/* debugging / A-B aid: potentials a barostat works on follow small box changes without rebuilding their neighbor list
 * (EXPERIMENTS.md, History, item 6).  0 turns that off process-wide (every box change rebuilds, as in the reference);
 * results are bit-identical either way -- the test suite checks exactly that. */
int tm_debug_set_box_scaling_reuse(int enabled);
/* debugging / A-B aid: a MonteCarloBarostat attempt inside a Context evaluates both energies on the nonbonded potential's CURRENT
 * neighbor list and sorted records and commits an accepted proposal into them (four launches + the list launch; csrc/barostat.hip)
 * whenever the potentials' state allows; 0 = always the reference-shaped attempt (barostat.cu:154-246: copy, centroids, rescale, two
 * full evaluations, decision).  Process-wide; *previous (may be NULL) receives the old value.  Energies, decisions and trajectories
 * are bit-identical either way. */
int tm_debug_set_barostat_fast_path(int enabled, int *previous);
/* test hooks of the locking contract (see "Threading" at the top of this header): tm_debug_set_thread_lock_device makes the calling
 * thread take device `device`'s lock whatever HIP's current device is (-1: back to hipGetDevice; a box without a GPU has no device to
 * make current); tm_debug_hold_api_lock enters the ABI like any entry point, holds the lock for `milliseconds` and reports the largest
 * number of threads that were ever inside it at once (a negative duration resets that figure).  Two threads on two devices: 2; on one: 1. */
int tm_debug_set_thread_lock_device(int device);
int tm_debug_hold_api_lock(int milliseconds, int *max_concurrent);
/* diagnostic: device time (HIP events on the call's stream) of the EVALUATIONS of this process's last tm_potential_execute*_f64 /
 * tm_bound_potential_execute*_f64 call -- behind the staging copy, in front of the conversion and the copy back (bench.py --mode potentials) */
int tm_debug_last_host_call_device_ms(double *ms);
/* debugging / A-B aid: the energy memo (tm_nonbonded_all_pairs_get_memo_stats) on / off; process-wide (TM_AMD_NO_ENERGY_MEMO in the
 * environment sets the initial value to 0); *previous (may be NULL) receives the old value.  Energies are bit-identical either way. */
int tm_debug_set_energy_memo(int enabled, int *previous);
/* debugging / A-B aid: the same-frame hint of the batch entry points (tm_nonbonded_all_pairs_get_same_frame_skips) on / off;
 * process-wide; *previous (may be NULL) receives the old value.  Results are bit-identical either way. */
int tm_debug_set_same_frame_hint(int enabled, int *previous);
/* debugging / A-B aid: forces-only and energy-only plans (MD steps, barostat attempts, Summed / Fanout energy calls) run an all-pairs
 * potential and an interaction group whose columns are exactly its atoms as ONE pipeline (see tm_nonbonded_all_pairs_get_merged_stats);
 * 0 = each keeps its own list, launch and hand-over (rounds 1-5).  Process-wide (TM_AMD_NO_MERGE in the environment sets the initial
 * value to 0); *previous (may be NULL) receives the old value.  Forces, energies and trajectories are bit-identical either way. */
int tm_debug_set_merge_producers(int enabled, int *previous);
/* debugging / A-B aid: nonbonded potentials over at most `max_atoms` atoms keep a STATIC, complete interaction list (every column
 * block listed for every row block: nothing can invalidate it, no list kernel runs on MD steps; EXPERIMENTS.md, History, item 11).
 * Process-wide; applies to potentials at their next call; 0 turns it off; *previous (may be NULL) receives the old value.
 * Results are bit-identical either way. */
int tm_debug_set_static_list_max_k(int max_atoms, int *previous);
/* debugging / A-B aid, VARIANT LIBRARY ONLY (libtimemachine_amd_rowblock.so, built with -DTM_ROWBLOCK by csrc/build.py; the product
 * library answers anything but INT_MAX with an error): forces-only nonbonded launches over at least `min_atoms` atoms run the
 * row-block kernel (one workgroup per row block and column range, lane-owned columns regrouped by hit count:
 * csrc/kernels_nonbonded_rowblock.hip.hpp) -- a second, independent implementation of k_nonbonded_unified
 * (cpp/src/kernels/k_nonbonded.cuh:109-327) that the parity tests compare with the product kernel bit for bit; smaller launches run the
 * wave-per-item kernel.  Process-wide; applies from the next call; 0 = always, INT_MAX = never; *previous (may be NULL) receives
 * the old value.  tm_debug_rowblock_available: 1 iff the loaded library carries the kernel. */
int tm_debug_rowblock_available(int *available);
int tm_debug_set_rowblock_min_k(int min_atoms, int *previous);
/* host only: the electrostatic force-factor table the f64 nonbonded kernels use for `beta` (csrc/nb_es_table.hip.hpp):
 * 256 intervals (32 per binade of d^2 from 2^-7 to 2) x 6 monomial coefficients in the in-interval position t in [0, 1).
 * out: double[1536].  The analytic function it replaces: k_nonbonded_common.cuh:16-94 (real_es_factor / d). */
int tm_es_force_table(double beta, double *out);
/* the same layout for the energy factor G(d^2) = erfc(beta d) S(d) / d that calls asking for energies or du/dp read
 * (u_es = charge_scale q_i q_j G; k_nonbonded_common.cuh:184-212).  out: double[1536]. */
int tm_es_energy_table(double beta, double *out);

/* ---- HREX: a batch of neighbour-swap Metropolis moves on the state -> replica permutation (host only; no device work)
 * replaces the jitted loop timemachine/md/hrex.py:50-130 (_run_neighbor_swaps): for attempt t, pair k = pair_idxs[t] =
 * (s_a, s_b); accept iff uniform_samples[t] < exp(min(0, log_q[r_a, s_b] + log_q[r_b, s_a] - log_q[r_a, s_a] - log_q[r_b, s_b]))
 * (false for NaN).  log_q_kl is [n_replicas, n_states] row-major; proposed / accepted are per pair. */
int tm_hrex_run_neighbor_swaps(int n_replicas, int n_states, const int64_t *replica_idx_by_state, int n_pairs, const int64_t *neighbor_pairs,
                               const double *log_q_kl, int n_attempts, const int64_t *pair_idxs, const double *uniform_samples,
                               int64_t *out_replica_idx_by_state, uint32_t *proposed, uint32_t *accepted);

/* debug: run the DEVICE fixed-point conversions (the functions the kernels inline) on caller-supplied values.
 *   kind 0: FLOAT_TO_FIXED (2^36; forces of bonded terms, du/dq, du/dw)      k_fixed_point.cuh:56-71
 *   kind 1 / 2: FLOAT_TO_FIXED_DU_DP with 2^37 (du/dsig) / 2^38 (du/deps)    fixed_point.hpp:8-11
 *   kind 3: the nonbonded force form FIX(prefactor * delta); `in` holds n (prefactor, delta) pairs   k_nonbonded.cuh:244-252
 * values are cast to the precision's Real first (f32: to float).  tm_debug_float_to_fixed_energy: FLOAT_TO_FIXED_ENERGY
 * (k_fixed_point.cuh:88-98: non-finite / beyond the int64 range -> LLONG_MAX). */
int tm_debug_float_to_fixed(int precision, int kind, const double *in, int n, uint64_t *out);
int tm_debug_float_to_fixed_energy(int precision, const double *in, int n, tm_int128 *out);
/* debug builds (-DTM_GUARD: guard zones around every device buffer): number of violated zones; -1 in product builds */
int tm_debug_check_guards(int *violations);

#ifdef __cplusplus
}
#endif
#endif /* TIMEMACHINE_AMD_H */
