/* modalhip -- C ABI of the MI355X-native modal-audio hot path (libmodalhip.so).
 *
 * The reference (khiner/MeshEditor) has no FFI layer for this path: it is reached through C++ free functions and
 * plain structs (src/audio/mesh2modes.h:77-88, src/audio/ModalAudio.h:294-315).  This header is the thin C ABI the
 * C++ mirror of those functions (mesheditor_amd/cpp/) and the Python binding (mesheditor_amd/api.py) sit on: opaque
 * handles, plain pointers and sizes, int status codes, no exceptions and no torch types across the boundary.
 * Every entry point names the reference interface it replaces.  All functions return MH_OK (0) or an MH_E* code;
 * mh_last_error() gives the message of the calling context's last failure.
 *
 * Threading: a context owns one HIP stream, its library handles and its device memory pool.  Calls on different
 * contexts may run concurrently from different host threads (the reference solves several entities concurrently,
 * src/audio/AudioSystem.cpp:812,865); calls on one context must be serialised by the caller.
 */
#ifndef MODALHIP_H
#define MODALHIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    MH_OK = 0,
    MH_EINVAL = 1, /* bad argument */
    MH_EHIP = 2, /* HIP / rocBLAS / rocSOLVER failure (no GPU, out of memory, ...) */
    MH_ECANCELLED = 3, /* JobMonitor::Cancelled() was observed (src/Job.h:13-19) -> empty result */
    MH_ENOTCONVERGED = 4, /* eigensolver hit its iteration limit -> empty result (mesh2modes.cpp:490) */
    MH_EFACTOR = 5, /* shifted operator not positive definite -> std::runtime_error (CholeskyShiftInvert.cpp:44) */
    MH_EEMPTY = 6 /* no tets left / no modes in band */
};

typedef struct mh_context mh_context;
typedef struct mh_mesh mh_mesh;
typedef struct mh_system mh_system;
typedef struct mh_bank mh_bank;

/* AcousticMaterialProperties, src/audio/AcousticMaterialProperties.h:6-16 */
typedef struct {
    double density, young_modulus, poisson_ratio, alpha, beta;
} mh_material;

/* modal::SolverConfig, src/audio/mesh2modes.h:17-26 */
typedef struct {
    float min_mode_freq, max_mode_freq;
    uint32_t num_modes, num_fem_modes;
    double tolerance, warm_tolerance;
    uint32_t max_restarts;
    int32_t has_fundamental;
    float fundamental_freq;
} mh_solver_config;

/* modal::SolveProfile, src/audio/mesh2modes.h:29-34.  Seconds per stage (HIP-event timed on the context's stream).
 * factorize = preconditioner set-up (the reference's Cholesky factorisation slot); op_solve = preconditioner
 * applications (the reference's triangular solves); op_applications = preconditioned block columns;
 * restarts = LOBPCG iterations. */
typedef struct {
    double mass_props, quad_mesh, assemble, sample_excite, factorize, iterate, op_solve, extract;
    uint32_t dofs, stiffness_nonzeros, op_applications, restarts;
    /* health of the device solve (no reference counterpart; zero / rounding level on a healthy run): Rayleigh-Ritz steps that were redone by a
     * fall-back because the multi-workgroup tridiagonalisation timed out, and the worst sampled residual of the steps' self-check against the
     * saved Rayleigh-Ritz matrix, max_i |A z - theta z|_i over max_i (sum_c |a_ic z_c| + |theta z_i|) -- 1e-13 ... 3e-12 measured; a solve
     * whose check exceeds 1e-6 fails with MH_EHIP */
    uint32_t sytrd_redos;
    /* wanted pairs that the LAST RESORT handed on with a residual still above the tolerance (below ten times it, unchanged for thirty iterations: the
     * rounding floor of a mesh at the edge of double precision -- raw Delaunay fills with cells at 1e-8); 0 on every solve that converged */
    uint32_t pairs_at_floor;
    double rr_selfcheck;
} mh_profile;

/* MassProperties, src/audio/ContactModel.h:16-23 (quaternion as w, x, y, z) */
typedef struct {
    double mass;
    float center_of_mass[3], inertia_diagonal[3], inertia_orientation_wxyz[4];
} mh_mass_props;

/* sizeof(mh_profile), sizeof(mh_solver_config), sizeof(mh_material), sizeof(mh_mass_props) as this library was built: a binding checks its own
 * struct images against them (needs no GPU) */
void mh_abi_struct_sizes(uint32_t out[4]);
int mh_context_create(int device, mh_context **out);
void mh_context_destroy(mh_context *);
const char *mh_last_error(const mh_context *);
/* Blocks until everything queued on the context's stream has finished. */
int mh_context_synchronize(mh_context *);
/* The context's hipStream_t, for callers that time with HIP events. */
void *mh_context_stream(mh_context *);
void mh_default_config(mh_solver_config *);
/* Kernel timing of the path itself (no reference counterpart; the one instrumentation hook that has to live in the product,
 * because it brackets the product's own launches in place -- bench.py's roofline objects read it; timing loops and experiments
 * are in libmodalhip_lab.so): when enabled, every launch of the path's named kernels is bracketed by
 * HIP events on the context's stream.  Kernel classes and their algorithmic work unit:
 *   MH_KERNEL_SPMM     the operator products y = (K - sigma M) x over n x w panels (both levels, all precisions);
 *                      bytes: (9 values + 1 index) per node block + 4 B per row pointer + every panel pass
 *   MH_KERNEL_ASSEMBLY the K/M assembly kernel of the quadratic level; bytes: per tet 16 B corners + 96 B coordinates
 *                      + 40 B node ids read, 80 B per node block written
 *   MH_KERNEL_BANK     the resonator kernel (RenderObjectFast); flops: 11 per rendered mode-sample
 *   MH_KERNEL_COMBINE  the basis updates out = [X | W | P] C of the eigensolver (k_combine, fp64 MFMA; the vendor dgemm for blocks of
 *                      >= 400 basis columns); flops: 2 n (basis columns) (output columns) per launch.  MH_KERNEL_COMBINE_BYTES carries
 *                      the same launches' algorithmic bytes (8 n (basis + output columns)) as its work, no time of its own.
 *                      MH_KERNEL_COMBINE_FULL: the subset of those launches with >= 200 basis and >= 128 output columns (an iteration's
 *                      full-size update of X and P together: 240 -> 160 columns on the 65-pair solve), same work unit.
 * Stats are the totals since the last enable: launches, summed device milliseconds, summed work. */
enum { MH_KERNEL_SPMM = 0, MH_KERNEL_ASSEMBLY = 1, MH_KERNEL_BANK = 2, MH_KERNEL_COMBINE = 3, MH_KERNEL_COMBINE_BYTES = 4, MH_KERNEL_COMBINE_FULL = 5, MH_KERNEL_CLASSES = 6 };
int mh_context_time_kernels(mh_context *, int enable);
int mh_context_kernel_stats(mh_context *, uint64_t *launches, double *total_ms, double *total_bytes); /* MH_KERNEL_SPMM */
int mh_context_kernel_class_stats(mh_context *, int kernel_class, uint64_t *launches, double *total_ms, double *total_work);

/* ---- analysis half: modal::mesh2modes (src/audio/mesh2modes.cpp:605-658), stage by stage ---- */

/* TetMesh (src/mesh/TetMesh.h:10-13) -> HBM.  points: n_points x 3 doubles (AoS dvec3); tets: n_tets x 4 uint32. */
int mh_mesh_create(mh_context *, uint32_t n_points, const double *points_xyz, uint32_t n_tets, const uint32_t *tets, mh_mesh **out);
void mh_mesh_destroy(mh_mesh *);

/* FilterDegenerate + BuildQuadMesh + ComputeElementBases + AssembleQuadratic (mesh2modes.cpp:42-60,137-165,246-264,
 * 273-327) on the device: K as 3x3 node blocks (BSR), M as one scalar per node block (M = M_node (x) I3). */
int mh_assemble(mh_context *, const mh_mesh *, const mh_material *, mh_system **out);
void mh_system_destroy(mh_system *);
int mh_system_dims(const mh_system *, uint32_t *dofs, uint32_t *node_count, uint32_t *kept_tets, uint64_t *node_blocks);
/* QuadMesh::ElementNodes in the reference's numbering (corners, then midside ids in first-encounter order). */
int mh_system_element_nodes(const mh_system *, uint32_t *out_kept_tets_x10);
/* Every stored node block as (row node, col node, 9 row-major K entries, M scalar), reference numbering, full
 * (both triangles).  Arrays sized by mh_system_dims' node_blocks. */
int mh_system_export_blocks(const mh_system *, uint32_t *row_node, uint32_t *col_node, double *k_blocks, double *m_blocks);
/* y = K x (which = 0), M x (which = 1) or (K - sigma M) x at the reference's shift (which = 2) for `width` vectors, x and y column-major n x width in the reference's
 * DOF order (3*node + component).  The SpMM kernel of the eigensolver, exposed for parity and roofline measurement.
 * which = 3 / 4: the same shifted product as the preconditioner's smoothers form it -- single-precision values and panel
 * (3), double-precision values over a single-precision panel (4, width % 4 == 0) -- so those kernels can be checked too. */
int mh_system_matvec(mh_system *, int which, const double *x, double *y, uint32_t width);
/* The reference's shift-invert operator as an OPERATION (src/audio/CholeskyShiftInvert.h:11-30: set_shift + perform_op / solve_panel):
 * x = (K - sigma M)^-1 b for `width` right-hand sides, b and x column-major n x width in the reference's DOF order.  There is no
 * factorisation here: the panel is solved by preconditioned conjugate gradients (the eigensolver's three-level cycle as the
 * preconditioner) to a relative residual `rel_tol` per column (0 = 1e-11), at most `max_iters` steps (0 = 200).  sigma must be negative
 * (MH_EFACTOR otherwise: the reference's "Modal shift-invert factorization failed.", CholeskyShiftInvert.cpp:44).  *iterations and
 * *worst_relative_residual (nullable) report the solve.  The hierarchy of the shift is built on first use and kept with the system, as
 * set_shift keeps its factor.  mh_eigs is the fast path to the eigenpairs; this exists for callers that drive their own Lanczos. */
int mh_system_shift_invert(mh_system *, double sigma, const double *b, double *x, uint32_t width, double rel_tol, uint32_t max_iters, uint32_t *iterations,
                           double *worst_relative_residual);

/* The nearest tet point to each excitation position, first minimum wins (mesh2modes.cpp:626-636). */
int mh_nearest_points(mh_context *, const mh_mesh *, uint32_t n, const float *positions_xyz, uint32_t *nearest);

/* ComputeModes' eigensolve (mesh2modes.cpp:441-497): the `nev` lowest eigenpairs of K x = lambda M x, ascending,
 * M-orthonormal.  Replaces Spectra SymGEigsShiftSolver + CholeskyShiftInvert (cold) and SubspaceIterate (warm, when
 * seed_basis has n rows and >= nev columns; column-major float as ModalResult::Basis) with a block LOBPCG on the
 * shifted pencil (K - sigma M, M) preconditioned by a three-level cycle.  residual_tol is the relative residual
 * ||K x - lambda M x|| / (|lambda - sigma| ||M x||) every returned pair meets.  cancel (nullable; one byte, the storage of JobMonitor's
 * std::atomic<bool>) is polled between iterations; progress (nullable) receives 0.3 + 0.65 * converged / nev as the reference's warm path does. */
int mh_eigs(mh_system *, uint32_t nev, double sigma, double residual_tol, uint32_t max_iters,
            const float *seed_basis, uint32_t seed_rows, uint32_t seed_cols,
            const volatile unsigned char *cancel, volatile float *progress, double *eigenvalues, mh_profile *profile);
/* shapes[node][column][xyz] = eigenvector rows 3*node + {0,1,2} (mesh2modes.cpp:498-504), as float. */
int mh_system_gather_shapes(const mh_system *, uint32_t n_nodes, const uint32_t *nodes, uint32_t n_cols, float *shapes);
/* ModalResult::Basis: n x n_cols column-major float, rows in the reference's DOF order (mesh2modes.cpp:509). */
int mh_system_basis(const mh_system *, uint32_t n_cols, float *basis);
/* The same in double, for parity tests. */
int mh_system_eigenvectors(const mh_system *, uint32_t n_cols, double *vectors);
/* What `residual_tol` of mh_eigs meant in the last solve (no reference counterpart: Spectra's tolerance is on Ritz values).
 * On a mesh WITHOUT near-degenerate elements a pair is accepted at ||K x - lambda M x||_2 < residual_tol |lambda - sigma| ||M x||_2.
 * On a mesh WITH them (elements of shape measure < 0.02: the sliver patches of the preconditioner exist) the same test is made
 * in the Jacobi-scaled norm ||.||_{D^-1}, D = diag(K - sigma M): the rounding noise eps ||A|| |x| of the slivers' rows alone
 * exceeds the tolerance in the 2-norm, while in the scaled norm those rows count as little as their noise means (identical on
 * a uniform mesh).  *worst_plain_residual then holds the worst 2-norm relative residual among the returned elastic pairs,
 * measured once after convergence (-1 when the 2-norm was the criterion); eigenvalue accuracy is checked against the oracle
 * on every committed scan fixture (2e-11 ... 3e-9).  dropped_patches[2]: sliver patches of the P2 / P1 level whose block was
 * not safely positive definite and contribute nothing to the smoother. */
int mh_system_residual_report(const mh_system *, double *worst_plain_residual, uint32_t dropped_patches[2]);

/* Host-side scalar stages of the path (the reference runs them on the calling thread as well). */
/* ComputeMassProperties (mesh2modes.cpp:73-126) */
int mh_compute_mass_properties(uint32_t n_points, const double *points_xyz, uint32_t n_tets, const uint32_t *tets, double density,
                       const float baked_scale[3], double length_to_si, mh_mass_props *out);
/* modal::PostprocessModes (mesh2modes.cpp:515-588).  shapes: [position][eigenpair][xyz].  Returns the mode count in
 * *n_modes (0 = the reference's empty result); outputs hold at most n_eigs modes. */
int mh_postprocess_modes(uint32_t n_eigs, const double *eigenvalues, uint32_t n_pos, const float *shapes, float shape_scale,
                         const mh_material *, const mh_solver_config *, uint32_t *n_modes, float *freqs, float *t60s,
                         float *shapes_out, float *original_fundamental);
/* modal::RescaleModes (mesh2modes.cpp:590-603).  *scalable = 0 when the Poisson ratio differs (std::nullopt). */
int mh_rescale_modes(uint32_t n_eigs, const double *eigenvalues, uint32_t n_pos, const float *summary_shapes,
                     const mh_material *solved, const mh_material *edited, const mh_solver_config *, int *scalable,
                     uint32_t *n_modes, float *freqs, float *t60s, float *shapes_out, float *original_fundamental);

/* ---- synthesis half: the resonator bank (src/audio/ModalAudio.{h,cpp}) ---- */

/* ModalBank::ActiveImpact, src/audio/ModalAudio.h:150-162.  The host mirror owns the event queue, the impact list
 * and the object deal (ModalAudio.cpp:28-82,430-461); the device renders a block from them. */
typedef struct {
    uint32_t object, ex_pos, samples_left, reserved;
    /* Real-valued fields travel as double so that the fp64 bank keeps its recurrences exact between blocks; the fp32
     * bank's values are floats and round-trip through double unchanged. */
    double jx, jy, jz;
    double phase_re, phase_im, rot_re, rot_im;
    double gamma, accel_amp;
    double click_b0, click_a1, click_a2, click_z1, click_z2;
} mh_impact;

/* Device mirror of a published ModalBank (InstallModalBank, ModalAudio.cpp:277-289).  Per-mode columns are n_modes
 * long, shape columns n_shapes long, per-object columns n_objects long; layouts as ModalAudio.h:103-166.
 * use_double selects the fp64 bank (same layout, double state and arithmetic). */
int mh_bank_create(mh_context *, int use_double, uint32_t n_objects, uint32_t n_modes, uint32_t n_shapes,
                   const uint32_t *mode_offset, const uint32_t *mode_count, const uint32_t *shape_offset,
                   const float *shape_x, const float *shape_y, const float *shape_z, mh_bank **out);
void mh_bank_destroy(mh_bank *);
/* TuneModalObject's device effect (ModalAudio.cpp:340-393): overwrite coefficient columns [first, first+count). */
/* Column arrays are float for an fp32 bank and double for an fp64 bank. */
int mh_bank_set_coefficients(mh_bank *, uint32_t first, uint32_t count, const void *coeff_re, const void *coeff_im,
                             const void *radiation_gain, const void *out_phase_im, const void *out_phase_re);
/* SetModalObjectShapes (ModalAudio.cpp:395-410) */
int mh_bank_set_shapes(mh_bank *, uint32_t first, uint32_t count, const float *x, const float *y, const float *z);
/* SilenceObject's state clear (ModalAudio.cpp:53-56) */
int mh_bank_zero_state(mh_bank *, uint32_t first_mode, uint32_t count);
/* One block of RenderModal (ModalAudio.cpp:486-555): force curves + click per impact (:504-538), RenderObjectFast per
 * dealt object (:86-147), mix in renderer order (:553-555).  `out` (frames samples, float or double per the bank) is
 * ADDED to, as the reference does.  The deal arrives flattened: renderer r renders objects
 * deal_objects[deal_offset[r] .. deal_offset[r+1]) in that order, each with mode count render_count[i] (Tuned when the
 * object has impacts, Live otherwise).  impacts is updated in place (phase, samples_left, click state).
 * Per dealt object the device returns its post-block energy, its audible prefix (chunk-granular `live`) and whether
 * it fell silent (no impacts and gain-weighted energy below 1e-12: its state was zeroed, ModalAudio.cpp:141-144), plus
 * its share of the modal-energy diagnostic over its tuned_count modes (ModalAudio.cpp:564-577; nullable). */
int mh_bank_render(mh_bank *, uint32_t frames, float click_gain, uint32_t n_impacts, mh_impact *impacts,
                   uint32_t n_renderers, const uint32_t *deal_offset, const uint32_t *deal_objects, const uint32_t *render_count,
                   const uint32_t *tuned_count, const float *out_gain, const float *listener_gain, void *out, double *object_energy,
                   uint32_t *object_live, uint8_t *object_silenced, double *object_modal_energy);
/* Read back state columns (for parity tests and the modal-energy diagnostic, ModalAudio.cpp:564-577). */
int mh_bank_read_state(const mh_bank *, uint32_t first, uint32_t count, double *state_re, double *state_im);

#ifdef __cplusplus
}
#endif
#endif
