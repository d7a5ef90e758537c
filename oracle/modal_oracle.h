/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's modal-audio hot path (khiner/MeshEditor, src/audio/mesh2modes.cpp,
 * CholeskyShiftInvert.cpp, ModalAudio.cpp, ContactModel.cpp).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product (libmodalhip.so) never links or calls it.
 *
 * Parity pins (tests/test_oracle_*.py): closed-form bar frequencies of the reference's ModalSolverTest,
 * the modal models embedded in the reference's sample glTFs (tests/golden/gltf_modal_models.json),
 * the ContactModelTest known answers, and SciPy ARPACK shift-invert on the same matrices.
 * Third-party arithmetic the reference calls but does not vendor (Spectra, Eigen, Accelerate Sparse, glm)
 * is restated from the published algorithms -- see the .cpp headers.
 */
#ifndef MODAL_ORACLE_H
#define MODAL_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    double density, young_modulus, poisson_ratio, alpha, beta;
} mo_material; /* AcousticMaterialProperties, src/audio/AcousticMaterialProperties.h:6-16 */

typedef struct {
    float min_mode_freq, max_mode_freq; /* mesh2modes.h:18-19 */
    uint32_t num_modes, num_fem_modes; /* :20-21 */
    double tolerance, warm_tolerance; /* :22-23 */
    uint32_t max_restarts; /* :24 */
    int32_t has_fundamental; /* :25 std::optional<float> */
    float fundamental_freq;
} mo_solver_config;

typedef struct {
    double mass_props, quad_mesh, assemble, sample_excite, factorize, iterate, op_solve, extract;
    uint32_t dofs, stiffness_nonzeros, op_applications, restarts;
} mo_profile; /* modal::SolveProfile, mesh2modes.h:29-34 */

void mo_default_config(mo_solver_config *cfg);
/* OpenMP team size for the sparse factorisation and solves (the arithmetic does not depend on it). */
void mo_set_threads(int n);
int mo_max_threads(void);

/* ---- whole path: modal::mesh2modes (mesh2modes.cpp:605-658) ---- */
typedef struct mo_result mo_result;
mo_result *mo_mesh2modes(uint32_t n_points, const double *points_xyz, uint32_t n_tets, const uint32_t *tets,
                         const mo_material *material, uint32_t n_excite, const float *excite_xyz,
                         const float baked_scale[3], const mo_solver_config *config,
                         const float *seed_basis, uint32_t seed_rows, uint32_t seed_cols, int keep_basis,
                         const volatile int *cancel_flag);
void mo_result_free(mo_result *);
uint32_t mo_result_num_modes(const mo_result *);
uint32_t mo_result_num_positions(const mo_result *);
uint32_t mo_result_num_eigenpairs(const mo_result *);
uint32_t mo_result_num_excitations(const mo_result *);
uint32_t mo_result_num_summary_points(const mo_result *r);
/* shapes: [position][mode][xyz]; positions: [position][xyz] */
void mo_result_modes(const mo_result *, float *freqs, float *t60s, float *shapes, float *positions, float *original_fundamental);
/* ModalEigenSummary: eigenvalues ascending, shapes [position][eigenpair][xyz] */
void mo_result_summary(const mo_result *, double *eigenvalues, float *shapes);
/* quat order: w, x, y, z */
void mo_result_mass_props(const mo_result *, double *mass, float com[3], float inertia_diag[3], float quat_wxyz[4]);
void mo_result_profile(const mo_result *, mo_profile *);
void mo_result_sample_point_of_excitation(const mo_result *, uint32_t *out);
uint32_t mo_result_basis_rows(const mo_result *);
uint32_t mo_result_basis_cols(const mo_result *);
void mo_result_basis(const mo_result *, float *col_major);

/* ---- stages, for kernel-level parity ---- */
typedef struct mo_system mo_system;
/* FilterDegenerate + BuildQuadMesh + AssembleQuadratic (mesh2modes.cpp:42-60,246-264,273-327). */
mo_system *mo_assemble(uint32_t n_points, const double *points_xyz, uint32_t n_tets, const uint32_t *tets, const mo_material *material);
void mo_system_free(mo_system *);
uint32_t mo_system_dofs(const mo_system *);
uint32_t mo_system_node_count(const mo_system *);
uint32_t mo_system_kept_tets(const mo_system *);
void mo_system_kept_tet_indices(const mo_system *, uint32_t *out); /* indices into the input tet list */
void mo_system_element_nodes(const mo_system *, uint32_t *out); /* kept_tets x 10 */
uint64_t mo_system_nnz(const mo_system *, int which); /* 0 = K, 1 = M; lower triangle */
void mo_system_csc(const mo_system *, int which, int64_t *colptr, int32_t *rows, double *vals);
/* QuadBasis tables (mesh2modes.cpp:209-237): mass[10][10], grad[10][4][10][4] */
void mo_quad_basis(double *mass100, double *grad1600);
/* Shift-invert Lanczos on (K - sigma M)^-1 M, the restatement of the cold branch (mesh2modes.cpp:470,485-491).
 * evecs may be NULL.  Returns 0 on success, 1 not converged, 2 factorisation failed. */
int mo_system_eigs(const mo_system *, uint32_t nev, uint32_t ncv, double sigma, double tol, uint32_t max_restarts,
                   double *evals, double *evecs_col_major, mo_profile *profile);
/* y = K x or M x (symmetric product from the lower triangle). */
void mo_system_matvec(const mo_system *, int which, const double *x, double *y);

/* modal::PostprocessModes (mesh2modes.cpp:515-588).  shapes_in: [position][eigenpair][xyz].
 * Outputs sized for n_eigs modes; returns the number of modes kept (0 = empty result). */
uint32_t mo_postprocess_modes(uint32_t n_eigs, const double *eigenvalues, uint32_t n_pos, const float *shapes_in,
                              float shape_scale, const mo_material *material, const mo_solver_config *config,
                              float *freqs, float *t60s, float *shapes_out, float *original_fundamental);
/* modal::RescaleModes (mesh2modes.cpp:590-603).  Returns modes kept, or UINT32_MAX when the edit is not scalable. */
uint32_t mo_rescale_modes(uint32_t n_eigs, const double *eigenvalues, uint32_t n_pos, const float *summary_shapes,
                          const mo_material *solved, const mo_material *edited, const mo_solver_config *config,
                          float *freqs, float *t60s, float *shapes_out, float *original_fundamental);
/* ComputeMassProperties (mesh2modes.cpp:73-126). */
void mo_mass_properties(uint32_t n_points, const double *points_xyz, uint32_t n_tets, const uint32_t *tets, double density,
                        const float scale[3], double length_to_si, double *mass, float com[3], float inertia_diag[3], float quat_wxyz[4]);

#ifdef __cplusplus
}
#endif
#endif
