// modal::mesh2modes over libmodalhip (reference orchestration: src/audio/mesh2modes.cpp:605-658 and :441-512).
#include "modal/solver.hpp"

#include "modalhip.h"

#include <chrono>
#include <cmath>
#include <stdexcept>
#include <unordered_map>

namespace {
thread_local int t_device = 0;
struct ThreadContext {
    mh_context *ctx{nullptr};
    int device{-1};
    ~ThreadContext() { mh_context_destroy(ctx); }
    mh_context *get() {
        if (!ctx || device != t_device) {
            mh_context_destroy(ctx);
            ctx = nullptr;
            if (mh_context_create(t_device, &ctx) != MH_OK) throw std::runtime_error("modalhip: no MI355X context (there is no CPU fallback)");
            device = t_device;
        }
        return ctx;
    }
};
thread_local ThreadContext t_context;

mh_material ToC(const AcousticMaterialProperties &m) { return {m.Density, m.YoungModulus, m.PoissonRatio, m.Alpha, m.Beta}; }
mh_solver_config ToC(const modal::SolverConfig &c) {
    return {c.MinModeFreq, c.MaxModeFreq, c.NumModes, c.NumFemModes, c.Tolerance, c.WarmTolerance, c.MaxRestarts, c.FundamentalFreq ? 1 : 0,
            c.FundamentalFreq.value_or(0.f)};
}
std::vector<float> Flatten(const std::vector<std::vector<vec3>> &shapes, size_t n_eigs) {
    std::vector<float> flat(shapes.size() * n_eigs * 3);
    for (size_t p = 0; p < shapes.size(); ++p)
        for (size_t k = 0; k < n_eigs; ++k) {
            flat[(p * n_eigs + k) * 3] = shapes[p][k].x;
            flat[(p * n_eigs + k) * 3 + 1] = shapes[p][k].y;
            flat[(p * n_eigs + k) * 3 + 2] = shapes[p][k].z;
        }
    return flat;
}
std::vector<std::vector<vec3>> Unflatten(const float *flat, size_t n_pos, size_t n_modes) {
    std::vector<std::vector<vec3>> shapes(n_pos, std::vector<vec3>(n_modes));
    for (size_t p = 0; p < n_pos; ++p)
        for (size_t k = 0; k < n_modes; ++k) shapes[p][k] = {flat[(p * n_modes + k) * 3], flat[(p * n_modes + k) * 3 + 1], flat[(p * n_modes + k) * 3 + 2]};
    return shapes;
}
double Since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }

struct Handles { // destroy in reverse order of creation
    mh_system *sys{nullptr};
    mh_mesh *mesh{nullptr};
    ~Handles() {
        mh_system_destroy(sys);
        mh_mesh_destroy(mesh);
    }
};
} // namespace

namespace modal {
void SetDevice(int device) { t_device = device; }

ModalModes PostprocessModes(std::span<const double> eigenvalues, const std::vector<std::vector<vec3>> &shapes, float shape_scale,
                            const AcousticMaterialProperties &material, const SolverConfig &config, std::vector<vec3> positions) {
    const auto n = uint32_t(eigenvalues.size());
    const auto flat = Flatten(shapes, n);
    std::vector<float> freqs(n), t60s(n), out(flat.size());
    uint32_t kept = 0;
    float original = 0;
    const auto mat = ToC(material);
    const auto cfg = ToC(config);
    mh_postprocess_modes(n, eigenvalues.data(), uint32_t(shapes.size()), flat.data(), shape_scale, &mat, &cfg, &kept, freqs.data(), t60s.data(), out.data(), &original);
    if (kept == 0 && original == 0) return {};
    freqs.resize(kept);
    t60s.resize(kept);
    ModalModes modes;
    modes.Freqs = std::move(freqs);
    modes.T60s = std::move(t60s);
    modes.Shapes = Unflatten(out.data(), shapes.size(), kept);
    modes.Positions = std::move(positions);
    modes.OriginalFundamentalFreq = original;
    return modes;
}

std::optional<ModalModes> RescaleModes(const ModalEigenSummary &summary, const ModalModes &current, const AcousticMaterialProperties &material, SolverConfig config) {
    if (summary.Eigenvalues.empty() || material.PoissonRatio != summary.SolvedMaterial.PoissonRatio) return {};
    const double rho_ratio = material.Density / summary.SolvedMaterial.Density;
    const double eigenvalue_scale = (material.YoungModulus / summary.SolvedMaterial.YoungModulus) / rho_ratio;
    auto eigenvalues = summary.Eigenvalues;
    for (auto &v : eigenvalues) v *= eigenvalue_scale;
    auto modes = PostprocessModes(eigenvalues, summary.Shapes, float(1 / std::sqrt(rho_ratio)), material, config, current.Positions);
    modes.Vertices = current.Vertices;
    modes.Indices = current.Indices;
    modes.BakedScale = current.BakedScale;
    return modes;
}

ModalResult mesh2modes(const TetMesh &tets, const AcousticMaterialProperties &material, const std::vector<vec3> &excite_positions, vec3 baked_scale,
                       SolverConfig config, SolveReuse reuse, JobMonitor *monitor) {
    mh_context *ctx = t_context.get();
    SolveProfile profile;
    Handles h;
    static_assert(sizeof(dvec3) == 3 * sizeof(double) && sizeof(std::array<uint32_t, 4>) == 4 * sizeof(uint32_t));
    const auto *pts = reinterpret_cast<const double *>(tets.Points.data());
    const auto *idx = reinterpret_cast<const uint32_t *>(tets.Tets.data());
    if (mh_mesh_create(ctx, uint32_t(tets.Points.size()), pts, uint32_t(tets.Tets.size()), idx, &h.mesh) != MH_OK) return {};

    const double length_to_si = (double(baked_scale.x) + baked_scale.y + baked_scale.z) / 3.0;
    auto t0 = std::chrono::steady_clock::now();
    mh_mass_props mp{};
    const float scale[3] = {baked_scale.x, baked_scale.y, baked_scale.z};
    mh_compute_mass_properties(uint32_t(tets.Points.size()), pts, uint32_t(tets.Tets.size()), idx, material.Density, scale, length_to_si, &mp);
    profile.MassProps = Since(t0);
    MassProperties mass_props{mp.mass, {mp.center_of_mass[0], mp.center_of_mass[1], mp.center_of_mass[2]},
                              {mp.inertia_diagonal[0], mp.inertia_diagonal[1], mp.inertia_diagonal[2]},
                              {mp.inertia_orientation_wxyz[0], mp.inertia_orientation_wxyz[1], mp.inertia_orientation_wxyz[2], mp.inertia_orientation_wxyz[3]}};

    if (monitor) monitor->Progress.store(0.1f, std::memory_order_relaxed);
    const auto mat = ToC(material);
    t0 = std::chrono::steady_clock::now();
    if (mh_assemble(ctx, h.mesh, &mat, &h.sys) != MH_OK) return {};
    profile.Assemble = Since(t0); // BuildQuadMesh + AssembleQuadratic, fused on the device
    uint32_t dofs = 0, node_count = 0, kept = 0;
    uint64_t blocks = 0;
    mh_system_dims(h.sys, &dofs, &node_count, &kept, &blocks);
    profile.Dofs = dofs;
    if (monitor && monitor->Cancelled()) return {};

    // Nearest tet point per excitation position; positions reaching the same point share one sample point.
    t0 = std::chrono::steady_clock::now();
    std::vector<uint32_t> nearest(excite_positions.size());
    static_assert(sizeof(vec3) == 3 * sizeof(float));
    if (mh_nearest_points(ctx, h.mesh, uint32_t(excite_positions.size()), reinterpret_cast<const float *>(excite_positions.data()), nearest.data()) != MH_OK) return {};
    std::vector<uint32_t> points, remap(excite_positions.size());
    std::vector<vec3> local;
    std::unordered_map<uint32_t, uint32_t> sample_point_at;
    const dvec3 inv_scale{1.0 / baked_scale.x, 1.0 / baked_scale.y, 1.0 / baked_scale.z};
    for (size_t i = 0; i < nearest.size(); ++i) {
        const auto [entry, first] = sample_point_at.emplace(nearest[i], uint32_t(points.size()));
        if (first) {
            points.push_back(nearest[i]);
            local.emplace_back(tets.Points[nearest[i]] * inv_scale);
        }
        remap[i] = entry->second;
    }
    profile.SampleExcite = Since(t0);

    const uint32_t n = dofs;
    const uint32_t nev = std::min(config.NumFemModes, n - 1);
    const double sigma = -std::pow(2 * M_PI * config.MinModeFreq, 2);
    const bool warm = reuse.SeedBasis && reuse.SeedBasis->rows() == std::ptrdiff_t(n) && reuse.SeedBasis->cols() >= std::ptrdiff_t(nev);
    // Spectra's tolerance bounds the Ritz-value error; a relative residual r gives an eigenvalue error ~r^2.
    const double tol = warm ? std::clamp(std::sqrt(config.WarmTolerance) * 1e-2, 1e-9, 1e-2) : std::clamp(0.1 * std::sqrt(config.Tolerance), 1e-9, 1e-4);
    if (monitor && monitor->Cancelled()) return {};
    std::vector<double> eigenvalues(nev);
    mh_profile dev{};
    // JobMonitor's atomics are lock-free single words: the device loop polls / writes their storage directly.
    static_assert(sizeof(std::atomic<bool>) == 1 && sizeof(std::atomic<float>) == sizeof(float));
    const volatile unsigned char *cancel = monitor ? reinterpret_cast<const volatile unsigned char *>(&monitor->CancelRequested) : nullptr;
    volatile float *progress = monitor ? reinterpret_cast<volatile float *>(&monitor->Progress) : nullptr;
    const int rc = mh_eigs(h.sys, nev, sigma, tol, std::max(config.MaxRestarts, 1u) * 3, warm ? reuse.SeedBasis->data() : nullptr, warm ? n : 0,
                           warm ? uint32_t(reuse.SeedBasis->cols()) : 0, cancel, progress, eigenvalues.data(), &dev);
    if (rc == MH_EFACTOR) throw std::runtime_error("Modal shift-invert factorization failed.");
    if (rc != MH_OK) return {};
    profile.Factorize = dev.factorize;
    profile.Iterate = dev.iterate;
    profile.OpSolve = dev.op_solve;
    profile.OpApplications = dev.op_applications;
    profile.Restarts = dev.restarts;
    profile.StiffnessNonZeros = dev.stiffness_nonzeros;

    t0 = std::chrono::steady_clock::now();
    std::vector<float> flat(points.size() * nev * 3);
    if (mh_system_gather_shapes(h.sys, uint32_t(points.size()), points.data(), nev, flat.data()) != MH_OK) return {};
    ModalResult result;
    result.Summary.Eigenvalues = eigenvalues;
    result.Summary.Shapes = Unflatten(flat.data(), points.size(), nev);
    result.Summary.SolvedMaterial = material;
    if (reuse.KeepBasis) {
        result.Basis.resize(n, nev);
        mh_system_basis(h.sys, nev, result.Basis.data());
    }
    profile.Extract = Since(t0);
    result.Modes = PostprocessModes(result.Summary.Eigenvalues, result.Summary.Shapes, 1.f, material, config, std::move(local));
    result.MassProps = mass_props;
    result.Profile = profile;
    result.SamplePointOfExcitation = std::move(remap);
    return result;
}
} // namespace modal
