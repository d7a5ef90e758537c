// The solve path through the mirrored C++ API (modal::mesh2modes and friends), checked against the closed forms of
// the reference's bar tests (tests/ModalSolverTest.cpp:228-261) and its calling conventions: cancellation through a
// JobMonitor, warm start through SolveReuse, RescaleModes on a material edit.
#include "harness.hpp"

#include <audio/CholeskyShiftInvert.h>
#include <audio/mesh2modes.h>
#include <modalhip.h> // (the C ABI's product of the shifted operator: the check of the operator below)
#include <mesh/TetMesh.h>

#include <algorithm>
#include <array>
#include <map>
#include <numbers>
#include <vector>

namespace {
// A free prismatic beam along x and what it is made of.
struct Beam {
    double L, W, T; // length (x), width (y), thickness (z), m
    AcousticMaterialProperties Solid;
    double WaveSpeed() const { return std::sqrt(Solid.YoungModulus / Solid.Density); }
};

// nx x ny x nz grid cells, each cut into the six tetrahedra that share the cell diagonal 0-7 (Kuhn), all positively oriented.
TetMesh GridTets(const Beam &beam, int nx, int ny, int nz) {
    TetMesh mesh;
    const int stride_j = nz + 1, stride_i = (ny + 1) * stride_j;
    for (int i = 0; i <= nx; ++i)
        for (int j = 0; j <= ny; ++j)
            for (int k = 0; k <= nz; ++k) mesh.Points.push_back({beam.L * i / nx, beam.W * j / ny, beam.T * k / nz});
    // cell corner c has offsets (c & 1, c >> 1 & 1, c >> 2) in (i, j, k); the six paths 0 -> 7 of the Kuhn triangulation
    static constexpr int Paths[6][4]{{0, 1, 3, 7}, {0, 3, 2, 7}, {0, 2, 6, 7}, {0, 6, 4, 7}, {0, 4, 5, 7}, {0, 5, 1, 7}};
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < ny; ++j)
            for (int k = 0; k < nz; ++k) {
                const uint32_t origin = uint32_t(i * stride_i + j * stride_j + k);
                const auto corner = [&](int c) { return origin + uint32_t((c & 1) * stride_i + ((c >> 1) & 1) * stride_j + (c >> 2)); };
                for (const auto &path : Paths) mesh.Tets.push_back({corner(path[0]), corner(path[1]), corner(path[2]), corner(path[3])});
            }
    return mesh;
}

enum class Motion { Axial, Twist, Flex, FlexY, FlexZ, Mixed };

// Sort a mode by where its shape energy sits: along the axis, in rigid rotation of the cross sections about the axis
// (per x-station: (sum r x u)^2 / sum r^2), or sideways in one plane.  Thresholds as the reference's classifier.
Motion KindOfMotion(const ModalModes &modes, size_t mode, const Beam &beam, int stations) {
    struct Station {
        double Moment{0}, Spread{0};
    };
    std::vector<Station> along(size_t(stations) + 1);
    double energy[3]{0, 0, 0};
    for (size_t p = 0; p < modes.Positions.size(); ++p) {
        const vec3 at = modes.Positions[p], u = modes.Shapes[p][mode];
        for (int axis = 0; axis < 3; ++axis) energy[axis] += double(u[axis]) * double(u[axis]);
        const double off_y = at.y - beam.W / 2, off_z = at.z - beam.T / 2;
        Station &here = along[size_t(std::clamp<long>(std::lround(at.x * stations / beam.L), 0, stations))];
        here.Moment += off_y * u.z - off_z * u.y;
        here.Spread += off_y * off_y + off_z * off_z;
    }
    const double sideways = energy[1] + energy[2], all = energy[0] + sideways;
    if (!(all > 0)) return Motion::Mixed;
    double twisting = 0;
    for (const Station &st : along)
        if (st.Spread > 0) twisting += st.Moment * st.Moment / st.Spread;
    if (energy[0] > 0.85 * all) return Motion::Axial;
    if (twisting > 0.85 * all) return Motion::Twist;
    if (sideways > 0.6 * all && twisting < 0.5 * all) {
        if (energy[1] > 0.8 * sideways) return Motion::FlexY;
        if (energy[2] > 0.8 * sideways) return Motion::FlexZ;
        return Motion::Flex;
    }
    return Motion::Mixed;
}

// Solve the beam with default settings, every mesh point an excitation position, and bucket the frequencies by motion.
std::map<Motion, std::vector<double>> BeamSpectrum(const Beam &beam, int nx, int ny, int nz) {
    const TetMesh mesh = GridTets(beam, nx, ny, nz);
    std::vector<vec3> everywhere(mesh.Points.size());
    std::transform(mesh.Points.begin(), mesh.Points.end(), everywhere.begin(), [](const dvec3 &p) { return vec3{float(p.x), float(p.y), float(p.z)}; });
    const auto solved = modal::mesh2modes(mesh, beam.Solid, everywhere, vec3{1.f});
    EXPECT(!solved.Modes.Freqs.empty());
    EXPECT(solved.Profile.Dofs == 3u * uint32_t((2 * nx + 1) * (2 * ny + 1) * (2 * nz + 1))); // quadratic elements: a node per half cell
    std::map<Motion, std::vector<double>> spectrum;
    for (size_t k = 0; k < solved.Modes.Freqs.size(); ++k) spectrum[KindOfMotion(solved.Modes, k, beam, nx)].push_back(solved.Modes.Freqs[k]);
    return spectrum;
}

std::vector<double> Multiples(double fundamental, int count = 3) {
    std::vector<double> out;
    for (int n = 1; n <= count; ++n) out.push_back(n * fundamental);
    return out;
}
// Free-free Euler-Bernoulli beam: f = (beta L)^2 / (2 pi L^2) * c * r_g with r_g = depth / sqrt(12); `copies` entries per
// root for cross sections whose two bending planes coincide.
std::vector<double> FlexuralTheory(const Beam &beam, double depth, int copies) {
    static constexpr double Roots[]{4.73004074, 7.85320462, 10.9956078};
    const double unit = beam.WaveSpeed() * (depth / std::sqrt(12.0)) / (2 * std::numbers::pi * beam.L * beam.L);
    std::vector<double> out;
    for (const double root : Roots) out.insert(out.end(), size_t(copies), root * root * unit);
    return out;
}
void Compare(const char *label, const std::vector<double> &fem, const std::vector<double> &theory, double tolerance, size_t at_least = 2) {
    const size_t n = std::min(fem.size(), theory.size());
    EXPECT_NOTE(n >= at_least, label);
    for (size_t i = 0; i < n; ++i) {
        const double ratio = fem[i] / theory[i];
        std::printf("%14s %zu: theory %9.2f Hz, FEM %9.2f Hz, ratio %.4f\n", label, i + 1, theory[i], fem[i], ratio);
        EXPECT_NOTE(std::abs(ratio - 1.0) < tolerance, label);
    }
}
} // namespace

// The reference's two known-answer beams (tests/ModalSolverTest.cpp:228-261): sizes, materials, grids and tolerances are its.
CASE(square_bar_modes_match_closed_forms) {
    const Beam beam{0.3, 0.05, 0.05, {.Density = 1000, .YoungModulus = 1e7, .PoissonRatio = 0, .Alpha = 0, .Beta = 0}};
    auto spectrum = BeamSpectrum(beam, 20, 4, 4);
    Compare("longitudinal", spectrum[Motion::Axial], Multiples(beam.WaveSpeed() / (2 * beam.L)), 0.01);
    // torsion of a square section: J / I_p = 0.140577 x 6
    const double twist_speed = std::sqrt(beam.Solid.Mu() / beam.Solid.Density * 0.140577 * 6);
    Compare("torsional", spectrum[Motion::Twist], Multiples(twist_speed / (2 * beam.L)), 0.05);
    std::vector<double> flex;
    for (const Motion m : {Motion::Flex, Motion::FlexY, Motion::FlexZ}) flex.insert(flex.end(), spectrum[m].begin(), spectrum[m].end());
    std::sort(flex.begin(), flex.end());
    if (flex.size() > 2) flex.resize(2); // the first (degenerate) pair
    Compare("bending", flex, FlexuralTheory(beam, beam.T, 2), 0.10);
}

CASE(thin_bar_bending_matches_closed_forms) {
    const Beam beam{0.3, 0.05, 0.01, {.Density = 1000, .YoungModulus = 1e9, .PoissonRatio = 0, .Alpha = 0, .Beta = 0}};
    auto spectrum = BeamSpectrum(beam, 30, 5, 1);
    Compare("longitudinal", spectrum[Motion::Axial], Multiples(beam.WaveSpeed() / (2 * beam.L)), 0.01);
    auto edgewise = FlexuralTheory(beam, beam.W, 1);
    edgewise.resize(1);
    Compare("bending-y", spectrum[Motion::FlexY], edgewise, 0.10, 1);
    Compare("bending-z", spectrum[Motion::FlexZ], FlexuralTheory(beam, beam.T, 1), 0.05);
}

CASE(cancelled_and_degenerate_solves_return_empty_results) {
    const auto mesh = GridTets(Beam{0.1, 0.1, 0.1, {}}, 3, 3, 3);
    const auto &material = materials::acoustic::Ceramic.Properties;
    JobMonitor monitor;
    monitor.RequestCancel();
    const auto cancelled = modal::mesh2modes(mesh, material, {vec3{0.f}}, vec3{1.f}, {}, {}, &monitor);
    EXPECT(cancelled.Modes.Freqs.empty());
    EXPECT(cancelled.Summary.Eigenvalues.empty());
    TetMesh flat; // every tet degenerate: nothing survives the filter
    flat.Points = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {1, 1, 0}};
    flat.Tets = {{0, 1, 2, 3}};
    EXPECT(modal::mesh2modes(flat, material, {vec3{0.f}}, vec3{1.f}).Modes.Freqs.empty());
}

CASE(warm_start_and_rescale_agree_with_a_cold_solve) {
    const auto mesh = GridTets(Beam{0.12, 0.08, 0.05, {}}, 6, 4, 3);
    const auto &ceramic = materials::acoustic::Ceramic.Properties;
    const std::vector<vec3> excite{vec3{0.f, 0.f, 0.f}, vec3{0.12f, 0.08f, 0.05f}};
    const auto cold = modal::mesh2modes(mesh, ceramic, excite, vec3{1.f}, {}, {.SeedBasis = nullptr, .KeepBasis = true});
    EXPECT(!cold.Modes.Freqs.empty());
    EXPECT(cold.Basis.rows() == std::ptrdiff_t(cold.Profile.Dofs));
    EXPECT(cold.MassProps.Mass > 0);
    EXPECT(check::near(cold.MassProps.Mass, 2700 * 0.12 * 0.08 * 0.05, 1e-6));
    // a stiffer body of the same shape, seeded with the basis of the first solve
    auto stiffer = ceramic;
    stiffer.YoungModulus *= 1.21;
    const auto warm = modal::mesh2modes(mesh, stiffer, excite, vec3{1.f}, {}, {.SeedBasis = &cold.Basis, .KeepBasis = false});
    const auto again = modal::mesh2modes(mesh, stiffer, excite, vec3{1.f});
    EXPECT(warm.Modes.Freqs.size() == again.Modes.Freqs.size());
    for (size_t k = 0; k < std::min(warm.Modes.Freqs.size(), again.Modes.Freqs.size()); ++k)
        EXPECT(check::near(warm.Modes.Freqs[k], again.Modes.Freqs[k], 1e-4));
    // the same edit re-derived without a solve: frequencies scale with sqrt(E'/E)
    const auto rescaled = modal::RescaleModes(cold.Summary, cold.Modes, stiffer);
    EXPECT(rescaled.has_value());
    if (rescaled) {
        EXPECT(rescaled->Freqs.size() == again.Modes.Freqs.size());
        for (size_t k = 0; k < std::min(rescaled->Freqs.size(), again.Modes.Freqs.size()); ++k)
            EXPECT(check::near(rescaled->Freqs[k], again.Modes.Freqs[k], 1e-4));
    }
    auto other_nu = ceramic;
    other_nu.PoissonRatio = 0.3;
    EXPECT(!modal::RescaleModes(cold.Summary, cold.Modes, other_nu).has_value());
}

// The reference's shift-invert operator concept (src/audio/CholeskyShiftInvert.h:11-30) over the device path: rows / cols / set_shift /
// perform_op / solve_panel with the reference's argument meaning; (K - sigma M) applied to what it returns gives the input back; a panel
// solve agrees with its columns solved one by one; the operator is symmetric; a non-negative shift throws the reference's error.
CASE(the_shift_invert_operator_concept_over_the_device_path) {
    const auto mesh = GridTets(Beam{0.12, 0.08, 0.05, {}}, 6, 4, 3);
    const auto &ceramic = materials::acoustic::Ceramic.Properties;
    double factorize = 0, solve = 0;
    CholeskyShiftInvert op(mesh, ceramic, factorize, solve);
    const auto n = size_t(op.rows());
    EXPECT(op.rows() == op.cols() && n > 3 * mesh.Points.size()); // (quadratic elements: corner and midside nodes)
    bool threw = false;
    try {
        op.set_shift(1.0);
    } catch (const std::runtime_error &e) { threw = std::string(e.what()) == "Modal shift-invert factorization failed."; }
    EXPECT(threw);
    const double sigma = -15791.367041742974; // -(2 pi 20 Hz)^2, the reference's default shift (mesh2modes.cpp:460)
    op.set_shift(sigma);
    EXPECT(factorize > 0);
    std::vector<double> b(3 * n), x(3 * n), one(n);
    uint64_t st = 0x2545F4914F6CDD1Dull;
    for (auto &v : b) {
        st ^= st << 13, st ^= st >> 7, st ^= st << 17;
        v = double(st >> 11) / 9007199254740992.0 - 0.5;
    }
    op.solve_panel(b.data(), x.data(), 3);
    EXPECT(op.LastIterations > 0 && op.LastResidual < 1e-8 && solve > 0);
    // (K - sigma M) x = b through the C ABI's own product on a second system of the same mesh
    {
        mh_context *ctx = nullptr;
        mh_mesh *dm = nullptr;
        mh_system *ds = nullptr;
        EXPECT(mh_context_create(0, &ctx) == MH_OK);
        const mh_material mat{ceramic.Density, ceramic.YoungModulus, ceramic.PoissonRatio, ceramic.Alpha, ceramic.Beta};
        EXPECT(mh_mesh_create(ctx, uint32_t(mesh.Points.size()), reinterpret_cast<const double *>(mesh.Points.data()), uint32_t(mesh.Tets.size()),
                              reinterpret_cast<const uint32_t *>(mesh.Tets.data()), &dm) == MH_OK);
        EXPECT(mh_assemble(ctx, dm, &mat, &ds) == MH_OK);
        std::vector<double> back(3 * n);
        EXPECT(mh_system_matvec(ds, 2, x.data(), back.data(), 3) == MH_OK);
        double worst = 0, scale = 0;
        for (size_t i = 0; i < 3 * n; ++i) worst = std::max(worst, std::fabs(back[i] - b[i])), scale = std::max(scale, std::fabs(b[i]));
        EXPECT(worst < 1e-7 * scale);
        mh_system_destroy(ds);
        mh_mesh_destroy(dm);
        mh_context_destroy(ctx);
    }
    // one column at a time gives the panel's columns; the operator is symmetric: b0 . op(b1) = b1 . op(b0)
    op.perform_op(b.data() + n, one.data());
    double diff = 0, size = 0;
    for (size_t i = 0; i < n; ++i) diff = std::max(diff, std::fabs(one[i] - x[n + i])), size = std::max(size, std::fabs(x[n + i]));
    EXPECT(diff < 1e-7 * size);
    double s01 = 0, s10 = 0;
    for (size_t i = 0; i < n; ++i) s01 += b[i] * x[n + i], s10 += b[n + i] * x[i];
    EXPECT(check::near(s01, s10, 1e-7));
}

int main() { return check::run_all(); }
