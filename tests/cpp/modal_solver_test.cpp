// The solve path through the mirrored C++ API (modal::mesh2modes and friends), checked against the closed forms of
// the reference's bar tests (tests/ModalSolverTest.cpp:228-261) and its calling conventions: cancellation through a
// JobMonitor, warm start through SolveReuse, RescaleModes on a material edit.
#include "harness.hpp"

#include <audio/mesh2modes.h>
#include <mesh/TetMesh.h>

#include <algorithm>
#include <array>
#include <map>
#include <numbers>

namespace {
struct Bar {
    double Length, Width, Thickness;
    AcousticMaterialProperties Material;
};

// Grid cells cut into six tetrahedra around the cell diagonal 0-7, all positively oriented.
TetMesh BoxTets(double lx, double ly, double lz, int nx, int ny, int nz) {
    TetMesh mesh;
    const auto vid = [&](int i, int j, int k) { return uint32_t((i * (ny + 1) + j) * (nz + 1) + k); };
    for (int i = 0; i <= nx; ++i)
        for (int j = 0; j <= ny; ++j)
            for (int k = 0; k <= nz; ++k) mesh.Points.push_back({lx * i / nx, ly * j / ny, lz * k / nz});
    static constexpr int Corner[6][4]{{0, 1, 3, 7}, {0, 3, 2, 7}, {0, 2, 6, 7}, {0, 6, 4, 7}, {0, 4, 5, 7}, {0, 5, 1, 7}};
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < ny; ++j)
            for (int k = 0; k < nz; ++k) {
                const uint32_t c[8]{vid(i, j, k), vid(i + 1, j, k), vid(i, j + 1, k), vid(i + 1, j + 1, k),
                                    vid(i, j, k + 1), vid(i + 1, j, k + 1), vid(i, j + 1, k + 1), vid(i + 1, j + 1, k + 1)};
                for (const auto &t : Corner) mesh.Tets.push_back({c[t[0]], c[t[1]], c[t[2]], c[t[3]]});
            }
    return mesh;
}

enum class Family { Longitudinal, Torsional, Bending, BendingY, BendingZ, Other };

// Which way a mode moves, from the share of its shape energy that is axial, rigid rotation of the cross sections
// about the bar axis, or lateral in one plane.
Family Classify(const ModalModes &modes, size_t mode, const Bar &bar, int nx) {
    double axial = 0, lat_y = 0, lat_z = 0;
    std::map<long, std::pair<double, double>> slice; // x slice -> (sum r x u, sum r^2)
    for (size_t p = 0; p < modes.Positions.size(); ++p) {
        const auto pos = modes.Positions[p];
        const auto u = modes.Shapes[p][mode];
        axial += double(u.x) * u.x;
        lat_y += double(u.y) * u.y;
        lat_z += double(u.z) * u.z;
        const double ry = pos.y - bar.Width / 2, rz = pos.z - bar.Thickness / 2;
        auto &s = slice[std::lround(pos.x * nx / bar.Length)];
        s.first += ry * u.z - rz * u.y;
        s.second += ry * ry + rz * rz;
    }
    const double total = axial + lat_y + lat_z;
    if (total <= 0) return Family::Other;
    double rotation = 0;
    for (const auto &[key, s] : slice)
        if (s.second > 0) rotation += s.first * s.first / s.second;
    if (axial / total > 0.85) return Family::Longitudinal;
    if (rotation / total > 0.85) return Family::Torsional;
    const double lateral = lat_y + lat_z;
    if (lateral / total > 0.6 && rotation / total < 0.5) {
        if (lat_y / lateral > 0.8) return Family::BendingY;
        if (lat_z / lateral > 0.8) return Family::BendingZ;
        return Family::Bending;
    }
    return Family::Other;
}

std::map<Family, std::vector<double>> SolveBar(const Bar &bar, int nx, int ny, int nz) {
    const auto mesh = BoxTets(bar.Length, bar.Width, bar.Thickness, nx, ny, nz);
    std::vector<vec3> excite;
    for (const auto &p : mesh.Points) excite.emplace_back(float(p.x), float(p.y), float(p.z)); // every point is an excitation position
    const auto result = modal::mesh2modes(mesh, bar.Material, excite, vec3{1.f});
    std::map<Family, std::vector<double>> fam;
    EXPECT(!result.Modes.Freqs.empty());
    EXPECT(result.Profile.Dofs == 3u * uint32_t((2 * nx + 1) * (2 * ny + 1) * (2 * nz + 1)));
    for (size_t k = 0; k < result.Modes.Freqs.size(); ++k) fam[Classify(result.Modes, k, bar, nx)].push_back(result.Modes.Freqs[k]);
    return fam;
}

std::vector<double> Harmonics(double f1) { return {f1, 2 * f1, 3 * f1}; }
// Free-free Euler-Bernoulli roots, each listed per_root times (degenerate planes)
std::vector<double> BendingTheory(const Bar &bar, double thickness, int per_root) {
    const double rg = thickness / std::sqrt(12.0);
    const double base = std::sqrt(bar.Material.YoungModulus / bar.Material.Density) * rg / (2 * std::numbers::pi * bar.Length * bar.Length);
    std::vector<double> out;
    for (const double bl : {4.73004074, 7.85320462, 10.9956078})
        for (int r = 0; r < per_root; ++r) out.push_back(bl * bl * base);
    return out;
}
void CheckFamily(const char *name, const std::vector<double> &fem, const std::vector<double> &theory, double tol, size_t min_count = 2) {
    const auto count = std::min(fem.size(), theory.size());
    EXPECT_NOTE(count >= min_count, name);
    for (size_t i = 0; i < count; ++i) {
        std::printf("%14s %zu: theory %9.2f Hz, FEM %9.2f Hz, ratio %.4f\n", name, i + 1, theory[i], fem[i], fem[i] / theory[i]);
        EXPECT_NOTE(std::abs(fem[i] / theory[i] - 1.0) < tol, name);
    }
}
} // namespace

CASE(square_bar_modes_match_closed_forms) {
    const Bar bar{.Length = 0.3, .Width = 0.05, .Thickness = 0.05, .Material = {.Density = 1000, .YoungModulus = 1e7, .PoissonRatio = 0, .Alpha = 0, .Beta = 0}};
    const double speed = std::sqrt(bar.Material.YoungModulus / bar.Material.Density);
    const double torsion_f1 = std::sqrt(bar.Material.Mu() / bar.Material.Density * 0.140577 * 6) / (2 * bar.Length);
    auto fem = SolveBar(bar, 20, 4, 4);
    auto bending = fem[Family::Bending];
    for (const auto f : {Family::BendingY, Family::BendingZ}) bending.insert(bending.end(), fem[f].begin(), fem[f].end());
    std::ranges::sort(bending);
    CheckFamily("longitudinal", fem[Family::Longitudinal], Harmonics(speed / (2 * bar.Length)), 0.01);
    CheckFamily("torsional", fem[Family::Torsional], Harmonics(torsion_f1), 0.05);
    bending.resize(std::min<size_t>(bending.size(), 2));
    CheckFamily("bending", bending, BendingTheory(bar, bar.Thickness, 2), 0.10);
}

CASE(thin_bar_bending_matches_closed_forms) {
    const Bar bar{.Length = 0.3, .Width = 0.05, .Thickness = 0.01, .Material = {.Density = 1000, .YoungModulus = 1e9, .PoissonRatio = 0, .Alpha = 0, .Beta = 0}};
    const double speed = std::sqrt(bar.Material.YoungModulus / bar.Material.Density);
    auto fem = SolveBar(bar, 30, 5, 1);
    CheckFamily("longitudinal", fem[Family::Longitudinal], Harmonics(speed / (2 * bar.Length)), 0.01);
    auto stiff = BendingTheory(bar, bar.Width, 1);
    stiff.resize(1);
    CheckFamily("bending-y", fem[Family::BendingY], stiff, 0.10, 1);
    CheckFamily("bending-z", fem[Family::BendingZ], BendingTheory(bar, bar.Thickness, 1), 0.05);
}

CASE(cancelled_and_degenerate_solves_return_empty_results) {
    const auto mesh = BoxTets(0.1, 0.1, 0.1, 3, 3, 3);
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
    const auto mesh = BoxTets(0.12, 0.08, 0.05, 6, 4, 3);
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

int main() { return check::run_all(); }
