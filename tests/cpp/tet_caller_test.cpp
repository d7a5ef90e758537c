// The tet front end as the reference's own callers use it: this file includes the mirror through the
// reference's header paths (mesh/Tets.h, mesh/Tetrahedralize.h, audio/mesh2modes.h) and restates, in the reference's idiom, the two call
// sites a maintainer would recompile against it --
//   tests/ModalSolveTool.cpp:72-77    auto tets = GenerateTets(positions, indices, {.Quality = ...}); if (!tets) ... tets.error();
//                                     modal::mesh2modes(tets->Mesh, material, positions, vec3{1}, config)
//   tests/ModalSolverBench.cpp:285-327 tets->Profile and the counters and stage seconds the bench prints
// -- so that it compiling IS the drop-in check; the assertions hold the Profile's arithmetic.  Host code: runs without a GPU (the solve
// itself is only bound, not called).
#include "harness.hpp"

#include <audio/mesh2modes.h>
#include <mesh/TetMesh.h>
#include <mesh/Tetrahedralize.h>
#include <mesh/Tets.h>

#include <chrono>
#include <string>
#include <type_traits>
#include <vector>

namespace {
struct Surface {
    std::vector<vec3> Positions;
    std::vector<uint32_t> TriangleIndices;
};
// a box with n x n x n quads a side, welded
Surface GridBox(int n, float side) {
    Surface s;
    std::vector<int> id(size_t(n + 1) * size_t(n + 1) * size_t(n + 1), -1);
    const auto at = [&](int i, int j, int k) {
        int &slot = id[size_t((i * (n + 1) + j) * (n + 1) + k)];
        if (slot < 0) {
            slot = int(s.Positions.size());
            s.Positions.push_back({side * float(i) / float(n), side * float(j) / float(n), side * float(k) / float(n)});
        }
        return uint32_t(slot);
    };
    const auto quad = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t d) { s.TriangleIndices.insert(s.TriangleIndices.end(), {a, b, c, a, c, d}); };
    for (int u = 0; u < n; ++u)
        for (int v = 0; v < n; ++v) {
            quad(at(0, u, v), at(0, u, v + 1), at(0, u + 1, v + 1), at(0, u + 1, v));
            quad(at(n, u, v), at(n, u + 1, v), at(n, u + 1, v + 1), at(n, u, v + 1));
            quad(at(u, 0, v), at(u + 1, 0, v), at(u + 1, 0, v + 1), at(u, 0, v + 1));
            quad(at(u, n, v), at(u, n, v + 1), at(u + 1, n, v + 1), at(u + 1, n, v));
            quad(at(u, v, 0), at(u, v + 1, 0), at(u + 1, v + 1, 0), at(u + 1, v, 0));
            quad(at(u, v, n), at(u + 1, v, n), at(u + 1, v + 1, n), at(u, v + 1, n));
        }
    return s;
}

// The reference's declarations, token for token (src/mesh/Tetrahedralize.h:61, src/mesh/Tets.h:16) with std::expected spelled as the alias
// that IS std::expected under a C++23 library: taking the functions' addresses at these types fails to compile if a signature drifts.
using TetrahedralizeFn = tetra::Expected<tetra::Result> (*)(std::span<const dvec3>, std::span<const uint32_t>, tetra::Options);
using GenerateTetsFn = tetra::Expected<tetra::Result> (*)(std::vector<vec3>, std::vector<uint32_t>, tetra::Options);
[[maybe_unused]] constexpr TetrahedralizeFn kTetrahedralize = &tetra::Tetrahedralize;
[[maybe_unused]] constexpr GenerateTetsFn kGenerateTets = &GenerateTets;
static_assert(std::is_same_v<decltype(tetra::Result::Mesh), TetMesh>);
static_assert(std::is_same_v<decltype(tetra::Result::Profile), tetra::Profile>);
static_assert(std::is_same_v<decltype(tetra::Options::Quality), bool> && std::is_same_v<decltype(tetra::Options::MaxVolume), double>);

// tests/ModalSolveTool.cpp:57-77, restated: never run here (no GPU in the build container), compiled and linked against the mirror
[[maybe_unused]] int SolveToolBody(const Surface *mesh, bool quality_flag) {
    const AcousticMaterialProperties material{.Density = 2700, .YoungModulus = 7.2e10, .PoissonRatio = 0.19, .Alpha = 5, .Beta = 2e-8};
    const modal::SolverConfig config{.MinModeFreq = 20, .MaxModeFreq = 16'000, .NumModes = 30, .NumFemModes = 45};
    auto tets = GenerateTets(mesh->Positions, mesh->TriangleIndices, {.Quality = quality_flag});
    if (!tets) {
        std::fprintf(stderr, "Tetrahedralization failed: %s\n", tets.error().c_str());
        return 1;
    }
    const auto result = modal::mesh2modes(tets->Mesh, material, mesh->Positions, vec3{1}, config);
    return result.Modes.Freqs.empty() ? 1 : 0;
}
} // namespace

CASE(generate_tets_binds_as_the_reference_declares_it_and_fills_its_profile) {
    const Surface box = GridBox(6, 0.1f);
    for (const bool quality : {false, true}) {
        const auto t0 = std::chrono::steady_clock::now();
        const auto tets = GenerateTets(box.Positions, box.TriangleIndices, {.Quality = quality});
        const double tets_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        EXPECT(bool(tets));
        if (!tets) {
            std::printf("FAILED: %s\n", tets.error().c_str());
            continue;
        }
        const auto &p = tets->Profile; // (tests/ModalSolverBench.cpp:290)
        // the line the bench prints (ModalSolverBench.cpp:306, 323)
        std::printf("        grid box%s %6zu pts %6zu tris -> %7u tets, %5u steiner | %8.5f s | flips %6u splits %4u missE %4u missF %4u\n", quality ? " q" : "  ", box.Positions.size(),
                    box.TriangleIndices.size() / 3, p.TetCount, p.SteinerCount, tets_seconds, p.FlipCount, p.SplitCount, p.MissingEdgeCount, p.MissingFaceCount);
        std::printf("        tets: delaunay %.3f s, recover %.3f s, carve %.3f s, refine %.3f s, %u flips, %u splits, %u missing edges, %u missing faces, %u steiner (%u on the surface, %u moved inside), %u builds\n",
                    p.DelaunaySeconds, p.RecoverSeconds, p.CarveSeconds, p.RefineSeconds, p.FlipCount, p.SplitCount, p.MissingEdgeCount, p.MissingFaceCount, p.SteinerCount, p.BdrySteinerCount, p.VolSteinerCount, p.Builds);
        EXPECT(p.TetCount == tets->Mesh.Tets.size() && p.TetCount > 0);
        EXPECT(p.SteinerCount == tets->Mesh.Points.size() - box.Positions.size());
        EXPECT(p.DelaunayTetCount > 0); // (a convex body: the hull's Delaunay cells are the body's)
        EXPECT(p.Builds >= 1 && p.Builds <= 2);
        EXPECT(p.BdrySteinerCount == 0); // every input triangle is a boundary face as given
        EXPECT(p.BdrySteinerCount + p.VolSteinerCount <= p.SplitCount);
        EXPECT(p.VolSteinerCount + p.ShellPointCount + p.QualityPointCount + p.FlatCellPointCount <= p.SteinerCount + p.BdrySteinerCount);
        EXPECT(p.FlipCount >= p.SliverExchangeCount);
        const double stages = p.DelaunaySeconds + p.RecoverSeconds + p.CarveSeconds + p.RefineSeconds;
        EXPECT(p.DelaunaySeconds > 0 && p.RecoverSeconds >= 0 && p.CarveSeconds > 0 && p.RefineSeconds >= 0);
        EXPECT(stages <= tets_seconds * 1.001 + 1e-4);
        EXPECT(p.SegmentSeconds + p.FaceSeconds <= p.RecoverSeconds * 1.001 + 1e-4 && p.SuppressSeconds <= p.RefineSeconds * 1.001 + 1e-4);
        if (quality) EXPECT(p.QualityPointCount > 0);
        else EXPECT(p.QualityPointCount == 0);
        // input vertex i keeps index i (Tetrahedralize.h:59)
        bool kept = true;
        for (size_t i = 0; i < box.Positions.size(); ++i) kept = kept && float(tets->Mesh.Points[i].x) == box.Positions[i].x && float(tets->Mesh.Points[i].y) == box.Positions[i].y && float(tets->Mesh.Points[i].z) == box.Positions[i].z;
        EXPECT(kept);
    }
}

CASE(an_unrecoverable_surface_comes_back_as_the_error_alternative) {
    Surface open = GridBox(2, 1.f);
    open.TriangleIndices.resize(open.TriangleIndices.size() - 3); // one triangle short of closed
    const auto tets = GenerateTets(open.Positions, open.TriangleIndices);
    EXPECT(!tets);
    EXPECT(!tets.has_value());
    if (!tets) EXPECT(tets.error().find("open") != std::string::npos);
    // the value alternative default-constructs (Result{}: an empty mesh), as std::expected's does
    const tetra::Expected<tetra::Result> blank;
    EXPECT(blank.has_value() && blank->Mesh.Tets.empty() && (*blank).Profile.TetCount == 0);
    const std::vector<dvec3> none;
    EXPECT(!tetra::Tetrahedralize(none, {}));
}

int main() { return check::run_all(); }
