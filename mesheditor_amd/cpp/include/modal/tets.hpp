// The tet-generation front end of the path, under the contract of the reference's
// tetra::Tetrahedralize (src/mesh/Tetrahedralize.h:49-61): input vertex i keeps index i, triangle winding is ignored, every
// tet is positively oriented, the tets fill exactly the enclosed volume, and an open or unrecoverable surface returns an error
// string.  "Every input triangle is a boundary face, and added (Steiner) points lie strictly inside" (Tetrahedralize.h:59) holds
// when Profile::BdrySteinerCount is 0 -- the usual outcome: the recovery first adds points ON the surface (edge bisections),
// and a second pass (Options::InteriorSteiner, on by default) takes them off it again, last one first, by moving each inside and
// filling the two thin wedges that open under its restored triangles with a tetrahedron each (src/tetrahedralize.cpp,
// LiftBoundaryPoints).  A point for which no valid inner position exists stays on the surface and is counted.
// Where the general fill still DEPARTS from the reference's contract:
//   * non-manifold input is accepted since round 4 (internal walls attached along seams of three triangles, fins with a free
//     border: every triangle a constraint, "inside" = what the outside cannot reach), but a recovery point on a WALL stays on it
//     (counted in BdrySteinerCount) and nested cavities are only understood on manifold input (parity rule); duplicate
//     positions are rejected as "point coincides";
//   * sliver repair = connectivity changes (edge removal, 2-3 flips: Options::RepairSlivers) in turn with smoothing of the ADDED points
//     (round 4); the reference's Options::Quality / MaxVolume exist with the same meaning since round 5 (a constrained Delaunay
//     refinement of the finished fill: RefineQuality);
//   * tetrahedra are not the reference's: same contract, other interior (it flips and carves, this refines, re-tiles and repairs).
// Two fills:
//   tetra::Tetrahedralize   any closed, non-self-intersecting surface -- non-convex, non-star-shaped, any genus, nested
//                           cavities: a conforming Delaunay tetrahedralisation on exact predicates.  Surface triangles the
//                           Delaunay mesh lacks are recovered by bisecting their edges; the added points then move inside.
//                           Where that refinement runs away (coarse triangles on thin walls, needle fans: quadric-decimated
//                           scans, UV-sphere poles) a constrained recovery re-tiles the cells a missing edge or triangle cuts
//                           through instead, adding a point only where no tiling exists (src/tetrahedralize.cpp, step 2b).
//   tetra::FillStarShaped   surfaces star-shaped about their centroid: layered shells, no point on the surface is added,
//                           well-shaped elements (the Delaunay fill of a bare surface has long interior tets).
#pragma once
#include "expected.hpp"
#include "types.hpp"

#include <filesystem>
#include <optional>
#include <span>
#include <string>
#include <vector>

namespace tetra {
struct Options {
    // The reference's two options (src/mesh/Tetrahedralize.h:17-27), same meaning.  Quality: interior points are inserted until every
    // tetrahedron meets a circumradius-to-shortest-edge ratio of 2, where the fixed surface allows (a circumcentre the surface cuts off
    // is not inserted); it gates the refinement alone -- sliver repair and the smoothing of added points run either way.  MaxVolume: any
    // tetrahedron larger than this absolute volume (the input's own units) is split; setting it turns Quality on; 0 = unconstrained.
    // Every point they add lies strictly inside: the boundary stays the input triangulation (src/tetrahedralize.cpp, RefineQuality).
    bool Quality{false};
    double MaxVolume{0};
    size_t MaxRefinePoints{0}; // budget of the quality arm; 0 = max(20 000, 40 x the points of the fill)
    size_t MaxSteinerPoints{0}; // boundary-recovery budget; 0 = 2 x the input vertices + 4096
    bool InteriorSteiner{true}; // after the fill, move the recovery's points off the surface (every input triangle a boundary face again)
    bool RepairSlivers{true}; // connectivity-only sliver repair afterwards (the reference repairs slivers whatever its options: Tetrahedralize.h:20)
    double SliverTarget{0.25}; // tetrahedra with a shape measure below this are worked on (1 = regular, 0 = flat)
    // One interior point under every surface vertex (with RepairSlivers).  WhenFlat: only if the fill is left with flat cells at the
    // surface (a 96 x 48 UV sphere: cells of shape 1e-9, on which no iterative eigensolver converges) or is poorly shaped throughout
    // (10th-percentile shape measure below 0.08: a thick body without interior points of its own -- the shell then HALVES the solve time
    // at 2.4 x the unknowns and lowers the P2 frequencies towards their limit); thin-walled scans and grid bodies are left as they are.
    // The reference's default fill adds no such points (its quality arm does): a fill with a shell has more points than the reference's.
    enum class Shell { Never, WhenFlat, Always };
    Shell InteriorShell{Shell::WhenFlat};
    // With RepairSlivers: whatever cell every other repair leaves with a shape measure below 1e-2 gets an interior point beside it (the
    // quality arm run locally, src/tetrahedralize.cpp: BreakFlatCells) -- no fill leaves here with a cell flat to 1e-9, whatever the other
    // options (round 6; the reference's repair and vertex optimisation run "either way", Tetrahedralize.h:19-20).  Off: for tests that
    // want such a mesh (a caller's own TetMesh need not be well shaped).
    bool BreakFlatCells{true};
};
// Wall-clock seconds per stage, with size and effort counters: the reference's tetra::Profile (src/mesh/Tetrahedralize.h:29-44), field
// for field, filled with what the corresponding stage of THIS fill did (the bench prints them: tests/ModalSolverBench.cpp:297-327).
struct Profile {
    double DelaunaySeconds{}, RecoverSeconds{}, CarveSeconds{}, RefineSeconds{}; // 1 Delaunay of the vertices; 2 / 2b boundary recovery; 3 inside / outside and output; 4 everything after (lifting, repair, smoothing, shell, quality arm, flat cells)
    double SegmentSeconds{}, FaceSeconds{}, SuppressSeconds{}; // within recovery: edges, faces; moving the recovery's points off the surface (LiftBoundaryPoints)
    uint32_t TetCount{}, SteinerCount{}; // the mesh returned: tetrahedra, points beyond the input's
    uint32_t DelaunayTetCount{}; // tetrahedra of the Delaunay tetrahedralisation of the input points (their convex hull), before the surface is met
    uint32_t BdrySteinerCount{}, VolSteinerCount{}; // recovery points left ON the surface (0 = every input triangle is a boundary face as given), and those moved inside instead
    uint32_t FlipCount{}, SplitCount{}, MissingEdgeCount{}, MissingFaceCount{}, Builds{}; // exchanges (recovery re-tilings + sliver repair), edge bisections, constraints the Delaunay mesh lacked, attempts (1 conforming, 2 = the constrained recovery took over)
    // beyond the reference's fields: interior points by origin
    uint32_t ShellPointCount{}; // under the surface (Options::InteriorShell)
    uint32_t QualityPointCount{}; // the quality arm (Options::Quality / MaxVolume)
    uint32_t FlatCellPointCount{}; // beside cells every other repair had left flat (always on, like the reference's repair)
    uint32_t SliverExchangeCount{}; // edge removals and 2-3 flips of the sliver repair alone
};
struct Result {
    TetMesh Mesh;
    tetra::Profile Profile;
};
template <class T> using Expected = modal_compat::expected<T, std::string>; // std::expected<T, std::string> with a C++23 library
// The reference's signature (src/mesh/Tetrahedralize.h:61): an error string for open, self-intersecting or otherwise unrecoverable surfaces.
Expected<Result> Tetrahedralize(std::span<const dvec3> points, std::span<const uint32_t> triangle_indices, Options options = {});
// `layers` shells between the surface and the centroid (0: a plain fan of one tet per triangle).  Each layer is a copy of
// the surface shrunk towards the centroid; the prisms between consecutive shells are cut into three tets with the
// smallest-index diagonal rule, so neighbouring prisms agree on their shared faces.
Expected<Result> FillStarShaped(std::span<const dvec3> points, std::span<const uint32_t> triangle_indices, uint32_t layers = 2);
} // namespace tetra

// Surface meshes: positions + triangles of a Wavefront .obj (v / f records; polygons are fanned, negative and
// v/vt/vn indices understood, positions welded bit-exactly as the reference's loader does for its solve tool).
struct ObjSurface {
    std::vector<vec3> Positions;
    std::vector<uint32_t> TriangleIndices;
};
// The reference's SimplifySurface (src/mesh/Tets.h:8-10): quadric edge-collapse of the surface to `ratio` of its triangles, in
// place, then unreferenced vertices are dropped; a no-op when ratio >= 1.  The surface stays a closed 2-manifold; a collapse that
// would turn a triangle over or fold the surface through its own neighbourhood is refused and that neighbourhood keeps its
// resolution (src/simplify.cpp).
void SimplifySurface(std::vector<vec3> &positions, std::vector<uint32_t> &triangle_indices, float ratio);
// The reference's GenerateTets (src/mesh/Tets.h:16, Tets.cpp:265-268), same signature: float surface in, tet mesh out -- the general fill.
tetra::Expected<tetra::Result> GenerateTets(std::vector<vec3> positions, std::vector<uint32_t> triangle_indices, tetra::Options options = {});
// With a layer count (not in the reference): the layered fill when the surface is star-shaped about its centroid, the general fill otherwise.
tetra::Expected<tetra::Result> GenerateTets(const std::vector<vec3> &positions, const std::vector<uint32_t> &triangle_indices, uint32_t layers);
std::optional<ObjSurface> LoadObj(const std::filesystem::path &);
