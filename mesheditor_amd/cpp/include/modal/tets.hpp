// A tet-generation front end for the path: fills a closed triangle surface that is
// star-shaped about its centroid -- convex bodies, blobs, the boxes and spheres of the reference's own tests -- with
// positively oriented tetrahedra.  It honours the contract of the reference's tetra::Tetrahedralize
// (src/mesh/Tetrahedralize.h:49-61): input vertex i keeps index i, every input triangle is a boundary face, added points
// lie strictly inside, triangle winding is ignored, and an open or unsuitable surface returns an error string.  The
// reference's constrained Delaunay tetrahedraliser (10 k lines, any closed surface) is not rebuilt.
#pragma once
#include "types.hpp"

#include <filesystem>
#include <optional>
#include <span>
#include <string>
#include <vector>

namespace tetra {
struct Result {
    TetMesh Mesh;
    std::string Error; // empty on success
    explicit operator bool() const { return Error.empty(); }
};
// `layers` shells between the surface and the centroid (0: a plain fan of one tet per triangle).  Each layer is a copy of
// the surface shrunk towards the centroid; the prisms between consecutive shells are cut into three tets with the
// smallest-index diagonal rule, so neighbouring prisms agree on their shared faces.
Result FillStarShaped(std::span<const dvec3> points, std::span<const uint32_t> triangle_indices, uint32_t layers = 2);
} // namespace tetra

// Surface meshes: positions + triangles of a Wavefront .obj (v / f records; polygons are fanned, negative and
// v/vt/vn indices understood, positions welded bit-exactly as the reference's loader does for its solve tool).
struct ObjSurface {
    std::vector<vec3> Positions;
    std::vector<uint32_t> TriangleIndices;
};
// The reference's GenerateTets (src/mesh/Tets.cpp:265): float surface in, tet mesh out.
tetra::Result GenerateTets(const std::vector<vec3> &positions, const std::vector<uint32_t> &triangle_indices, uint32_t layers = 2);
std::optional<ObjSurface> LoadObj(const std::filesystem::path &);
