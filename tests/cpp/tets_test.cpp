// The tet-generation front end against the structural properties the reference checks of its own
// tetrahedraliser (tests/ValidateTetMesh.h:47-140): input vertices unmoved, every tet positively oriented, faces paired,
// the boundary exactly the input triangles, and the filled volume equal to the surface's; plus its error returns.
#include "harness.hpp"

#include <modal/tets.hpp>

#include <algorithm>
#include <array>
#include <fstream>
#include <map>
#include <set>

namespace {
double Vol6(const dvec3 &a, const dvec3 &b, const dvec3 &c, const dvec3 &d) {
    const dvec3 u = b - a, v = c - a, w = d - a;
    return u.x * (v.y * w.z - v.z * w.y) - u.y * (v.x * w.z - v.z * w.x) + u.z * (v.x * w.y - v.y * w.x);
}
struct Surface {
    std::vector<dvec3> P;
    std::vector<uint32_t> T;
};
// a box surface with n x n quads a side, every other triangle wound the wrong way round
Surface BoxSurface(double lx, double ly, double lz, int n) {
    Surface s;
    std::map<std::array<int, 3>, uint32_t> ids;
    const auto vid = [&](int i, int j, int k) {
        const auto [it, fresh] = ids.try_emplace({i, j, k}, uint32_t(s.P.size()));
        if (fresh) s.P.push_back({lx * i / n, ly * j / n, lz * k / n});
        return it->second;
    };
    size_t count = 0;
    const auto quad = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
        if (count++ % 2) s.T.insert(s.T.end(), {a, b, c, a, c, d});
        else s.T.insert(s.T.end(), {a, c, b, a, d, c});
    };
    for (int axis = 0; axis < 3; ++axis)
        for (int side = 0; side <= n; side += n)
            for (int u = 0; u < n; ++u)
                for (int v = 0; v < n; ++v) {
                    const auto at = [&](int uu, int vv) {
                        int c[3];
                        c[axis] = side, c[(axis + 1) % 3] = uu, c[(axis + 2) % 3] = vv;
                        return vid(c[0], c[1], c[2]);
                    };
                    quad(at(u, v), at(u + 1, v), at(u + 1, v + 1), at(u, v + 1));
                }
    return s;
}
Surface Octahedron() {
    Surface s;
    s.P = {{1, 0, 0}, {-1, 0, 0}, {0, 1.5, 0}, {0, -1.5, 0}, {0, 0, 2}, {0, 0, -2}};
    for (uint32_t x : {0u, 1u})
        for (uint32_t y : {2u, 3u})
            for (uint32_t z : {4u, 5u}) s.T.insert(s.T.end(), {x, y, z});
    return s;
}
// an L-shaped prism: its vertex centroid lies outside the solid
Surface LPrism() {
    Surface s;
    const double xy[6][2]{{0, 0}, {4, 0}, {4, 1}, {1, 1}, {1, 4}, {0, 4}};
    for (int z = 0; z < 2; ++z)
        for (const auto &p : xy) s.P.push_back({p[0], p[1], double(z)});
    const uint32_t cap[4][3]{{0, 1, 2}, {0, 2, 3}, {0, 3, 4}, {0, 4, 5}};
    for (const auto &t : cap) {
        s.T.insert(s.T.end(), {t[0], t[2], t[1]});
        s.T.insert(s.T.end(), {t[0] + 6, t[1] + 6, t[2] + 6});
    }
    for (uint32_t i = 0; i < 6; ++i) {
        const uint32_t j = (i + 1) % 6;
        s.T.insert(s.T.end(), {i, j, j + 6, i, j + 6, i + 6});
    }
    return s;
}
std::array<uint32_t, 3> Sorted(uint32_t a, uint32_t b, uint32_t c) {
    std::array<uint32_t, 3> t{a, b, c};
    std::sort(t.begin(), t.end());
    return t;
}
// empty when the mesh is a valid fill of the surface, else the first defect
std::string Validate(const Surface &s, const TetMesh &mesh, double surface_volume) {
    for (size_t i = 0; i < s.P.size(); ++i)
        if (mesh.Points[i].x != s.P[i].x || mesh.Points[i].y != s.P[i].y || mesh.Points[i].z != s.P[i].z) return "input vertex moved";
    std::map<std::array<uint32_t, 3>, int> faces;
    double volume = 0;
    for (const auto &t : mesh.Tets) {
        const double v = Vol6(mesh.Points[t[0]], mesh.Points[t[1]], mesh.Points[t[2]], mesh.Points[t[3]]);
        if (v <= 0) return "non-positive tet";
        volume += v / 6;
        static constexpr int F[4][3]{{1, 3, 2}, {0, 2, 3}, {0, 3, 1}, {0, 1, 2}};
        for (const auto &f : F) ++faces[Sorted(t[f[0]], t[f[1]], t[f[2]])];
    }
    std::set<std::array<uint32_t, 3>> boundary, input;
    for (const auto &[f, count] : faces) {
        if (count > 2) return "a face shared by more than two tets";
        if (count == 1) boundary.insert(f);
    }
    for (size_t i = 0; i < s.T.size(); i += 3) input.insert(Sorted(s.T[i], s.T[i + 1], s.T[i + 2]));
    if (boundary != input) return "the boundary is not the input surface";
    if (!check::near(volume, surface_volume, 1e-12)) return "volume mismatch: " + std::to_string(volume);
    // added points strictly inside: none of them is a boundary vertex
    for (const auto &f : boundary)
        for (const auto v : f)
            if (v >= s.P.size()) return "an added point lies on the boundary";
    return {};
}
} // namespace

CASE(a_box_surface_fills_for_every_layer_count) {
    const auto s = BoxSurface(2, 1, 0.5, 3);
    for (uint32_t layers : {0u, 1u, 2u, 5u}) {
        const auto r = tetra::FillStarShaped(s.P, s.T, layers);
        EXPECT_NOTE(bool(r), r.Error);
        EXPECT(r.Mesh.Tets.size() == (s.T.size() / 3) * (3 * layers + 1));
        EXPECT(r.Mesh.Points.size() == s.P.size() * (layers + 1) + 1);
        const auto defect = Validate(s, r.Mesh, 1.0);
        EXPECT_NOTE(defect.empty(), defect);
    }
}

CASE(an_octahedron_fills) {
    const auto s = Octahedron();
    const auto r = tetra::FillStarShaped(s.P, s.T, 2);
    EXPECT_NOTE(bool(r), r.Error);
    const auto defect = Validate(s, r.Mesh, 8 * (1 * 1.5 * 2) / 6);
    EXPECT_NOTE(defect.empty(), defect);
}

CASE(unsuitable_surfaces_return_an_error) {
    auto open = BoxSurface(1, 1, 1, 2);
    open.T.resize(open.T.size() - 3);
    EXPECT(!tetra::FillStarShaped(open.P, open.T));
    EXPECT(tetra::FillStarShaped(open.P, open.T).Error.find("open") != std::string::npos);
    const auto bent = LPrism();
    const auto r = tetra::FillStarShaped(bent.P, bent.T);
    EXPECT(!r);
    EXPECT(r.Error.find("star-shaped") != std::string::npos);
    auto bad = BoxSurface(1, 1, 1, 1);
    bad.T[0] = 99;
    EXPECT(!tetra::FillStarShaped(bad.P, bad.T));
    EXPECT(!tetra::FillStarShaped({}, {}));
}

CASE(an_obj_file_loads_welded_and_fanned) {
    const char *path = "/tmp/modalhip_tets_test.obj";
    {
        std::ofstream out{path};
        out << "# a unit cube as quads, one corner listed twice\n"
               "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nv 1 0 1\nv 1 1 1\nv 0 1 1\nv 0 0 0\n"
               "vn 0 0 1\n"
               "f 9 4 3 2\nf 5//1 6//1 7//1 8//1\nf 1/1 2/1 6/1 5/1\nf -7 -6 -2 -3\nf 4 1 5 8\nf 2 3 7 6\n";
    }
    const auto obj = LoadObj(path);
    EXPECT(obj.has_value());
    EXPECT(obj->Positions.size() == 8);
    EXPECT(obj->TriangleIndices.size() == 36);
    const auto r = GenerateTets(obj->Positions, obj->TriangleIndices, 1);
    EXPECT_NOTE(bool(r), r.Error);
    Surface s;
    for (const auto &p : obj->Positions) s.P.push_back({p.x, p.y, p.z});
    s.T = obj->TriangleIndices;
    const auto defect = Validate(s, r.Mesh, 1.0);
    EXPECT_NOTE(defect.empty(), defect);
    EXPECT(!LoadObj("/nonexistent/file.obj"));
}

int main() { return check::run_all(); }
