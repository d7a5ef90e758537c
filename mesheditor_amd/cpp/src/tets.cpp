// Star-shaped tetrahedral fill and a small .obj reader (see modal/tets.hpp).
#include "modal/tets.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace {
double SignedVolume6(const dvec3 &a, const dvec3 &b, const dvec3 &c, const dvec3 &d) { // 6 * volume of tet abcd
    const dvec3 u = b - a, v = c - a, w = d - a;
    return u.x * (v.y * w.z - v.z * w.y) - u.y * (v.x * w.z - v.z * w.x) + u.z * (v.x * w.y - v.y * w.x);
}
} // namespace

namespace tetra {
Expected<Result> FillStarShaped(std::span<const dvec3> points, std::span<const uint32_t> tris, uint32_t layers) {
    Result out;
    const auto fail = [](const char *why) { return modal_compat::unexpected<std::string>(std::string(why)); };
    const size_t nv = points.size(), nt = tris.size() / 3;
    if (nv < 4 || nt < 4 || tris.size() % 3 != 0) return fail("the surface needs at least four vertices and four triangles");
    for (const auto i : tris)
        if (i >= nv) return fail("triangle index out of range");
    // closed and manifold: every undirected edge belongs to exactly two triangles
    std::map<std::pair<uint32_t, uint32_t>, int> edges;
    for (size_t t = 0; t < nt; ++t)
        for (int e = 0; e < 3; ++e) {
            const uint32_t a = tris[3 * t + e], b = tris[3 * t + (e + 1) % 3];
            if (a == b) return fail("degenerate triangle");
            ++edges[{std::min(a, b), std::max(a, b)}];
        }
    for (const auto &[edge, count] : edges)
        if (count != 2) return fail(count == 1 ? "the surface is open (an edge with a single triangle)" : "non-manifold edge: this fill needs a simple closed surface");
    // centroid of the referenced vertices
    std::vector<uint8_t> used(nv, 0);
    for (const auto i : tris) used[i] = 1;
    dvec3 c{0, 0, 0};
    size_t n_used = 0;
    for (size_t i = 0; i < nv; ++i)
        if (used[i]) { c = c + points[i]; ++n_used; }
    c = c * (1.0 / double(n_used));
    // winding is ignored: every triangle is turned so that (a, b, c, centroid) is positive; star-shapedness = none is flat
    double scale = 0;
    for (size_t i = 0; i < nv; ++i)
        if (used[i]) { const dvec3 d = points[i] - c; scale = std::max(scale, std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z)); }
    std::vector<std::array<uint32_t, 3>> oriented(nt);
    size_t flipped = 0;
    for (size_t t = 0; t < nt; ++t) {
        std::array<uint32_t, 3> f{tris[3 * t], tris[3 * t + 1], tris[3 * t + 2]};
        const double vol = SignedVolume6(points[f[0]], points[f[1]], points[f[2]], c);
        if (std::abs(vol) <= 1e-12 * scale * scale * scale) return fail("the surface is not star-shaped about its centroid (a triangle is seen edge-on)");
        if (vol < 0) { std::swap(f[1], f[2]); ++flipped; }
        oriented[t] = f;
    }
    // a consistently wound input is either all flipped or none; a mix means the centroid sees some triangles from behind
    {
        // consistency of the ORIENTED set: neighbours must traverse their shared edge in opposite directions
        std::map<std::pair<uint32_t, uint32_t>, int> directed;
        for (const auto &f : oriented)
            for (int e = 0; e < 3; ++e) ++directed[{f[e], f[(e + 1) % 3]}];
        for (const auto &[edge, count] : directed)
            if (count != 1 || !directed.count({edge.second, edge.first})) return fail("the surface is not star-shaped about its centroid");
    }
    (void)flipped;
    // vertices: shell 0 = the input (same indices), shells 1..layers shrunk towards the centroid, then the centroid
    auto &mesh = out.Mesh;
    mesh.Points.assign(points.begin(), points.end());
    for (uint32_t s = 1; s <= layers; ++s) {
        const double f = 1.0 - double(s) / double(layers + 1);
        for (size_t i = 0; i < nv; ++i) mesh.Points.push_back(c + (points[i] - c) * f);
    }
    const uint32_t centre = uint32_t(mesh.Points.size());
    mesh.Points.push_back(c);
    const auto add = [&](uint32_t a, uint32_t b, uint32_t cc, uint32_t d) {
        if (SignedVolume6(mesh.Points[a], mesh.Points[b], mesh.Points[cc], mesh.Points[d]) < 0) std::swap(cc, d);
        mesh.Tets.push_back({a, b, cc, d});
    };
    for (const auto &f : oriented) {
        for (uint32_t s = 0; s < layers; ++s) {
            // prism between shell s (p) and shell s + 1 (q); columns sorted by vertex id, diagonals from the smaller id's
            // bottom vertex to the larger id's top vertex on every side face
            std::array<uint32_t, 3> v = f;
            std::sort(v.begin(), v.end());
            const uint32_t lo = uint32_t(s * nv), hi = uint32_t((s + 1) * nv);
            const uint32_t pa = lo + v[0], pb = lo + v[1], pc = lo + v[2], qa = hi + v[0], qb = hi + v[1], qc = hi + v[2];
            add(pa, pb, pc, qc);
            add(pa, pb, qb, qc);
            add(pa, qa, qb, qc);
        }
        const uint32_t in = uint32_t(layers * nv);
        add(in + f[0], in + f[1], in + f[2], centre);
    }
    out.Profile.TetCount = uint32_t(mesh.Tets.size());
    out.Profile.SteinerCount = uint32_t(mesh.Points.size() - nv);
    out.Profile.Builds = 1;
    return out;
}
} // namespace tetra

namespace {
std::vector<dvec3> Widened(const std::vector<vec3> &positions) {
    std::vector<dvec3> points(positions.size());
    for (size_t i = 0; i < positions.size(); ++i) points[i] = {double(positions[i].x), double(positions[i].y), double(positions[i].z)};
    return points;
}
} // namespace

tetra::Expected<tetra::Result> GenerateTets(std::vector<vec3> positions, std::vector<uint32_t> triangle_indices, tetra::Options options) {
    return tetra::Tetrahedralize(Widened(positions), triangle_indices, options);
}

tetra::Expected<tetra::Result> GenerateTets(const std::vector<vec3> &positions, const std::vector<uint32_t> &triangle_indices, uint32_t layers) {
    const auto points = Widened(positions);
    auto layered = tetra::FillStarShaped(points, triangle_indices, layers);
    if (layered || layered.error().find("star-shaped") == std::string::npos) return layered;
    return tetra::Tetrahedralize(points, triangle_indices); // not star-shaped: the general fill
}

std::optional<ObjSurface> LoadObj(const std::filesystem::path &path) {
    std::ifstream in{path};
    if (!in) return std::nullopt;
    ObjSurface s;
    std::vector<vec3> raw;
    std::vector<uint32_t> weld; // raw position -> welded index
    std::map<std::array<uint32_t, 3>, uint32_t> seen; // bit patterns of x, y, z
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls{line};
        std::string tag;
        if (!(ls >> tag)) continue;
        if (tag == "v") {
            vec3 p;
            if (!(ls >> p.x >> p.y >> p.z)) return std::nullopt;
            std::array<uint32_t, 3> key;
            std::memcpy(key.data(), &p.x, 12);
            const auto [it, inserted] = seen.try_emplace(key, uint32_t(s.Positions.size()));
            if (inserted) s.Positions.push_back(p);
            weld.push_back(it->second);
            raw.push_back(p);
        } else if (tag == "f") {
            std::vector<uint32_t> poly;
            std::string item;
            while (ls >> item) {
                const long i = std::strtol(item.c_str(), nullptr, 10); // "v", "v/vt", "v//vn", "v/vt/vn"
                const long idx = i < 0 ? long(raw.size()) + i : i - 1;
                if (idx < 0 || idx >= long(raw.size())) return std::nullopt;
                poly.push_back(weld[size_t(idx)]);
            }
            for (size_t k = 1; k + 1 < poly.size(); ++k)
                if (poly[0] != poly[k] && poly[k] != poly[k + 1] && poly[0] != poly[k + 1]) s.TriangleIndices.insert(s.TriangleIndices.end(), {poly[0], poly[k], poly[k + 1]});
        }
    }
    if (s.Positions.empty()) return std::nullopt;
    return s;
}
