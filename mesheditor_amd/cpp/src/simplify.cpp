// SimplifySurface (modal/tets.hpp; the reference's src/mesh/Tets.h:8-10): quadric edge-collapse of a closed triangle surface to
// a fraction of its triangles, in place, then the unreferenced vertices are dropped.  Written from the published method
// (Garland & Heckbert 1997: per-vertex plane quadrics, collapse cost = the quadric of the merged vertex at its best position),
// with the three guards a surface that is going to be tetrahedralised needs:
//   * the link condition (the two end points share exactly the two vertices opposite the edge): the surface stays a
//     2-manifold, no pinched vertices, no doubled faces;
//   * no triangle around the merged vertex may turn over (normal against its old normal) or degenerate;
//   * no new triangle may pass through a triangle of the surrounding two-ring that it does not share a vertex with -- the
//     local form of "a collapse that folds the surface through itself": the collapse is refused and its neighbourhood frozen
//     (the edge is not tried again), so a region that cannot be coarsened safely keeps its resolution.
// A far-apart self-contact (two sheets approaching each other) is outside the two-ring and is not looked for.
#include "modal/tets.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <queue>
#include <unordered_map>
#include <vector>

namespace {
struct Quadric { // symmetric 4 x 4: a (3 x 3, upper), b (3), c
    double a00{0}, a01{0}, a02{0}, a11{0}, a12{0}, a22{0}, b0{0}, b1{0}, b2{0}, c{0};
    void AddPlane(const dvec3 &n, double d, double weight) { // weight * (n.x + d)^2
        a00 += weight * n.x * n.x, a01 += weight * n.x * n.y, a02 += weight * n.x * n.z;
        a11 += weight * n.y * n.y, a12 += weight * n.y * n.z, a22 += weight * n.z * n.z;
        b0 += weight * n.x * d, b1 += weight * n.y * d, b2 += weight * n.z * d;
        c += weight * d * d;
    }
    Quadric operator+(const Quadric &o) const {
        return {a00 + o.a00, a01 + o.a01, a02 + o.a02, a11 + o.a11, a12 + o.a12, a22 + o.a22, b0 + o.b0, b1 + o.b1, b2 + o.b2, c + o.c};
    }
    double At(const dvec3 &p) const {
        return p.x * (a00 * p.x + 2 * a01 * p.y + 2 * a02 * p.z) + p.y * (a11 * p.y + 2 * a12 * p.z) + a22 * p.z * p.z + 2 * (b0 * p.x + b1 * p.y + b2 * p.z) + c;
    }
    bool Minimum(dvec3 &out) const { // A p = -b when A is comfortably invertible
        const double det = a00 * (a11 * a22 - a12 * a12) - a01 * (a01 * a22 - a12 * a02) + a02 * (a01 * a12 - a11 * a02);
        const double scale = std::abs(a00) + std::abs(a11) + std::abs(a22);
        if (!(std::abs(det) > 1e-9 * scale * scale * scale)) return false;
        const double i00 = a11 * a22 - a12 * a12, i01 = a02 * a12 - a01 * a22, i02 = a01 * a12 - a02 * a11;
        const double i11 = a00 * a22 - a02 * a02, i12 = a01 * a02 - a00 * a12, i22 = a00 * a11 - a01 * a01;
        out = {-(i00 * b0 + i01 * b1 + i02 * b2) / det, -(i01 * b0 + i11 * b1 + i12 * b2) / det, -(i02 * b0 + i12 * b1 + i22 * b2) / det};
        return true;
    }
};

dvec3 Cross(const dvec3 &u, const dvec3 &v) { return {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x}; }
double Dot(const dvec3 &u, const dvec3 &v) { return u.x * v.x + u.y * v.y + u.z * v.z; }
double Orient(const dvec3 &a, const dvec3 &b, const dvec3 &c, const dvec3 &d) { return Dot(Cross(b - a, c - a), d - a); }

// Does segment pq cross the interior of triangle abc?  (floating point; touching counts as not crossing)
bool SegmentCrossesTriangle(const dvec3 &p, const dvec3 &q, const dvec3 &a, const dvec3 &b, const dvec3 &c) {
    const double sp = Orient(a, b, c, p), sq = Orient(a, b, c, q);
    if (!((sp > 0 && sq < 0) || (sp < 0 && sq > 0))) return false;
    const double s1 = Orient(p, q, a, b), s2 = Orient(p, q, b, c), s3 = Orient(p, q, c, a);
    return (s1 > 0 && s2 > 0 && s3 > 0) || (s1 < 0 && s2 < 0 && s3 < 0);
}
bool TrianglesCross(const std::array<dvec3, 3> &s, const std::array<dvec3, 3> &t) {
    for (int e = 0; e < 3; ++e) {
        if (SegmentCrossesTriangle(s[e], s[(e + 1) % 3], t[0], t[1], t[2])) return true;
        if (SegmentCrossesTriangle(t[e], t[(e + 1) % 3], s[0], s[1], s[2])) return true;
    }
    return false;
}
} // namespace

void SimplifySurface(std::vector<vec3> &positions, std::vector<uint32_t> &triangle_indices, float ratio) {
    const size_t nt0 = triangle_indices.size() / 3;
    if (!(ratio < 1.f) || nt0 < 8) return;
    const size_t target = std::max<size_t>(4, size_t(std::llround(double(std::max(ratio, 0.f)) * double(nt0))));
    const size_t nv = positions.size();
    std::vector<dvec3> pos(positions.begin(), positions.end());
    std::vector<std::array<uint32_t, 3>> tri(nt0);
    std::vector<uint8_t> alive(nt0, 1);
    std::vector<std::vector<uint32_t>> fans(nv); // vertex -> triangles (live or stale entries: filtered on use)
    for (size_t t = 0; t < nt0; ++t) {
        tri[t] = {triangle_indices[3 * t], triangle_indices[3 * t + 1], triangle_indices[3 * t + 2]};
        if (tri[t][0] == tri[t][1] || tri[t][1] == tri[t][2] || tri[t][0] == tri[t][2]) { alive[t] = 0; continue; }
        for (const uint32_t v : tri[t]) fans[v].push_back(uint32_t(t));
    }
    std::vector<Quadric> quadric(nv);
    for (size_t t = 0; t < nt0; ++t) {
        if (!alive[t]) continue;
        const dvec3 n = Cross(pos[tri[t][1]] - pos[tri[t][0]], pos[tri[t][2]] - pos[tri[t][0]]);
        const double len = std::sqrt(Dot(n, n));
        if (!(len > 0)) continue;
        const dvec3 unit = n * (1.0 / len);
        for (const uint32_t v : tri[t]) quadric[v].AddPlane(unit, -Dot(unit, pos[tri[t][0]]), 0.5 * len); // area-weighted
    }
    std::vector<uint32_t> version(nv, 0);
    std::vector<uint8_t> frozen(nv, 0);
    struct Candidate {
        double Cost;
        uint32_t A, B, VersionA, VersionB;
        dvec3 Where;
        bool operator<(const Candidate &o) const { return Cost > o.Cost || (Cost == o.Cost && (A > o.A || (A == o.A && B > o.B))); } // min-heap, deterministic ties
    };
    std::priority_queue<Candidate> heap;
    const auto live_fan = [&](uint32_t v) {
        auto &f = fans[v];
        f.erase(std::remove_if(f.begin(), f.end(), [&](uint32_t t) { return !alive[t] || (tri[t][0] != v && tri[t][1] != v && tri[t][2] != v); }), f.end());
        std::sort(f.begin(), f.end());
        f.erase(std::unique(f.begin(), f.end()), f.end());
        return f;
    };
    const auto propose = [&](uint32_t a, uint32_t b) {
        if (a > b) std::swap(a, b);
        const Quadric q = quadric[a] + quadric[b];
        dvec3 best;
        const dvec3 mid = (pos[a] + pos[b]) * 0.5;
        if (!q.Minimum(best) || Dot(best - mid, best - mid) > 4 * Dot(pos[a] - pos[b], pos[a] - pos[b])) { // singular, or far off the edge: best of three
            best = mid;
            for (const dvec3 &c : {pos[a], pos[b]})
                if (q.At(c) < q.At(best)) best = c;
        }
        heap.push({std::max(q.At(best), 0.0), a, b, version[a], version[b], best});
    };
    for (size_t t = 0; t < nt0; ++t)
        if (alive[t])
            for (int e = 0; e < 3; ++e)
                if (tri[t][e] < tri[t][(e + 1) % 3]) propose(tri[t][e], tri[t][(e + 1) % 3]); // every edge of a closed surface once
    size_t live = size_t(std::count(alive.begin(), alive.end(), uint8_t(1)));
    std::vector<uint32_t> ring_a, ring_b, around;
    while (live > target && !heap.empty()) {
        const Candidate c = heap.top();
        heap.pop();
        const uint32_t a = c.A, b = c.B;
        if (version[a] != c.VersionA || version[b] != c.VersionB || frozen[a] || frozen[b]) continue;
        const std::vector<uint32_t> fa = live_fan(a), fb = live_fan(b);
        // link condition: the vertices adjacent to both a and b are exactly the apexes of the triangles on the edge
        const auto ring_of = [&](uint32_t v, const std::vector<uint32_t> &fan, std::vector<uint32_t> &ring) {
            ring.clear();
            for (const uint32_t t : fan)
                for (const uint32_t w : tri[t])
                    if (w != v) ring.push_back(w);
            std::sort(ring.begin(), ring.end());
            ring.erase(std::unique(ring.begin(), ring.end()), ring.end());
        };
        ring_of(a, fa, ring_a);
        ring_of(b, fb, ring_b);
        std::vector<uint32_t> shared;
        std::set_intersection(ring_a.begin(), ring_a.end(), ring_b.begin(), ring_b.end(), std::back_inserter(shared));
        std::vector<uint32_t> on_edge;
        for (const uint32_t t : fa)
            if (tri[t][0] == b || tri[t][1] == b || tri[t][2] == b) on_edge.push_back(t);
        if (on_edge.size() != 2 || shared.size() != 2 || ring_a.size() + ring_b.size() < 7) continue; // (a tetrahedron-sized piece is left alone)
        // the triangles that survive, with the merged vertex at c.Where: none may turn over or degenerate
        bool ok = true;
        std::vector<std::array<dvec3, 3>> fresh;
        std::vector<std::array<uint32_t, 3>> fresh_ids;
        for (const std::vector<uint32_t> *fan : {&fa, &fb}) {
            for (const uint32_t t : *fan) {
                if (t == on_edge[0] || t == on_edge[1]) continue;
                std::array<dvec3, 3> before, after;
                for (int k = 0; k < 3; ++k) {
                    before[k] = pos[tri[t][k]];
                    after[k] = (tri[t][k] == a || tri[t][k] == b) ? c.Where : pos[tri[t][k]];
                }
                const dvec3 n0 = Cross(before[1] - before[0], before[2] - before[0]), n1 = Cross(after[1] - after[0], after[2] - after[0]);
                const double l0 = std::sqrt(Dot(n0, n0)), l1 = std::sqrt(Dot(n1, n1));
                if (!(l1 > 1e-12 * (l0 + 1e-300)) || Dot(n0, n1) < 0.2 * l0 * l1) { ok = false; break; }
                // no needles: a new triangle may not be far thinner than the one it replaces (compactness 4 sqrt(3) area / sum of
                // squared edges, 1 for an equilateral triangle) -- fans of needle triangles are what a Delaunay fill chokes on
                const auto compact = [](const std::array<dvec3, 3> &t, double twice_area) {
                    double e2 = 0;
                    for (int k = 0; k < 3; ++k) e2 += Dot(t[k] - t[(k + 1) % 3], t[k] - t[(k + 1) % 3]);
                    return e2 > 0 ? 3.4641016151377544 * twice_area / e2 : 0.0;
                };
                const double q0 = compact(before, l0), q1 = compact(after, l1);
                if (q1 < 0.15 && q1 < 0.7 * q0) { ok = false; break; }
                fresh.push_back(after);
                fresh_ids.push_back(tri[t]);
            }
            if (!ok) break;
        }
        if (ok) { // the local fold test: the new triangles against the two-ring's triangles they share no vertex with
            around.clear();
            for (const std::vector<uint32_t> *ring : {&ring_a, &ring_b})
                for (const uint32_t v : *ring)
                    for (const uint32_t t : live_fan(v)) around.push_back(t);
            std::sort(around.begin(), around.end());
            around.erase(std::unique(around.begin(), around.end()), around.end());
            for (size_t k = 0; k < fresh.size() && ok; ++k) {
                for (const uint32_t t : around) {
                    bool touches = false;
                    for (const uint32_t v : tri[t]) touches = touches || v == a || v == b || v == fresh_ids[k][0] || v == fresh_ids[k][1] || v == fresh_ids[k][2];
                    if (touches) continue;
                    if (TrianglesCross(fresh[k], {pos[tri[t][0]], pos[tri[t][1]], pos[tri[t][2]]})) { ok = false; break; }
                }
            }
            if (!ok) frozen[a] = frozen[b] = 1; // this neighbourhood keeps its resolution
        }
        if (!ok) continue;
        // collapse b into a
        alive[on_edge[0]] = alive[on_edge[1]] = 0;
        live -= 2;
        for (const uint32_t t : fb) {
            if (!alive[t]) continue;
            for (uint32_t &v : tri[t])
                if (v == b) v = a;
            fans[a].push_back(t);
        }
        fans[b].clear();
        pos[a] = c.Where;
        quadric[a] = quadric[a] + quadric[b];
        ++version[a];
        ++version[b];
        ring_of(a, live_fan(a), ring_a);
        for (const uint32_t w : ring_a) propose(a, w);
    }
    // compact: referenced vertices only, in their old order
    std::vector<uint32_t> remap(nv, UINT32_MAX);
    std::vector<vec3> out_pos;
    std::vector<uint32_t> out_tri;
    for (size_t t = 0; t < nt0; ++t) {
        if (!alive[t]) continue;
        for (const uint32_t v : tri[t]) {
            if (remap[v] == UINT32_MAX) remap[v] = 0; // mark
        }
    }
    for (size_t v = 0; v < nv; ++v)
        if (remap[v] != UINT32_MAX) {
            remap[v] = uint32_t(out_pos.size());
            out_pos.push_back(vec3{float(pos[v].x), float(pos[v].y), float(pos[v].z)});
        }
    for (size_t t = 0; t < nt0; ++t)
        if (alive[t])
            for (const uint32_t v : tri[t]) out_tri.push_back(remap[v]);
    positions.swap(out_pos);
    triangle_indices.swap(out_tri);
}
