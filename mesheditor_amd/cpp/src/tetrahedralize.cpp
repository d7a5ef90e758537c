// General tet generation for closed triangle surfaces, under the contract of the reference's
// tetra::Tetrahedralize (src/mesh/Tetrahedralize.h:49-61): input vertex i keeps index i, every input triangle appears on
// the boundary (whole, or refined by points lying on it -- the reference's validator, tests/ValidateTetMesh.h:47-140,
// accepts refinements and counts them as boundary Steiner points), every tet is positively oriented, the tets fill exactly
// the enclosed volume, winding is ignored, and an open / self-intersecting / unrecoverable surface yields an error string.
//
// Method -- a conforming Delaunay tetrahedralisation with a constrained recovery behind it, written from the textbook
// algorithms, not from the reference's 10 k-line TetGen rewrite:
//   1. Delaunay tetrahedralisation of the surface vertices by incremental Bowyer-Watson insertion inside a far enclosing
//      tetrahedron, on exact predicates (predicates.hpp): visibility walk to the containing tet, cavity = tets whose open
//      circumball holds the point, re-triangulated as a fan.  Exact arithmetic keeps every cavity star-shaped, so fully
//      degenerate input (grid boxes: everything coplanar or cospherical) needs no perturbation.
//   2. Boundary recovery by refinement, one cut at a time: a surface edge that is not an edge of the tetrahedralisation is
//      bisected (both surface triangles on it are split with it); a surface triangle whose edges are present but whose face
//      is not has its longest edge bisected; everything near the new point is re-examined before the next cut.  The edge
//      actually cut is found by longest-edge propagation (Rivara): from the wanted edge on to the longest edge of a
//      neighbouring surface triangle while that is longer -- arbitrary-edge bisection breeds ever thinner pieces whose new
//      edges are again not Delaunay.  Cut points are the EXACT midpoints (coordinates kept as floating-point expansions), so
//      the pieces of an edge stay exactly collinear and the pieces of a triangle exactly coplanar.
//      On degenerate input (grid boxes: the cells' corners cospherical) the Delaunay tetrahedralisation is not unique; before a
//      point is added the degenerate cells around a missing edge are re-tiled among their own vertices (FlipIn): one-cell-thick
//      grid bodies -- the reference's sample boxes -- need no point at all.
//      Limits: coarse triangles on a thin wall and fans of needle triangles (quadric-decimated scans) make the refinement run
//      away; a cap on the added points (the input vertices + 2048) stops it and step 2b takes over.
//  2b. Constrained recovery (round 4), from the plain Delaunay tetrahedralisation of the vertices again: the cells a missing edge
//      passes through / a missing triangle cuts through are re-tiled among their own vertices so that they hold it (a backtracking
//      advancing front over positively oriented tetrahedra, pruned by what must stay: every surface edge and face already
//      there; the cells' neighbours join when no tiling exists).  The mesh stops being Delaunay, so the rare point that is still
//      needed -- a Schoenhardt-like pocket admits no tiling -- is inserted by splitting the cells that hold it, not by a cavity.
//      Decimated scan surfaces (25 %, 10 % of the triangles) fill in 1-3 s with 1-10 added points.
//   3. Inside / outside by parity: a flood from the enclosing tetrahedron that flips each time it crosses a surface face;
//      an inconsistent parity means the surface does not close.  Non-manifold input (internal walls): the flood stops at every
//      triangle and what it never reaches is inside.
//   4. The recovery's points are moved off the surface (LiftBoundaryPoints), then slivers are repaired by edge removal and 2-3
//      flips and the added points smoothed (RepairSlivers, SmoothAddedPoints) -- as the reference does whatever its options.
// Non-star-shaped, non-convex and higher-genus bodies (brackets, tori, bowls) go through unchanged code paths.
#include "modal/tets.hpp"

#include "predicates.hpp"

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstring>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <map>
#include <queue>
#include <set>
#include <unordered_map>
#include <unordered_set>

namespace tetra {
namespace {
// One attempt's outcome (internal): the mesh or an error, and the counters that become tetra::Profile.
struct Attempt {
    TetMesh Mesh;
    std::string Error; // empty on success
    uint32_t BoundarySteinerCount{0}; // added points left ON the surface (input triangles they refine are not boundary faces)
    uint32_t SliverExchanges{0}; // edge removals and 2-3 flips the sliver repair made
    uint32_t ShellPoints{0}, QualityPoints{0}, FlatCellPoints{0}; // interior points by origin
    tetra::Profile Profile; // seconds and recovery counters, filled as the stages run
    explicit operator bool() const { return Error.empty(); }
};
using Tri = std::array<uint32_t, 3>;
Tri Sorted(uint32_t a, uint32_t b, uint32_t c) {
    Tri t{a, b, c};
    std::sort(t.begin(), t.end());
    return t;
}
uint64_t EdgeKey(uint32_t a, uint32_t b) { return (uint64_t(std::min(a, b)) << 32) | std::max(a, b); }

// Face i of a tet = the three vertices other than vertex i, ordered so that vertex i lies on their positive side.
constexpr int FaceOf[4][3]{{1, 3, 2}, {0, 2, 3}, {0, 3, 1}, {0, 1, 2}};

struct Cell {
    uint32_t V[4];
    int32_t N[4]; // neighbour across face i, -1 outside the enclosing tetrahedron
    bool Alive{true};
    uint32_t Stamp{0}; // cavity marker of the insertion that last looked at the cell
};

class DelaunayMesh {
public:
    std::vector<dvec3> Points; // rounded positions (filters, output)
    std::vector<std::array<exact::Sum, 3>> Fine; // exact coordinates of the points from FirstFine on (boundary midpoints)
    uint32_t FirstFine{UINT32_MAX};
    std::vector<Cell> Cells;
    std::vector<int32_t> CellOf; // one live cell per vertex (every vertex of a removed cell is a vertex of a cell that replaces it)
    std::vector<uint32_t> Touched; // vertices of the cells the last insertion created
    std::string Error;

    // Is {u, v} an edge / {u, v, w} a face of the mesh?  Walks the cells around u.
    bool HasEdge(uint32_t u, uint32_t v) const { return StarHas(u, v, v); }
    bool HasFace(uint32_t u, uint32_t v, uint32_t w) const { return StarHas(u, v, w); }

    exact::Point At(uint32_t i) const { return {Points[i], i >= FirstFine ? Fine[i - FirstFine].data() : nullptr}; }
    // The exact midpoint of two points becomes a new point; returns its id.
    uint32_t AddMidpoint(uint32_t u, uint32_t v) {
        if (FirstFine == UINT32_MAX) FirstFine = uint32_t(Points.size());
        const exact::Point a = At(u), b = At(v);
        std::array<exact::Sum, 3> fine;
        dvec3 rounded;
        for (int k = 0; k < 3; ++k) {
            fine[k] = (a.X(k) + b.X(k)).Halved();
            rounded[k] = fine[k].Rounded();
        }
        Points.push_back(rounded);
        Fine.push_back(std::move(fine));
        return uint32_t(Points.size() - 1);
    }

    // The four corners of the enclosing tetrahedron take the ids [first_free, first_free + 4).
    void Enclose(const dvec3 &lo, const dvec3 &hi) {
        const dvec3 mid{(lo.x + hi.x) / 2, (lo.y + hi.y) / 2, (lo.z + hi.z) / 2};
        const double r = 4096.0 * std::max({hi.x - lo.x, hi.y - lo.y, hi.z - lo.z, 1e-30});
        const uint32_t base = uint32_t(Points.size());
        Points.push_back({mid.x - r, mid.y - r, mid.z - r});
        Points.push_back({mid.x + r, mid.y + r, mid.z - r});
        Points.push_back({mid.x + r, mid.y - r, mid.z + r});
        Points.push_back({mid.x - r, mid.y + r, mid.z + r});
        Cell c{{base, base + 1, base + 2, base + 3}, {-1, -1, -1, -1}};
        if (exact::Orient3D(At(c.V[0]), At(c.V[1]), At(c.V[2]), At(c.V[3])) < 0) std::swap(c.V[0], c.V[1]);
        Cells.push_back(c);
        CellOf.assign(Points.size(), -1);
        for (const uint32_t v : c.V) CellOf[v] = 0;
        Last = 0;
    }

    bool Insert(uint32_t p) {
        const exact::Point x = At(p);
        const int32_t start = Locate(x);
        if (start < 0) return false;
        ++Epoch;
        Touched.clear();
        // cavity: the containing cell and every cell reachable through faces whose open circumball holds the point
        std::vector<int32_t> cavity{start}, stack{start};
        Cells[start].Stamp = Epoch;
        while (!stack.empty()) {
            const int32_t c = stack.back();
            stack.pop_back();
            for (const int32_t n : Cells[c].N) {
                if (n < 0 || Cells[n].Stamp == Epoch) continue;
                const Cell &o = Cells[n];
                if (exact::InSphere(At(o.V[0]), At(o.V[1]), At(o.V[2]), At(o.V[3]), x) > 0) {
                    Cells[n].Stamp = Epoch;
                    cavity.push_back(n);
                    stack.push_back(n);
                }
            }
        }
        // fan of new cells over the cavity's boundary faces
        struct Opening {
            int32_t Cell, Face;
        };
        std::unordered_map<uint64_t, Opening> open_edges; // boundary edge -> the new cell face waiting for its twin
        const size_t first_new = Cells.size();
        for (const int32_t c : cavity) {
            for (int i = 0; i < 4; ++i) {
                const int32_t outside = Cells[c].N[i];
                if (outside >= 0 && Cells[outside].Stamp == Epoch) continue;
                const uint32_t f0 = Cells[c].V[FaceOf[i][0]], f1 = Cells[c].V[FaceOf[i][1]], f2 = Cells[c].V[FaceOf[i][2]];
                if (exact::Orient3D(At(f0), At(f1), At(f2), x) <= 0) {
                    Error = "point coincides with an existing vertex or the triangulation lost convexity";
                    return false;
                }
                Cell fresh{{f0, f1, f2, p}, {-1, -1, -1, outside}};
                const int32_t id = int32_t(Cells.size());
                if (outside >= 0) // the outside cell's face that looked at the cavity now looks at the new cell
                    for (int j = 0; j < 4; ++j)
                        if (Cells[outside].N[j] == c) Cells[outside].N[j] = id;
                Cells.push_back(fresh);
                if (CellOf.size() < Points.size()) CellOf.resize(Points.size(), -1);
                for (const uint32_t v : fresh.V) CellOf[v] = id, Touched.push_back(v);
                // the three faces containing p pair up across the boundary edges: face j (opposite f_j) holds edge {f_k, f_l}
                for (int j = 0; j < 3; ++j) {
                    const uint64_t key = EdgeKey(fresh.V[(j + 1) % 3], fresh.V[(j + 2) % 3]);
                    const auto it = open_edges.find(key);
                    if (it == open_edges.end()) {
                        open_edges.emplace(key, Opening{id, j});
                    } else {
                        Cells[id].N[j] = it->second.Cell;
                        Cells[it->second.Cell].N[it->second.Face] = id;
                        open_edges.erase(it);
                    }
                }
            }
        }
        if (!open_edges.empty()) {
            Error = "cavity boundary is not closed (inconsistent adjacency)";
            return false;
        }
        for (const int32_t c : cavity) Cells[c].Alive = false;
        Last = int32_t(first_new);
        return true;
    }

    // Replaces the cells of `region` (boundary faces kept: a cell counts as inside the region while its Stamp equals Epoch) by
    // another tiling on the region's own vertices that holds the edges `must_edges`, the faces `must_faces` and every face / edge of
    // the old tiling for which keep_face / keep_edge answer true: a backtracking advancing front over positively oriented
    // tetrahedra (with `delaunay`: only those whose circumsphere holds no region vertex, and the exchange must be locally Delaunay
    // against the outside as well).  A set of positively oriented tetrahedra whose faces pair up and whose outer faces are the
    // region's boundary tiles the region (the map has degree one).  False, mesh untouched, when the search finds none in `budget` steps.
    template <class KeepEdge, class KeepFace>
    bool Retile(const std::vector<int32_t> &region, const std::vector<std::array<uint32_t, 2>> &must_edges, const std::vector<Tri> &must_faces, const KeepEdge &keep_edge,
                const KeepFace &keep_face, bool delaunay, size_t max_vertices, size_t budget) {
        std::vector<uint32_t> verts;
        for (const int32_t id : region)
            for (const uint32_t v : Cells[size_t(id)].V)
                if (std::find(verts.begin(), verts.end(), v) == verts.end()) verts.push_back(v);
        if (verts.size() > max_vertices) return false;
        std::sort(verts.begin(), verts.end());
        // faces: which of the two sides already has its tetrahedron (bit 0: the negative side of the sorted triple, bit 1: the positive one)
        const auto side_bit = [&](const Tri &f, uint32_t v) { return exact::Orient3D(At(f[0]), At(f[1]), At(f[2]), At(v)) > 0 ? 2 : 1; };
        std::map<Tri, int> used;
        std::vector<Tri> kept_faces; // faces and edges inside the region that the new tiling must hold
        std::vector<std::array<uint32_t, 2>> kept_edges = must_edges;
        kept_faces = must_faces;
        for (const int32_t id : region) {
            const Cell &t = Cells[size_t(id)];
            for (int i = 0; i < 4; ++i) {
                const Tri f = Sorted(t.V[FaceOf[i][0]], t.V[FaceOf[i][1]], t.V[FaceOf[i][2]]);
                if (t.N[i] < 0 || Cells[size_t(t.N[i])].Stamp != Epoch) used[f] |= 3 ^ side_bit(f, t.V[i]); // boundary: the outside is taken
                else if (keep_face(f[0], f[1], f[2])) kept_faces.push_back(f);
            }
            for (int i = 0; i < 4; ++i)
                for (int j = i + 1; j < 4; ++j)
                    if (keep_edge(t.V[i], t.V[j])) kept_edges.push_back({t.V[i], t.V[j]});
        }
        std::map<std::array<uint32_t, 4>, bool> empty_ball; // candidate tetrahedron (sorted) -> no region vertex strictly inside its circumsphere
        const auto admissible = [&](const Tri &f, uint32_t v) {
            std::array<uint32_t, 4> key{f[0], f[1], f[2], v};
            std::sort(key.begin(), key.end());
            const auto it = empty_ball.find(key);
            if (it != empty_ball.end()) return it->second;
            std::array<uint32_t, 4> t = key;
            if (exact::Orient3D(At(t[0]), At(t[1]), At(t[2]), At(t[3])) < 0) std::swap(t[0], t[1]);
            bool ok = true;
            for (const uint32_t w : verts)
                if (delaunay && ok && w != t[0] && w != t[1] && w != t[2] && w != t[3] && exact::InSphere(At(t[0]), At(t[1]), At(t[2]), At(t[3]), At(w)) > 0) ok = false;
            // a tetrahedron that a wanted edge passes through, or that reaches through a wanted face, is in no tiling that holds them
            const auto has = [&](uint32_t x) { return t[0] == x || t[1] == x || t[2] == x || t[3] == x; };
            for (size_t e = 0; e < kept_edges.size() && ok; ++e) {
                const uint32_t c = kept_edges[e][0], d = kept_edges[e][1];
                if (has(c) && has(d)) continue;
                for (int i = 0; i < 4 && ok; ++i) {
                    const uint32_t p = t[size_t(i + 1) & 3], q = t[size_t(i + 2) & 3], r = t[size_t(i + 3) & 3];
                    if (p == c || q == c || r == c || p == d || q == d || r == d) continue; // (a face at an end of the edge is met there only, or along it: the other faces tell)
                    const int sc = exact::Orient3D(At(p), At(q), At(r), At(c)), sd = exact::Orient3D(At(p), At(q), At(r), At(d));
                    if (sc == 0 || sd == 0 || sc == sd) continue;
                    const int s1 = exact::Orient3D(At(c), At(d), At(p), At(q)), s2 = exact::Orient3D(At(c), At(d), At(q), At(r)), s3 = exact::Orient3D(At(c), At(d), At(r), At(p));
                    if ((s1 >= 0 && s2 >= 0 && s3 >= 0) || (s1 <= 0 && s2 <= 0 && s3 <= 0)) ok = false;
                }
            }
            for (size_t k = 0; k < kept_faces.size() && ok; ++k) {
                const Tri &g = kept_faces[k];
                if (has(g[0]) && has(g[1]) && has(g[2])) continue;
                for (int i = 0; i < 4 && ok; ++i)
                    for (int j = i + 1; j < 4 && ok; ++j) {
                        const uint32_t u = t[size_t(i)], w = t[size_t(j)];
                        if (u == g[0] || u == g[1] || u == g[2] || w == g[0] || w == g[1] || w == g[2]) continue;
                        const int su = exact::Orient3D(At(g[0]), At(g[1]), At(g[2]), At(u)), sw = exact::Orient3D(At(g[0]), At(g[1]), At(g[2]), At(w));
                        if (su == 0 || sw == 0 || su == sw) continue;
                        const int t1 = exact::Orient3D(At(u), At(w), At(g[0]), At(g[1])), t2 = exact::Orient3D(At(u), At(w), At(g[1]), At(g[2])), t3 = exact::Orient3D(At(u), At(w), At(g[2]), At(g[0]));
                        if (t1 != 0 && t1 == t2 && t2 == t3) ok = false;
                    }
            }
            return empty_ball[key] = ok;
        };
        std::vector<std::array<uint32_t, 4>> tiling;
        const auto complete = [&] {
            for (const uint32_t w : verts) { // a vertex whose whole star lies inside the region must not drop out of the mesh
                bool held = false;
                for (const auto &t : tiling) held = held || t[0] == w || t[1] == w || t[2] == w || t[3] == w;
                if (!held) return false;
            }
            for (const auto &e : kept_edges) {
                bool held = false;
                for (const auto &t : tiling)
                    held = held || (std::find(t.begin(), t.end(), e[0]) != t.end() && std::find(t.begin(), t.end(), e[1]) != t.end());
                if (!held) return false;
            }
            for (const Tri &f : kept_faces) {
                bool held = false;
                for (const auto &t : tiling)
                    held = held || (std::find(t.begin(), t.end(), f[0]) != t.end() && std::find(t.begin(), t.end(), f[1]) != t.end() && std::find(t.begin(), t.end(), f[2]) != t.end());
                if (!held) return false;
            }
            return true;
        };
        const auto advance = [&](auto &&self) -> bool {
            // the open face with the fewest tetrahedra that could still close it goes first (none: this branch is dead)
            Tri f{};
            int want = 0;
            size_t fewest = SIZE_MAX;
            std::vector<uint32_t> apexes, best_apexes;
            for (const auto &[g, bits] : used) {
                if (bits != 1 && bits != 2) continue;
                apexes.clear();
                for (const uint32_t v : verts) {
                    if (v == g[0] || v == g[1] || v == g[2]) continue;
                    if (exact::Orient3D(At(g[0]), At(g[1]), At(g[2]), At(v)) == 0 || side_bit(g, v) != (3 ^ bits) || !admissible(g, v)) continue;
                    apexes.push_back(v);
                    if (apexes.size() >= fewest) break;
                }
                if (apexes.size() < fewest) fewest = apexes.size(), f = g, want = 3 ^ bits, best_apexes = apexes;
                if (fewest == 0) return false;
            }
            if (fewest == SIZE_MAX) return complete();
            if (budget == 0) return false;
            --budget;
            (void)want;
            for (const uint32_t v : best_apexes) {
                const std::array<uint32_t, 4> t{f[0], f[1], f[2], v};
                std::array<std::pair<Tri, int>, 4> marks;
                bool fits = true;
                for (int i = 0; i < 4 && fits; ++i) {
                    const Tri g = Sorted(t[(i + 1) & 3], t[(i + 2) & 3], t[(i + 3) & 3]);
                    marks[size_t(i)] = {g, side_bit(g, t[size_t(i)])};
                    const auto it = used.find(g);
                    fits = it == used.end() || !(it->second & marks[size_t(i)].second);
                }
                if (!fits) continue;
                for (const auto &[g, bit] : marks) used[g] |= bit;
                tiling.push_back(t);
                if (self(self)) return true;
                tiling.pop_back();
                for (const auto &[g, bit] : marks) {
                    const auto it = used.find(g);
                    it->second &= ~bit;
                    if (it->second == 0) used.erase(it);
                }
            }
            return false;
        };
        if (!advance(advance)) return false;
        std::vector<Cell> fresh;
        for (const auto &t : tiling) {
            Cell cell{{t[0], t[1], t[2], t[3]}, {-1, -1, -1, -1}};
            if (exact::Orient3D(At(cell.V[0]), At(cell.V[1]), At(cell.V[2]), At(cell.V[3])) < 0) std::swap(cell.V[0], cell.V[1]);
            fresh.push_back(cell);
        }
        return Exchange(region, fresh, delaunay);
    }

    // Brings the missing edge {c, d} into the mesh WITHOUT adding a point, when the mesh can stay Delaunay: on degenerate
    // input (grid boxes: the corners of every cell cospherical, the corners of every surface quad concyclic) the Delaunay
    // tetrahedralisation is not unique, and the insertion order picked the other diagonal {a, b} of the planar quad
    // (a, c, b, d).  The cells around {a, b}, together with every cell that shares a circumsphere with one of them (the
    // whole degenerate Delaunay cells: a grid cell on the inside, the cells towards the enclosing tetrahedron outside a hull
    // face), are one region whose boundary faces stay; the region is tetrahedralised anew on its own vertices by a
    // backtracking advancing front over the tetrahedra with empty circumspheres, and the first tiling that holds {c, d}
    // and every face / edge for which `keep_face` / `keep_edge` answer true replaces the old cells -- if it is locally
    // Delaunay against the cells outside as well (exact), so later Bowyer-Watson insertions still see a Delaunay mesh.
    // (A set of positively oriented tetrahedra whose faces pair up and whose outer faces are the region's boundary is a
    // tiling of the region: the map has degree one.)  Returns false, mesh untouched, when {c, d} does not cross exactly one
    // edge in that manner, the region is too large to search, or no such tiling exists.
    template <class KeepEdge, class KeepFace> bool FlipIn(uint32_t c, uint32_t d, const KeepEdge &keep_edge, const KeepFace &keep_face) {
        // the edge {a, b} that {c, d} crosses, among the faces opposite c in the cells around c
        uint32_t a = 0, b = 0;
        int32_t first = -1;
        ForStar(c, [&](int32_t id) {
            const Cell &t = Cells[size_t(id)];
            uint32_t o[3];
            int n = 0;
            for (const uint32_t v : t.V)
                if (v != c) o[n++] = v;
            for (int k = 0; k < 3 && first < 0; ++k) {
                const uint32_t x = o[k], y = o[(k + 1) % 3], z = o[(k + 2) % 3];
                if (x == d || y == d) continue;
                if (exact::Orient3D(At(c), At(d), At(x), At(y)) != 0) continue; // c, d, x, y in one plane; z is off it
                const int sx = exact::Orient3D(At(c), At(d), At(z), At(x)), sy = exact::Orient3D(At(c), At(d), At(z), At(y));
                if (sx == 0 || sx != -sy) continue; // x and y strictly on opposite sides of the line c d
                const int sc = exact::Orient3D(At(x), At(y), At(z), At(c)), sd = exact::Orient3D(At(x), At(y), At(z), At(d));
                if (sc == 0 || sc != -sd) continue; // c and d strictly on opposite sides of the line x y
                a = x, b = y, first = id;
            }
            return first < 0;
        });
        if (first < 0 || keep_edge(a, b)) return false;
        // the cells around {a, b} ...
        constexpr size_t MaxRegionCells = 24, MaxRegionVertices = 14; // two grid cells and the cells outside a hull face: 16 cells, 14 vertices
        std::vector<int32_t> region;
        ++Epoch;
        {
            int32_t cell = first;
            uint32_t from = c;
            for (size_t guard = 0; guard < MaxRegionCells; ++guard) {
                const Cell &t = Cells[size_t(cell)];
                uint32_t to = UINT32_MAX;
                int i_from = -1;
                for (int i = 0; i < 4; ++i) {
                    if (t.V[i] == from) i_from = i;
                    else if (t.V[i] != a && t.V[i] != b) to = t.V[i];
                }
                if (i_from < 0 || to == UINT32_MAX) return false;
                region.push_back(cell);
                Cells[size_t(cell)].Stamp = Epoch;
                cell = t.N[i_from]; // across the face (a, b, to)
                from = to;
                if (cell < 0) return false; // (only at the enclosing tetrahedron: never for an edge of real vertices)
                if (cell == first) break;
            }
            if (cell != first) return false;
        }
        // ... and everything cospherical with them
        const auto flood = [&](std::vector<int32_t> &cells, size_t at) {
            for (; at < cells.size(); ++at) {
                const Cell t = Cells[size_t(cells[at])];
                for (int i = 0; i < 4; ++i) {
                    const int32_t n = t.N[i];
                    if (n < 0 || Cells[size_t(n)].Stamp == Epoch) continue;
                    const Cell &o = Cells[size_t(n)];
                    uint32_t facing = 0;
                    for (int j = 0; j < 4; ++j)
                        if (o.N[j] == cells[at]) facing = o.V[j];
                    if (exact::InSphere(At(t.V[0]), At(t.V[1]), At(t.V[2]), At(t.V[3]), At(facing)) != 0) continue;
                    if (cells.size() >= MaxRegionCells) return false;
                    Cells[size_t(n)].Stamp = Epoch;
                    cells.push_back(n);
                }
            }
            return true;
        };
        if (!flood(region, 0)) return false;
        const std::vector<std::array<uint32_t, 2>> wanted{{c, d}};
        const auto attempt = [&](const std::vector<int32_t> &cells) { return Retile(cells, wanted, {}, keep_edge, keep_face, true, MaxRegionVertices, 8000); };
        if (attempt(region)) return true;
        // no tiling with the region's own boundary: the faces it shares with neighbouring degenerate cells may be what stands in
        // the way (a grid cell's side faces were triangulated for the neighbours' convenience).  The neighbours join, one at a
        // time first, then two at a time.
        std::vector<std::vector<int32_t>> families;
        for (size_t at = 0; at < region.size(); ++at) {
            const Cell t = Cells[size_t(region[at])];
            for (int i = 0; i < 4; ++i) {
                const int32_t n = t.N[i];
                if (n < 0 || Cells[size_t(n)].Stamp == Epoch) continue;
                if (keep_face(t.V[FaceOf[i][0]], t.V[FaceOf[i][1]], t.V[FaceOf[i][2]])) continue;
                bool known = false;
                for (const auto &f : families) known = known || std::find(f.begin(), f.end(), n) != f.end();
                if (known) continue;
                std::vector<int32_t> family{n};
                Cells[size_t(n)].Stamp = Epoch;
                const bool whole = flood(family, 0);
                for (const int32_t id : family) Cells[size_t(id)].Stamp = 0;
                if (whole) families.push_back(std::move(family));
            }
        }
        const auto with = [&](std::initializer_list<size_t> chosen) {
            std::vector<int32_t> wider = region;
            for (const size_t f : chosen) wider.insert(wider.end(), families[f].begin(), families[f].end());
            if (wider.size() > MaxRegionCells) return false;
            const uint32_t epoch = Epoch; // (a successful attempt ends in Exchange, which moves the epoch on)
            for (size_t k = region.size(); k < wider.size(); ++k) Cells[size_t(wider[k])].Stamp = epoch;
            if (attempt(wider)) return true;
            for (size_t k = region.size(); k < wider.size(); ++k) Cells[size_t(wider[k])].Stamp = 0;
            return false;
        };
        for (size_t f = 0; f < families.size(); ++f)
            if (with({f})) return true;
        for (size_t f = 0; f < families.size(); ++f)
            for (size_t g = f + 1; g < families.size(); ++g)
                if (with({f, g})) return true;
        return false;
    }

    // ---- constrained recovery (no points added, the mesh stops being Delaunay) ------------------------------------------
    // The cells that the open segment c d meets, found by a flood from the cells around c through every face the segment
    // touches (exact tests; a segment that runs through an edge or a vertex takes all the cells around it along).  False when the
    // segment lies in the plane of a face it touches (a degenerate position this recovery does not handle) or the set outgrows `cap`.
    bool CellsAlongSegment(uint32_t c, uint32_t d, std::vector<int32_t> &cells, size_t cap) {
        ++Epoch;
        cells.clear();
        bool flat = false;
        ForStar(c, [&](int32_t id) {
            const Cell &t = Cells[size_t(id)];
            int at = 0;
            for (int i = 0; i < 4; ++i)
                if (t.V[i] == c) at = i;
            const uint32_t f0 = t.V[FaceOf[at][0]], f1 = t.V[FaceOf[at][1]], f2 = t.V[FaceOf[at][2]];
            if (f0 == d || f1 == d || f2 == d) return true; // (the edge exists: not asked for)
            const int ref = exact::Orient3D(At(c), At(f0), At(f1), At(f2));
            const int s1 = exact::Orient3D(At(c), At(f0), At(f1), At(d)), s2 = exact::Orient3D(At(c), At(f1), At(f2), At(d)), s3 = exact::Orient3D(At(c), At(f2), At(f0), At(d));
            if ((s1 == ref || s1 == 0) && (s2 == ref || s2 == 0) && (s3 == ref || s3 == 0)) {
                if (s1 == 0 && s2 == 0 && s3 == 0) flat = true; // (cannot happen for distinct points)
                Cells[size_t(id)].Stamp = Epoch;
                cells.push_back(id);
            }
            return true;
        });
        if (flat || cells.empty()) return false;
        for (size_t k = 0; k < cells.size(); ++k) {
            const Cell t = Cells[size_t(cells[k])];
            bool holds_d = false;
            for (const uint32_t v : t.V) holds_d = holds_d || v == d;
            if (holds_d) continue; // the far end: the segment stops here
            for (int i = 0; i < 4; ++i) {
                const int32_t n = t.N[i];
                if (n < 0 || Cells[size_t(n)].Stamp == Epoch) continue;
                const uint32_t p = t.V[FaceOf[i][0]], q = t.V[FaceOf[i][1]], r = t.V[FaceOf[i][2]];
                if (p == c || q == c || r == c) continue; // faces at c are met at c only (or lie along the segment: the cone test took their cells)
                const int sc = exact::Orient3D(At(p), At(q), At(r), At(c)), sd = exact::Orient3D(At(p), At(q), At(r), At(d));
                if (sc == 0 || sd == 0) {
                    if (sc == 0 && sd == 0) return false; // the segment lies in this face's plane
                    if (sd == 0 && (p == d || q == d || r == d)) { // the face holds the far end: its neighbour does too
                        Cells[size_t(n)].Stamp = Epoch;
                        cells.push_back(n);
                    }
                    continue;
                }
                if (sc == sd) continue;
                const int s1 = exact::Orient3D(At(c), At(d), At(p), At(q)), s2 = exact::Orient3D(At(c), At(d), At(q), At(r)), s3 = exact::Orient3D(At(c), At(d), At(r), At(p));
                const bool touches = (s1 >= 0 && s2 >= 0 && s3 >= 0) || (s1 <= 0 && s2 <= 0 && s3 <= 0);
                if (!touches) continue;
                if (cells.size() >= cap) return false;
                Cells[size_t(n)].Stamp = Epoch;
                cells.push_back(n);
            }
        }
        return true;
    }

    // The cells that the open triangle (a, b, c) cuts through -- every cell with an edge that pierces it -- when its three edges
    // are edges of the mesh already.  False when no cell is cut (a vertex in the triangle's plane is the only other way to miss it) or the
    // set outgrows `cap`.
    bool CellsAcrossTriangle(uint32_t a, uint32_t b, uint32_t c, std::vector<int32_t> &cells, size_t cap) {
        ++Epoch;
        cells.clear();
        const auto pierced = [&](const Cell &t) { // does an edge of the cell pass through the triangle's interior?
            for (int i = 0; i < 4; ++i)
                for (int j = i + 1; j < 4; ++j) {
                    const uint32_t u = t.V[i], v = t.V[j];
                    if (u == a || u == b || u == c || v == a || v == b || v == c) continue;
                    const int su = exact::Orient3D(At(a), At(b), At(c), At(u)), sv = exact::Orient3D(At(a), At(b), At(c), At(v));
                    if (su == 0 || sv == 0 || su == sv) continue;
                    const int t1 = exact::Orient3D(At(u), At(v), At(a), At(b)), t2 = exact::Orient3D(At(u), At(v), At(b), At(c)), t3 = exact::Orient3D(At(u), At(v), At(c), At(a));
                    if (t1 != 0 && t1 == t2 && t2 == t3) return true;
                }
            return false;
        };
        for (const uint32_t corner : {a, b, c})
            ForStar(corner, [&](int32_t id) {
                if (Cells[size_t(id)].Stamp != Epoch && pierced(Cells[size_t(id)])) {
                    Cells[size_t(id)].Stamp = Epoch;
                    cells.push_back(id);
                }
                return true;
            });
        for (size_t k = 0; k < cells.size(); ++k) {
            const Cell t = Cells[size_t(cells[k])];
            for (int i = 0; i < 4; ++i) {
                const int32_t n = t.N[i];
                if (n < 0 || Cells[size_t(n)].Stamp == Epoch || !pierced(Cells[size_t(n)])) continue;
                if (cells.size() >= cap) return false;
                Cells[size_t(n)].Stamp = Epoch;
                cells.push_back(n);
            }
        }
        return !cells.empty();
    }

    // Adds to `cells` (all stamped with the current Epoch) every live cell that shares a face with one of them, except across faces
    // for which keep_face answers true.
    template <class KeepFace> bool Widen(std::vector<int32_t> &cells, const KeepFace &keep_face, size_t cap) {
        const size_t own = cells.size();
        for (size_t k = 0; k < own; ++k) {
            const Cell t = Cells[size_t(cells[k])];
            for (int i = 0; i < 4; ++i) {
                const int32_t n = t.N[i];
                if (n < 0 || Cells[size_t(n)].Stamp == Epoch) continue;
                if (keep_face(t.V[FaceOf[i][0]], t.V[FaceOf[i][1]], t.V[FaceOf[i][2]])) continue;
                if (cells.size() >= cap) return false;
                Cells[size_t(n)].Stamp = Epoch;
                cells.push_back(n);
            }
        }
        return cells.size() > own;
    }

    // Inserts the point m, which lies on the open segment u v (its exact midpoint), WITHOUT keeping the mesh Delaunay: the cells whose
    // closure holds m -- one, the two on a face, or the ring around an edge -- are replaced by the cone from m over their outer
    // faces.  The constrained recovery's way of adding a point: nothing outside those cells changes, so what has been recovered stays.
    bool InsertBySplitting(uint32_t m, uint32_t u, uint32_t v) {
        const auto holds = [&](const Cell &t) {
            for (int i = 0; i < 4; ++i)
                if (exact::Orient3D(At(t.V[FaceOf[i][0]]), At(t.V[FaceOf[i][1]]), At(t.V[FaceOf[i][2]]), At(m)) < 0) return false;
            return true;
        };
        std::vector<int32_t> seeds;
        if (!CellsAlongSegment(u, v, seeds, 4096) || seeds.empty()) { // (the edge may exist already: the cells around it)
            seeds.clear();
            ForStar(u, [&](int32_t id) {
                for (const uint32_t x : Cells[size_t(id)].V)
                    if (x == v) seeds.push_back(id);
                return true;
            });
        }
        ++Epoch;
        std::vector<int32_t> region;
        for (const int32_t id : seeds)
            if (Cells[size_t(id)].Stamp != Epoch && holds(Cells[size_t(id)])) Cells[size_t(id)].Stamp = Epoch, region.push_back(id);
        for (size_t k = 0; k < region.size(); ++k) {
            const Cell t = Cells[size_t(region[k])];
            for (const int32_t n : t.N)
                if (n >= 0 && Cells[size_t(n)].Stamp != Epoch && holds(Cells[size_t(n)])) Cells[size_t(n)].Stamp = Epoch, region.push_back(n);
        }
        if (region.empty()) return Error = "a boundary point lies in no cell of the constrained mesh", false;
        std::vector<Cell> fresh;
        for (const int32_t id : region) {
            const Cell &t = Cells[size_t(id)];
            for (int i = 0; i < 4; ++i) {
                if (t.N[i] >= 0 && Cells[size_t(t.N[i])].Stamp == Epoch) continue;
                const uint32_t f0 = t.V[FaceOf[i][0]], f1 = t.V[FaceOf[i][1]], f2 = t.V[FaceOf[i][2]];
                const int s = exact::Orient3D(At(f0), At(f1), At(f2), At(m));
                if (s < 0) return Error = "a boundary point is not seen by the cells around it", false;
                if (s == 0) continue; // m lies in this outer face's plane: only possible on the enclosing tetrahedron's hull
                fresh.push_back(Cell{{f0, f1, f2, m}, {-1, -1, -1, -1}});
            }
        }
        if (CellOf.size() < Points.size()) CellOf.resize(Points.size(), -1);
        return Exchange(region, fresh, false);
    }

    // Forces the edge {c, d} / the face {a, b, c} into the mesh by re-tiling the cells it cuts through (then those and their
    // neighbours), giving up the Delaunay property.  False, mesh untouched, when no tiling is found.
    template <class KeepEdge, class KeepFace> bool ConstrainEdge(uint32_t c, uint32_t d, const KeepEdge &keep_edge, const KeepFace &keep_face) {
        std::vector<int32_t> cells;
        if (!CellsAlongSegment(c, d, cells, 48)) return false;
        const std::vector<std::array<uint32_t, 2>> wanted{{c, d}};
        for (int widen = 0; widen < 3; ++widen) { // (a search that runs out of steps costs seconds on 40 vertices; a point on the edge costs nothing)
            if (Retile(cells, wanted, {}, keep_edge, keep_face, false, 32, 3000)) return true;
            if (!Widen(cells, keep_face, 96)) return false;
        }
        return false;
    }
    template <class KeepEdge, class KeepFace> bool ConstrainFace(uint32_t a, uint32_t b, uint32_t c, const KeepEdge &keep_edge, const KeepFace &keep_face) {
        std::vector<int32_t> cells;
        if (!CellsAcrossTriangle(a, b, c, cells, 48)) return false;
        const std::vector<Tri> wanted{Sorted(a, b, c)};
        for (int widen = 0; widen < 3; ++widen) {
            if (Retile(cells, {}, wanted, keep_edge, keep_face, false, 32, 3000)) return true;
            if (!Widen(cells, keep_face, 96)) return false;
        }
        return false;
    }

private:
    int32_t Last{0};
    uint32_t Epoch{0};

    // Calls visit(cell id) for the live cells around vertex u until it returns false.
    template <class Visit> void ForStar(uint32_t u, const Visit &visit) const {
        if (u >= CellOf.size() || CellOf[u] < 0) return;
        if (StarSeen.size() < Cells.size()) StarSeen.resize(Cells.size(), 0);
        ++StarEpoch;
        StarStack.assign(1, CellOf[u]);
        StarSeen[size_t(CellOf[u])] = StarEpoch;
        while (!StarStack.empty()) {
            const int32_t id = StarStack.back();
            StarStack.pop_back();
            const Cell c = Cells[size_t(id)];
            for (int i = 0; i < 4; ++i) {
                if (c.V[i] == u || c.N[i] < 0) continue;
                if (StarSeen[size_t(c.N[i])] == StarEpoch) continue;
                StarSeen[size_t(c.N[i])] = StarEpoch;
                StarStack.push_back(c.N[i]);
            }
            if (!visit(id)) return;
        }
    }

    // Replaces the cells `old` by `fresh` (same region, positively oriented, adjacency not yet set) if every fresh cell is
    // locally Delaunay; false and nothing changed otherwise.
    bool Exchange(const std::vector<int32_t> &old, std::vector<Cell> &fresh, bool delaunay = true) {
        enum class Kind { Unmatched, Hull, Outside, Fresh };
        struct Side {
            Kind What{Kind::Unmatched};
            int32_t Cell{-1}, Face{-1}; // Outside: a live cell beyond the region and its face looking at it; Fresh: index into `fresh`
        };
        std::map<Tri, Side> open;
        ++Epoch;
        for (const int32_t c : old) Cells[size_t(c)].Stamp = Epoch;
        for (const int32_t c : old)
            for (int i = 0; i < 4; ++i) {
                const Cell &t = Cells[size_t(c)];
                const int32_t outside = t.N[i];
                if (outside >= 0 && Cells[size_t(outside)].Stamp == Epoch) continue;
                Side side{outside >= 0 ? Kind::Outside : Kind::Hull, outside, -1};
                if (outside >= 0)
                    for (int j = 0; j < 4; ++j)
                        if (Cells[size_t(outside)].N[j] == c) side.Face = j;
                open[Sorted(t.V[FaceOf[i][0]], t.V[FaceOf[i][1]], t.V[FaceOf[i][2]])] = side;
            }
        std::vector<std::array<Side, 4>> across(fresh.size());
        for (size_t f = 0; f < fresh.size(); ++f)
            for (int i = 0; i < 4; ++i) {
                const Tri key = Sorted(fresh[f].V[FaceOf[i][0]], fresh[f].V[FaceOf[i][1]], fresh[f].V[FaceOf[i][2]]);
                const auto it = open.find(key);
                if (it == open.end()) {
                    open[key] = Side{Kind::Fresh, int32_t(f), i};
                    continue;
                }
                across[f][size_t(i)] = it->second;
                if (it->second.What == Kind::Fresh) across[size_t(it->second.Cell)][size_t(it->second.Face)] = Side{Kind::Fresh, int32_t(f), i};
                open.erase(it);
            }
        if (!open.empty()) return false; // the fresh cells do not tile the region
        for (size_t f = 0; f < fresh.size(); ++f)
            for (int i = 0; i < 4; ++i) {
                const Side &s = across[f][size_t(i)];
                if (s.What == Kind::Unmatched) return false;
                if (s.What == Kind::Hull) continue;
                const uint32_t facing = s.What == Kind::Outside ? Cells[size_t(s.Cell)].V[s.Face] : fresh[size_t(s.Cell)].V[s.Face];
                if (delaunay && exact::InSphere(At(fresh[f].V[0]), At(fresh[f].V[1]), At(fresh[f].V[2]), At(fresh[f].V[3]), At(facing)) > 0) return false;
            }
        const int32_t base = int32_t(Cells.size());
        Touched.clear();
        for (size_t f = 0; f < fresh.size(); ++f)
            for (int i = 0; i < 4; ++i) {
                const Side &s = across[f][size_t(i)];
                if (s.What == Kind::Outside) {
                    fresh[f].N[i] = s.Cell;
                    Cells[size_t(s.Cell)].N[s.Face] = base + int32_t(f);
                } else if (s.What == Kind::Fresh) {
                    fresh[f].N[i] = base + s.Cell;
                }
            }
        for (const int32_t c : old) Cells[size_t(c)].Alive = false;
        for (size_t f = 0; f < fresh.size(); ++f) {
            Cells.push_back(fresh[f]);
            for (const uint32_t v : fresh[f].V) CellOf[v] = base + int32_t(f), Touched.push_back(v);
        }
        Last = base;
        return true;
    }
    mutable std::vector<int32_t> StarStack;
    mutable std::vector<uint32_t> StarSeen;
    mutable uint32_t StarEpoch{0};

    bool StarHas(uint32_t u, uint32_t v, uint32_t w) const {
        if (u >= CellOf.size() || CellOf[u] < 0) return false;
        if (StarSeen.size() < Cells.size()) StarSeen.resize(Cells.size(), 0);
        ++StarEpoch;
        StarStack.assign(1, CellOf[u]);
        StarSeen[size_t(CellOf[u])] = StarEpoch;
        while (!StarStack.empty()) {
            const Cell &c = Cells[size_t(StarStack.back())];
            StarStack.pop_back();
            bool has_v = false, has_w = false;
            for (const uint32_t x : c.V) has_v = has_v || x == v, has_w = has_w || x == w;
            if (has_v && has_w) return true;
            for (int i = 0; i < 4; ++i) {
                if (c.V[i] == u || c.N[i] < 0) continue; // the faces that contain u lead to the other cells around u
                if (StarSeen[size_t(c.N[i])] == StarEpoch) continue;
                StarSeen[size_t(c.N[i])] = StarEpoch;
                StarStack.push_back(c.N[i]);
            }
        }
        return false;
    }

    // Visibility walk: step through any face that has the point strictly on its far side.
    int32_t Locate(const exact::Point &x) {
        int32_t c = Last;
        while (c >= 0 && !Cells[c].Alive) c = c + 1 < int32_t(Cells.size()) ? c + 1 : -1;
        if (c < 0) {
            for (c = int32_t(Cells.size()) - 1; c >= 0 && !Cells[c].Alive; --c) {}
        }
        size_t steps = 0;
        const size_t limit = 8 * Cells.size() + 64;
        uint32_t spin = 0;
        while (c >= 0 && steps++ < limit) {
            const Cell &t = Cells[c];
            int32_t next = -2;
            for (int k = 0; k < 4 && next == -2; ++k) {
                const int i = int((k + spin) & 3u); // vary the face order so that a walk cannot circle
                if (exact::Orient3D(At(t.V[FaceOf[i][0]]), At(t.V[FaceOf[i][1]]), At(t.V[FaceOf[i][2]]), x) < 0) next = t.N[i];
            }
            ++spin;
            if (next == -2) return c;
            c = next;
        }
        Error = c < 0 ? "point outside the enclosing tetrahedron" : "point location did not terminate";
        return -1;
    }
};

struct SurfaceEdgeUse {
    std::vector<uint32_t> Triangles; // surface triangles on the edge
};
} // namespace

// The point farthest inside a set of half-spaces n . x + d >= 0 (|n| = 1), no further than `reach` from `old`: a linear programme in
// (position, depth), solved by enumeration of the vertices of its feasible set -- four planes at a time -- over a growing subset of the planes.
// Returns the depth (0: none found) and the point.  Rounded arithmetic: the caller's exact predicates have the last word.
struct HalfSpace {
    dvec3 n;
    double d;
};
static double ChebyshevVertex(const std::vector<HalfSpace> &planes, const std::vector<uint32_t> &active, const dvec3 &old, double reach, dvec3 &centre) {
    // the deepest vertex of { (x, depth) : n_i . x + d_i >= depth, i in active } within `reach` of `old`: every four of the planes, one 4 x 4 system each
    double best_depth = 0;
    const size_t na = active.size();
    for (size_t i0 = 0; i0 < na; ++i0)
        for (size_t i1 = i0 + 1; i1 < na; ++i1)
            for (size_t i2 = i1 + 1; i2 < na; ++i2)
                for (size_t i3 = i2 + 1; i3 < na; ++i3) {
                    // n_i . x - depth = -d_i for the four planes: a 4 x 4 system in (x, depth)
                    const HalfSpace *pl[4] = {&planes[active[i0]], &planes[active[i1]], &planes[active[i2]], &planes[active[i3]]};
                    double A[4][5];
                    for (int r = 0; r < 4; ++r) A[r][0] = pl[r]->n.x, A[r][1] = pl[r]->n.y, A[r][2] = pl[r]->n.z, A[r][3] = -1.0, A[r][4] = -pl[r]->d;
                    bool singular = false;
                    for (int col = 0; col < 4 && !singular; ++col) {
                        int piv = col;
                        for (int r = col + 1; r < 4; ++r)
                            if (std::fabs(A[r][col]) > std::fabs(A[piv][col])) piv = r;
                        if (std::fabs(A[piv][col]) < 1e-12) { singular = true; break; }
                        if (piv != col)
                            for (int cc = 0; cc < 5; ++cc) std::swap(A[piv][cc], A[col][cc]);
                        for (int r = 0; r < 4; ++r) {
                            if (r == col) continue;
                            const double f = A[r][col] / A[col][col];
                            for (int cc = col; cc < 5; ++cc) A[r][cc] -= f * A[col][cc];
                        }
                    }
                    if (singular) continue;
                    const dvec3 x{A[0][4] / A[0][0], A[1][4] / A[1][1], A[2][4] / A[2][2]};
                    const double depth = A[3][4] / A[3][3];
                    if (!(depth > best_depth)) continue;
                    const dvec3 move = x - old;
                    if (move.x * move.x + move.y * move.y + move.z * move.z > reach * reach) continue;
                    bool feasible = true;
                    for (size_t j = 0; j < na && feasible; ++j) {
                        const HalfSpace &h = planes[active[j]];
                        feasible = h.n.x * x.x + h.n.y * x.y + h.n.z * x.z + h.d >= depth * (1 - 1e-9) - 1e-300;
                    }
                    if (feasible) best_depth = depth, centre = x;
                }
    return best_depth;
}
static double ChebyshevCentre(const std::vector<HalfSpace> &planes, const dvec3 &old, double reach, dvec3 &centre) {
    const size_t np_ = planes.size();
    if (np_ < 4 || np_ > 4096) return 0;
    // Cutting planes: the optimum hangs on four of the planes, nearly always among those nearest to the point as it stands.  Start from the
    // sixteen nearest, solve, take in the planes the answer violates, solve again; with none violated the answer is the whole set's (and,
    // the active planes kept in their own order, the same four systems in the same arithmetic as the enumeration of all C(n, 4) would
    // reach -- at a fiftieth of the work for the fifty planes of a point's star).
    std::vector<uint32_t> order(np_);
    for (uint32_t i = 0; i < np_; ++i) order[i] = i;
    const auto slack = [&](uint32_t i, const dvec3 &x) { return planes[i].n.x * x.x + planes[i].n.y * x.y + planes[i].n.z * x.z + planes[i].d; };
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return slack(a, old) < slack(b, old); });
    std::vector<uint8_t> in(np_, 0);
    std::vector<uint32_t> active;
    for (size_t k = 0; k < std::min<size_t>(np_, 16); ++k) in[order[k]] = 1;
    for (int round = 0; round < 16; ++round) {
        active.clear();
        for (uint32_t i = 0; i < np_; ++i)
            if (in[i]) active.push_back(i);
        dvec3 x = old;
        const double depth = ChebyshevVertex(planes, active, old, reach, x);
        if (!(depth > 0)) {
            if (active.size() == np_) return 0;
            // (no vertex within reach on so few planes: the sub-problem is unbounded that way) -- eight more, nearest first
            size_t taken = 0;
            for (size_t k = 0; k < np_ && taken < 8; ++k)
                if (!in[order[k]]) in[order[k]] = 1, ++taken;
            continue;
        }
        std::vector<std::pair<double, uint32_t>> violated;
        for (uint32_t i = 0; i < np_; ++i)
            if (!in[i] && !(slack(i, x) >= depth * (1 - 1e-9) - 1e-300)) violated.emplace_back(slack(i, x), i);
        if (violated.empty()) {
            centre = x;
            return depth;
        }
        std::sort(violated.begin(), violated.end());
        for (size_t k = 0; k < std::min<size_t>(violated.size(), 8); ++k) in[violated[k].second] = 1;
    }
    if (np_ > 64) return 0; // (sixteen rounds of cuts did not settle it: the full enumeration is C(n, 4) systems, affordable for a few dozen planes only)
    active.clear();
    for (uint32_t i = 0; i < np_; ++i) active.push_back(i);
    return ChebyshevVertex(planes, active, old, reach, centre);
}

// The recovery above leaves its points ON the surface: the boundary of the mesh refines the input triangulation.  The reference's
// contract (src/mesh/Tetrahedralize.h:59) wants every input triangle a boundary face and added points strictly inside, so the
// points are taken off the surface again, last one first.  The last point m bisected a surface edge (a, b) with the apexes c, d
// of its two triangles, and nothing later touched its four boundary triangles (a, m, c), (m, b, c), (a, m, d), (m, b, d): moving
// m to a position m' strictly inside and adding the tetrahedra (a, b, c, m') and (a, b, d, m') fills exactly the two thin
// wedges that open between the old faces and the restored triangles (a, b, c), (a, b, d).  With m gone from the surface the
// point before it is in the same situation, and so on.  m' is taken along the bisector of the two triangles' inward normals or
// towards the centroid of m's tetrahedra, as far in as keeps every tetrahedron at m positively oriented (exact predicates) and
// gives the best worst-volume among them and the two new ones.  Returns how many points had to stay on the surface.
static uint32_t LiftBoundaryPoints(TetMesh &mesh, uint32_t n_input, const std::vector<std::array<uint32_t, 2>> &split_edge) {
    auto &P = mesh.Points;
    auto &T = mesh.Tets;
    const auto face_key = [](uint32_t a, uint32_t b, uint32_t c) {
        if (a > b) std::swap(a, b);
        if (b > c) std::swap(b, c);
        if (a > b) std::swap(a, b);
        return (uint64_t(a) << 42) | (uint64_t(b) << 21) | uint64_t(c); // (< 2^21 points: checked by the caller of this pass)
    };
    if (P.size() >= (size_t(1) << 21)) return uint32_t(P.size() - n_input);
    // faces of the mesh: how many tetrahedra share each, and one of them (a boundary face has exactly one)
    std::unordered_map<uint64_t, std::pair<uint32_t, uint32_t>> faces; // key -> (count, a tet)
    faces.reserve(T.size() * 2);
    std::vector<std::vector<uint32_t>> star(P.size() - n_input); // tets of every added point
    const auto add_tet = [&](uint32_t t) {
        const auto &v = T[t];
        for (int i = 0; i < 4; ++i) {
            auto &f = faces[face_key(v[(i + 1) & 3], v[(i + 2) & 3], v[(i + 3) & 3])];
            ++f.first;
            f.second = t;
            if (v[i] >= n_input) star[v[i] - n_input].push_back(t);
        }
    };
    for (uint32_t t = 0; t < T.size(); ++t) add_tet(t);
    const auto orient = [&](const dvec3 &a, const dvec3 &b, const dvec3 &c, const dvec3 &d) { return exact::Orient3D(a, b, c, d); };
    const auto volume6 = [](const dvec3 &a, const dvec3 &b, const dvec3 &c, const dvec3 &d) {
        const dvec3 u = b - a, v = c - a, w = d - a;
        return u.x * (v.y * w.z - v.z * w.y) - u.y * (v.x * w.z - v.z * w.x) + u.z * (v.x * w.y - v.y * w.x);
    };
    uint32_t stuck = 0;
    size_t why[4] = {0, 0, 0, 0};
    for (size_t k = split_edge.size(); k-- > 0;) {
        const uint32_t m = n_input + uint32_t(k), a = split_edge[k][0], b = split_edge[k][1];
        // the boundary triangles at m: exactly (a, m, c), (m, b, c), (a, m, d), (m, b, d)
        uint32_t apex[2] = {0, 0};
        int n_apex = 0, n_boundary = 0;
        bool pattern = true;
        for (const uint32_t t : star[k]) {
            const auto &v = T[t];
            for (int i = 0; i < 4 && pattern; ++i) {
                if (v[i] == m) continue; // the face opposite v[i] contains m
                const uint32_t f[3] = {v[(i + 1) & 3], v[(i + 2) & 3], v[(i + 3) & 3]};
                if (faces[face_key(f[0], f[1], f[2])].first != 1) continue;
                ++n_boundary;
                uint32_t others[2];
                int no = 0;
                for (const uint32_t x : f)
                    if (x != m) others[no++] = x;
                const bool has_a = others[0] == a || others[1] == a, has_b = others[0] == b || others[1] == b;
                if (has_a == has_b) { pattern = false; break; } // each boundary triangle at m holds exactly one end of the edge
                const uint32_t w = (others[0] == a || others[0] == b) ? others[1] : others[0];
                if (n_apex < 2 && (n_apex == 0 || apex[0] != w) && (n_apex < 2 || apex[1] != w)) {
                    if (n_apex == 0 || apex[0] != w) apex[n_apex++] = w;
                } else if (!(apex[0] == w || (n_apex == 2 && apex[1] == w))) pattern = false;
            }
        }
        if (!pattern || n_boundary != 4 || n_apex != 2) { ++stuck; ++why[0]; continue; }
        const uint32_t c = apex[0], d = apex[1];
        for (const auto key : {face_key(a, m, c), face_key(m, b, c), face_key(a, m, d), face_key(m, b, d)})
            if (!faces.count(key) || faces[key].first != 1) pattern = false;
        if (!pattern) { ++stuck; ++why[1]; continue; }
        // which side of the restored triangles is inside: the side of the fourth vertex of the tet on (a, m, c) / (a, m, d)
        const auto fourth = [&](uint32_t t, uint32_t x, uint32_t y, uint32_t z) {
            for (const uint32_t v : T[t])
                if (v != x && v != y && v != z) return v;
            return x;
        };
        const uint32_t ec = fourth(faces[face_key(a, m, c)].second, a, m, c), ed = fourth(faces[face_key(a, m, d)].second, a, m, d);
        const int in_c = orient(P[a], P[m], P[c], P[ec]), in_d = orient(P[a], P[m], P[d], P[ed]);
        if (in_c == 0 || in_d == 0) { ++stuck; ++why[2]; continue; }
        const auto cross = [](const dvec3 &u, const dvec3 &v) { return dvec3{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x}; };
        const auto unit = [](dvec3 v) {
            const double l = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
            return l > 0 ? dvec3{v.x / l, v.y / l, v.z / l} : v;
        };
        dvec3 nc = unit(cross(P[b] - P[a], P[c] - P[a])), nd = unit(cross(P[b] - P[a], P[d] - P[a]));
        if (volume6(P[a], P[b], P[c], P[ec]) < 0) nc = dvec3{-nc.x, -nc.y, -nc.z};
        if (volume6(P[a], P[b], P[d], P[ed]) < 0) nd = dvec3{-nd.x, -nd.y, -nd.z};
        dvec3 centroid{0, 0, 0};
        for (const uint32_t t : star[k])
            for (const uint32_t v : T[t]) centroid = centroid + P[v] * (0.25 / double(star[k].size()));
        const dvec3 old = P[m];
        const dvec3 ab = P[b] - P[a];
        const double h = 0.5 * std::sqrt(ab.x * ab.x + ab.y * ab.y + ab.z * ab.z);
        const dvec3 towards = centroid - old;
        const double reach = std::sqrt(towards.x * towards.x + towards.y * towards.y + towards.z * towards.z);
        const dvec3 dir_n = unit(nc + nd), dir_c = unit(towards), dir_mix = unit(dir_n + dir_c);
        const auto valid_quality = [&](const dvec3 &x, double &worst) { // every tet at m stays positive, the two new ones fit
            worst = 1e300;
            for (const uint32_t t : star[k]) {
                dvec3 q[4];
                for (int i = 0; i < 4; ++i) q[i] = T[t][i] == m ? x : P[T[t][i]];
                if (orient(q[0], q[1], q[2], q[3]) <= 0) return false;
                worst = std::min(worst, volume6(q[0], q[1], q[2], q[3]));
            }
            // (a, b, c, x): x on the inner side of the restored triangle, b beyond the face (a, x, c) as seen from inside, a beyond (x, b, c)
            if (orient(P[a], P[b], P[c], x) != in_c || orient(P[a], P[b], P[d], x) != in_d) return false;
            if (orient(P[a], x, P[c], P[b]) != -in_c || orient(x, P[b], P[c], P[a]) != -orient(P[m], P[b], P[c], P[fourth(faces[face_key(m, b, c)].second, m, b, c)])) return false;
            if (orient(P[a], x, P[d], P[b]) != -in_d || orient(x, P[b], P[d], P[a]) != -orient(P[m], P[b], P[d], P[fourth(faces[face_key(m, b, d)].second, m, b, d)])) return false;
            const int sc = orient(P[a], P[b], x, P[c]), sd = orient(P[a], P[b], x, P[d]);
            if (sc == 0 || sc != -sd) return false; // c and d on opposite sides of the face the two new tets share
            worst = std::min({worst, std::fabs(volume6(P[a], P[b], P[c], x)), std::fabs(volume6(P[a], P[b], P[d], x))});
            return true;
        };
        dvec3 best = old;
        double best_q = 0;
        for (const dvec3 &dir : {dir_n, dir_c, dir_mix})
            for (const double step : {1.0, 0.8, 0.6, 0.4, 0.25, 0.15, 0.08, 0.04, 0.02, 0.01, 0.003}) { // (shallow moves cost the eigensolver iterations: 39 at <= 0.6, 56 at <= 0.08, 79 at <= 0.02 on the 30k-tet scan)
                const double len = step * std::min(h, reach > 0 ? 2 * reach : h);
                const dvec3 x = old + dir * len;
                double q;
                if (valid_quality(x, q) && q > best_q) best_q = q, best = x;
            }
        if (!(best_q > 0)) {
            // None of the sampled positions is valid: on a finely tessellated smooth surface the tetrahedra at m include caps on nearly coplanar
            // surface vertices, and the region from which m sees every face of its link AND fits the two new tetrahedra is a sliver of space
            // the three sampled directions miss.  Every condition above is the sign of a determinant that is AFFINE in the position, so that
            // region is a polyhedron: its Chebyshev centre (the point farthest inside all the planes) is a small linear programme -- solved
            // here by enumeration of the vertices of the feasible set of (position, depth), some forty constraints at most -- and the exact
            // predicates then have the last word on it.
            std::vector<HalfSpace> planes;
            const double eps_len = h > 0 ? h : 1.0;
            const auto add_condition = [&](const auto &g, int want) { // g(x): an orientation determinant as a function of the position; want: its required sign
                if (want == 0) return;
                const double g0 = g(old);
                const dvec3 grad{(g(old + dvec3{eps_len, 0, 0}) - g0) / eps_len, (g(old + dvec3{0, eps_len, 0}) - g0) / eps_len, (g(old + dvec3{0, 0, eps_len}) - g0) / eps_len};
                const double len = std::sqrt(grad.x * grad.x + grad.y * grad.y + grad.z * grad.z);
                if (!(len > 0)) return;
                const double sgn = want > 0 ? 1.0 : -1.0;
                const dvec3 n = grad * (sgn / len);
                planes.push_back({n, sgn * g0 / len - (n.x * old.x + n.y * old.y + n.z * old.z)});
            };
            for (const uint32_t t : star[k]) {
                const auto tet = T[t];
                add_condition([&](const dvec3 &x) {
                    dvec3 q[4];
                    for (int i = 0; i < 4; ++i) q[i] = tet[size_t(i)] == m ? x : P[tet[size_t(i)]];
                    return volume6(q[0], q[1], q[2], q[3]);
                }, 1);
            }
            add_condition([&](const dvec3 &x) { return volume6(P[a], P[b], P[c], x); }, in_c);
            add_condition([&](const dvec3 &x) { return volume6(P[a], P[b], P[d], x); }, in_d);
            add_condition([&](const dvec3 &x) { return volume6(P[a], x, P[c], P[b]); }, -in_c);
            add_condition([&](const dvec3 &x) { return volume6(P[a], x, P[d], P[b]); }, -in_d);
            add_condition([&](const dvec3 &x) { return volume6(x, P[b], P[c], P[a]); }, -orient(P[m], P[b], P[c], P[fourth(faces[face_key(m, b, c)].second, m, b, c)]));
            add_condition([&](const dvec3 &x) { return volume6(x, P[b], P[d], P[a]); }, -orient(P[m], P[b], P[d], P[fourth(faces[face_key(m, b, d)].second, m, b, d)]));
            // (c and d on opposite sides of the face (a, b, x): implied for x inside both restored triangles' inner half-spaces near the edge; the exact test below checks it)
            dvec3 centre = old;
            const double best_depth = ChebyshevCentre(planes, old, 2 * h, centre);
            double q;
            if (best_depth > 0 && valid_quality(centre, q)) best_q = q, best = centre;
            if (!(best_q > 0)) { ++stuck; ++why[3]; continue; }
        }
        // apply: m moves; the four old boundary faces become interior, two tetrahedra restore (a, b, c) and (a, b, d)
        P[m] = best;
        for (const uint32_t w : {c, d}) {
            std::array<uint32_t, 4> tet{a, b, w, m};
            if (orient(P[tet[0]], P[tet[1]], P[tet[2]], P[tet[3]]) < 0) std::swap(tet[0], tet[1]);
            T.push_back(tet);
            add_tet(uint32_t(T.size() - 1));
        }
    }
    if (stuck && std::getenv("MH_TET_DEBUG")) std::fprintf(stderr, "[tets] lifting: %u of %zu points stay on the surface: %zu not the four-triangle pattern, %zu pattern faces not on the boundary, %zu degenerate side, %zu no valid inner position\n", stuck, split_edge.size(), why[0], why[1], why[2], why[3]);
    return stuck;
}

// MH_TET_DEBUG: every face on at most two tetrahedra, every tetrahedron positively oriented -- said per stage, so that a defect names its origin
static void DebugValidate(const TetMesh &mesh, const char *stage) {
    static const bool on = std::getenv("MH_TET_DEBUG") != nullptr;
    if (!on) return;
    std::map<Tri, int> count;
    size_t inverted = 0;
    for (const auto &t : mesh.Tets) {
        if (exact::Orient3D(mesh.Points[t[0]], mesh.Points[t[1]], mesh.Points[t[2]], mesh.Points[t[3]]) <= 0) ++inverted;
        for (int i = 0; i < 4; ++i) ++count[Sorted(t[size_t(i + 1) & 3], t[size_t(i + 2) & 3], t[size_t(i + 3) & 3])];
    }
    size_t over = 0;
    for (const auto &[f, c] : count) over += c > 2;
    std::fprintf(stderr, "[tets] %-28s %zu tets, %zu points: %zu faces on more than two tetrahedra, %zu tetrahedra not positively oriented\n", stage, mesh.Tets.size(), mesh.Points.size(), over, inverted);
}

// Sliver repair (the reference's tetrahedraliser repairs slivers whatever its options say: src/mesh/Tetrahedralize.h:20).  A
// Delaunay fill of a bare surface leaves flat tetrahedra -- four points of one latitude ring of a UV sphere, the wedges under a
// recovered edge -- whose stiffness entries dwarf their neighbours' and stall iterative eigensolvers (shape 2e-8 on the reference's
// sample sphere: ||A|| / theta ~ 1e13).  This pass changes the CONNECTIVITY only, by hill climbing on the worst shape measure of
// the tetrahedra involved (shape = 6 sqrt 2 V / l_rms^3: 1 for the regular tetrahedron, 0 for a flat one):
//   * edge removal: the n <= 12 tetrahedra around an interior edge {u, v} are replaced by the best triangulation of their link
//     polygon coned to u and to v (Klincsek's dynamic programme over the polygon: n = 3 is the 3-2 flip, n = 4 the 4-4 flip);
//   * the 2-3 flip of an interior face.
// Boundary faces are untouched (an edge on the boundary has an open ring and is skipped), no point is added or moved, every new
// tetrahedron is positively oriented (exact), and an exchange is made only if the worst shape among the new tetrahedra exceeds
// the worst among the old ones -- so the pass terminates and can only improve the mesh's worst elements.
static uint32_t RepairSlivers(TetMesh &mesh, double target, const std::set<Tri> *walls = nullptr) { // walls: faces INSIDE the mesh that must stay (non-manifold input)
    auto &P = mesh.Points;
    auto &T = mesh.Tets;
    struct FaceHash {
        size_t operator()(const Tri &f) const { return (size_t(f[0]) * 0x9E3779B97F4A7C15ull) ^ (size_t(f[1]) * 0xC2B2AE3D27D4EB4Full) ^ (size_t(f[2]) * 0x165667B19E3779F9ull); }
    };
    std::unordered_map<Tri, std::array<int32_t, 2>, FaceHash> faces; // sorted face -> the (at most two) live tets on it
    faces.reserve(T.size() * 2);
    std::vector<uint8_t> alive(T.size(), 1);
    const auto link = [&](int32_t t, bool add) {
        const auto &v = T[size_t(t)];
        for (int i = 0; i < 4; ++i) {
            const Tri key = Sorted(v[(i + 1) & 3], v[(i + 2) & 3], v[(i + 3) & 3]);
            if (add) {
                auto [it, fresh] = faces.try_emplace(key, std::array<int32_t, 2>{t, -1});
                if (!fresh) (it->second[0] < 0 ? it->second[0] : it->second[1]) = t;
            } else {
                auto it = faces.find(key);
                if (it == faces.end()) continue;
                if (it->second[0] == t) it->second[0] = it->second[1];
                it->second[1] = -1;
                if (it->second[0] < 0) faces.erase(it);
            }
        }
    };
    for (size_t t = 0; t < T.size(); ++t) link(int32_t(t), true);
    const auto across = [&](int32_t t, uint32_t a, uint32_t b, uint32_t c) -> int32_t { // the other tet on face {a, b, c}, or -1
        const auto it = faces.find(Sorted(a, b, c));
        if (it == faces.end()) return -1;
        return it->second[0] == t ? it->second[1] : it->second[0];
    };
    const auto shape_of = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
        const dvec3 &pa = P[a], &pb = P[b], &pc = P[c], &pd = P[d];
        const dvec3 u = pb - pa, v = pc - pa, w = pd - pa;
        const double vol6 = std::fabs(u.x * (v.y * w.z - v.z * w.y) - u.y * (v.x * w.z - v.z * w.x) + u.z * (v.x * w.y - v.y * w.x));
        double l2 = 0;
        const dvec3 *q[4] = {&pa, &pb, &pc, &pd};
        for (int i = 0; i < 4; ++i)
            for (int j = i + 1; j < 4; ++j) {
                const dvec3 e = *q[i] - *q[j];
                l2 += e.x * e.x + e.y * e.y + e.z * e.z;
            }
        const double lrms = std::sqrt(l2 / 6);
        return lrms > 0 ? 1.4142135623730951 * vol6 / (lrms * lrms * lrms) : 0.0;
    };
    const auto shape = [&](int32_t t) { return shape_of(T[size_t(t)][0], T[size_t(t)][1], T[size_t(t)][2], T[size_t(t)][3]); };
    const auto positive = [&](std::array<uint32_t, 4> t) {
        if (exact::Orient3D(P[t[0]], P[t[1]], P[t[2]], P[t[3]]) < 0) std::swap(t[0], t[1]);
        return t;
    };
    struct Plan {
        double Worst{0};
        std::vector<int32_t> Old;
        std::vector<std::array<uint32_t, 4>> Fresh;
    };
    constexpr int MaxRing = 12;
    // best re-tiling of the tets around the edge {u, v} of tet t; Worst = 0 when there is none
    const auto remove_edge = [&](int32_t t, uint32_t u, uint32_t v) {
        Plan plan;
        uint32_t ring_v[MaxRing];
        int32_t ring_t[MaxRing];
        int n = 0;
        uint32_t from = UINT32_MAX, to = UINT32_MAX;
        for (const uint32_t x : T[size_t(t)])
            if (x != u && x != v) (from == UINT32_MAX ? from : to) = x;
        int32_t cell = t;
        double old_worst = 1e300;
        for (;;) {
            if (n == MaxRing) return plan;
            ring_v[n] = from, ring_t[n] = cell, ++n;
            old_worst = std::min(old_worst, shape(cell));
            const int32_t next = across(cell, u, v, to);
            if (next < 0) return plan; // the edge lies on the boundary
            if (walls && walls->count(Sorted(u, v, to))) return plan; // ... or on an internal wall
            uint32_t beyond = UINT32_MAX;
            for (const uint32_t x : T[size_t(next)])
                if (x != u && x != v && x != to) beyond = x;
            from = to, to = beyond, cell = next;
            if (cell == t) break;
        }
        if (n < 3) return plan;
        const int turn = exact::Orient3D(P[ring_v[0]], P[ring_v[1]], P[v], P[u]);
        if (turn == 0) return plan;
        double best[MaxRing][MaxRing];
        int apex[MaxRing][MaxRing];
        for (int len = 1; len < n; ++len)
            for (int i = 0; i + len < n; ++i) {
                const int j = i + len;
                if (len == 1) { best[i][j] = 1e300; continue; }
                best[i][j] = -1, apex[i][j] = -1;
                for (int k = i + 1; k < j; ++k) {
                    if (best[i][k] <= old_worst || best[k][j] <= old_worst) continue;
                    const uint32_t a = ring_v[i], b = ring_v[k], c = ring_v[j];
                    if (exact::Orient3D(P[a], P[b], P[c], P[u]) != turn || exact::Orient3D(P[a], P[b], P[c], P[v]) != -turn) continue;
                    const double q = std::min({best[i][k], best[k][j], shape_of(a, b, c, u), shape_of(a, b, c, v)});
                    if (q > best[i][j]) best[i][j] = q, apex[i][j] = k;
                }
            }
        if (!(best[0][n - 1] > old_worst * 1.001 + 1e-14)) return plan;
        plan.Worst = best[0][n - 1];
        plan.Old.assign(ring_t, ring_t + n);
        const auto emit = [&](auto &&self, int i, int j) -> void {
            if (j == i + 1) return;
            const int k = apex[i][j];
            plan.Fresh.push_back(positive({ring_v[i], ring_v[k], ring_v[j], u}));
            plan.Fresh.push_back(positive({ring_v[i], ring_v[k], ring_v[j], v}));
            self(self, i, k);
            self(self, k, j);
        };
        emit(emit, 0, n - 1);
        return plan;
    };
    // the 2-3 flip of the face of t opposite its vertex `at`
    const auto flip_face = [&](int32_t t, int at) {
        Plan plan;
        const auto &v = T[size_t(t)];
        const uint32_t a = v[size_t(at)], p = v[size_t(at + 1) & 3], q = v[size_t(at + 2) & 3], r = v[size_t(at + 3) & 3];
        const int32_t o = across(t, p, q, r);
        if (o < 0 || (walls && walls->count(Sorted(p, q, r)))) return plan;
        uint32_t b = UINT32_MAX;
        for (const uint32_t x : T[size_t(o)])
            if (x != p && x != q && x != r) b = x;
        // a and b on opposite sides of (p, q, r) already; the segment a b must pass through the triangle's interior
        const int s0 = exact::Orient3D(P[a], P[b], P[p], P[q]), s1 = exact::Orient3D(P[a], P[b], P[q], P[r]), s2 = exact::Orient3D(P[a], P[b], P[r], P[p]);
        if (s0 == 0 || s0 != s1 || s1 != s2) return plan;
        const double old_worst = std::min(shape(t), shape(o));
        const double fresh_worst = std::min({shape_of(a, b, p, q), shape_of(a, b, q, r), shape_of(a, b, r, p)});
        if (!(fresh_worst > old_worst * 1.001 + 1e-14)) return plan;
        plan.Worst = fresh_worst;
        plan.Old = {t, o};
        plan.Fresh = {positive({a, b, p, q}), positive({a, b, q, r}), positive({a, b, r, p})};
        return plan;
    };
    std::vector<int32_t> work;
    for (size_t t = 0; t < T.size(); ++t)
        if (shape(int32_t(t)) < target) work.push_back(int32_t(t));
    uint32_t exchanges = 0;
    for (int pass = 0; pass < 8 && !work.empty(); ++pass) {
        std::sort(work.begin(), work.end(), [&](int32_t x, int32_t y) { return shape(x) < shape(y); });
        std::vector<int32_t> next;
        for (const int32_t t : work) {
            if (!alive[size_t(t)]) continue;
            Plan best;
            const auto v = T[size_t(t)];
            for (int i = 0; i < 4; ++i) {
                for (int j = i + 1; j < 4; ++j) {
                    Plan plan = remove_edge(t, v[size_t(i)], v[size_t(j)]);
                    if (plan.Worst > best.Worst) best = std::move(plan);
                }
                Plan plan = flip_face(t, i);
                if (plan.Worst > best.Worst) best = std::move(plan);
            }
            if (best.Fresh.empty()) continue;
            for (const int32_t o : best.Old) link(o, false), alive[size_t(o)] = 0;
            for (const auto &f : best.Fresh) {
                T.push_back(f);
                alive.push_back(1);
                link(int32_t(T.size() - 1), true);
                if (shape(int32_t(T.size() - 1)) < target) next.push_back(int32_t(T.size() - 1));
            }
            ++exchanges;
        }
        work.swap(next);
    }
    std::vector<std::array<uint32_t, 4>> kept;
    kept.reserve(T.size());
    for (size_t t = 0; t < T.size(); ++t)
        if (alive[t]) kept.push_back(T[t]);
    T.swap(kept);
    return exchanges;
}

// Caps and other flat cells of a bare surface's fill.  A cap is a flat tetrahedron on TWO boundary faces -- the two triangles of a (nearly) planar surface quad joined into one cell, the
// rule on a UV sphere, whose latitude-longitude quads are planar trapezoids -- cannot be exchanged away: its only removable edge is
// the quad's other diagonal {b, d}, and the cells around that edge seldom re-tile without it.  With shape measures of 1e-7 ... 1e-9
// such cells inflate ||K|| by as much and stall the eigensolver (measured: a 96 x 48 UV sphere did not converge at all).  Here the
// cells around {b, d} are replaced by the cone over their outer faces from a new point m just inside the body under the middle of
// {b, d}: every cell around an edge is star-shaped about the edge's interior, so for m close enough the cone is a valid tiling --
// checked exactly -- and the cap becomes (a, b, c, m), (a, c, d, m) of height |m - quad|.  Both surface triangles stay.  The sliver
// repair and the smoothing that follow do the rest.  The same is done at an edge of a flat cell with one boundary face (a triangle
// of a pole fan and a fourth point of the first ring: the fan is all but planar) or with none.  Returns the number of points added.
static uint32_t BreakCaps(TetMesh &mesh, double flat) {
    auto &P = mesh.Points;
    auto &T = mesh.Tets;
    struct FaceHash {
        size_t operator()(const Tri &f) const { return (size_t(f[0]) * 0x9E3779B97F4A7C15ull) ^ (size_t(f[1]) * 0xC2B2AE3D27D4EB4Full) ^ (size_t(f[2]) * 0x165667B19E3779F9ull); }
    };
    const auto shape = [&](const std::array<uint32_t, 4> &t) {
        const dvec3 u = P[t[1]] - P[t[0]], v = P[t[2]] - P[t[0]], w = P[t[3]] - P[t[0]];
        const double vol6 = std::fabs(u.x * (v.y * w.z - v.z * w.y) - u.y * (v.x * w.z - v.z * w.x) + u.z * (v.x * w.y - v.y * w.x));
        double l2 = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = i + 1; j < 4; ++j) {
                const dvec3 e = P[t[size_t(i)]] - P[t[size_t(j)]];
                l2 += e.x * e.x + e.y * e.y + e.z * e.z;
            }
        const double lrms = std::sqrt(l2 / 6);
        return lrms > 0 ? 1.4142135623730951 * vol6 / (lrms * lrms * lrms) : 0.0;
    };
    uint32_t added = 0;
    static const bool dbg = std::getenv("MH_TET_DEBUG") != nullptr;
    size_t why[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int pass = 0; pass < 4; ++pass) {
        std::unordered_map<Tri, std::array<int32_t, 2>, FaceHash> faces;
        faces.reserve(T.size() * 2);
        for (size_t t = 0; t < T.size(); ++t)
            for (int i = 0; i < 4; ++i) {
                auto [it, fresh] = faces.try_emplace(Sorted(T[t][size_t(i + 1) & 3], T[t][size_t(i + 2) & 3], T[t][size_t(i + 3) & 3]), std::array<int32_t, 2>{int32_t(t), -1});
                if (!fresh) it->second[1] = int32_t(t);
            }
        const auto across = [&](int32_t t, uint32_t x, uint32_t y, uint32_t z) -> int32_t {
            const auto it = faces.find(Sorted(x, y, z));
            if (it == faces.end()) return -1;
            return it->second[0] == t ? it->second[1] : it->second[0];
        };
        const size_t n_before = T.size();
        std::vector<uint8_t> dead(n_before, 0);
        uint32_t broken = 0;
        for (size_t t = 0; t < n_before; ++t) {
            if (dead[t] || shape(T[t]) >= flat) continue;
            const auto tet = T[t];
            bool open[4];
            int n_open = 0;
            for (int i = 0; i < 4; ++i) n_open += (open[i] = across(int32_t(t), tet[size_t(i + 1) & 3], tet[size_t(i + 2) & 3], tet[size_t(i + 3) & 3]) < 0);
            ++why[0];
            if (n_open == 0) { ++why[1]; continue; } // (an interior sliver split at an edge only breeds more of them: those are the exchanges' business)
            // the edges of the cell that lie on none of its boundary faces: for a cap the quad's other diagonal; for a flat cell on one
            // boundary face (a pole fan's triangle and a fourth ring point) the three edges at its fourth vertex
            bool done = false;
            for (int i = 0; i < 4 && !done; ++i)
                for (int j = i + 1; j < 4 && !done; ++j) {
                    bool on_boundary = false;
                    for (int f = 0; f < 4; ++f) on_boundary = on_boundary || (open[f] && f != i && f != j); // face f (opposite vertex f) holds both i and j
                    if (on_boundary) continue;
                    const uint32_t b = tet[size_t(i)], d = tet[size_t(j)];
                    // the cells around {b, d}
                    std::vector<int32_t> ring;
                    std::vector<std::array<uint32_t, 2>> far; // per ring cell its two vertices other than b, d
                    uint32_t from = UINT32_MAX, to = UINT32_MAX;
                    for (const uint32_t x : tet)
                        if (x != b && x != d) (from == UINT32_MAX ? from : to) = x;
                    int32_t cell = int32_t(t);
                    bool closed = false, clean = true;
                    for (int guard = 0; guard < 32; ++guard) {
                        ring.push_back(cell);
                        far.push_back({from, to});
                        clean = clean && !dead[size_t(cell)];
                        const int32_t next = across(cell, b, d, to);
                        if (next < 0) break;
                        uint32_t beyond = UINT32_MAX;
                        for (const uint32_t x : T[size_t(next)])
                            if (x != b && x != d && x != to) beyond = x;
                        from = to, to = beyond, cell = next;
                        if (cell == int32_t(t)) { closed = true; break; }
                    }
                    if (!closed || !clean) { ++why[closed ? 3 : 2]; continue; }
                    // away from the flat cell: from the middle of {b, d} towards the ring's vertices that are not the cell's
                    const dvec3 mid = (P[b] + P[d]) * 0.5;
                    dvec3 inner{0, 0, 0};
                    double count = 0;
                    for (const auto &f : far)
                        for (const uint32_t x : f)
                            if (x != tet[0] && x != tet[1] && x != tet[2] && x != tet[3]) inner = inner + P[x], count += 1;
                    if (count == 0) continue;
                    const dvec3 towards = inner * (1.0 / count) - mid;
                    const dvec3 bd = P[d] - P[b];
                    const double edge2 = bd.x * bd.x + bd.y * bd.y + bd.z * bd.z;
                    // candidate positions: towards the ring's other vertices, and straight down from the flat cell (its normal, on their side)
                    dvec3 down{0, 0, 0};
                    {
                        const dvec3 ac = P[far[0][1]] - P[far[0][0]];
                        dvec3 n{ac.y * bd.z - ac.z * bd.y, ac.z * bd.x - ac.x * bd.z, ac.x * bd.y - ac.y * bd.x};
                        const double nl = std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
                        if (nl > 0) {
                            if (n.x * towards.x + n.y * towards.y + n.z * towards.z < 0) n = n * -1.0;
                            down = n * (std::sqrt(edge2) / nl);
                        }
                    }
                    const dvec3 offsets[7] = {towards * 0.3, towards * 0.15, down * 0.3, down * 0.18, towards * 0.5, towards * 0.07, down * 0.11};
                    // the position whose WORST new cell is best shaped; taken only if no new cell is flat itself (breaking a cap into
                    // cells below the threshold bred more flat cells per pass than it removed: 925 -> 1 191 over four passes)
                    double best_worst = flat;
                    dvec3 best_m{0, 0, 0};
                    for (const dvec3 &offset : offsets) {
                        if (offset.x * offset.x + offset.y * offset.y + offset.z * offset.z < 0.01 * edge2) continue; // (closer to the edge than a tenth of its length the new cells are flat themselves)
                        const dvec3 m = mid + offset;
                        bool valid = true;
                        double worst = 1e300;
                        for (size_t k = 0; k < ring.size() && valid; ++k)
                            for (const uint32_t pole : {b, d}) { // the face of the cell that holds `pole` but not the other end of the edge
                                const auto &cell = T[size_t(ring[k])];
                                const uint32_t gone = pole == b ? d : b;
                                dvec3 q[4];
                                for (int c = 0; c < 4; ++c) q[c] = cell[size_t(c)] == gone ? m : P[cell[size_t(c)]];
                                if (exact::Orient3D(q[0], q[1], q[2], q[3]) <= 0) { valid = false; break; }
                                const dvec3 u = q[1] - q[0], v = q[2] - q[0], w = q[3] - q[0];
                                const double vol6 = std::fabs(u.x * (v.y * w.z - v.z * w.y) - u.y * (v.x * w.z - v.z * w.x) + u.z * (v.x * w.y - v.y * w.x));
                                double l2 = 0;
                                for (int x = 0; x < 4; ++x)
                                    for (int y = x + 1; y < 4; ++y) {
                                        const dvec3 e = q[x] - q[y];
                                        l2 += e.x * e.x + e.y * e.y + e.z * e.z;
                                    }
                                const double lrms = std::sqrt(l2 / 6);
                                worst = std::min(worst, lrms > 0 ? 1.4142135623730951 * vol6 / (lrms * lrms * lrms) : 0.0);
                            }
                        if (valid && worst > best_worst) best_worst = worst, best_m = m;
                    }
                    if (best_worst > flat) {
                        const uint32_t id = uint32_t(P.size());
                        P.push_back(best_m);
                        for (const int32_t r : ring) {
                            for (const uint32_t gone : {d, b}) {
                                std::array<uint32_t, 4> piece = T[size_t(r)];
                                for (int c = 0; c < 4; ++c)
                                    if (piece[size_t(c)] == gone) piece[size_t(c)] = id;
                                T.push_back(piece);
                            }
                            dead[size_t(r)] = 1;
                        }
                        ++broken;
                        done = true;
                    } else ++why[4];
                }
            if (!done) ++why[5];
        }
        if (dbg) std::fprintf(stderr, "BreakCaps pass %d: flat %zu interior %zu ring-open %zu ring-unclean %zu no-position %zu unbroken %zu broken %u\n", pass, why[0], why[1], why[2], why[3], why[4], why[5], broken);
        for (auto &w : why) w = 0;
        if (!broken) break;
        added += broken;
        std::vector<std::array<uint32_t, 4>> kept;
        kept.reserve(T.size());
        for (size_t t = 0; t < T.size(); ++t)
            if (t >= n_before || !dead[t]) kept.push_back(T[t]);
        T.swap(kept);
    }
    return added;
}

// A shell of interior points under the surface.  A finely tessellated SMOOTH surface filled without interior points has flat cells
// everywhere -- wherever four of its vertices lie within a patch that is nearly planar at the mesh's own scale (two adjacent surface
// triangles at a dihedral angle of 176 degrees and a neighbour's vertex; a pole fan) -- and no exchange helps, because every vertex
// available sits on the same nearly flat patch.  The cure is a vertex UNDER the patch: one point per surface vertex, offset inwards along
// the vertex normal by 0.45 ... 0.75 of the mean length of the surface edges at it (the reference's own quality arm refines with hundreds of
// such points: tests/fixtures/TetCorpusSnapshot.txt, the q rows).  A point is inserted by splitting the cell that holds it (found
// among the cells around its surface vertex), so no face of the mesh is touched; the sliver repair and the smoothing that follow
// turn the long cells towards the interior into well-shaped ones.  Returns the number of points added.
static uint32_t AddInteriorShell(TetMesh &mesh, uint32_t n_surface_vertices) {
    auto &P = mesh.Points;
    auto &T = mesh.Tets;
    struct FaceHash {
        size_t operator()(const Tri &f) const { return (size_t(f[0]) * 0x9E3779B97F4A7C15ull) ^ (size_t(f[1]) * 0xC2B2AE3D27D4EB4Full) ^ (size_t(f[2]) * 0x165667B19E3779F9ull); }
    };
    std::unordered_map<Tri, int, FaceHash> face_count;
    face_count.reserve(T.size() * 2);
    for (const auto &t : T)
        for (int i = 0; i < 4; ++i) ++face_count[Sorted(t[size_t(i + 1) & 3], t[size_t(i + 2) & 3], t[size_t(i + 3) & 3])];
    // outward normals and edge lengths at the surface vertices, from the boundary faces (each seen from its one cell)
    std::vector<dvec3> normal(n_surface_vertices, dvec3{0, 0, 0});
    std::vector<double> length(n_surface_vertices, 0.0), edges(n_surface_vertices, 0.0);
    for (const auto &t : T)
        for (int i = 0; i < 4; ++i) {
            const uint32_t a = t[size_t(i + 1) & 3], b = t[size_t(i + 2) & 3], c = t[size_t(i + 3) & 3];
            if (face_count[Sorted(a, b, c)] != 1) continue;
            const dvec3 u = P[b] - P[a], v = P[c] - P[a];
            dvec3 n{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x}; // twice the area, either way round
            const dvec3 in = P[t[size_t(i)]] - P[a];
            if (n.x * in.x + n.y * in.y + n.z * in.z > 0) n = n * -1.0; // away from the cell's fourth vertex
            for (const uint32_t x : {a, b, c})
                if (x < n_surface_vertices) normal[x] = normal[x] + n;
            const uint32_t tri[3] = {a, b, c};
            for (int e = 0; e < 3; ++e) {
                const dvec3 d = P[tri[e]] - P[tri[(e + 1) % 3]];
                const double l = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
                for (const uint32_t x : {tri[e], tri[(e + 1) % 3]})
                    if (x < n_surface_vertices) length[x] += l, edges[x] += 1;
            }
        }
    // cells by face, kept up to date as cells are split: the point is located by a walk that starts at a cell around its vertex
    std::unordered_map<Tri, std::array<int32_t, 2>, FaceHash> cells_on;
    cells_on.reserve(T.size() * 2);
    const auto link = [&](int32_t t, bool add) {
        const auto &v = T[size_t(t)];
        for (int i = 0; i < 4; ++i) {
            const Tri key = Sorted(v[size_t(i + 1) & 3], v[size_t(i + 2) & 3], v[size_t(i + 3) & 3]);
            if (add) {
                auto [it, fresh] = cells_on.try_emplace(key, std::array<int32_t, 2>{t, -1});
                if (!fresh) (it->second[0] < 0 ? it->second[0] : it->second[1]) = t;
            } else {
                auto it = cells_on.find(key);
                if (it == cells_on.end()) continue;
                if (it->second[0] == t) it->second[0] = it->second[1];
                it->second[1] = -1;
                if (it->second[0] < 0) cells_on.erase(it);
            }
        }
    };
    for (size_t t = 0; t < T.size(); ++t) link(int32_t(t), true);
    std::vector<int32_t> cell_at(n_surface_vertices, -1); // one live cell per surface vertex
    for (size_t t = 0; t < T.size(); ++t)
        for (const uint32_t x : T[t])
            if (x < n_surface_vertices) cell_at[x] = int32_t(t);
    std::vector<uint8_t> dead(T.size(), 0);
    uint32_t added = 0;
    for (uint32_t v = 0; v < n_surface_vertices; ++v) {
        const double nl = std::sqrt(normal[v].x * normal[v].x + normal[v].y * normal[v].y + normal[v].z * normal[v].z);
        if (!(nl > 0) || !(edges[v] > 0) || cell_at[v] < 0) continue;
        const dvec3 inward = normal[v] * (-1.0 / nl);
        const double h = length[v] / edges[v];
        // (depths staggered from vertex to vertex -- a deterministic hash, 0.45 ... 0.75: points at ONE depth form a surface parallel
        // to the input, and the quads between an edge and its offset copy are planar: the flat cells would be back one layer down)
        const double stagger = 0.45 + 0.3 * double((v * 2654435761u) >> 8 & 0xffffu) / 65536.0;
        for (const double depth : {stagger, 0.6 * stagger, 1.4 * stagger}) {
            const dvec3 p = P[v] + inward * (depth * h);
            // visibility walk from a cell at v: through any face that has p strictly on its far side
            int32_t at = cell_at[v];
            bool found = false;
            for (int step = 0; step < 200 && at >= 0; ++step) {
                const auto cell = T[size_t(at)];
                int32_t next = -2;
                bool strictly = true;
                for (int k = 0; k < 4; ++k) {
                    const int i = (k + step) & 3; // (vary the order: a walk must not circle)
                    dvec3 q[4];
                    for (int j = 0; j < 4; ++j) q[j] = j == i ? p : P[cell[size_t(j)]];
                    const int side = exact::Orient3D(q[0], q[1], q[2], q[3]);
                    if (side < 0 && next == -2) {
                        const auto it = cells_on.find(Sorted(cell[size_t(i + 1) & 3], cell[size_t(i + 2) & 3], cell[size_t(i + 3) & 3]));
                        next = it == cells_on.end() ? -1 : (it->second[0] == at ? it->second[1] : it->second[0]);
                    }
                    strictly = strictly && side > 0;
                }
                if (next == -2) { found = strictly; break; } // inside (or on a face: another depth)
                at = next; // -1: p is outside the body
            }
            if (!found || at < 0) continue;
            const auto cell = T[size_t(at)];
            const uint32_t id = uint32_t(P.size());
            P.push_back(p);
            link(at, false);
            dead[size_t(at)] = 1;
            for (int i = 0; i < 4; ++i) {
                std::array<uint32_t, 4> piece = cell;
                piece[size_t(i)] = id;
                T.push_back(piece);
                dead.push_back(0);
                link(int32_t(T.size() - 1), true);
                for (const uint32_t x : piece)
                    if (x < n_surface_vertices) cell_at[x] = int32_t(T.size() - 1);
            }
            ++added;
            break;
        }
    }
    if (added) {
        std::vector<std::array<uint32_t, 4>> kept;
        kept.reserve(T.size());
        for (size_t t = 0; t < T.size(); ++t)
            if (!dead[t]) kept.push_back(T[t]);
        T.swap(kept);
    }
    return added;
}

// Point insertion into a finished fill: the constrained Bowyer-Watson step on exact predicates, shared by the quality arm (RefineQuality)
// and the flat-cell pass (BreakFlatCells).  The cavity is what the insphere test collects from the containing cell without crossing the
// boundary or a wall, shrunk until the point sees every face of its hull strictly from the inside, then fanned.  Every point inserted is
// strictly interior: no boundary face and no wall face is ever touched.
class FillEditor {
public:
    struct FaceHash {
        size_t operator()(const Tri &f) const { return (size_t(f[0]) * 0x9E3779B97F4A7C15ull) ^ (size_t(f[1]) * 0xC2B2AE3D27D4EB4Full) ^ (size_t(f[2]) * 0x165667B19E3779F9ull); }
    };
    FillEditor(TetMesh &mesh, const std::set<Tri> *walls) : P(mesh.Points), T(mesh.Tets), Walls(walls), Dead(mesh.Tets.size(), 0), Mark(mesh.Tets.size(), 0),
                                                           Neighbour(mesh.Tets.size(), std::array<int32_t, 4>{-1, -1, -1, -1}) {
        CellsOn.reserve(T.size() * 2);
        for (size_t t = 0; t < T.size(); ++t) Link(int32_t(t), true);
    }
    static Tri FaceOf(const std::array<uint32_t, 4> &v, int i) { return Sorted(v[size_t(i + 1) & 3], v[size_t(i + 2) & 3], v[size_t(i + 3) & 3]); }
    // the cell across face i of cell t, or -1 at the boundary, -2 behind a wall
    int32_t Across(int32_t t, int i) const { return Neighbour[size_t(t)][size_t(i)]; } // (kept by Link: the face map's answer, without the look-up -- half a fine ellipsoid's fill was that look-up)
    bool IsDead(size_t t) const { return Dead[t] != 0; }
    // the cell that holds p strictly inside, by a walk from `from` that never crosses the boundary or a wall; -1: the surface cuts p
    // off, or p lies on a face or an edge (not this point)
    int32_t Locate(const dvec3 &p, int32_t from, std::vector<int32_t> *path = nullptr) const { // path: the cells the walk went through, `from` first
        int32_t at = from;
        for (int step = 0; step < 400 && at >= 0; ++step) {
            if (path) path->push_back(at);
            const auto cell = T[size_t(at)];
            int32_t next = -3;
            bool strictly = true;
            for (int k = 0; k < 4 && next == -3; ++k) {
                const int i = (k + step) & 3;
                dvec3 q[4];
                for (int j = 0; j < 4; ++j) q[j] = j == i ? p : P[cell[size_t(j)]];
                const int side = exact::Orient3D(q[0], q[1], q[2], q[3]);
                if (side < 0) next = Across(at, i);
                strictly = strictly && side > 0;
            }
            if (next == -3) return strictly ? at : -1;
            at = next;
        }
        return -1;
    }
    // The cavity of p around the cell `at` that holds it: insphere from the containing cell, never across a constraint; at most 512 cells
    // (the long cells of a bare surface's Delaunay fill all hold an interior point in their circumspheres: the full cavity of an early
    // point is most of the mesh, and a local one serves as well -- the exchanges of the sliver repair afterwards do not need a Delaunay
    // mesh).  Returns the cells to be replaced; empty when the containing cell itself cannot see p through one of its hull faces.
    // `also`: a cell that joins the cavity whatever the insphere test says, if it shares a face with it (a FLAT cell's circumsphere is a
    // half-space whose side is decided by the last bits of its vertices: a point just under a cap may lie outside it) -- the star-shape
    // test below still has the last word on it.
    std::vector<int32_t> Cavity(const dvec3 &p, int32_t at, int32_t also = -1, double flat_below = 0, const std::vector<int32_t> *seeds = nullptr) {
        if (Mark.size() < T.size()) Mark.resize(T.size() + T.size() / 2, 0);
        ++Stamp;
        std::vector<int32_t> cavity{at};
        Mark[size_t(at)] = Stamp;
        if (seeds) // (the cells the walk from the flat cell to the point's cell went through: they tie the two together; the star-shape test decides)
            for (const int32_t c : *seeds)
                if (c >= 0 && Mark[size_t(c)] != Stamp) Mark[size_t(c)] = Stamp, cavity.push_back(c);
        for (size_t head = 0; head < cavity.size() && cavity.size() < 512; ++head)
            for (int i = 0; i < 4; ++i) {
                const int32_t o = Across(cavity[head], i);
                if (o < 0 || Mark[size_t(o)] == Stamp) continue;
                const auto &ov = T[size_t(o)];
                // (flat_below: with `also`, every FLAT neighbour of the cavity joins as well -- caps come stacked, and the one next to the cap in hand
                // decides whether the cap's inner face is a face of the cavity's hull, which the point would have to see from the wrong side)
                if (o == also || exact::InSphere(P[ov[0]], P[ov[1]], P[ov[2]], P[ov[3]], p) > 0 || (flat_below > 0 && FlatBelow(ov, flat_below))) Mark[size_t(o)] = Stamp, cavity.push_back(o);
            }
        // star-shaped hull: every hull face must see p strictly from the inside; a cell whose face does not leaves the cavity
        bool ok = true;
        for (bool changed = true; changed && ok;) {
            changed = false;
            for (const int32_t c : cavity) {
                if (!Inside(c)) continue;
                const auto &cv = T[size_t(c)];
                for (int i = 0; i < 4; ++i) {
                    if (Inside(Across(c, i))) continue; // interior face of the cavity
                    dvec3 q[4];
                    for (int j = 0; j < 4; ++j) q[j] = j == i ? p : P[cv[size_t(j)]];
                    if (exact::Orient3D(q[0], q[1], q[2], q[3]) > 0) continue;
                    if (Trace) std::fprintf(stderr, "        cell %d (%u %u %u %u) shape %.1e cannot show face %d to the point (%s)\n", c, cv[0], cv[1], cv[2], cv[3], FlatBelow(cv, 0) ? 0.0 : ShapeValue(cv), i, c == at ? "the containing cell" : "dropped");
                    if (c == at) ok = false;
                    else Mark[size_t(c)] = 0, changed = true;
                    break;
                }
                if (!ok) break;
            }
            // the cavity must stay connected to the containing cell
            if (ok && changed) {
                ++Stamp;
                std::vector<int32_t> reach{at};
                const uint32_t old = Stamp - 1;
                Mark[size_t(at)] = Stamp;
                for (size_t head = 0; head < reach.size(); ++head)
                    for (int i = 0; i < 4; ++i) {
                        const int32_t o = Across(reach[head], i);
                        if (o >= 0 && Mark[size_t(o)] == old) Mark[size_t(o)] = Stamp, reach.push_back(o);
                    }
                cavity.swap(reach);
            }
        }
        std::vector<int32_t> in;
        if (!ok) return in;
        for (const int32_t c : cavity)
            if (Inside(c)) in.push_back(c);
        return in;
    }
    // the cells that fanning the cavity (as returned by Cavity, still marked) from a new point would make; the point's id is P.size()
    std::vector<std::array<uint32_t, 4>> Fan(const std::vector<int32_t> &in) const {
        const uint32_t id = uint32_t(P.size());
        std::vector<std::array<uint32_t, 4>> fresh;
        for (const int32_t c : in) {
            const auto &cv = T[size_t(c)];
            for (int i = 0; i < 4; ++i) {
                if (Inside(Across(c, i))) continue;
                std::array<uint32_t, 4> piece = cv;
                piece[size_t(i)] = id; // (p on the side of the vertex it replaces: the orientation stays positive)
                fresh.push_back(piece);
            }
        }
        return fresh;
    }
    // replaces the cavity by the fan; returns the index of the first new cell
    size_t Commit(const dvec3 &p, const std::vector<int32_t> &in, const std::vector<std::array<uint32_t, 4>> &fresh) {
        P.push_back(p);
        for (const int32_t c : in) Link(c, false), Dead[size_t(c)] = 1;
        const size_t first = T.size();
        for (const auto &piece : fresh) {
            T.push_back(piece);
            Dead.push_back(0);
            Neighbour.push_back({-1, -1, -1, -1});
            Link(int32_t(T.size() - 1), true);
        }
        return first;
    }
    bool Trace{false};
    // the cells around the edge (u, v), walked from cell t; empty unless the ring closes without meeting the boundary or a wall (an edge of
    // the surface, or one in a wall, is nobody's to split) and within 64 cells
    std::vector<int32_t> Ring(int32_t t, uint32_t u, uint32_t v) const {
        std::vector<int32_t> ring;
        int32_t at = t;
        uint32_t from = UINT32_MAX; // the third vertex of the face the walk came in through
        for (int step = 0; step < 64; ++step) {
            ring.push_back(at);
            const auto &c = T[size_t(at)];
            int others[2], k = 0;
            for (int i = 0; i < 4; ++i)
                if (c[size_t(i)] != u && c[size_t(i)] != v) {
                    if (k == 2) return {};
                    others[k++] = i;
                }
            if (k != 2) return {};
            const int behind = from == UINT32_MAX || c[size_t(others[0])] == from ? others[0] : others[1];
            const int ahead = behind == others[0] ? others[1] : others[0];
            const int32_t next = Across(at, behind); // (the face opposite `behind` holds u, v and `ahead`)
            if (next < 0) return {};
            if (next == t) return ring;
            from = c[size_t(ahead)];
            at = next;
        }
        return {};
    }
    double ShapeValue(const std::array<uint32_t, 4> &t) const {
        double lo = 0, hi = 1;
        for (int k = 0; k < 40; ++k) (FlatBelow(t, 0.5 * (lo + hi)) ? hi : lo) = 0.5 * (lo + hi);
        return hi;
    }
    bool FlatBelow(const std::array<uint32_t, 4> &t, double bound) const {
        const dvec3 u = P[t[1]] - P[t[0]], v = P[t[2]] - P[t[0]], w = P[t[3]] - P[t[0]];
        const double vol6 = std::fabs(u.x * (v.y * w.z - v.z * w.y) - u.y * (v.x * w.z - v.z * w.x) + u.z * (v.x * w.y - v.y * w.x));
        double l2 = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = i + 1; j < 4; ++j) {
                const dvec3 e = P[t[size_t(i)]] - P[t[size_t(j)]];
                l2 += e.x * e.x + e.y * e.y + e.z * e.z;
            }
        const double lrms = std::sqrt(l2 / 6);
        return !(lrms > 0) || 1.4142135623730951 * vol6 / (lrms * lrms * lrms) < bound;
    }
    void Compact() {
        std::vector<std::array<uint32_t, 4>> kept;
        kept.reserve(T.size());
        for (size_t t = 0; t < T.size(); ++t)
            if (!Dead[t]) kept.push_back(T[t]);
        T.swap(kept);
    }
    std::vector<dvec3> &P;
    std::vector<std::array<uint32_t, 4>> &T;

private:
    bool Inside(int32_t c) const { return c >= 0 && Mark[size_t(c)] == Stamp; }
    // the index of the face `key` in cell o (the vertex of o that the face does not hold)
    int FaceIndexIn(int32_t o, const Tri &key) const {
        const auto &v = T[size_t(o)];
        for (int j = 0; j < 4; ++j)
            if (v[size_t(j)] != key[0] && v[size_t(j)] != key[1] && v[size_t(j)] != key[2]) return j;
        return 0;
    }
    void Link(int32_t t, bool add) {
        for (int i = 0; i < 4; ++i) {
            const Tri key = FaceOf(T[size_t(t)], i);
            const bool wall = Walls && Walls->count(key);
            if (add) {
                auto [it, fresh] = CellsOn.try_emplace(key, std::array<int32_t, 2>{t, -1});
                if (!fresh) (it->second[0] < 0 ? it->second[0] : it->second[1]) = t;
                const int32_t other = it->second[0] == t ? it->second[1] : it->second[0];
                Neighbour[size_t(t)][size_t(i)] = wall ? -2 : other;
                if (other >= 0) Neighbour[size_t(other)][size_t(FaceIndexIn(other, key))] = wall ? -2 : t;
            } else {
                auto it = CellsOn.find(key);
                if (it == CellsOn.end()) continue;
                if (it->second[0] == t) it->second[0] = it->second[1];
                it->second[1] = -1;
                if (it->second[0] >= 0) Neighbour[size_t(it->second[0])][size_t(FaceIndexIn(it->second[0], key))] = wall ? -2 : -1;
                if (it->second[0] < 0) CellsOn.erase(it);
            }
        }
    }
    const std::set<Tri> *Walls;
    std::unordered_map<Tri, std::array<int32_t, 2>, FaceHash> CellsOn;
    std::vector<uint8_t> Dead;
    std::vector<uint32_t> Mark; // cavity membership by stamp
    std::vector<std::array<int32_t, 4>> Neighbour; // per live cell and face: the cell across, -1 at the boundary, -2 behind a wall
    uint32_t Stamp{0};
};

static double ShapeOf(const std::vector<dvec3> &P, const std::array<uint32_t, 4> &t, const dvec3 *fresh = nullptr) {
    const auto at = [&](uint32_t v) -> const dvec3 & { return v < P.size() ? P[v] : *fresh; };
    const dvec3 u = at(t[1]) - at(t[0]), v = at(t[2]) - at(t[0]), w = at(t[3]) - at(t[0]);
    const double vol6 = std::fabs(u.x * (v.y * w.z - v.z * w.y) - u.y * (v.x * w.z - v.z * w.x) + u.z * (v.x * w.y - v.y * w.x));
    double l2 = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = i + 1; j < 4; ++j) {
            const dvec3 e = at(t[size_t(i)]) - at(t[size_t(j)]);
            l2 += e.x * e.x + e.y * e.y + e.z * e.z;
        }
    const double lrms = std::sqrt(l2 / 6);
    return lrms > 0 ? 1.4142135623730951 * vol6 / (lrms * lrms * lrms) : 0.0;
}

// The quality arm (the reference's Options::Quality / MaxVolume, src/mesh/Tetrahedralize.h:17-27): interior points are inserted until
// every tetrahedron has a circumradius-to-shortest-edge ratio of at most `ratio_bound` (2) and a volume of at most `max_volume` (0: no
// bound) -- "where the fixed surface allows": the point of a bad tetrahedron is its circumcentre, reached by a walk from the tetrahedron
// that may not cross the boundary or a wall; a circumcentre the surface cuts off is not inserted (the tetrahedron stays; one that is only
// too LARGE gets its centroid instead).  Insertion: FillEditor.  Returns the number of points added; `budget` bounds it.
static uint32_t RefineQuality(TetMesh &mesh, bool quality, double ratio_bound, double max_volume, const std::set<Tri> *walls, size_t budget) {
    FillEditor ed(mesh, walls);
    auto &P = mesh.Points;
    auto &T = mesh.Tets;
    std::vector<uint8_t> hopeless(T.size(), 0);
    struct Measure { double ratio, volume; dvec3 centre; bool ok; };
    const auto measure = [&](const std::array<uint32_t, 4> &v) {
        const dvec3 &a = P[v[0]];
        const dvec3 u = P[v[1]] - a, w = P[v[2]] - a, x = P[v[3]] - a;
        const auto cross = [](const dvec3 &p, const dvec3 &q) { return dvec3{p.y * q.z - p.z * q.y, p.z * q.x - p.x * q.z, p.x * q.y - p.y * q.x}; };
        const auto dot = [](const dvec3 &p, const dvec3 &q) { return p.x * q.x + p.y * q.y + p.z * q.z; };
        const double det = dot(u, cross(w, x));
        Measure m{0, std::fabs(det) / 6, a, false};
        if (!(std::fabs(det) > 0)) return m;
        const dvec3 num = cross(w, x) * dot(u, u) + cross(x, u) * dot(w, w) + cross(u, w) * dot(x, x);
        const dvec3 off = num * (0.5 / det);
        m.centre = a + off;
        double shortest = 1e300;
        for (int i = 0; i < 4; ++i)
            for (int j = i + 1; j < 4; ++j) {
                const dvec3 e = P[v[size_t(i)]] - P[v[size_t(j)]];
                shortest = std::min(shortest, dot(e, e));
            }
        m.ratio = std::sqrt(dot(off, off) / shortest);
        m.ok = std::isfinite(m.ratio);
        return m;
    };
    const auto is_bad = [&](const Measure &m) { return (quality && m.ratio > ratio_bound) || (max_volume > 0 && m.volume > max_volume); };
    uint32_t added = 0;
    for (int pass = 0; pass < 64 && added < budget; ++pass) {
        std::vector<std::pair<double, int32_t>> work; // worst first: the volume excess counts like a ratio
        for (size_t t = 0; t < T.size(); ++t) {
            if (ed.IsDead(t) || hopeless[t]) continue;
            const Measure m = measure(T[t]);
            if (!m.ok || !is_bad(m)) continue;
            work.emplace_back(std::max(m.ratio / ratio_bound, max_volume > 0 ? std::cbrt(m.volume / max_volume) : 0.0), int32_t(t));
        }
        if (work.empty()) break;
        std::sort(work.begin(), work.end(), [](const auto &l, const auto &r) { return l.first != r.first ? l.first > r.first : l.second < r.second; });
        uint32_t added_this_pass = 0;
        for (const auto &[badness, t0] : work) {
            if (added >= budget) break;
            if (ed.IsDead(size_t(t0))) continue;
            const Measure m = measure(T[size_t(t0)]);
            if (!m.ok || !is_bad(m)) continue;
            const bool too_large = max_volume > 0 && m.volume > max_volume;
            bool inserted = false;
            if (too_large && !quality) {
                // the volume bound alone (the front end's last word): of the circumcentre, the centroid and the midpoint of the longest edge's
                // opposite pair -- whichever fan has the best worst cell; a flat cell made here would stay
                const auto &v = T[size_t(t0)];
                std::vector<dvec3> tries{m.centre, (P[v[0]] + P[v[1]] + P[v[2]] + P[v[3]]) * 0.25};
                tries.push_back((tries[0] + tries[1]) * 0.5);
                double best = -1;
                dvec3 best_p{0, 0, 0};
                std::vector<int32_t> best_in;
                std::vector<std::array<uint32_t, 4>> best_fresh;
                for (const dvec3 &p : tries) {
                    const int32_t at = ed.Locate(p, t0);
                    if (at < 0) continue;
                    const std::vector<int32_t> in = ed.Cavity(p, at);
                    if (in.empty() || std::find(in.begin(), in.end(), t0) == in.end()) continue;
                    const auto fresh = ed.Fan(in);
                    double worst = 1e300;
                    for (const auto &piece : fresh) worst = std::min(worst, ShapeOf(P, piece, &p));
                    if (worst > best) best = worst, best_p = p, best_in = in, best_fresh = fresh;
                }
                if (best > 0) {
                    ed.Commit(best_p, best_in, best_fresh);
                    hopeless.resize(T.size(), 0);
                    ++added, ++added_this_pass;
                    inserted = true;
                }
            }
            for (int attempt = 0; attempt < 2 && !inserted; ++attempt) {
                // attempt 0: the circumcentre; attempt 1 (a tetrahedron that is too large only): its centroid, which it always contains
                if (attempt == 1 && !too_large) break;
                dvec3 p = m.centre;
                if (attempt == 1) {
                    const auto &v = T[size_t(t0)];
                    p = (P[v[0]] + P[v[1]] + P[v[2]] + P[v[3]]) * 0.25;
                }
                const int32_t at = ed.Locate(p, t0);
                if (at < 0) continue; // the surface cuts p off
                const std::vector<int32_t> in = ed.Cavity(p, at);
                if (in.empty()) continue;
                // not on top of a vertex of the cavity (a point that close makes an edge shorter than the ones that called for it)
                {
                    double nearest = 1e300, shortest = 1e300;
                    for (const int32_t c : in)
                        for (const uint32_t v : T[size_t(c)]) {
                            const dvec3 e = P[v] - p;
                            nearest = std::min(nearest, e.x * e.x + e.y * e.y + e.z * e.z);
                        }
                    const auto &v0 = T[size_t(t0)];
                    for (int i = 0; i < 4; ++i)
                        for (int j = i + 1; j < 4; ++j) {
                            const dvec3 e = P[v0[size_t(i)]] - P[v0[size_t(j)]];
                            shortest = std::min(shortest, e.x * e.x + e.y * e.y + e.z * e.z);
                        }
                    if (attempt == 0 && nearest < shortest) continue; // no new edge shorter than the shortest edge of the tetrahedron that called for the point: the smallest edge length of the mesh never falls, so the refinement ends (the centroid of a tetrahedron that is too LARGE goes in regardless)
                }
                const auto fresh = ed.Fan(in);
                ed.Commit(p, in, fresh);
                hopeless.resize(T.size(), 0);
                ++added, ++added_this_pass;
                inserted = true;
            }
            if (!inserted) hopeless[size_t(t0)] = 1; // (the surface cuts its circumcentre off: it stays as it is)
        }
        if (!added_this_pass) break;
    }
    if (added) ed.Compact();
    return added;
}

// Flat cells that the exchanges, the caps' apexes and the shell have left (the reference repairs slivers and optimises vertices "either
// way", src/mesh/Tetrahedralize.h:19-20; a cell flat to 1e-9 inflates ||K|| by as much and no iterative eigensolver converges on it): the
// quality arm run LOCALLY.  Every cell with a shape measure below `floor` gets a point beside it -- under its boundary face(s) when it has
// any (a cap on a surface quad, a pole fan's triangle with a ring point), otherwise off its own plane on either side -- at a distance of
// the order of its edges, inserted by the constrained Bowyer-Watson step (FillEditor): the flat cell's circumsphere is huge, so it and its
// flat neighbours fall into the cavity together and are replaced by cells of the new point's height.  A position is taken only when the
// flat cell goes and the WORST cell of the fan is better than the worst cell it replaces; the best of the candidate positions wins.
// Points are strictly interior; the boundary and the walls are untouched.  Returns the number of points added.
static uint32_t BreakFlatCells(TetMesh &mesh, double floor, const std::set<Tri> *walls, size_t budget, uint32_t n_input, uint32_t *moved_out = nullptr) {
    FillEditor ed(mesh, walls);
    auto &P = mesh.Points;
    auto &T = mesh.Tets;
    const auto cross = [](const dvec3 &p, const dvec3 &q) { return dvec3{p.y * q.z - p.z * q.y, p.z * q.x - p.x * q.z, p.x * q.y - p.y * q.x}; };
    const auto dot = [](const dvec3 &p, const dvec3 &q) { return p.x * q.x + p.y * q.y + p.z * q.z; };
    uint32_t added = 0, moved = 0;
    // A flat cell that holds an ADDED interior point is flat because that point sits almost in the plane of the other three (a recovery point
    // lifted only a hair off the surface): the cure is to move the point, not to add another beside it.  Candidates along the cell's normal
    // (both ways) and towards the centroid of the point's neighbours; taken where every tetrahedron at the point stays positively oriented
    // (exact) and the worst of them improves; the best candidate wins.  Input vertices, points on the boundary and on walls never move.
    std::unordered_set<uint64_t> centre_tried;
    // the cells at every ADDED point (relocate's stars; a scan of all cells per point was a fifth of a fine ellipsoid's fill): built per pass, kept up by note_new_cells
    std::vector<std::vector<int32_t>> incident;
    const auto index_cells = [&] {
        incident.assign(P.size(), {});
        for (size_t t = 0; t < T.size(); ++t)
            if (!ed.IsDead(t))
                for (const uint32_t v : T[t])
                    if (v >= n_input) incident[v].push_back(int32_t(t));
    };
    const auto note_new_cells = [&](size_t first) {
        incident.resize(P.size());
        for (size_t t = first; t < T.size(); ++t)
            for (const uint32_t v : T[t])
                if (v >= n_input) incident[v].push_back(int32_t(t));
    };
    const auto relocate = [&](int32_t t0) -> bool {
        const auto cell = T[size_t(t0)];
        for (int vi = 0; vi < 4; ++vi) {
            const uint32_t v = cell[size_t(vi)];
            if (v < n_input) continue;
            std::vector<int32_t> star;
            bool fixed = false;
            dvec3 centre{0, 0, 0};
            double weight = 0;
            for (const int32_t listed : incident[v]) { // (ascending, as a scan of all cells would meet them)
                if (fixed) break;
                const size_t t = size_t(listed);
                if (ed.IsDead(t)) continue;
                const auto &c = T[t];
                int at = -1;
                for (int i = 0; i < 4; ++i)
                    if (c[size_t(i)] == v) at = i;
                if (at < 0) continue;
                star.push_back(int32_t(t));
                for (int i = 0; i < 4; ++i) {
                    if (i == at) { // the faces that hold v are the other three: none may be a boundary or wall face
                        continue;
                    }
                    if (ed.Across(int32_t(t), i) < 0) fixed = true; // face i is opposite vertex i, i.e. it holds v when i != at
                    centre = centre + P[c[size_t(i)]];
                    weight += 1;
                }
            }
            if (std::getenv("MH_TET_DEBUG2")) std::fprintf(stderr, "  relocate: cell %d vertex %u: star of %zu, %s\n", t0, v, star.size(), fixed ? "on the boundary or a wall" : "free");
            if (fixed || star.empty()) continue;
            centre = centre * (1.0 / weight);
            const dvec3 &a = P[cell[size_t(vi + 1) & 3]], &b = P[cell[size_t(vi + 2) & 3]], &c3 = P[cell[size_t(vi + 3) & 3]];
            dvec3 n = cross(b - a, c3 - a);
            const double nl = std::sqrt(dot(n, n));
            if (!(nl > 0)) continue;
            n = n * (1.0 / nl);
            const double scale = std::sqrt(nl); // ~ the opposite face's edge length
            const auto worst_at = [&](const dvec3 &x, bool &valid) {
                double worst = 1e300;
                valid = true;
                for (const int32_t t : star) {
                    dvec3 q[4];
                    for (int i = 0; i < 4; ++i) q[i] = T[size_t(t)][size_t(i)] == v ? x : P[T[size_t(t)][size_t(i)]];
                    if (exact::Orient3D(q[0], q[1], q[2], q[3]) <= 0) { valid = false; return 0.0; }
                    const dvec3 u = q[1] - q[0], w = q[2] - q[0], z = q[3] - q[0];
                    const double vol6 = std::fabs(dot(u, cross(w, z)));
                    double l2 = 0;
                    for (int i = 0; i < 4; ++i)
                        for (int j = i + 1; j < 4; ++j) l2 += dot(q[i] - q[j], q[i] - q[j]);
                    const double lrms = std::sqrt(l2 / 6);
                    worst = std::min(worst, lrms > 0 ? 1.4142135623730951 * vol6 / (lrms * lrms * lrms) : 0.0);
                }
                return worst;
            };
            bool ok = false;
            double best = worst_at(P[v], ok);
            if (!ok) continue;
            dvec3 best_x = P[v];
            bool found = false;
            std::vector<dvec3> candidates;
            for (const double step : {0.15, 0.3, 0.5, 0.8}) {
                candidates.push_back(P[v] + n * (step * scale));
                candidates.push_back(P[v] - n * (step * scale));
            }
            for (const double step : {0.25, 0.5, 0.75, 1.0}) candidates.push_back(P[v] + (centre - P[v]) * step);
            for (const dvec3 &x : candidates) {
                bool valid = false;
                const double w = worst_at(x, valid);
                if (std::getenv("MH_TET_DEBUG2")) std::fprintf(stderr, "      position: %s, worst %.1e (now %.1e)\n", valid ? "valid" : "a cell inverts", w, best);
                if (valid && w > best * 1.5) best = w, best_x = x, found = true;
            }
            // (the linear programme is the costly step -- C(n, 4) small solves -- and the passes come back to the same cells: a star that has not changed
            // since it was last tried gives the same answer)
            uint64_t signature = 0xcbf29ce484222325ull ^ v;
            if (!found)
                for (const int32_t t : star)
                    for (const uint32_t x : T[size_t(t)]) {
                        uint64_t bits[3];
                        std::memcpy(bits, &P[x], sizeof bits);
                        for (const uint64_t word : {uint64_t(x), bits[0], bits[1], bits[2]}) signature = (signature ^ word) * 0x100000001b3ull;
                    }
            if (!found && star.size() <= 512 && centre_tried.insert(signature).second) {
                // No sampled position keeps every cell at the point positive (a point a hair off a surface EDGE has a star of forty cells, some
                // of them thin: the steps above overshoot them).  The positions that do are a polyhedron -- each cell's volume is affine in
                // the position -- and the point farthest inside it is ChebyshevCentre's small linear programme.
                std::vector<HalfSpace> planes;
                double reach = 0;
                for (const int32_t t : star) {
                    const auto &tet = T[size_t(t)];
                    const auto g = [&](const dvec3 &x) {
                        dvec3 q[4];
                        for (int i = 0; i < 4; ++i) q[i] = tet[size_t(i)] == v ? x : P[tet[size_t(i)]];
                        return dot(q[1] - q[0], cross(q[2] - q[0], q[3] - q[0]));
                    };
                    const double g0 = g(P[v]);
                    const dvec3 grad{(g(P[v] + dvec3{scale, 0, 0}) - g0) / scale, (g(P[v] + dvec3{0, scale, 0}) - g0) / scale, (g(P[v] + dvec3{0, 0, scale}) - g0) / scale};
                    const double len = std::sqrt(dot(grad, grad));
                    if (!(len > 0)) continue;
                    const dvec3 nrm = grad * (1.0 / len);
                    planes.push_back({nrm, g0 / len - dot(nrm, P[v])});
                    for (const uint32_t x : tet) reach = std::max(reach, std::sqrt(dot(P[x] - P[v], P[x] - P[v])));
                }
                dvec3 x = P[v];
                if (ChebyshevCentre(planes, P[v], reach, x) > 0) {
                    bool valid = false;
                    const double w = worst_at(x, valid);
                    if (std::getenv("MH_TET_DEBUG2")) std::fprintf(stderr, "      the star's centre: %s, worst %.1e (now %.1e)\n", valid ? "valid" : "a cell inverts", w, best);
                    if (valid && w > best * 1.5) best = w, best_x = x, found = true;
                }
            }
            if (found) {
                P[v] = best_x;
                return true;
            }
        }
        return false;
    };
    for (int pass = 0; pass < 6 && added < budget; ++pass) {
        std::vector<std::pair<double, int32_t>> work; // flattest first
        for (size_t t = 0; t < T.size(); ++t)
            if (!ed.IsDead(t)) {
                const double q = ShapeOf(P, T[t]);
                if (q < floor) work.emplace_back(q, int32_t(t));
            }
        if (work.empty()) break;
        std::sort(work.begin(), work.end());
        index_cells();
        uint32_t added_this_pass = 0;
        for (const auto &[q0, t0] : work) {
            if (added >= budget) break;
            if (ed.IsDead(size_t(t0))) continue;
            if (ShapeOf(P, T[size_t(t0)]) >= floor) continue; // (a neighbour's point has moved meanwhile)
            if (relocate(t0)) { ++moved; ++added_this_pass; continue; }
            const auto cell = T[size_t(t0)];
            // candidate positions
            std::vector<dvec3> candidates;
            double l2 = 0;
            for (int i = 0; i < 4; ++i)
                for (int j = i + 1; j < 4; ++j) {
                    const dvec3 e = P[cell[size_t(i)]] - P[cell[size_t(j)]];
                    l2 += dot(e, e);
                }
            const double lrms = std::sqrt(l2 / 6);
            const dvec3 centroid = (P[cell[0]] + P[cell[1]] + P[cell[2]] + P[cell[3]]) * 0.25;
            // the cell's own plane: the normal of its largest face
            dvec3 normal{0, 0, 0};
            double largest = -1;
            bool open[4];
            int n_open = 0;
            for (int i = 0; i < 4; ++i) {
                const dvec3 &a = P[cell[size_t(i + 1) & 3]], &b = P[cell[size_t(i + 2) & 3]], &c = P[cell[size_t(i + 3) & 3]];
                const dvec3 n = cross(b - a, c - a);
                const double area2 = dot(n, n);
                if (area2 > largest) largest = area2, normal = n;
                n_open += (open[i] = ed.Across(t0, i) == -1);
            }
            if (!(largest > 0)) continue;
            normal = normal * (1.0 / std::sqrt(largest));
            if (n_open > 0) {
                // inwards = the mean of the boundary faces' normals on the cell's side.  The cell is flat, so which side its fourth vertex is
                // on is asked of the exact predicate, not of a rounded dot product.
                dvec3 inward{0, 0, 0};
                for (int i = 0; i < 4; ++i) {
                    if (!open[i]) continue;
                    const dvec3 &a = P[cell[size_t(i + 1) & 3]], &b = P[cell[size_t(i + 2) & 3]], &c = P[cell[size_t(i + 3) & 3]];
                    dvec3 n = cross(b - a, c - a);
                    const double nl = std::sqrt(dot(n, n));
                    if (!(nl > 0)) continue;
                    n = n * (1.0 / nl);
                    const dvec3 probe = (a + b + c) * (1.0 / 3) + n * lrms;
                    if (exact::Orient3D(a, b, c, probe) != exact::Orient3D(a, b, c, P[cell[size_t(i)]])) n = n * -1.0;
                    inward = inward + n;
                }
                const double il = std::sqrt(dot(inward, inward));
                if (il > 0) {
                    inward = inward * (1.0 / il);
                    for (const double depth : {0.45, 0.3, 0.65, 0.2, 0.9, 0.12}) candidates.push_back(centroid + inward * (depth * lrms));
                }
            }
            for (const double depth : {0.45, 0.3, 0.65, 0.2, 0.9}) {
                candidates.push_back(centroid + normal * (depth * lrms));
                candidates.push_back(centroid - normal * (depth * lrms));
            }
            // the worst cell around: what the insertion must beat
            double best_gain = 0;
            dvec3 best_p{0, 0, 0};
            std::vector<int32_t> best_in;
            std::vector<std::array<uint32_t, 4>> best_fresh;
            static const bool dbg2 = std::getenv("MH_TET_DEBUG2") != nullptr;
            static const double dbg_below = std::getenv("MH_TET_DEBUG_BELOW") ? std::atof(std::getenv("MH_TET_DEBUG_BELOW")) : 1e-6; // (which cells the trace follows)
            for (const dvec3 &p : candidates) {
                std::vector<int32_t> path;
                const int32_t at = ed.Locate(p, t0, &path);
                if (dbg2 && q0 < dbg_below) std::fprintf(stderr, "  cell %d (%u %u %u %u) shape %.1e n_open %d candidate (%.5f %.5f %.5f): located in %d\n", t0, cell[0], cell[1], cell[2], cell[3], q0, n_open, p.x, p.y, p.z, at);
                if (at < 0) continue;
                ed.Trace = dbg2 && q0 < dbg_below && std::getenv("MH_TET_DEBUG3");
                std::vector<int32_t> in = ed.Cavity(p, at, t0);
                if (in.empty() || std::find(in.begin(), in.end(), t0) == in.end()) { // once more with the cap's flat neighbours taken in
                    in = ed.Cavity(p, at, t0, 0.05);
                    if ((in.empty() || std::find(in.begin(), in.end(), t0) == in.end()) && path.size() <= 12) in = ed.Cavity(p, at, t0, 0.05, &path); // ... and with the walk's cells
                    if (in.empty() || std::find(in.begin(), in.end(), t0) == in.end()) {
                        if (dbg2 && q0 < dbg_below) std::fprintf(stderr, "      cavity %s\n", in.empty() ? "empty (the containing cell does not see the point through its hull)" : "without the flat cell");
                        continue;
                    }
                }
                double worst_old = 1e300, worst_new = 1e300;
                for (const int32_t c : in) worst_old = std::min(worst_old, ShapeOf(P, T[size_t(c)]));
                const auto fresh = ed.Fan(in);
                for (const auto &piece : fresh) worst_new = std::min(worst_new, ShapeOf(P, piece, &p));
                {
                    // A cavity that took cells in by adjacency may SWALLOW a vertex (every tetrahedron at an interior point inside it): the fan would
                    // leave that point belonging to no tetrahedron (the round-6 soak: "mesh point(s) belong to no tetrahedron").  Not this position.
                    std::vector<uint32_t> before, after;
                    for (const int32_t c : in)
                        for (const uint32_t v : T[size_t(c)]) before.push_back(v);
                    for (const auto &piece : fresh)
                        for (const uint32_t v : piece) after.push_back(v);
                    std::sort(before.begin(), before.end());
                    before.erase(std::unique(before.begin(), before.end()), before.end());
                    std::sort(after.begin(), after.end());
                    after.erase(std::unique(after.begin(), after.end()), after.end());
                    bool swallowed = false;
                    for (const uint32_t v : before) swallowed = swallowed || !std::binary_search(after.begin(), after.end(), v);
                    if (swallowed) continue;
                }
                if (dbg2 && q0 < dbg_below) std::fprintf(stderr, "      cavity of %zu cells, worst old %.1e, worst new %.1e\n", in.size(), worst_old, worst_new);
                if (!(worst_new > worst_old)) continue;
                const double gain = worst_new;
                if (gain > best_gain) best_gain = gain, best_p = p, best_in = in, best_fresh = fresh;
                if (worst_new >= 0.05) break; // (good enough: the repair and the smoothing follow)
            }
            if (best_gain > 0) {
                // (Cavity's marks belong to the LAST candidate: commit needs only the lists)
                note_new_cells(ed.Commit(best_p, best_in, best_fresh));
                ++added, ++added_this_pass;
                continue;
            }
            // No position's cavity keeps the flat cell (caps come in fans round a surface vertex: the cavity unravels cell by cell under the
            // star-shape test).  Then one of the cell's INTERIOR edges is split instead, at a point pushed off the cell's plane: the cells round
            // the edge are halved, nothing else changes, and the flat cell's two halves get the point's height.  An edge qualifies when its
            // ring of cells closes inside the body; a position, when every half is positively oriented (exact) and the worst half is better
            // than the worst cell of the ring.
            {
                std::vector<int32_t> best_ring;
                uint32_t best_u = 0, best_v = 0;
                const uint32_t id = uint32_t(P.size());
                for (int i = 0; i < 4; ++i)
                    for (int j = i + 1; j < 4; ++j) {
                        const uint32_t u = cell[size_t(i)], v = cell[size_t(j)];
                        const std::vector<int32_t> ring = ed.Ring(t0, u, v);
                        if (dbg2 && q0 < dbg_below && ring.empty()) std::fprintf(stderr, "      edge (%u %u): its ring of cells is open (a surface or wall edge) or longer than 64\n", u, v);
                        if (ring.empty()) continue;
                        double worst_old = 1e300;
                        for (const int32_t c : ring) worst_old = std::min(worst_old, ShapeOf(P, T[size_t(c)]));
                        const dvec3 mid = (P[u] + P[v]) * 0.5;
                        // positions: from the edge's midpoint towards the middle of the ring's other vertices (the kernel of the ring is where
                        // the point may go: the cells round a surface vertex are thin, an offset of the cell's own size leaves them), then the
                        // offsets of the insertion above
                        std::vector<dvec3> positions;
                        {
                            dvec3 middle{0, 0, 0};
                            double count = 0;
                            for (const int32_t c : ring)
                                for (const uint32_t x : T[size_t(c)])
                                    if (x != u && x != v) middle = middle + P[x], count += 1;
                            middle = middle * (1.0 / count);
                            for (const double step : {0.35, 0.2, 0.5, 0.1, 0.7, 0.05}) positions.push_back(mid + (middle - mid) * step);
                            for (const dvec3 &full : candidates) positions.push_back(mid + (full - centroid));
                            for (const dvec3 &full : candidates) positions.push_back(mid + (full - centroid) * 0.2);
                            // ... and the point farthest inside the region where every half is positive (each half's volume is affine in the
                            // position: ChebyshevCentre) -- the kernel of a ring that hugs the surface is a sliver the samples miss
                            std::vector<HalfSpace> planes;
                            const double span = std::sqrt(dot(P[u] - P[v], P[u] - P[v]));
                            for (const int32_t c : ring)
                                for (const uint32_t gone : {u, v}) {
                                    const auto tet = T[size_t(c)];
                                    const auto g = [&](const dvec3 &x) {
                                        dvec3 q[4];
                                        for (int k = 0; k < 4; ++k) q[k] = tet[size_t(k)] == gone ? x : P[tet[size_t(k)]];
                                        return dot(q[1] - q[0], cross(q[2] - q[0], q[3] - q[0]));
                                    };
                                    const double g0 = g(mid);
                                    const dvec3 grad{(g(mid + dvec3{span, 0, 0}) - g0) / span, (g(mid + dvec3{0, span, 0}) - g0) / span, (g(mid + dvec3{0, 0, span}) - g0) / span};
                                    const double len = std::sqrt(dot(grad, grad));
                                    if (!(len > 0)) continue;
                                    const dvec3 nrm = grad * (1.0 / len);
                                    planes.push_back({nrm, g0 / len - dot(nrm, mid)});
                                }
                            dvec3 deepest = mid;
                            if (ChebyshevCentre(planes, mid, span, deepest) > 0) positions.push_back(deepest);
                        }
                        for (const dvec3 &p : positions) {
                            double worst_new = 1e300;
                            bool valid = true;
                            for (size_t r = 0; r < ring.size() && valid; ++r)
                                for (const uint32_t gone : {u, v}) {
                                    std::array<uint32_t, 4> piece = T[size_t(ring[r])];
                                    dvec3 q[4];
                                    for (int k = 0; k < 4; ++k) {
                                        if (piece[size_t(k)] == gone) piece[size_t(k)] = id;
                                        q[k] = piece[size_t(k)] == id ? p : P[piece[size_t(k)]];
                                    }
                                    if (exact::Orient3D(q[0], q[1], q[2], q[3]) <= 0) { valid = false; break; }
                                    worst_new = std::min(worst_new, ShapeOf(P, piece, &p));
                                }
                            if (dbg2 && q0 < dbg_below) std::fprintf(stderr, "      edge (%u %u) ring of %zu, offset position: %s, worst old %.1e, worst new %.1e\n", u, v, ring.size(), valid ? "valid" : "a half is inverted", worst_old, valid ? worst_new : 0.0);
                            if (!valid || !(worst_new > worst_old) || !(worst_new > best_gain)) continue;
                            best_gain = worst_new, best_p = p, best_ring = ring, best_u = u, best_v = v;
                        }
                    }
                if (best_gain > 0) {
                    std::vector<std::array<uint32_t, 4>> halves;
                    for (const int32_t c : best_ring)
                        for (const uint32_t gone : {best_u, best_v}) {
                            std::array<uint32_t, 4> piece = T[size_t(c)];
                            for (uint32_t &x : piece)
                                if (x == gone) x = id;
                            halves.push_back(piece);
                        }
                    note_new_cells(ed.Commit(best_p, best_ring, halves));
                    ++added, ++added_this_pass;
                }
            }
        }
        if (std::getenv("MH_TET_DEBUG")) std::fprintf(stderr, "BreakFlatCells pass %d: %zu below %.0e (flattest %.1e), %u points added or moved (%u moved so far)\n", pass, work.size(), floor, work.front().first, added_this_pass, moved);
        if (!added_this_pass) break;
    }
    if (added) ed.Compact();
    if (moved_out) *moved_out = moved;
    return added;
}

// Vertex smoothing of the ADDED points (the reference's "vertex optimisation" runs with its sliver repair: Tetrahedralize.h:20):
// an interior point moves towards the centroid of the vertices it is connected to, as far (1, 1/2, 1/4 of the way) as raises the
// worst shape measure of its tetrahedra while every one of them stays positively oriented (exact).  Input vertices never move;
// neither does an added point that is still on the boundary.  Returns the number of points moved.
static uint32_t SmoothAddedPoints(TetMesh &mesh, uint32_t n_input, const std::set<Tri> *walls = nullptr) {
    auto &P = mesh.Points;
    const auto &T = mesh.Tets;
    if (P.size() <= n_input) return 0;
    std::vector<std::vector<uint32_t>> star(P.size() - n_input);
    std::map<Tri, int> face_count;
    for (uint32_t t = 0; t < T.size(); ++t)
        for (int i = 0; i < 4; ++i) {
            if (T[t][size_t(i)] >= n_input) star[T[t][size_t(i)] - n_input].push_back(t);
        }
    // added points on the boundary: a face with one tet only
    std::vector<uint8_t> on_boundary(P.size() - n_input, 0);
    {
        std::unordered_map<uint64_t, uint32_t> count; // (only faces that hold an added point matter)
        const auto key = [](uint32_t a, uint32_t b, uint32_t c) {
            if (a > b) std::swap(a, b);
            if (b > c) std::swap(b, c);
            if (a > b) std::swap(a, b);
            return (uint64_t(a) << 42) | (uint64_t(b) << 21) | uint64_t(c);
        };
        if (P.size() >= (size_t(1) << 21)) return 0;
        for (const auto &v : T)
            for (int i = 0; i < 4; ++i) {
                const uint32_t a = v[size_t(i + 1) & 3], b = v[size_t(i + 2) & 3], c = v[size_t(i + 3) & 3];
                if (a >= n_input || b >= n_input || c >= n_input) ++count[key(a, b, c)];
            }
        for (const auto &v : T)
            for (int i = 0; i < 4; ++i) {
                const uint32_t f[3] = {v[size_t(i + 1) & 3], v[size_t(i + 2) & 3], v[size_t(i + 3) & 3]};
                if ((f[0] >= n_input || f[1] >= n_input || f[2] >= n_input) && count[key(f[0], f[1], f[2])] == 1)
                    for (const uint32_t x : f)
                        if (x >= n_input) on_boundary[x - n_input] = 1;
            }
    }
    if (walls)
        for (const Tri &f : *walls)
            for (const uint32_t x : f)
                if (x >= n_input) on_boundary[x - n_input] = 1; // a point on an internal wall stays in the wall
    const auto shape_at = [&](const std::array<uint32_t, 4> &t, uint32_t moved, const dvec3 &x) {
        dvec3 q[4];
        for (int i = 0; i < 4; ++i) q[i] = t[size_t(i)] == moved ? x : P[t[size_t(i)]];
        const dvec3 u = q[1] - q[0], v = q[2] - q[0], w = q[3] - q[0];
        const double vol6 = u.x * (v.y * w.z - v.z * w.y) - u.y * (v.x * w.z - v.z * w.x) + u.z * (v.x * w.y - v.y * w.x);
        double l2 = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = i + 1; j < 4; ++j) {
                const dvec3 e = q[i] - q[j];
                l2 += e.x * e.x + e.y * e.y + e.z * e.z;
            }
        const double lrms = std::sqrt(l2 / 6);
        return lrms > 0 ? 1.4142135623730951 * vol6 / (lrms * lrms * lrms) : 0.0; // signed: the tets are stored positively oriented
    };
    uint32_t moved = 0;
    for (int sweep = 0; sweep < 3; ++sweep)
        for (uint32_t k = 0; k < star.size(); ++k) {
            if (on_boundary[k] || star[k].empty()) continue;
            const uint32_t m = n_input + k;
            dvec3 centre{0, 0, 0};
            double weight = 0;
            double worst = 1e300;
            for (const uint32_t t : star[k]) {
                worst = std::min(worst, shape_at(T[t], m, P[m]));
                for (const uint32_t x : T[t])
                    if (x != m) centre = centre + P[x], weight += 1;
            }
            centre = centre * (1.0 / weight);
            for (const double step : {1.0, 0.5, 0.25}) {
                const dvec3 x = P[m] + (centre - P[m]) * step;
                double fresh = 1e300;
                bool valid = true;
                for (const uint32_t t : star[k]) {
                    fresh = std::min(fresh, shape_at(T[t], m, x));
                    if (!(fresh > worst)) { valid = false; break; }
                }
                if (!valid) continue;
                for (const uint32_t t : star[k]) { // exact orientation at the new position
                    dvec3 q[4];
                    for (int i = 0; i < 4; ++i) q[i] = T[t][size_t(i)] == m ? x : P[T[t][size_t(i)]];
                    if (exact::Orient3D(q[0], q[1], q[2], q[3]) <= 0) { valid = false; break; }
                }
                if (!valid) continue;
                P[m] = x;
                ++moved;
                break;
            }
        }
    return moved;
}

// One attempt.  `constrained`: no point is added; what the Delaunay tetrahedralisation of the vertices lacks is forced in by
// re-tiling the cells it cuts through (DelaunayMesh::ConstrainEdge / ConstrainFace).
static Attempt TetrahedralizeOnce(std::span<const dvec3> points, std::span<const uint32_t> triangle_indices, const Options &options, bool constrained) {
    Attempt out;
    using Clock = std::chrono::steady_clock;
    const auto seconds_since = [](Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); };
    auto stage_start = Clock::now();
    const uint32_t n_input = uint32_t(points.size());
    if (triangle_indices.size() < 12 || triangle_indices.size() % 3) return out.Error = "a closed surface needs at least four triangles (three indices each)", out;
    std::vector<Tri> surface; // current surface triangles (refined as recovery proceeds), winding as given
    for (size_t t = 0; t + 2 < triangle_indices.size(); t += 3) {
        const Tri tri{triangle_indices[t], triangle_indices[t + 1], triangle_indices[t + 2]};
        for (const uint32_t v : tri)
            if (v >= n_input) return out.Error = "triangle index out of range", out;
        if (tri[0] == tri[1] || tri[1] == tri[2] || tri[0] == tri[2]) return out.Error = "degenerate triangle (repeated vertex)", out;
        surface.push_back(tri);
    }
    // A closed 2-manifold has two triangles on every edge.  Any other count marks NON-MANIFOLD input, which the reference accepts
    // (src/mesh/Tetrahedralize.h:53-55: "an edge may be shared by more than two triangles (internal walls)"): a wall inside the
    // solid attached to the outer surface (three triangles on the seam), a fin with a free border (one).  Such input is filled with
    // every triangle a constraint and "inside" = whatever cannot be reached from the enclosing tetrahedron without crossing a
    // triangle (step 3); if the outer surface itself is open, that flood reaches everything and the fill reports it.
    bool manifold = true;
    {
        std::map<uint64_t, int> uses;
        for (const Tri &t : surface)
            for (int e = 0; e < 3; ++e) ++uses[EdgeKey(t[e], t[(e + 1) % 3])];
        for (const auto &[key, count] : uses)
            if (count != 2) manifold = false;
    }
    std::vector<uint8_t> used(n_input, 0);
    for (const Tri &t : surface)
        for (const uint32_t v : t) used[v] = 1;

    // 1. Delaunay tetrahedralisation of the surface's vertices
    DelaunayMesh dt;
    dt.Points.assign(points.begin(), points.end());
    dvec3 lo{1e300, 1e300, 1e300}, hi{-1e300, -1e300, -1e300};
    for (uint32_t i = 0; i < n_input; ++i) {
        if (!used[i]) continue;
        const dvec3 &p = points[i];
        if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) return out.Error = "non-finite vertex", out;
        lo = {std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z)};
        hi = {std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z)};
    }
    dt.Enclose(lo, hi);
    const uint32_t shell0 = n_input, first_steiner = n_input + 4; // ids of the enclosing corners, then Steiner points
    {
        // spatially coherent insertion order (coarse grid sort): short walks
        std::vector<uint32_t> order;
        for (uint32_t i = 0; i < n_input; ++i)
            if (used[i]) order.push_back(i);
        const double sx = 64.0 / std::max(hi.x - lo.x, 1e-300), sy = 64.0 / std::max(hi.y - lo.y, 1e-300), sz = 64.0 / std::max(hi.z - lo.z, 1e-300);
        const auto cell = [&](uint32_t i) {
            const dvec3 &p = points[i];
            const uint32_t cx = uint32_t(std::min(63.0, (p.x - lo.x) * sx)), cy = uint32_t(std::min(63.0, (p.y - lo.y) * sy)), cz = uint32_t(std::min(63.0, (p.z - lo.z) * sz));
            return (cx << 12) | ((cx & 1 ? 63 - cy : cy) << 6) | ((cx + cy) & 1 ? 63 - cz : cz); // boustrophedon: neighbours in the order are neighbours in space
        };
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cell(a) < cell(b); });
        for (const uint32_t i : order)
            if (!dt.Insert(i)) return out.Error = "Delaunay insertion failed: " + dt.Error, out;
    }
    for (const auto &cell : dt.Cells) {
        if (!cell.Alive) continue;
        bool hull = true;
        for (const uint32_t v : cell.V) hull = hull && !(v >= shell0 && v < first_steiner);
        out.Profile.DelaunayTetCount += hull;
    }
    out.Profile.DelaunaySeconds = seconds_since(stage_start);
    stage_start = Clock::now();

    // 2. boundary recovery by refinement, one split at a time: a split can knock neighbouring constraints out of the mesh (and
    //    make queued ones present again), so everything near the new point is re-examined before anything else is cut
    const size_t steiner_cap = options.MaxSteinerPoints ? options.MaxSteinerPoints : size_t(n_input) + 2048; // (surfaces that fill at all needed at most 0.4 x their vertices; a run-away refinement is cut short and the constrained recovery takes over)
    const auto length2 = [&](uint32_t a, uint32_t b) {
        const dvec3 d = dt.Points[a] - dt.Points[b];
        return d.x * d.x + d.y * d.y + d.z * d.z;
    };
    std::vector<uint8_t> alive(surface.size(), 1);
    std::unordered_map<uint64_t, std::vector<uint32_t>> on_edge; // surface edge -> live surface triangles on it
    std::vector<std::vector<uint32_t>> around; // vertex -> surface triangles that were created with it (live or not)
    const auto enlist = [&](uint32_t t) {
        for (int e = 0; e < 3; ++e) on_edge[EdgeKey(surface[t][e], surface[t][(e + 1) % 3])].push_back(t);
        for (const uint32_t v : surface[t]) {
            if (around.size() <= v) around.resize(size_t(v) + 1);
            around[v].push_back(t);
        }
    };
    for (uint32_t t = 0; t < surface.size(); ++t) enlist(t);
    std::deque<uint32_t> pending; // surface triangles whose edges and face want checking
    std::vector<uint8_t> queued(surface.size(), 1);
    for (uint32_t t = 0; t < surface.size(); ++t) pending.push_back(t);
    const auto requeue_near = [&](const std::vector<uint32_t> &vertices) {
        for (const uint32_t v : vertices) {
            if (v >= around.size()) continue;
            for (const uint32_t t : around[v])
                if (alive[t] && !queued[t]) queued[t] = 1, pending.push_back(t);
        }
    };
    std::vector<std::array<uint32_t, 2>> split_edge; // per Steiner point, in insertion order: the surface edge it bisected
    const auto split = [&](uint32_t u, uint32_t v) -> bool {
        const uint32_t m = dt.AddMidpoint(u, v); // exactly on the segment (exact coordinates), rounded only for output
        split_edge.push_back({u, v});
        const uint64_t key = EdgeKey(u, v);
        const std::vector<uint32_t> hit = on_edge[key];
        on_edge.erase(key);
        for (const uint32_t t_old : hit) {
            if (!alive[t_old]) continue;
            alive[t_old] = 0;
            const Tri tri = surface[t_old];
            int at = 0;
            for (int e = 0; e < 3; ++e)
                if (EdgeKey(tri[e], tri[(e + 1) % 3]) == key) at = e;
            const uint32_t p = tri[at], q = tri[(at + 1) % 3], w = tri[(at + 2) % 3]; // winding kept: (p, q, w) -> (p, m, w) + (m, q, w)
            for (const uint64_t other : {EdgeKey(q, w), EdgeKey(w, p)}) { // the old triangle leaves its two other edges
                auto &list = on_edge[other];
                list.erase(std::remove(list.begin(), list.end(), t_old), list.end());
            }
            for (const Tri &piece : {Tri{p, m, w}, Tri{m, q, w}}) {
                surface.push_back(piece);
                alive.push_back(1);
                queued.push_back(1);
                pending.push_back(uint32_t(surface.size() - 1));
                enlist(uint32_t(surface.size() - 1));
            }
        }
        if (!(constrained ? dt.InsertBySplitting(m, u, v) : dt.Insert(m))) return false;
        requeue_near(dt.Touched);
        return true;
    };
    const auto is_surface_edge = [&](uint32_t a, uint32_t b) {
        const auto it = on_edge.find(EdgeKey(a, b));
        return it != on_edge.end() && !it->second.empty();
    };
    const auto is_surface_face = [&](uint32_t a, uint32_t b, uint32_t c) {
        const auto it = on_edge.find(EdgeKey(a, b));
        if (it == on_edge.end()) return false;
        for (const uint32_t t : it->second)
            if (alive[t] && (surface[t][0] == c || surface[t][1] == c || surface[t][2] == c)) return true;
        return false;
    };
    while (!pending.empty()) {
        const uint32_t t = pending.front();
        pending.pop_front();
        queued[t] = 0;
        if (!alive[t]) continue;
        const Tri tri = surface[t];
        int cut = -1;
        bool flipped = false;
        const auto item_start = Clock::now();
        bool face_work = false;
        const auto account = [&] { (face_work ? out.Profile.FaceSeconds : out.Profile.SegmentSeconds) += seconds_since(item_start); };
        for (int e = 0; e < 3 && cut < 0 && !flipped; ++e)
            if (!dt.HasEdge(tri[e], tri[(e + 1) % 3])) {
                ++out.Profile.MissingEdgeCount;
                // first without a point: on degenerate input another Delaunay tetrahedralisation may hold the edge
                // (between input vertices only: the recovery's own points sit exactly in the planes of the triangles they split, where
                // a coplanar crossing is the rule and the search behind it costs more than the bisection it would save)
                const uint32_t eu = tri[e], ev = tri[(e + 1) % 3];
                if (eu < n_input && ev < n_input && dt.FlipIn(eu, ev, is_surface_edge, is_surface_face)) flipped = true;
                else if (constrained && dt.ConstrainEdge(eu, ev, is_surface_edge, is_surface_face)) flipped = true;
                else cut = e; // (constrained: a point on the edge, inserted by splitting the cells that hold it)
            }
        if (flipped) { // look at the triangle again, and at everything around the exchanged cells
            ++out.Profile.FlipCount;
            requeue_near(dt.Touched);
            if (!queued[t]) queued[t] = 1, pending.push_back(t);
            account();
            continue;
        }
        if (cut < 0 && constrained && !dt.HasFace(tri[0], tri[1], tri[2])) {
            face_work = true;
            ++out.Profile.MissingFaceCount;
            if (dt.ConstrainFace(tri[0], tri[1], tri[2], is_surface_edge, is_surface_face)) {
                ++out.Profile.FlipCount;
                requeue_near(dt.Touched);
                if (!queued[t]) queued[t] = 1, pending.push_back(t);
                account();
                continue;
            }
        }
        if (cut < 0 && !dt.HasFace(tri[0], tri[1], tri[2])) { // edges present, face absent: cut the longest edge
            if (!face_work) ++out.Profile.MissingFaceCount;
            face_work = true;
            cut = 0;
            for (int e = 1; e < 3; ++e)
                if (length2(tri[e], tri[(e + 1) % 3]) > length2(tri[cut], tri[(cut + 1) % 3])) cut = e;
        }
        if (cut < 0) { account(); continue; }
        if (dt.Points.size() - first_steiner >= steiner_cap)
            return out.Error = "boundary recovery did not converge (surface self-intersects or has very sharp wedges)", out;
        // Longest-edge propagation (Rivara): bisecting an arbitrary edge of a triangle makes thinner and thinner pieces, whose
        // new edges are in turn not Delaunay -- a cascade.  So the edge that is actually cut is the end of the path that starts
        // at the wanted edge and keeps moving to the longest edge of a neighbouring surface triangle while that one is longer;
        // the pieces then never get angles below half the smallest input angle.  The triangle comes back for another look.
        uint32_t cu = tri[cut], cv = tri[(cut + 1) % 3];
        for (int hop = 0; hop < 64 && !constrained; ++hop) { // (constrained: this edge itself -- nothing cascades there)
            bool moved = false;
            const auto it = on_edge.find(EdgeKey(cu, cv));
            if (it == on_edge.end()) break;
            for (const uint32_t nb : it->second) {
                if (!alive[nb]) continue;
                const Tri &o = surface[nb];
                int longest = 0;
                for (int e = 1; e < 3; ++e)
                    if (length2(o[e], o[(e + 1) % 3]) > length2(o[longest], o[(longest + 1) % 3])) longest = e;
                if (length2(o[longest], o[(longest + 1) % 3]) > length2(cu, cv) * (1 + 1e-12)) {
                    cu = o[longest], cv = o[(longest + 1) % 3];
                    moved = true;
                    break;
                }
            }
            if (!moved) break;
        }
        const bool same = EdgeKey(cu, cv) == EdgeKey(tri[cut], tri[(cut + 1) % 3]);
        if (!split(cu, cv)) return out.Error = "Delaunay insertion of a boundary point failed: " + dt.Error, out;
        if (!same && alive[t] && !queued[t]) queued[t] = 1, pending.push_back(t);
        account();
    }
    out.Profile.SplitCount = uint32_t(split_edge.size());
    out.Profile.RecoverSeconds = seconds_since(stage_start);
    stage_start = Clock::now();
    {
        std::vector<Tri> live;
        for (uint32_t t = 0; t < surface.size(); ++t)
            if (alive[t]) live.push_back(surface[t]);
        surface.swap(live);
    }

    // 3. inside / outside by parity across surface faces, flooding from the enclosing tetrahedron
    std::set<Tri> wall;
    for (const Tri &t : surface) {
        const Tri key = Sorted(t[0], t[1], t[2]);
        if (!wall.insert(key).second) wall.erase(key); // a face given twice is a zero-thickness flap: it separates nothing
    }
    std::vector<int8_t> side(dt.Cells.size(), -1);
    std::queue<int32_t> frontier;
    for (size_t c = 0; c < dt.Cells.size(); ++c) {
        if (!dt.Cells[c].Alive) continue;
        bool shell = false;
        for (const uint32_t v : dt.Cells[c].V) shell = shell || (v >= shell0 && v < first_steiner);
        if (shell) {
            side[c] = 0;
            frontier.push(int32_t(c));
        }
    }
    while (!frontier.empty()) {
        const int32_t c = frontier.front();
        frontier.pop();
        const Cell &cell = dt.Cells[c];
        for (int i = 0; i < 4; ++i) {
            const int32_t n = cell.N[i];
            if (n < 0) continue;
            const bool crosses = wall.count(Sorted(cell.V[FaceOf[i][0]], cell.V[FaceOf[i][1]], cell.V[FaceOf[i][2]])) != 0;
            if (!manifold) { // walls inside: the flood stops at every triangle; what it never reaches is inside
                if (!crosses && side[n] < 0) {
                    side[n] = 0;
                    frontier.push(n);
                }
                continue;
            }
            const int8_t want = int8_t(side[c] ^ (crosses ? 1 : 0));
            if (side[n] < 0) {
                side[n] = want;
                frontier.push(n);
            } else if (side[n] != want) {
                return out.Error = "surface does not separate inside from outside (open or self-intersecting)", out;
            }
        }
    }

    if (!manifold) {
        bool any_inside = false;
        for (size_t c = 0; c < dt.Cells.size(); ++c)
            if (dt.Cells[c].Alive && side[c] < 0) side[c] = 1, any_inside = true;
        if (!any_inside) return out.Error = "surface is open: nothing is enclosed (an edge borders one triangle, and the outside reaches every tetrahedron)", out;
    }
    // output: input points unchanged, Steiner points appended, inside cells only
    out.Mesh.Points.assign(points.begin(), points.end());
    out.Mesh.Points.insert(out.Mesh.Points.end(), dt.Points.begin() + first_steiner, dt.Points.end());
    out.BoundarySteinerCount = uint32_t(dt.Points.size() - first_steiner);
    const auto final_id = [&](uint32_t v) { return v < n_input ? v : v - 4; };
    for (size_t c = 0; c < dt.Cells.size(); ++c) {
        if (!dt.Cells[c].Alive || side[c] != 1) continue;
        const Cell &cell = dt.Cells[c];
        out.Mesh.Tets.push_back({final_id(cell.V[0]), final_id(cell.V[1]), final_id(cell.V[2]), final_id(cell.V[3])});
    }
    if (out.Mesh.Tets.empty()) return out.Error = "surface encloses no volume", out;
    out.Profile.CarveSeconds = seconds_since(stage_start);
    stage_start = Clock::now();
    const auto refine_start = stage_start; // everything from here on is the last stage
    DebugValidate(out.Mesh, "after the carve");
    // The recovery's points are exact midpoints (expansions) while it runs and ROUNDED on the way out: a tetrahedron between several of them
    // on one thin wall can come out flat or turned over in the rounded coordinates (found by the round-6 soak: a 13 mm scan wall at a 13 mm
    // lattice, 2 080 recovery points, 15 such tetrahedra -- and a crash in the sliver repair behind them).  Such a fill is not handed on:
    // the attempt reports that the refinement did not converge, and the constrained recovery (a handful of points) takes over.
    {
        size_t turned = 0;
        for (const auto &t : out.Mesh.Tets) turned += exact::Orient3D(out.Mesh.Points[t[0]], out.Mesh.Points[t[1]], out.Mesh.Points[t[2]], out.Mesh.Points[t[3]]) <= 0;
        if (turned) return out.Error = "boundary recovery did not converge: " + std::to_string(turned) + " tetrahedra flat or inverted once the recovery's points are rounded", out;
    }
    if (options.InteriorSteiner && out.BoundarySteinerCount) {
        for (auto &e : split_edge) e = {final_id(e[0]), final_id(e[1])};
        const uint32_t on_surface = out.BoundarySteinerCount;
        out.BoundarySteinerCount = LiftBoundaryPoints(out.Mesh, n_input, split_edge);
        out.Profile.VolSteinerCount = on_surface - out.BoundarySteinerCount;
        out.Profile.SuppressSeconds = seconds_since(stage_start);
    }
    DebugValidate(out.Mesh, "after lifting");
    if (options.RepairSlivers) {
        std::set<Tri> walls; // non-manifold input: the surface pieces that ended up between two tetrahedra
        if (!manifold)
            for (const Tri &t : surface) walls.insert(Sorted(final_id(t[0]), final_id(t[1]), final_id(t[2])));
        const std::set<Tri> *keep = manifold ? nullptr : &walls;
        // connectivity and positions in turn: a moved point opens exchanges, an exchange changes what a point is connected to
        out.SliverExchanges = RepairSlivers(out.Mesh, options.SliverTarget, keep);
        for (int round = 0; round < 2 && out.Mesh.Points.size() > n_input; ++round) {
            if (!SmoothAddedPoints(out.Mesh, n_input, keep)) break;
            out.SliverExchanges += RepairSlivers(out.Mesh, options.SliverTarget, keep);
        }
        // The quality arm, when asked for (the reference's Options::Quality / MaxVolume): interior points until the radius-edge ratio is at
        // most 2 and no tetrahedron is larger than MaxVolume, where the fixed surface allows; then the same repair and smoothing again.
        if (options.Quality || options.MaxVolume > 0) {
            const size_t budget = options.MaxRefinePoints ? options.MaxRefinePoints : std::max<size_t>(20000, 40 * out.Mesh.Points.size());
            out.QualityPoints = RefineQuality(out.Mesh, true, 2.0, options.MaxVolume, keep, budget);
            if (out.QualityPoints) {
                for (int round = 0; round < 2; ++round) {
                    out.SliverExchanges += RepairSlivers(out.Mesh, options.SliverTarget, keep);
                    if (!SmoothAddedPoints(out.Mesh, n_input, keep)) break;
                }
                out.SliverExchanges += RepairSlivers(out.Mesh, options.SliverTarget, keep);
                if (options.MaxVolume > 0 && out.QualityPoints < budget) out.QualityPoints += RefineQuality(out.Mesh, false, 2.0, options.MaxVolume, keep, budget - out.QualityPoints); // (an exchange may have merged cells past the bound; what is left of the budget)
            }
        }
        // Flat cells at the surface.  A few of them (planar surface quads joined into one cell: a coarse UV sphere) each get an apex
        // underneath (BreakCaps); if the fill is still left with flat cells after that, the surface is smooth at its own resolution
        // and every surface vertex needs a vertex underneath (AddInteriorShell, from the mesh as it was before the caps).
        const auto flat_at_surface = [&](const TetMesh &m) {
            const auto &P = m.Points;
            size_t flat = 0;
            for (const auto &t : m.Tets) {
                if ((t[0] >= n_input) + (t[1] >= n_input) + (t[2] >= n_input) + (t[3] >= n_input) > 1) continue; // (three or four surface vertices)
                const dvec3 u = P[t[1]] - P[t[0]], v = P[t[2]] - P[t[0]], w = P[t[3]] - P[t[0]];
                const double vol6 = std::fabs(u.x * (v.y * w.z - v.z * w.y) - u.y * (v.x * w.z - v.z * w.x) + u.z * (v.x * w.y - v.y * w.x));
                double l2 = 0;
                for (int i = 0; i < 4; ++i)
                    for (int j = i + 1; j < 4; ++j) {
                        const dvec3 e = P[t[size_t(i)]] - P[t[size_t(j)]];
                        l2 += e.x * e.x + e.y * e.y + e.z * e.z;
                    }
                const double lrms = std::sqrt(l2 / 6);
                flat += lrms > 0 && 1.4142135623730951 * vol6 / (lrms * lrms * lrms) < 1e-3;
            }
            return flat;
        };
        // ... or with long thin cells from the surface to the far interior throughout: a thick body without interior points (a fine UV
        // sphere, a scanned rock).  Measured on the device (tools/probe/shell_probe.py): UV sphere 80 x 40 40 iterations / 303 ms without
        // the shell, 23 / 132 ms with it at 2.4 x the unknowns (and a fundamental 0.6 % lower: P2 converges from above); a solid scan 30 /
        // 168 ms -> 18 / 153 ms; the thin-walled skillet, whose fill is well shaped as it is (10th percentile 0.28), 19 / 120 -> 17 / 175 ms:
        // the shell is for fills whose 10th-percentile shape measure is below 0.08.
        const auto poorly_shaped = [&](const TetMesh &m) {
            const auto &P = m.Points;
            std::vector<double> shapes;
            shapes.reserve(m.Tets.size());
            for (const auto &t : m.Tets) {
                const dvec3 u = P[t[1]] - P[t[0]], v = P[t[2]] - P[t[0]], w = P[t[3]] - P[t[0]];
                const double vol6 = std::fabs(u.x * (v.y * w.z - v.z * w.y) - u.y * (v.x * w.z - v.z * w.x) + u.z * (v.x * w.y - v.y * w.x));
                double l2 = 0;
                for (int i = 0; i < 4; ++i)
                    for (int j = i + 1; j < 4; ++j) {
                        const dvec3 e = P[t[size_t(i)]] - P[t[size_t(j)]];
                        l2 += e.x * e.x + e.y * e.y + e.z * e.z;
                    }
                const double lrms = std::sqrt(l2 / 6);
                shapes.push_back(lrms > 0 ? 1.4142135623730951 * vol6 / (lrms * lrms * lrms) : 0.0);
            }
            if (shapes.size() < 64) return false; // (a coarse primitive: a handful of cells say nothing about a percentile)
            const size_t tenth = shapes.size() / 10;
            std::nth_element(shapes.begin(), shapes.begin() + long(tenth), shapes.end());
            return shapes[tenth] < 0.08;
        };
        const bool quality_arm = options.Quality || options.MaxVolume > 0; // (asked for: it stands in for the shell heuristic)
        if (manifold && (flat_at_surface(out.Mesh) > 0 || (!quality_arm && options.InteriorShell == Options::Shell::WhenFlat && poorly_shaped(out.Mesh)))) {
            const TetMesh before = out.Mesh;
            const uint32_t exchanges_before = out.SliverExchanges;
            if (BreakCaps(out.Mesh, 1e-3)) {
                out.SliverExchanges += RepairSlivers(out.Mesh, options.SliverTarget, keep);
                SmoothAddedPoints(out.Mesh, n_input, keep);
                out.SliverExchanges += RepairSlivers(out.Mesh, options.SliverTarget, keep);
            }
            const bool shell = options.InteriorShell == Options::Shell::Always || // (asked for explicitly: also beside the quality arm)
                               (!quality_arm && options.InteriorShell == Options::Shell::WhenFlat && (flat_at_surface(out.Mesh) * 200 > out.Mesh.Tets.size() || poorly_shaped(out.Mesh))); // > 0.5 % of the cells
            if (shell) {
                out.Mesh = before;
                out.SliverExchanges = exchanges_before;
                if ((out.ShellPoints = AddInteriorShell(out.Mesh, n_input)) > 0) {
                    for (int round = 0; round < 3; ++round) {
                        out.SliverExchanges += RepairSlivers(out.Mesh, options.SliverTarget, keep);
                        if (!SmoothAddedPoints(out.Mesh, n_input, keep)) break;
                    }
                    out.SliverExchanges += RepairSlivers(out.Mesh, options.SliverTarget, keep);
                    if (BreakCaps(out.Mesh, 1e-3)) { // (the caps that no exchange could open: now with points underneath to aim at)
                        out.SliverExchanges += RepairSlivers(out.Mesh, options.SliverTarget, keep);
                        SmoothAddedPoints(out.Mesh, n_input, keep);
                        out.SliverExchanges += RepairSlivers(out.Mesh, options.SliverTarget, keep);
                    }
                }
            }
        } else if (manifold && options.InteriorShell == Options::Shell::Always && (out.ShellPoints = AddInteriorShell(out.Mesh, n_input)) > 0) {
            for (int round = 0; round < 3; ++round) {
                out.SliverExchanges += RepairSlivers(out.Mesh, options.SliverTarget, keep);
                if (!SmoothAddedPoints(out.Mesh, n_input, keep)) break;
            }
            out.SliverExchanges += RepairSlivers(out.Mesh, options.SliverTarget, keep);
        }
        // Whatever is still flat after all that (needle quads at a fine UV sphere's poles, cells between three points of one ring): the
        // quality arm run locally, then the repair and the smoothing again -- until no cell is below the floor or a round adds nothing.
        // The reference's repair and vertex optimisation run whatever its options (Tetrahedralize.h:19-20); so does this.
        const auto worst_shape = [&](const TetMesh &m) {
            double worst = 1e300;
            for (const auto &t : m.Tets) worst = std::min(worst, ShapeOf(m.Points, t));
            return worst;
        };
        constexpr double kShapeFloor = 1e-2; // cells below it are worked on; what the pass guarantees where it succeeds is 1e-3 (tests)
        for (int round = 0; round < 4 && options.BreakFlatCells && worst_shape(out.Mesh) < kShapeFloor; ++round) {
            uint32_t moved = 0;
            const uint32_t points = BreakFlatCells(out.Mesh, kShapeFloor, keep, std::max<size_t>(4096, out.Mesh.Points.size() / 4), n_input, &moved);
            if (!points && !moved) break;
            out.FlatCellPoints += points;
            for (int sweep = 0; sweep < 2; ++sweep) {
                out.SliverExchanges += RepairSlivers(out.Mesh, options.SliverTarget, keep);
                if (!SmoothAddedPoints(out.Mesh, n_input, keep)) break;
            }
            out.SliverExchanges += RepairSlivers(out.Mesh, options.SliverTarget, keep);
        }
    }
    // MaxVolume has the last word: the exchanges, the smoothing and the flat-cell pass that followed the quality arm merge and move cells, and a
    // tetrahedron may have grown past the bound again (the round-6 options fuzz: up to twice the bound on one fill in five); without
    // RepairSlivers the arm above has not run at all.  Points only from here on -- nothing after this changes a volume.
    if (options.MaxVolume > 0) {
        std::set<Tri> walls;
        if (!manifold)
            for (const Tri &t : surface) walls.insert(Sorted(final_id(t[0]), final_id(t[1]), final_id(t[2])));
        const std::set<Tri> *keep = manifold ? nullptr : &walls;
        const size_t cap = options.MaxRefinePoints ? options.MaxRefinePoints : std::max<size_t>(20000, 40 * out.Mesh.Points.size());
        for (int round = 0; round < 4; ++round) {
            if (out.QualityPoints < cap) out.QualityPoints += RefineQuality(out.Mesh, false, 2.0, options.MaxVolume, keep, cap - out.QualityPoints);
            // (a point that bounds a volume may leave a sliver between interior points: the flat-cell pass beside it -- points and moves only,
            // no exchange -- and the bound once more, which ends the loop)
            double worst = 1e300;
            for (const auto &t : out.Mesh.Tets) worst = std::min(worst, ShapeOf(out.Mesh.Points, t));
            if (round == 3 || !options.BreakFlatCells || !options.RepairSlivers || worst >= 1e-3) break;
            uint32_t moved = 0;
            const uint32_t points = BreakFlatCells(out.Mesh, 2e-3, keep, 4096, n_input, &moved);
            out.FlatCellPoints += points;
            if (!points && !moved) break;
        }
    }
    out.Profile.RefineSeconds = seconds_since(refine_start);
    DebugValidate(out.Mesh, "at the end");
    {
        std::vector<uint8_t> used(out.Mesh.Points.size(), 0);
        for (const auto &t : out.Mesh.Tets)
            for (const uint32_t v : t) used[v] = 1;
        for (size_t v = n_input; v < used.size(); ++v) // (an input vertex no triangle uses is the caller's business; an ADDED point must be in the mesh)
            if (!used[v]) return out.Error = "boundary recovery did not converge: an added point belongs to no tetrahedron", out;
    }
    // never hand on a mesh with a tetrahedron that is not positively oriented (the reference's validator, tests/ValidateTetMesh.h:47-140, rejects it)
    for (const auto &t : out.Mesh.Tets)
        if (exact::Orient3D(out.Mesh.Points[t[0]], out.Mesh.Points[t[1]], out.Mesh.Points[t[2]], out.Mesh.Points[t[3]]) <= 0)
            return out.Error = "boundary recovery did not converge: the finished fill holds a tetrahedron that is not positively oriented", out;
    return out;
}

// Conforming Delaunay first (well-shaped cells, what every surface of round 3 goes through); a surface whose refinement runs away
// -- coarse triangles on a thin wall, needle fans: quadric-decimated scans -- is filled by the constrained recovery instead.
static Attempt TetrahedralizeAttempts(std::span<const dvec3> points, std::span<const uint32_t> triangle_indices, const Options &options) {
    Attempt conforming = TetrahedralizeOnce(points, triangle_indices, options, false);
    conforming.Profile.Builds = 1;
    if (options.MaxSteinerPoints) return conforming; // (an explicit budget asks for the refinement alone)
    if (conforming && (conforming.BoundarySteinerCount == 0 || !options.InteriorSteiner)) return conforming;
    if (!conforming && conforming.Error.find("did not converge") == std::string::npos) return conforming;
    // the refinement ran away, or left points on the surface that could not be moved inside: the constrained recovery adds a point
    // only where no tiling exists without one
    Attempt constrained = TetrahedralizeOnce(points, triangle_indices, options, true);
    const auto both = [](Attempt &kept, const Attempt &other) { // the seconds of both attempts were spent
        auto &p = kept.Profile;
        const auto &o = other.Profile;
        p.DelaunaySeconds += o.DelaunaySeconds, p.RecoverSeconds += o.RecoverSeconds, p.CarveSeconds += o.CarveSeconds, p.RefineSeconds += o.RefineSeconds;
        p.SegmentSeconds += o.SegmentSeconds, p.FaceSeconds += o.FaceSeconds, p.SuppressSeconds += o.SuppressSeconds;
        p.Builds = 2;
    };
    if (constrained && (!conforming || constrained.BoundarySteinerCount < conforming.BoundarySteinerCount)) return both(constrained, conforming), constrained;
    if (conforming) return both(conforming, constrained), conforming;
    conforming.Error += "; " + constrained.Error;
    return conforming;
}

// Conforming Delaunay first (well-shaped cells, what every surface of round 3 goes through); a surface whose refinement runs away
// -- coarse triangles on a thin wall, needle fans: quadric-decimated scans -- is filled by the constrained recovery instead.
Expected<Result> Tetrahedralize(std::span<const dvec3> points, std::span<const uint32_t> triangle_indices, Options options) {
    Attempt done = TetrahedralizeAttempts(points, triangle_indices, options);
    if (!done) return modal_compat::unexpected<std::string>(std::move(done.Error));
    Result out{std::move(done.Mesh), done.Profile};
    auto &p = out.Profile;
    p.TetCount = uint32_t(out.Mesh.Tets.size());
    p.SteinerCount = uint32_t(out.Mesh.Points.size() - points.size());
    p.BdrySteinerCount = done.BoundarySteinerCount;
    p.FlipCount += done.SliverExchanges;
    p.SliverExchangeCount = done.SliverExchanges;
    p.ShellPointCount = done.ShellPoints, p.QualityPointCount = done.QualityPoints, p.FlatCellPointCount = done.FlatCellPoints;
    return out;
}
} // namespace tetra
