// Exact geometric predicates for the tetrahedraliser: the sign of orient3d and insphere determinants.
//
// A floating-point evaluation with a forward error bound decides almost every call; when the result is within the
// bound of zero the determinant is re-evaluated exactly with floating-point expansions (sums of non-overlapping doubles:
// Dekker / Knuth error-free transformations, Shewchuk 1997 "Adaptive precision floating-point arithmetic and fast robust
// geometric predicates" -- the published technique, written here as a small generic expansion type rather than as
// unrolled stages).  Exactness is what lets the Bowyer-Watson cavity stay star-shaped on fully degenerate input (grid
// boxes: every point set coplanar or cospherical).  Host code, no device counterpart.
#pragma once
#include "modal/math.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

namespace exact {
// A real number as an unevaluated sum of doubles, least significant first, components non-overlapping.
class Sum {
public:
    Sum() = default;
    explicit Sum(double v) {
        if (v != 0) Terms.push_back(v);
    }
    static Sum Difference(double a, double b) { // a - b exactly
        const double x = a - b, bv = a - x, av = x + bv;
        const double err = (a - av) + (bv - b);
        Sum s;
        if (err != 0) s.Terms.push_back(err);
        if (x != 0) s.Terms.push_back(x);
        return s;
    }
    static Sum Product(double a, double b) { // a * b exactly (fused multiply-add gives the rounding error)
        const double p = a * b, err = std::fma(a, b, -p);
        Sum s;
        if (err != 0) s.Terms.push_back(err);
        if (p != 0) s.Terms.push_back(p);
        return s;
    }
    Sum operator+(const Sum &o) const {
        Sum out = *this;
        for (const double t : o.Terms) out.Grow(t);
        return out;
    }
    Sum operator-() const {
        Sum out = *this;
        for (double &t : out.Terms) t = -t;
        return out;
    }
    Sum operator-(const Sum &o) const { return *this + (-o); }
    Sum operator*(const Sum &o) const {
        Sum out;
        for (const double t : o.Terms) out = out + Scaled(t);
        return out;
    }
    int Sign() const { return Terms.empty() ? 0 : (Terms.back() > 0 ? 1 : -1); } // the most significant component decides
    Sum Halved() const { // exact: a power of two scales every component without rounding
        Sum out = *this;
        for (double &t : out.Terms) t *= 0.5;
        return out;
    }
    double Rounded() const { // nearest double to the represented value, to within an ulp
        double v = 0;
        for (const double t : Terms) v += t;
        return v;
    }

private:
    std::vector<double> Terms;
    void Grow(double b) { // add one double, keeping the components non-overlapping (Shewchuk's GROW-EXPANSION with zero elimination)
        std::vector<double> out;
        out.reserve(Terms.size() + 1);
        double q = b;
        for (const double e : Terms) {
            const double x = q + e, bv = x - q, av = x - bv;
            const double err = (q - av) + (e - bv);
            if (err != 0) out.push_back(err);
            q = x;
        }
        if (q != 0) out.push_back(q);
        Terms.swap(out);
    }
    Sum Scaled(double b) const { // this * b: every component times b exactly, re-accumulated
        Sum out;
        for (const double e : Terms) {
            const double p = e * b, err = std::fma(e, b, -p);
            if (err != 0) out.Grow(err);
            if (p != 0) out.Grow(p);
        }
        return out;
    }
};

constexpr double Epsilon = 1.1102230246251565e-16; // 2^-53

// Sign of the signed volume (b - a) x (c - a) . (d - a): > 0 for a positively oriented tetrahedron in TetMesh's convention
// (the unit tet (0,0,0), (1,0,0), (0,1,0), (0,0,1) is positive), 0 when the four points are coplanar.  Internally the
// determinant det [a-d; b-d; c-d] is evaluated, which has the opposite sign.
inline int Orient3D(const dvec3 &a, const dvec3 &b, const dvec3 &c, const dvec3 &d) {
    const double adx = a.x - d.x, ady = a.y - d.y, adz = a.z - d.z;
    const double bdx = b.x - d.x, bdy = b.y - d.y, bdz = b.z - d.z;
    const double cdx = c.x - d.x, cdy = c.y - d.y, cdz = c.z - d.z;
    const double bc = bdx * cdy - cdx * bdy, ca = cdx * ady - adx * cdy, ab = adx * bdy - bdx * ady;
    const double det = adz * bc + bdz * ca + cdz * ab;
    const double permanent = (std::abs(bdx * cdy) + std::abs(cdx * bdy)) * std::abs(adz) + (std::abs(cdx * ady) + std::abs(adx * cdy)) * std::abs(bdz) +
                             (std::abs(adx * bdy) + std::abs(bdx * ady)) * std::abs(cdz);
    if (std::abs(det) > (7.0 + 56.0 * Epsilon) * Epsilon * permanent) return det > 0 ? -1 : 1;
    const auto diff = [](double p, double q) { return Sum::Difference(p, q); };
    const Sum ax = diff(a.x, d.x), ay = diff(a.y, d.y), az = diff(a.z, d.z);
    const Sum bx = diff(b.x, d.x), by = diff(b.y, d.y), bz = diff(b.z, d.z);
    const Sum cx = diff(c.x, d.x), cy = diff(c.y, d.y), cz = diff(c.z, d.z);
    return -(az * (bx * cy - cx * by) + bz * (cx * ay - ax * cy) + cz * (ax * by - bx * ay)).Sign();
}

// With a, b, c, d positively oriented (Orient3D(a, b, c, d) > 0): > 0 when e lies strictly inside their circumsphere,
// 0 on it, < 0 outside.
inline int InSphere(const dvec3 &a, const dvec3 &b, const dvec3 &c, const dvec3 &d, const dvec3 &e) {
    const double aex = a.x - e.x, aey = a.y - e.y, aez = a.z - e.z, bex = b.x - e.x, bey = b.y - e.y, bez = b.z - e.z;
    const double cex = c.x - e.x, cey = c.y - e.y, cez = c.z - e.z, dex = d.x - e.x, dey = d.y - e.y, dez = d.z - e.z;
    {
        const double aexbey = aex * bey, bexaey = bex * aey, bexcey = bex * cey, cexbey = cex * bey, cexdey = cex * dey, dexcey = dex * cey;
        const double dexaey = dex * aey, aexdey = aex * dey, aexcey = aex * cey, cexaey = cex * aey, bexdey = bex * dey, dexbey = dex * bey;
        const double ab = aexbey - bexaey, bc = bexcey - cexbey, cd = cexdey - dexcey, da = dexaey - aexdey, ac = aexcey - cexaey, bd = bexdey - dexbey;
        const double abc = aez * bc - bez * ac + cez * ab, bcd = bez * cd - cez * bd + dez * bc;
        const double cda = cez * da + dez * ac + aez * cd, dab = dez * ab + aez * bd + bez * da;
        const double alift = aex * aex + aey * aey + aez * aez, blift = bex * bex + bey * bey + bez * bez;
        const double clift = cex * cex + cey * cey + cez * cez, dlift = dex * dex + dey * dey + dez * dez;
        const double det = (dlift * abc - clift * dab) + (blift * cda - alift * bcd);
        const double A = std::abs(aez), B = std::abs(bez), C = std::abs(cez), D = std::abs(dez);
        const double pab = std::abs(aexbey) + std::abs(bexaey), pbc = std::abs(bexcey) + std::abs(cexbey), pcd = std::abs(cexdey) + std::abs(dexcey);
        const double pda = std::abs(dexaey) + std::abs(aexdey), pac = std::abs(aexcey) + std::abs(cexaey), pbd = std::abs(bexdey) + std::abs(dexbey);
        const double permanent = ((pcd * B + pbd * C + pbc * D) * alift + (pda * C + pac * D + pcd * A) * blift) +
                                 ((pab * D + pbd * A + pda * B) * clift + (pbc * A + pac * B + pab * C) * dlift);
        // the same expression as above gives the opposite sign convention of "inside" for positively oriented tets: fix it below
        if (std::abs(det) > (16.0 + 224.0 * Epsilon) * Epsilon * permanent) return det > 0 ? -1 : 1;
    }
    const auto diff = [](double p, double q) { return Sum::Difference(p, q); };
    const Sum ax = diff(a.x, e.x), ay = diff(a.y, e.y), az = diff(a.z, e.z), bx = diff(b.x, e.x), by = diff(b.y, e.y), bz = diff(b.z, e.z);
    const Sum cx = diff(c.x, e.x), cy = diff(c.y, e.y), cz = diff(c.z, e.z), dx = diff(d.x, e.x), dy = diff(d.y, e.y), dz = diff(d.z, e.z);
    const Sum ab = ax * by - bx * ay, bc = bx * cy - cx * by, cd = cx * dy - dx * cy, da = dx * ay - ax * dy, ac = ax * cy - cx * ay, bd = bx * dy - dx * by;
    const Sum abc = az * bc - bz * ac + cz * ab, bcd = bz * cd - cz * bd + dz * bc, cda = cz * da + dz * ac + az * cd, dab = dz * ab + az * bd + bz * da;
    const Sum al = ax * ax + ay * ay + az * az, bl = bx * bx + by * by + bz * bz, cl = cx * cx + cy * cy + cz * cz, dl = dx * dx + dy * dy + dz * dz;
    return -((dl * abc - cl * dab) + (bl * cda - al * bcd)).Sign();
}
// A point whose coordinates are either plain doubles (Fine == nullptr) or exact sums (the midpoints the boundary recovery
// inserts: exactly ON the segment they split, which no rounded double triple generally is -- with rounded midpoints the
// Delaunay mesh grows slivers of volume ~1e-17 between a segment and its bent copy).  R is the rounded position, used for the
// floating-point filter and for output.
struct Point {
    dvec3 R;
    const Sum *Fine{nullptr}; // three sums (x, y, z) or null
    Sum X(int k) const { return Fine ? Fine[k] : Sum(R[k]); }
    bool Simple() const { return Fine == nullptr; }
    // how far a coordinate of R may sit from the exact one: a few ulps of the coordinate's magnitude (not of a difference!)
    double Slop() const { return Fine ? 8.0 * Epsilon * std::max({std::abs(R.x), std::abs(R.y), std::abs(R.z)}) : 0.0; }
};

inline int Orient3D(const Point &a, const Point &b, const Point &c, const Point &d) {
    if (a.Simple() && b.Simple() && c.Simple() && d.Simple()) return Orient3D(a.R, b.R, c.R, d.R);
    {
        const double adx = a.R.x - d.R.x, ady = a.R.y - d.R.y, adz = a.R.z - d.R.z, bdx = b.R.x - d.R.x, bdy = b.R.y - d.R.y, bdz = b.R.z - d.R.z;
        const double cdx = c.R.x - d.R.x, cdy = c.R.y - d.R.y, cdz = c.R.z - d.R.z;
        const double det = adz * (bdx * cdy - cdx * bdy) + bdz * (cdx * ady - adx * cdy) + cdz * (adx * bdy - bdx * ady);
        const double permanent = (std::abs(bdx * cdy) + std::abs(cdx * bdy)) * std::abs(adz) + (std::abs(cdx * ady) + std::abs(adx * cdy)) * std::abs(bdz) +
                                 (std::abs(adx * bdy) + std::abs(bdx * ady)) * std::abs(cdz);
        // the rounded positions sit within Slop() of the exact ones; to first order that moves the determinant by at most
        // (entry error) x (sum of cofactor magnitudes) <= 9 E 2 M^2 with M the largest entry -- added to the rounding bound
        const double E = 2.0 * std::max({a.Slop(), b.Slop(), c.Slop(), d.Slop()});
        const double M = std::max({std::abs(adx), std::abs(ady), std::abs(adz), std::abs(bdx), std::abs(bdy), std::abs(bdz), std::abs(cdx), std::abs(cdy), std::abs(cdz)}) + E;
        if (std::abs(det) > 16.0 * Epsilon * permanent + 24.0 * E * M * M) return det > 0 ? -1 : 1;
    }
    const Sum ax = a.X(0) - d.X(0), ay = a.X(1) - d.X(1), az = a.X(2) - d.X(2), bx = b.X(0) - d.X(0), by = b.X(1) - d.X(1), bz = b.X(2) - d.X(2);
    const Sum cx = c.X(0) - d.X(0), cy = c.X(1) - d.X(1), cz = c.X(2) - d.X(2);
    return -(az * (bx * cy - cx * by) + bz * (cx * ay - ax * cy) + cz * (ax * by - bx * ay)).Sign();
}

inline int InSphere(const Point &a, const Point &b, const Point &c, const Point &d, const Point &e) {
    if (a.Simple() && b.Simple() && c.Simple() && d.Simple() && e.Simple()) return InSphere(a.R, b.R, c.R, d.R, e.R);
    {
        const double aex = a.R.x - e.R.x, aey = a.R.y - e.R.y, aez = a.R.z - e.R.z, bex = b.R.x - e.R.x, bey = b.R.y - e.R.y, bez = b.R.z - e.R.z;
        const double cex = c.R.x - e.R.x, cey = c.R.y - e.R.y, cez = c.R.z - e.R.z, dex = d.R.x - e.R.x, dey = d.R.y - e.R.y, dez = d.R.z - e.R.z;
        const double aexbey = aex * bey, bexaey = bex * aey, bexcey = bex * cey, cexbey = cex * bey, cexdey = cex * dey, dexcey = dex * cey;
        const double dexaey = dex * aey, aexdey = aex * dey, aexcey = aex * cey, cexaey = cex * aey, bexdey = bex * dey, dexbey = dex * bey;
        const double ab = aexbey - bexaey, bc = bexcey - cexbey, cd = cexdey - dexcey, da = dexaey - aexdey, ac = aexcey - cexaey, bd = bexdey - dexbey;
        const double abc = aez * bc - bez * ac + cez * ab, bcd = bez * cd - cez * bd + dez * bc, cda = cez * da + dez * ac + aez * cd, dab = dez * ab + aez * bd + bez * da;
        const double alift = aex * aex + aey * aey + aez * aez, blift = bex * bex + bey * bey + bez * bez, clift = cex * cex + cey * cey + cez * cez, dlift = dex * dex + dey * dey + dez * dez;
        const double det = (dlift * abc - clift * dab) + (blift * cda - alift * bcd);
        const double A = std::abs(aez), B = std::abs(bez), C = std::abs(cez), D = std::abs(dez);
        const double pab = std::abs(aexbey) + std::abs(bexaey), pbc = std::abs(bexcey) + std::abs(cexbey), pcd = std::abs(cexdey) + std::abs(dexcey);
        const double pda = std::abs(dexaey) + std::abs(aexdey), pac = std::abs(aexcey) + std::abs(cexaey), pbd = std::abs(bexdey) + std::abs(dexbey);
        const double permanent = ((pcd * B + pbd * C + pbc * D) * alift + (pda * C + pac * D + pcd * A) * blift) + ((pab * D + pbd * A + pda * B) * clift + (pbc * A + pac * B + pab * C) * dlift);
        // input perturbation as in Orient3D: twelve entries, each moving the degree-5 determinant by at most ~30 M^4 per unit
        const double E = 2.0 * std::max({a.Slop(), b.Slop(), c.Slop(), d.Slop(), e.Slop()});
        const double M = std::max({std::abs(aex), std::abs(aey), std::abs(aez), std::abs(bex), std::abs(bey), std::abs(bez), std::abs(cex), std::abs(cey), std::abs(cez),
                                   std::abs(dex), std::abs(dey), std::abs(dez)}) + E;
        if (std::abs(det) > 32.0 * Epsilon * permanent + 500.0 * E * M * M * M * M) return det > 0 ? -1 : 1;
    }
    const Sum ax = a.X(0) - e.X(0), ay = a.X(1) - e.X(1), az = a.X(2) - e.X(2), bx = b.X(0) - e.X(0), by = b.X(1) - e.X(1), bz = b.X(2) - e.X(2);
    const Sum cx = c.X(0) - e.X(0), cy = c.X(1) - e.X(1), cz = c.X(2) - e.X(2), dx = d.X(0) - e.X(0), dy = d.X(1) - e.X(1), dz = d.X(2) - e.X(2);
    const Sum ab = ax * by - bx * ay, bc = bx * cy - cx * by, cd = cx * dy - dx * cy, da = dx * ay - ax * dy, ac = ax * cy - cx * ay, bd = bx * dy - dx * by;
    const Sum abc = az * bc - bz * ac + cz * ab, bcd = bz * cd - cz * bd + dz * bc, cda = cz * da + dz * ac + az * cd, dab = dz * ab + az * bd + bz * da;
    const Sum al = ax * ax + ay * ay + az * az, bl = bx * bx + by * by + bz * bz, cl = cx * cx + cy * cy + cz * cz, dl = dx * dx + dy * dy + dz * dz;
    return -((dl * abc - cl * dab) + (bl * cda - al * bcd)).Sign();
}
} // namespace exact
