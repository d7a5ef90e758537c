// ORACLE -- TEST INFRASTRUCTURE ONLY (see modal_oracle.h).
//
// CPU restatement of the analysis half of the reference's modal path, function by function:
//   src/audio/mesh2modes.cpp (FilterDegenerate :42-60, ComputeMassProperties :73-126, ComputeElementBases
//   :137-165, GetQuadBasis :209-237, BuildQuadMesh :246-264, AssembleQuadratic :273-327, SubspaceIterate
//   :339-428, ComputeModes :441-512, PostprocessModes :515-588, RescaleModes :590-603, mesh2modes :605-658)
//   src/audio/CholeskyShiftInvert.cpp (:26-62) -> MultifrontalCholesky (sparse_chol.*)
// Third-party pieces restated from their published algorithms (sources not under /root/reference):
//   Spectra SymGEigsShiftSolver (lib/spectra, unpinned submodule): restarted Lanczos in the M inner product on
//     (K - sigma M)^-1 M with full re-orthogonalisation; implicit restart with exact shifts is restated as the
//     mathematically equivalent thick restart (Wu & Simon 2000).  Convergence test and the adjusted number of
//     retained Ritz vectors follow Spectra/ARPACK: |beta * s_last,i| < tol * max(eps^(2/3), |theta_i|).
//   Eigen setFromTriplets: duplicate (row, col) triplets are summed.
//   glm::quat_cast / normalize: standard matrix-to-quaternion with the largest component first.
#include <omp.h>
#include "modal_oracle.h"

#include "dense.h"
#include "sparse_chol.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <optional>
#include <random>
#include <unordered_map>
#include <vector>

namespace oracle {
using uint = uint32_t;

struct dvec3 {
    double x{0}, y{0}, z{0};
    double &operator[](int i) { return i == 0 ? x : i == 1 ? y : z; }
    double operator[](int i) const { return i == 0 ? x : i == 1 ? y : z; }
};
struct vec3 {
    float x{0}, y{0}, z{0};
    float &operator[](int i) { return i == 0 ? x : i == 1 ? y : z; }
    float operator[](int i) const { return i == 0 ? x : i == 1 ? y : z; }
};
inline dvec3 operator-(const dvec3 &a, const dvec3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline dvec3 operator+(const dvec3 &a, const dvec3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline dvec3 operator*(const dvec3 &a, const dvec3 &b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline dvec3 operator*(double s, const dvec3 &a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(const dvec3 &a, const dvec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline dvec3 cross(const dvec3 &a, const dvec3 &b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }

struct TetMesh { // src/mesh/TetMesh.h:10-13
    std::vector<dvec3> Points;
    std::vector<std::array<uint, 4>> Tets;
};

struct MassProperties { // src/audio/ContactModel.h:16-23
    double Mass{0};
    vec3 CenterOfMass{};
    vec3 InertiaDiagonal{};
    float Quat[4]{1, 0, 0, 0}; // w, x, y, z
};

struct ModalModes { // src/audio/ModalModes.h:7-20 (Vertices/Indices/BakedScale are the caller's to fill)
    std::vector<float> Freqs, T60s;
    std::vector<std::vector<vec3>> Shapes;
    std::vector<vec3> Positions;
    float OriginalFundamentalFreq{0};
};

struct ModalEigenSummary { // src/audio/ModalEigenSummary.h:12-23
    std::vector<double> Eigenvalues;
    std::vector<std::vector<vec3>> Shapes;
    mo_material SolvedMaterial{};
};

double Lambda(const mo_material &m) { return (m.poisson_ratio * m.young_modulus) / ((1 + m.poisson_ratio) * (1 - 2 * m.poisson_ratio)); }
double Mu(const mo_material &m) { return m.young_modulus / (2 * (1 + m.poisson_ratio)); }

double SecondsSince(std::chrono::steady_clock::time_point start) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
}

// mesh2modes.cpp:42-60
TetMesh FilterDegenerate(const TetMesh &tets, std::vector<uint> *kept = nullptr) {
    TetMesh clean;
    clean.Points = tets.Points;
    clean.Tets.reserve(tets.Tets.size());
    for (uint ti = 0; ti < tets.Tets.size(); ++ti) {
        const auto &t = tets.Tets[ti];
        const dvec3 &a = tets.Points[t[0]];
        const dvec3 r0 = tets.Points[t[1]] - a, r1 = tets.Points[t[2]] - a, r2 = tets.Points[t[3]] - a;
        const double det = std::abs(dot(r0, cross(r1, r2)));
        double lmax_sq = 0;
        for (uint i = 0; i < 4; ++i) {
            for (uint j = i + 1; j < 4; ++j) {
                const dvec3 d = tets.Points[t[i]] - tets.Points[t[j]];
                lmax_sq = std::max(lmax_sq, dot(d, d));
            }
        }
        if (det > 1e-12 * lmax_sq * std::sqrt(lmax_sq)) {
            clean.Tets.push_back(t);
            if (kept) kept->push_back(ti);
        }
    }
    return clean;
}

double GetTetDeterminant(const dvec3 &a, const dvec3 &b, const dvec3 &c, const dvec3 &d) { return dot(d - a, cross(b - a, c - a)); }
// mesh2modes.cpp:67-69: the one-sixth is a float constant.
double GetTetVolume(const dvec3 &a, const dvec3 &b, const dvec3 &c, const dvec3 &d) { return (1.f / 6.f) * fabs(GetTetDeterminant(a, b, c, d)); }

// mesh2modes.cpp:73-126
MassProperties ComputeMassProperties(const TetMesh &tets, double density, vec3 scale, double length_to_si) {
    const size_t nverts = tets.Points.size();
    const dvec3 inv_scale{1.0 / scale.x, 1.0 / scale.y, 1.0 / scale.z};
    std::vector<dvec3> pos(nverts);
    for (size_t i = 0; i < nverts; ++i) pos[i] = tets.Points[i] * inv_scale;

    std::vector<double> vol(nverts, 0.0);
    for (const auto &t : tets.Tets) {
        const double quarter = GetTetVolume(pos[t[0]], pos[t[1]], pos[t[2]], pos[t[3]]) * 0.25;
        for (int c = 0; c < 4; ++c) vol[t[c]] += quarter;
    }
    double total = 0;
    dvec3 com{};
    for (size_t i = 0; i < nverts; ++i) {
        total += vol[i];
        com = com + vol[i] * pos[i];
    }
    if (total <= 0) return {};
    com = {com.x / total, com.y / total, com.z / total};

    const double s = length_to_si;
    double inertia[9] = {0}; // column-major 3x3
    auto I = [&](int r, int c) -> double & { return inertia[c * 3 + r]; };
    for (size_t i = 0; i < nverts; ++i) {
        const dvec3 r = pos[i] - com;
        const double rr = dot(r, r);
        I(0, 0) += vol[i] * (rr - r.x * r.x);
        I(1, 1) += vol[i] * (rr - r.y * r.y);
        I(2, 2) += vol[i] * (rr - r.z * r.z);
        I(0, 1) -= vol[i] * r.x * r.y;
        I(0, 2) -= vol[i] * r.x * r.z;
        I(1, 2) -= vol[i] * r.y * r.z;
    }
    I(1, 0) = I(0, 1);
    I(2, 0) = I(0, 2);
    I(2, 1) = I(1, 2);
    const double k = density * s * s * s * s * s;
    for (double &v : inertia) v *= k;

    double evals[3], evecs[9];
    sym_eig(3, inertia, evals, evecs);
    // axes[c][r] = float(evecs(r, c)); flip the first axis when the frame is left-handed (:118).
    float m[3][3]; // m[col][row]
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) m[c][r] = float(evecs[c * 3 + r]);
    const float det = m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2]) - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2]) +
        m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]);
    if (det < 0)
        for (int r = 0; r < 3; ++r) m[0][r] = -m[0][r];

    // glm::quat_cast(mat3), then glm::normalize.
    const float fx = m[0][0] - m[1][1] - m[2][2], fy = m[1][1] - m[0][0] - m[2][2], fz = m[2][2] - m[0][0] - m[1][1], fw = m[0][0] + m[1][1] + m[2][2];
    int biggest = 0;
    float big = fw;
    if (fx > big) { big = fx; biggest = 1; }
    if (fy > big) { big = fy; biggest = 2; }
    if (fz > big) { big = fz; biggest = 3; }
    const float bv = std::sqrt(big + 1.f) * 0.5f, mult = 0.25f / bv;
    float qw, qx, qy, qz;
    switch (biggest) {
        case 0: qw = bv; qx = (m[1][2] - m[2][1]) * mult; qy = (m[2][0] - m[0][2]) * mult; qz = (m[0][1] - m[1][0]) * mult; break;
        case 1: qw = (m[1][2] - m[2][1]) * mult; qx = bv; qy = (m[0][1] + m[1][0]) * mult; qz = (m[2][0] + m[0][2]) * mult; break;
        case 2: qw = (m[2][0] - m[0][2]) * mult; qx = (m[0][1] + m[1][0]) * mult; qy = bv; qz = (m[1][2] + m[2][1]) * mult; break;
        default: qw = (m[0][1] - m[1][0]) * mult; qx = (m[2][0] + m[0][2]) * mult; qy = (m[1][2] + m[2][1]) * mult; qz = bv; break;
    }
    const float qn = std::sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    MassProperties mp;
    mp.Mass = density * total * s * s * s;
    mp.CenterOfMass = {float(com.x), float(com.y), float(com.z)};
    mp.InertiaDiagonal = {float(evals[0]), float(evals[1]), float(evals[2])};
    if (qn > 0) {
        mp.Quat[0] = qw / qn; mp.Quat[1] = qx / qn; mp.Quat[2] = qy / qn; mp.Quat[3] = qz / qn;
    }
    return mp;
}

constexpr uint NEV = 4;
struct ElementBasis {
    double Volume;
    dvec3 Phig[NEV];
};

// mesh2modes.cpp:137-165
std::vector<ElementBasis> ComputeElementBases(const TetMesh &tets) {
    std::vector<ElementBasis> elements(tets.Tets.size());
    dvec3 columns[2];
    for (uint el = 0; el < elements.size(); ++el) {
        auto &element = elements[el];
        auto vert = [&](uint v) -> const dvec3 & { return tets.Points[tets.Tets[el][v]]; };
        const double det = GetTetDeterminant(vert(0), vert(1), vert(2), vert(3));
        element.Volume = fabs(det / 6);
        for (uint i = 0; i < NEV; ++i) {
            for (uint j = 0; j < 3; ++j) {
                uint ni = 0;
                for (uint ii = 0; ii < NEV; ++ii) {
                    if (ii == i) continue;
                    uint nj = 0;
                    for (uint jj = 0; jj < 3; ++jj) {
                        if (jj != j) {
                            columns[nj][ni] = vert(ii)[jj];
                            nj++;
                        }
                    }
                    ++ni;
                }
                const int sign = (i + j) % 2 == 0 ? -1 : 1;
                element.Phig[i][j] = sign * dot(dvec3{1, 1, 1}, cross(columns[0], columns[1])) / det;
            }
        }
    }
    return elements;
}

// ---- Quadratic (10-node) elements: mesh2modes.cpp:167-264 ----
struct BaryTerm {
    double Coeff;
    std::array<int, 4> Exp;
};
using BaryPoly = std::vector<BaryTerm>;

BaryPoly Multiply(const BaryPoly &a, const BaryPoly &b) {
    BaryPoly product;
    for (const auto &ta : a)
        for (const auto &tb : b)
            product.push_back({ta.Coeff * tb.Coeff, {ta.Exp[0] + tb.Exp[0], ta.Exp[1] + tb.Exp[1], ta.Exp[2] + tb.Exp[2], ta.Exp[3] + tb.Exp[3]}});
    return product;
}

double UnitIntegral(const BaryPoly &p) {
    static constexpr double Factorial[]{1, 1, 2, 6, 24, 120, 720, 5040};
    double sum = 0;
    for (const auto &t : p) {
        const auto &e = t.Exp;
        sum += t.Coeff * 6 * Factorial[e[0]] * Factorial[e[1]] * Factorial[e[2]] * Factorial[e[3]] / Factorial[e[0] + e[1] + e[2] + e[3] + 3];
    }
    return sum;
}

constexpr uint NumQuadNodes = 10;
constexpr uint EdgeCorners[6][2]{{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};

struct QuadBasis {
    double Mass[NumQuadNodes][NumQuadNodes];
    double Grad[NumQuadNodes][4][NumQuadNodes][4];
};

const QuadBasis &GetQuadBasis() {
    static const QuadBasis basis = [] {
        std::array<BaryPoly, NumQuadNodes> n;
        std::array<std::array<BaryPoly, 4>, NumQuadNodes> dn;
        for (int i = 0; i < 4; ++i) {
            n[i] = {{2, {2 * (i == 0), 2 * (i == 1), 2 * (i == 2), 2 * (i == 3)}}, {-1, {i == 0, i == 1, i == 2, i == 3}}};
            dn[i][i] = {{4, {i == 0, i == 1, i == 2, i == 3}}, {-1, {0, 0, 0, 0}}};
        }
        for (uint e = 0; e < 6; ++e) {
            const int i = EdgeCorners[e][0], j = EdgeCorners[e][1];
            n[4 + e] = {{4, {i == 0 || j == 0, i == 1 || j == 1, i == 2 || j == 2, i == 3 || j == 3}}};
            dn[4 + e][i] = {{4, {j == 0, j == 1, j == 2, j == 3}}};
            dn[4 + e][j] = {{4, {i == 0, i == 1, i == 2, i == 3}}};
        }
        QuadBasis b;
        for (uint a = 0; a < NumQuadNodes; ++a) {
            for (uint c = 0; c < NumQuadNodes; ++c) {
                b.Mass[a][c] = UnitIntegral(Multiply(n[a], n[c]));
                for (int k = 0; k < 4; ++k)
                    for (int l = 0; l < 4; ++l)
                        b.Grad[a][k][c][l] = dn[a][k].empty() || dn[c][l].empty() ? 0 : UnitIntegral(Multiply(dn[a][k], dn[c][l]));
            }
        }
        return b;
    }();
    return basis;
}

struct QuadMesh {
    std::vector<std::array<uint, NumQuadNodes>> ElementNodes;
    uint NodeCount;
};

// mesh2modes.cpp:246-264: midside ids in first-encounter order.
QuadMesh BuildQuadMesh(const TetMesh &tets) {
    QuadMesh quad;
    quad.ElementNodes.resize(tets.Tets.size());
    quad.NodeCount = uint(tets.Points.size());
    std::unordered_map<uint64_t, uint> edge_nodes;
    edge_nodes.reserve(tets.Tets.size() * 2);
    for (uint el = 0; el < uint(tets.Tets.size()); ++el) {
        auto &nodes = quad.ElementNodes[el];
        for (uint c = 0; c < 4; ++c) nodes[c] = tets.Tets[el][c];
        for (uint e = 0; e < 6; ++e) {
            const uint a = nodes[EdgeCorners[e][0]], b = nodes[EdgeCorners[e][1]];
            const uint64_t key = (uint64_t(std::min(a, b)) << 32) | std::max(a, b);
            const auto [it, inserted] = edge_nodes.try_emplace(key, quad.NodeCount);
            if (inserted) ++quad.NodeCount;
            nodes[4 + e] = it->second;
        }
    }
    return quad;
}

struct Triplet {
    uint row, col;
    double value;
};

// Eigen::SparseMatrix::setFromTriplets restated: column-major, duplicates summed, rows ascending.
CscLower FromTriplets(uint n, std::vector<Triplet> &t) {
    CscLower m;
    m.n = int(n);
    m.colptr.assign(n + 1, 0);
    std::stable_sort(t.begin(), t.end(), [](const Triplet &a, const Triplet &b) { return a.col != b.col ? a.col < b.col : a.row < b.row; });
    for (size_t i = 0; i < t.size();) {
        size_t j = i;
        double s = 0;
        while (j < t.size() && t[j].row == t[i].row && t[j].col == t[i].col) s += t[j++].value;
        m.row.push_back(int(t[i].row));
        m.val.push_back(s);
        ++m.colptr[t[i].col + 1];
        i = j;
    }
    for (uint c = 0; c < n; ++c) m.colptr[c + 1] += m.colptr[c];
    return m;
}

struct MassStiffness {
    CscLower Mass, Stiffness;
};

// mesh2modes.cpp:273-327
MassStiffness AssembleQuadratic(const TetMesh &tets, const QuadMesh &quad, const mo_material &material) {
    const auto &basis = GetQuadBasis();
    const auto coeffs = ComputeElementBases(tets);
    const double lambda = Lambda(material), mu = Mu(material);
    std::vector<Triplet> mass_triplets, stiffness_triplets;
    mass_triplets.reserve(quad.ElementNodes.size() * NumQuadNodes * (NumQuadNodes + 1) / 2 * 3);
    stiffness_triplets.reserve(quad.ElementNodes.size() * NumQuadNodes * (NumQuadNodes + 1) / 2 * 9);
    for (uint el = 0; el < quad.ElementNodes.size(); ++el) {
        const auto &ed = coeffs[el];
        const auto &nodes = quad.ElementNodes[el];
        double outer[4][4][3][3];
        for (int k = 0; k < 4; ++k)
            for (int l = 0; l < 4; ++l)
                for (uint p = 0; p < 3; ++p)
                    for (uint q = 0; q < 3; ++q) outer[k][l][p][q] = ed.Phig[k][p] * ed.Phig[l][q];
        for (uint a = 0; a < NumQuadNodes; ++a) {
            for (uint c = 0; c < NumQuadNodes; ++c) {
                const uint row = 3 * nodes[a], col = 3 * nodes[c];
                if (row < col) continue;
                const double m = material.density * ed.Volume * basis.Mass[a][c];
                for (uint k = 0; k < 3; ++k) mass_triplets.push_back({row + k, col + k, m});
                double g[3][3]{};
                for (int k = 0; k < 4; ++k) {
                    for (int l = 0; l < 4; ++l) {
                        const double w = basis.Grad[a][k][c][l];
                        if (w == 0) continue;
                        for (uint p = 0; p < 3; ++p)
                            for (uint q = 0; q < 3; ++q) g[p][q] += w * outer[k][l][p][q];
                    }
                }
                const double trace = g[0][0] + g[1][1] + g[2][2];
                for (uint p = 0; p < 3; ++p)
                    for (uint q = 0; q < (row == col ? p + 1 : 3u); ++q)
                        stiffness_triplets.push_back({row + p, col + q, ed.Volume * (lambda * g[p][q] + mu * g[q][p] + (p == q ? mu * trace : 0))});
            }
        }
    }
    const uint n = 3 * quad.NodeCount;
    return {FromTriplets(n, mass_triplets), FromTriplets(n, stiffness_triplets)};
}

// y = A x for a symmetric matrix stored as its lower triangle (Eigen selfadjointView<Lower>).
void SymMatVec(const CscLower &a, const double *x, double *y) {
    std::fill(y, y + a.n, 0.0);
    for (int c = 0; c < a.n; ++c) {
        const double xc = x[c];
        double acc = 0;
        for (int64_t p = a.colptr[c]; p < a.colptr[c + 1]; ++p) {
            const int r = a.row[p];
            const double v = a.val[p];
            if (r == c) {
                y[c] += v * xc;
            } else {
                y[r] += v * xc;
                acc += v * x[r];
            }
        }
        y[c] += acc;
    }
}

// K - sigma*M with the union pattern (CholeskyShiftInvert.cpp:28).
CscLower Shifted(const CscLower &k, const CscLower &m, double sigma) {
    CscLower s;
    s.n = k.n;
    s.colptr.assign(k.n + 1, 0);
    for (int c = 0; c < k.n; ++c) {
        int64_t pk = k.colptr[c], pm = m.colptr[c];
        const int64_t ek = k.colptr[c + 1], em = m.colptr[c + 1];
        while (pk < ek || pm < em) {
            const int rk = pk < ek ? k.row[pk] : std::numeric_limits<int>::max();
            const int rm = pm < em ? m.row[pm] : std::numeric_limits<int>::max();
            const int r = std::min(rk, rm);
            double v = 0;
            if (rk == r) v += k.val[pk++];
            if (rm == r) v -= sigma * m.val[pm++];
            s.row.push_back(r);
            s.val.push_back(v);
        }
        s.colptr[c + 1] = int64_t(s.row.size());
    }
    return s;
}

// The reference's operator (CholeskyShiftInvert.h:11-30) over the restated factorisation.
struct ShiftInvertOp {
    const CscLower &K, &M;
    double &FactorizeSeconds, &SolveSeconds;
    MultifrontalCholesky Factor;
    bool Ok{false};
    int rows() const { return K.n; }
    void set_shift(double sigma) {
        const auto start = std::chrono::steady_clock::now();
        Ok = Factor.factorize(Shifted(K, M, sigma));
        FactorizeSeconds += SecondsSince(start);
    }
    void perform_op(const double *x, double *y) {
        const auto start = std::chrono::steady_clock::now();
        Factor.solve(x, y, 1);
        SolveSeconds += SecondsSince(start);
    }
    void solve_panel(const double *b, double *x, int width) {
        const auto start = std::chrono::steady_clock::now();
        Factor.solve(b, x, width);
        SolveSeconds += SecondsSince(start);
    }
};

struct EigResult {
    std::vector<double> Eigenvalues; // ascending; empty on failure
    std::vector<double> Eigenvectors; // n x nev column-major, M-orthonormal
    uint OpApplications{0}, Restarts{0};
};

// Long-vector kernels of the Lanczos loop on the OpenMP team.  Dot products are summed over fixed blocks whose partial
// sums are then added in block order, so the value does not depend on the team size.
double BlockedDot(const double *a, const double *b, size_t n) {
    constexpr size_t Block = 4096;
    const size_t blocks = (n + Block - 1) / Block;
    std::vector<double> partial(blocks);
#pragma omp parallel for schedule(static) if (blocks >= 8)
    for (size_t k = 0; k < blocks; ++k) {
        const size_t lo = k * Block, hi = std::min(n, lo + Block);
        double s = 0;
        for (size_t r = lo; r < hi; ++r) s += a[r] * b[r];
        partial[k] = s;
    }
    double total = 0;
    for (const double v : partial) total += v;
    return total;
}
void Axpy(double *y, double alpha, const double *x, size_t n) { // y += alpha x
#pragma omp parallel for schedule(static) if (n >= 32768)
    for (size_t r = 0; r < n; ++r) y[r] += alpha * x[r];
}

// One restarted Lanczos run on OP = (K - sigma M)^-1 M in the M inner product, optionally deflated against the
// M-orthonormal vectors Q (every Krylov vector is kept M-orthogonal to them).  Returns the `nev` Ritz pairs of
// largest |theta| (theta = 1/(lambda - sigma)) in `theta_out` / `vec_out` (n x nev), or false when not converged.
bool LanczosRun(ShiftInvertOp &op, const CscLower &M, uint nev, uint ncv, double tol, uint max_restarts, const std::vector<double> &Q,
                const std::vector<double> &MQ, uint nq, uint64_t seed, std::vector<double> &theta_out, std::vector<double> &vec_out, uint &ops, uint &restarts) {
    const size_t n = size_t(op.rows());
    std::vector<double> V(n * (ncv + 1)), MV(n * (ncv + 1)), T(size_t(ncv) * ncv, 0.0);
    std::vector<double> w(n), Mw(n), h(ncv + 1), theta(ncv), S(size_t(ncv) * ncv);
    auto col = [&](std::vector<double> &a, size_t j) { return a.data() + j * n; };
    auto mnorm = [&](const double *x, const double *mx) { return std::sqrt(std::max(BlockedDot(x, mx, n), 0.0)); };
    auto deflate = [&](double *x) { // x -= Q (MQ^T x), twice
        for (int pass = 0; pass < 2 && nq; ++pass)
            for (uint i = 0; i < nq; ++i) {
                const double *mq = MQ.data() + size_t(i) * n, *q = Q.data() + size_t(i) * n;
                Axpy(x, -BlockedDot(mq, x, n), q, n);
            }
    };
    // Start vector: fixed-seed uniform noise pushed through the operator once (into the range of OP).
    {
        std::mt19937_64 rng{seed};
        std::uniform_real_distribution<double> uni(-0.5, 0.5);
        for (size_t i = 0; i < n; ++i) w[i] = uni(rng);
        SymMatVec(M, w.data(), Mw.data());
        op.perform_op(Mw.data(), col(V, 0));
        ++ops;
        deflate(col(V, 0));
        SymMatVec(M, col(V, 0), col(MV, 0));
        const double nrm = mnorm(col(V, 0), col(MV, 0));
        for (size_t i = 0; i < n; ++i) {
            col(V, 0)[i] /= nrm;
            col(MV, 0)[i] /= nrm;
        }
    }
    const double eps23 = std::pow(std::numeric_limits<double>::epsilon(), 2.0 / 3.0);
    uint k = 0; // retained Ritz vectors at the head of V
    double beta_last = 0;
    std::vector<uint> wanted(ncv);
    uint local_restarts = 0;
    for (;;) {
        for (uint j = k; j < ncv; ++j) {
            op.perform_op(col(MV, j), w.data());
            ++ops;
            deflate(w.data());
            // Two passes of classical Gram-Schmidt in the M inner product against everything so far.
            std::fill(h.begin(), h.end(), 0.0);
            for (int pass = 0; pass < 2; ++pass) {
                for (uint i = 0; i <= j; ++i) {
                    const double s = BlockedDot(col(MV, i), w.data(), n);
                    h[i] += s;
                    Axpy(w.data(), -s, col(V, i), n);
                }
            }
            T[size_t(j) * ncv + j] = h[j];
            SymMatVec(M, w.data(), Mw.data());
            const double beta = mnorm(w.data(), Mw.data());
            const double inv = beta > 0 ? 1.0 / beta : 0.0;
            double *vn = col(V, j + 1), *mvn = col(MV, j + 1);
            for (size_t r = 0; r < n; ++r) {
                vn[r] = w[r] * inv;
                mvn[r] = Mw[r] * inv;
            }
            if (j + 1 < ncv) {
                T[size_t(j) * ncv + j + 1] = T[size_t(j + 1) * ncv + j] = beta;
            } else {
                beta_last = beta;
            }
        }
        if (!sym_eig(int(ncv), T.data(), theta.data(), S.data())) return false;
        std::iota(wanted.begin(), wanted.end(), 0u);
        std::stable_sort(wanted.begin(), wanted.end(), [&](uint a, uint b) { return std::abs(theta[a]) > std::abs(theta[b]); });
        uint nconv = 0;
        for (uint i = 0; i < nev; ++i) {
            const uint c = wanted[i];
            const double res = std::abs(beta_last * S[size_t(c) * ncv + (ncv - 1)]);
            if (res < tol * std::max(eps23, std::abs(theta[c]))) ++nconv;
        }
        if (nconv >= nev || local_restarts >= max_restarts) {
            if (nconv < nev) return false;
            break;
        }
        ++local_restarts;
        ++restarts;
        uint keep = nev + std::min(nconv, (ncv - nev) / 2);
        if (nev == 1 && ncv >= 6) keep = ncv / 2;
        else if (nev == 1 && ncv > 2) keep = 2;
        keep = std::min(keep, ncv - 1);
        // Thick restart: V <- V S(:, wanted[0..keep)), T <- diag(theta) bordered by the residual couplings.
        std::vector<double> Vn(n * keep), MVn(n * keep);
#pragma omp parallel for schedule(dynamic, 1)
        for (uint c = 0; c < keep; ++c) {
            const double *s = S.data() + size_t(wanted[c]) * ncv;
            double *vo = Vn.data() + size_t(c) * n, *mo = MVn.data() + size_t(c) * n;
            for (uint i = 0; i < ncv; ++i) {
                const double si = s[i];
                if (si == 0) continue;
                const double *v = col(V, i), *mv = col(MV, i);
                for (size_t r = 0; r < n; ++r) {
                    vo[r] += si * v[r];
                    mo[r] += si * mv[r];
                }
            }
        }
        std::vector<double> vnext(col(V, ncv), col(V, ncv) + n), mvnext(col(MV, ncv), col(MV, ncv) + n);
        std::copy(Vn.begin(), Vn.end(), V.begin());
        std::copy(MVn.begin(), MVn.end(), MV.begin());
        std::copy(vnext.begin(), vnext.end(), col(V, keep));
        std::copy(mvnext.begin(), mvnext.end(), col(MV, keep));
        std::fill(T.begin(), T.end(), 0.0);
        for (uint c = 0; c < keep; ++c) {
            T[size_t(c) * ncv + c] = theta[wanted[c]];
            const double coupling = beta_last * S[size_t(wanted[c]) * ncv + (ncv - 1)];
            T[size_t(c) * ncv + keep] = T[size_t(keep) * ncv + c] = coupling;
        }
        k = keep;
    }
    theta_out.resize(nev);
    vec_out.assign(n * nev, 0.0);
#pragma omp parallel for schedule(dynamic, 1)
    for (uint c = 0; c < nev; ++c) {
        theta_out[c] = theta[wanted[c]];
        const double *s = S.data() + size_t(wanted[c]) * ncv;
        double *out = vec_out.data() + size_t(c) * n;
        for (uint i = 0; i < ncv; ++i) {
            const double si = s[i];
            const double *v = col(V, i);
            for (size_t r = 0; r < n; ++r) out[r] += si * v[r];
        }
    }
    return true;
}

// Restarted shift-invert Lanczos (Spectra SymGEigsShiftSolver<..., ShiftInvert>::compute(LargestMagn, maxit, tol, SmallestAlge)).
//
// Multiplicity safeguard (not part of Spectra): a single-vector Krylov space holds one copy of a repeated
// eigenvalue, and the test bodies here (cubes, square bars, balls) have exact symmetries.  After the main run, short
// verification runs deflated against everything found so far look for eigenvalues below the largest one returned;
// any they find replace the tail.  On meshes without exact multiplicities the first verification run finds nothing.
EigResult ShiftInvertLanczos(ShiftInvertOp &op, const CscLower &M, uint nev, uint ncv, double sigma, double tol, uint max_restarts) {
    const size_t n = size_t(op.rows());
    EigResult result;
    std::vector<double> theta, vecs;
    if (!LanczosRun(op, M, nev, ncv, tol, max_restarts, {}, {}, 0, 0, theta, vecs, result.OpApplications, result.Restarts)) return {};
    auto mprod = [&](const std::vector<double> &X, uint cols) {
        std::vector<double> Y(n * cols);
        for (uint j = 0; j < cols; ++j) SymMatVec(M, X.data() + size_t(j) * n, Y.data() + size_t(j) * n);
        return Y;
    };
    const uint nv = std::min(8u, nev), ncv_v = uint(std::min<size_t>(std::max(nv + 20u, 20u), n > nev ? n - nev : 1));
    for (uint round = 0; round < 16 && ncv_v > nv && size_t(nev) + ncv_v < n; ++round) {
        const std::vector<double> MQ = mprod(vecs, uint(theta.size()));
        std::vector<double> th2, v2;
        if (!LanczosRun(op, M, nv, ncv_v, tol, max_restarts, vecs, MQ, uint(theta.size()), 1 + round, th2, v2, result.OpApplications, result.Restarts)) break;
        // smallest theta currently kept among the first nev (largest lambda)
        std::vector<uint> order(theta.size());
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint a, uint b) { return theta[a] > theta[b]; });
        const double cutoff = theta[order[nev - 1]];
        uint added = 0;
        for (uint i = 0; i < nv; ++i) {
            if (th2[i] > cutoff * (1 + 1e-9)) {
                theta.push_back(th2[i]);
                vecs.insert(vecs.end(), v2.begin() + size_t(i) * n, v2.begin() + size_t(i + 1) * n);
                ++added;
            }
        }
        if (added == 0) break;
    }
    // lambda = sigma + 1/theta for the nev largest theta, ascending in lambda.
    std::vector<uint> order(theta.size());
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint a, uint b) { return theta[a] > theta[b]; });
    result.Eigenvalues.resize(nev);
    result.Eigenvectors.assign(n * nev, 0.0);
    for (uint c = 0; c < nev; ++c) {
        result.Eigenvalues[c] = sigma + 1.0 / theta[order[c]];
        std::copy(vecs.begin() + size_t(order[c]) * n, vecs.begin() + size_t(order[c] + 1) * n, result.Eigenvectors.begin() + size_t(c) * n);
    }
    return result;
}

// mesh2modes.cpp:339-428
EigResult SubspaceIterate(ShiftInvertOp &op, const CscLower &M, uint nev, uint p, double sigma, double tol, uint max_iters,
                          const float *x0, uint x0_cols, const volatile int *cancel) {
    const size_t n = size_t(M.n);
    auto matmat = [&](const std::vector<double> &X, uint cols, std::vector<double> &Y) {
        Y.resize(n * cols);
        for (uint j = 0; j < cols; ++j) SymMatVec(M, X.data() + size_t(j) * n, Y.data() + size_t(j) * n);
    };
    // C (ra x ca) = A^T B with A n x ra, B n x ca.
    auto atb = [&](const double *A, uint ra, const double *B, uint ca) {
        std::vector<double> C(size_t(ra) * ca);
        for (uint j = 0; j < ca; ++j)
            for (uint i = 0; i < ra; ++i) {
                const double *a = A + size_t(i) * n, *b = B + size_t(j) * n;
                double s = 0;
                for (size_t r = 0; r < n; ++r) s += a[r] * b[r];
                C[size_t(j) * ra + i] = s;
            }
        return C;
    };
    // Y (n x cb) += alpha * A (n x ra) * B (ra x cb)
    auto axpy_mm = [&](double *Y, const double *A, uint ra, const double *B, uint cb, double alpha) {
        for (uint j = 0; j < cb; ++j)
            for (uint i = 0; i < ra; ++i) {
                const double s = alpha * B[size_t(j) * ra + i];
                if (s == 0) continue;
                const double *a = A + size_t(i) * n;
                double *y = Y + size_t(j) * n;
                for (size_t r = 0; r < n; ++r) y[r] += s * a[r];
            }
    };
    std::vector<double> MX;
    {
        std::vector<double> X(n * p);
        std::mt19937_64 rng{20260710};
        std::normal_distribution<double> gauss;
        const uint seeded = std::min(x0_cols, p);
        for (uint j = 0; j < seeded; ++j)
            for (size_t i = 0; i < n; ++i) X[size_t(j) * n + i] = double(x0[size_t(j) * n + i]);
        for (uint j = seeded; j < p; ++j)
            for (size_t i = 0; i < n; ++i) X[size_t(j) * n + i] = gauss(rng);
        matmat(X, p, MX);
    }
    EigResult result;
    std::vector<double> XL(n * nev, 0.0), MXL(n * nev, 0.0), theta_locked(nev, 0.0);
    uint c = 0;
    std::vector<double> prev_lambda(nev, std::numeric_limits<double>::max());
    for (uint iter = 0; iter < max_iters; ++iter) {
        if (cancel && *cancel) return {};
        const uint w = p - c;
        std::vector<double> Xbar(n * w);
        op.solve_panel(MX.data(), Xbar.data(), int(w));
        result.OpApplications += w;
        std::vector<double> Kr = atb(Xbar.data(), w, MX.data(), w);
        std::vector<double> MXbar;
        matmat(Xbar, w, MXbar);
        if (c > 0) {
            const std::vector<double> C = atb(XL.data(), c, MXbar.data(), w); // c x w
            axpy_mm(Xbar.data(), XL.data(), c, C.data(), w, -1.0);
            axpy_mm(MXbar.data(), MXL.data(), c, C.data(), w, -1.0);
            for (uint j = 0; j < w; ++j)
                for (uint i = 0; i < w; ++i) {
                    double s = 0;
                    for (uint l = 0; l < c; ++l) s += C[size_t(i) * c + l] * theta_locked[l] * C[size_t(j) * c + l];
                    Kr[size_t(j) * w + i] -= s;
                }
        }
        std::vector<double> Mr = atb(Xbar.data(), w, MXbar.data(), w);
        for (uint j = 0; j < w; ++j)
            for (uint i = 0; i < j; ++i) {
                const double ks = 0.5 * (Kr[size_t(j) * w + i] + Kr[size_t(i) * w + j]);
                Kr[size_t(j) * w + i] = Kr[size_t(i) * w + j] = ks;
                const double ms = 0.5 * (Mr[size_t(j) * w + i] + Mr[size_t(i) * w + j]);
                Mr[size_t(j) * w + i] = Mr[size_t(i) * w + j] = ms;
            }
        std::vector<double> dscale(w);
        for (uint i = 0; i < w; ++i) dscale[i] = 1.0 / std::sqrt(Mr[size_t(i) * w + i]);
        for (uint j = 0; j < w; ++j)
            for (uint i = 0; i < w; ++i) {
                Kr[size_t(j) * w + i] *= dscale[i] * dscale[j];
                Mr[size_t(j) * w + i] *= dscale[i] * dscale[j];
            }
        std::vector<double> evals(w), q(size_t(w) * w);
        if (!gen_sym_eig(int(w), Kr.data(), Mr.data(), evals.data(), q.data())) return result;
        for (uint j = 0; j < w; ++j)
            for (uint i = 0; i < w; ++i) q[size_t(j) * w + i] *= dscale[i];

        uint newly_locked = 0;
        for (uint i = 0; i < w && c + i < nev; ++i) {
            const double lambda = evals[i] + sigma;
            const double rel = std::abs(lambda - prev_lambda[c + i]) / std::max(std::abs(lambda), std::abs(sigma));
            prev_lambda[c + i] = lambda;
            if (newly_locked == i && rel < tol) ++newly_locked;
        }
        if (newly_locked > 0) {
            std::fill(XL.begin() + size_t(c) * n, XL.begin() + size_t(c + newly_locked) * n, 0.0);
            std::fill(MXL.begin() + size_t(c) * n, MXL.begin() + size_t(c + newly_locked) * n, 0.0);
            axpy_mm(XL.data() + size_t(c) * n, Xbar.data(), w, q.data(), newly_locked, 1.0);
            axpy_mm(MXL.data() + size_t(c) * n, MXbar.data(), w, q.data(), newly_locked, 1.0);
            for (uint i = 0; i < newly_locked; ++i) theta_locked[c + i] = evals[i];
            c += newly_locked;
        }
        result.Restarts = iter + 1;
        if (c >= nev) {
            result.Eigenvalues = prev_lambda;
            result.Eigenvectors = std::move(XL);
            return result;
        }
        const uint rest = w - newly_locked;
        std::vector<double> next(n * rest, 0.0);
        axpy_mm(next.data(), MXbar.data(), w, q.data() + size_t(newly_locked) * w, rest, 1.0);
        MX = std::move(next);
    }
    return result;
}

ModalModes PostprocessModes(const std::vector<double> &eigenvalues, const std::vector<std::vector<vec3>> &shapes, float shape_scale,
                            const mo_material &material, const mo_solver_config &config, std::vector<vec3> positions);

struct ModalResult {
    ModalModes Modes;
    MassProperties MassProps;
    mo_profile Profile{};
    ModalEigenSummary Summary;
    std::vector<float> Basis; // n x cols column-major
    uint BasisRows{0}, BasisCols{0};
    std::vector<uint32_t> SamplePointOfExcitation;
};

// mesh2modes.cpp:441-512
ModalModes ComputeModes(const CscLower &M, const CscLower &K, uint num_vertices, uint vertex_dim, const mo_solver_config &config,
                        const std::vector<uint> &ex_pos, std::vector<vec3> positions, const mo_material &material,
                        const float *seed, uint seed_rows, uint seed_cols, const volatile int *cancel,
                        mo_profile &profile, ModalEigenSummary &summary_out, ModalResult *basis_out) {
    const uint n = num_vertices * vertex_dim;
    const uint fem_n_modes = std::min(config.num_fem_modes, n - 1);
    const uint basis_size = std::min(std::max(fem_n_modes + 20, 20u), n);
    const double sigma = -pow(2 * M_PI * config.min_mode_freq, 2);
    if (cancel && *cancel) return {};
    ShiftInvertOp op{K, M, profile.factorize, profile.op_solve, {}, false};
    const bool use_subspace = seed != nullptr && seed_rows == n && seed_cols >= fem_n_modes;
    EigResult eig;
    const auto eig_start = std::chrono::steady_clock::now();
    op.set_shift(sigma);
    if (!op.Ok) return {}; // the reference throws std::runtime_error here (CholeskyShiftInvert.cpp:44)
    if (use_subspace) {
        eig = SubspaceIterate(op, M, fem_n_modes, std::min(fem_n_modes + 15, n), sigma, config.warm_tolerance, config.max_restarts, seed, seed_cols, cancel);
        if (eig.Eigenvalues.empty()) return {};
    } else {
        if (cancel && *cancel) return {};
        eig = ShiftInvertLanczos(op, M, fem_n_modes, basis_size, sigma, config.tolerance, config.max_restarts);
        if (eig.Eigenvalues.empty()) return {};
    }
    profile.op_applications = eig.OpApplications;
    profile.restarts = eig.Restarts;
    profile.iterate = SecondsSince(eig_start) - profile.factorize;
    const auto ext_start = std::chrono::steady_clock::now();
    const std::vector<double> &eigenvectors = eig.Eigenvectors;
    profile.extract = SecondsSince(ext_start);
    std::vector<std::vector<vec3>> shapes(ex_pos.size(), std::vector<vec3>(fem_n_modes));
    for (size_t ex = 0; ex < shapes.size(); ++ex) {
        const uint ev_i = vertex_dim * ex_pos[ex];
        for (uint mode = 0; mode < fem_n_modes; ++mode)
            for (uint vi = 0; vi < vertex_dim; ++vi) shapes[ex][mode][int(vi)] = float(eigenvectors[size_t(mode) * n + ev_i + vi]);
    }
    summary_out.Eigenvalues = eig.Eigenvalues;
    summary_out.Shapes = shapes;
    summary_out.SolvedMaterial = material;
    if (basis_out) {
        basis_out->BasisRows = n;
        basis_out->BasisCols = fem_n_modes;
        basis_out->Basis.resize(size_t(n) * fem_n_modes);
        for (size_t i = 0; i < basis_out->Basis.size(); ++i) basis_out->Basis[i] = float(eigenvectors[i]);
    }
    return PostprocessModes(summary_out.Eigenvalues, shapes, 1.f, material, config, std::move(positions));
}

// mesh2modes.cpp:515-588
ModalModes PostprocessModes(const std::vector<double> &eigenvalues, const std::vector<std::vector<vec3>> &shapes, float shape_scale,
                            const mo_material &material, const mo_solver_config &config, std::vector<vec3> positions) {
    const uint fem_n_modes = uint(eigenvalues.size());
    std::vector<float> mode_freqs(fem_n_modes), mode_t60s(fem_n_modes);
    std::vector<double> omega_undamped(fem_n_modes);
    const double lambda_eps = pow(2 * M_PI * config.min_mode_freq, 2) * 1e-10;
    for (uint mode = 0; mode < fem_n_modes; ++mode) {
        const double lambda_i = eigenvalues[mode];
        omega_undamped[mode] = lambda_i > lambda_eps ? std::sqrt(lambda_i) : 0;
    }
    const auto c_from_omega = [&material](double omega) { return material.alpha + material.beta * (omega * omega); };
    const auto damped_hz = [&](double omega, double c) {
        const double omega_d_sq = omega * omega - 0.25 * c * c;
        return omega_d_sq > 0 ? std::sqrt(omega_d_sq) / (2 * M_PI) : 0;
    };
    uint lowest_mode_i = fem_n_modes;
    float lowest_mode_freq_orig{0};
    for (uint mode = 0; mode < fem_n_modes; ++mode) {
        const double omega_i = omega_undamped[mode];
        if (omega_i <= 0) {
            mode_freqs[mode] = mode_t60s[mode] = 0.f;
            continue;
        }
        mode_freqs[mode] = float(damped_hz(omega_i, c_from_omega(omega_i)));
        if (lowest_mode_i == fem_n_modes && mode_freqs[mode] >= config.min_mode_freq) {
            lowest_mode_i = mode;
            lowest_mode_freq_orig = mode_freqs[mode];
        }
    }
    if (lowest_mode_i == fem_n_modes) return {};

    static const double ln_1000 = std::log(1000);
    const float freq_scale = config.has_fundamental ? config.fundamental_freq / lowest_mode_freq_orig : 1.f;
    for (uint mode = lowest_mode_i; mode < fem_n_modes; ++mode) {
        const double omega_s = omega_undamped[mode] * freq_scale;
        const double c = c_from_omega(omega_s);
        mode_freqs[mode] = float(damped_hz(omega_s, c));
        mode_t60s[mode] = float(c > 0 ? (2 * ln_1000) / c : 0);
    }
    const float max_mode_freq = config.max_mode_freq * std::max(1.f, freq_scale);
    uint highest_mode_i = fem_n_modes;
    while (highest_mode_i > lowest_mode_i && mode_freqs[highest_mode_i - 1] > max_mode_freq) --highest_mode_i;

    const uint n_modes = std::min({config.num_modes, fem_n_modes, highest_mode_i - lowest_mode_i});
    mode_freqs.erase(mode_freqs.begin(), mode_freqs.begin() + lowest_mode_i);
    mode_freqs.resize(n_modes);
    mode_t60s.erase(mode_t60s.begin(), mode_t60s.begin() + lowest_mode_i);
    mode_t60s.resize(n_modes);

    std::vector<std::vector<vec3>> out_shapes(shapes.size(), std::vector<vec3>(n_modes));
    for (size_t ex = 0; ex < shapes.size(); ++ex)
        for (uint mode = 0; mode < n_modes; ++mode) {
            const vec3 &s = shapes[ex][mode + lowest_mode_i];
            out_shapes[ex][mode] = {s.x * shape_scale, s.y * shape_scale, s.z * shape_scale};
        }
    ModalModes out;
    out.Freqs = std::move(mode_freqs);
    out.T60s = std::move(mode_t60s);
    out.Shapes = std::move(out_shapes);
    out.Positions = std::move(positions);
    out.OriginalFundamentalFreq = lowest_mode_freq_orig;
    return out;
}

// mesh2modes.cpp:590-603
std::optional<ModalModes> RescaleModes(const ModalEigenSummary &summary, const std::vector<vec3> &current_positions,
                                       const mo_material &material, const mo_solver_config &config) {
    if (summary.Eigenvalues.empty() || material.poisson_ratio != summary.SolvedMaterial.poisson_ratio) return {};
    const double rho_ratio = material.density / summary.SolvedMaterial.density;
    const double eigenvalue_scale = (material.young_modulus / summary.SolvedMaterial.young_modulus) / rho_ratio;
    auto eigenvalues = summary.Eigenvalues;
    for (auto &v : eigenvalues) v *= eigenvalue_scale;
    return PostprocessModes(eigenvalues, summary.Shapes, float(1 / std::sqrt(rho_ratio)), material, config, current_positions);
}

// mesh2modes.cpp:605-658
ModalResult Mesh2Modes(const TetMesh &input_tets, const mo_material &material, const std::vector<vec3> &excite_positions, vec3 baked_scale,
                       const mo_solver_config &config, const float *seed, uint seed_rows, uint seed_cols, bool keep_basis, const volatile int *cancel) {
    const TetMesh tets = FilterDegenerate(input_tets);
    mo_profile profile{};
    const double length_to_si = (double(baked_scale.x) + baked_scale.y + baked_scale.z) / 3.0;
    auto t0 = std::chrono::steady_clock::now();
    auto mass_props = ComputeMassProperties(tets, material.density, baked_scale, length_to_si);
    profile.mass_props = SecondsSince(t0);
    t0 = std::chrono::steady_clock::now();
    const auto quad = BuildQuadMesh(tets);
    profile.quad_mesh = SecondsSince(t0);
    t0 = std::chrono::steady_clock::now();
    const auto ms = AssembleQuadratic(tets, quad, material);
    profile.assemble = SecondsSince(t0);
    profile.dofs = 3 * quad.NodeCount;
    profile.stiffness_nonzeros = uint(ms.Stiffness.row.size());
    if (cancel && *cancel) return {};

    t0 = std::chrono::steady_clock::now();
    const dvec3 inv_scale{1.0 / baked_scale.x, 1.0 / baked_scale.y, 1.0 / baked_scale.z};
    std::vector<uint> points;
    std::vector<vec3> local;
    std::vector<uint32_t> remap(excite_positions.size());
    std::unordered_map<uint, uint32_t> sample_point_at;
    for (size_t i = 0; i < excite_positions.size(); ++i) {
        const dvec3 p{excite_positions[i].x, excite_positions[i].y, excite_positions[i].z};
        double best = std::numeric_limits<double>::max();
        uint nearest = 0;
        for (uint v = 0; v < uint(tets.Points.size()); ++v) {
            const dvec3 d = p - tets.Points[v];
            if (const double d2 = dot(d, d); d2 < best) {
                best = d2;
                nearest = v;
            }
        }
        const auto [entry, first] = sample_point_at.emplace(nearest, uint32_t(points.size()));
        if (first) {
            points.push_back(nearest);
            const dvec3 l = tets.Points[nearest] * inv_scale;
            local.push_back({float(l.x), float(l.y), float(l.z)});
        }
        remap[i] = entry->second;
    }
    profile.sample_excite = SecondsSince(t0);

    ModalResult result;
    ModalEigenSummary summary;
    auto modes = ComputeModes(ms.Mass, ms.Stiffness, quad.NodeCount, 3, config, points, std::move(local), material, seed, seed_rows, seed_cols,
                              cancel, profile, summary, keep_basis ? &result : nullptr);
    result.Modes = std::move(modes);
    result.MassProps = mass_props;
    result.Profile = profile;
    result.Summary = std::move(summary);
    result.SamplePointOfExcitation = std::move(remap);
    return result;
}
} // namespace oracle

// ---------------------------------------------------------------- C API
using namespace oracle;

struct mo_result {
    ModalResult r;
};
struct mo_system {
    TetMesh tets;
    std::vector<uint32_t> kept;
    QuadMesh quad;
    MassStiffness ms;
};

namespace {
TetMesh MakeMesh(uint32_t n_points, const double *p, uint32_t n_tets, const uint32_t *t) {
    TetMesh m;
    m.Points.resize(n_points);
    for (uint32_t i = 0; i < n_points; ++i) m.Points[i] = {p[3 * i], p[3 * i + 1], p[3 * i + 2]};
    m.Tets.resize(n_tets);
    for (uint32_t i = 0; i < n_tets; ++i) m.Tets[i] = {t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]};
    return m;
}
std::vector<std::vector<vec3>> UnpackShapes(uint32_t n_pos, uint32_t n_eigs, const float *s) {
    std::vector<std::vector<vec3>> shapes(n_pos, std::vector<vec3>(n_eigs));
    for (uint32_t p = 0; p < n_pos; ++p)
        for (uint32_t k = 0; k < n_eigs; ++k) shapes[p][k] = {s[(size_t(p) * n_eigs + k) * 3], s[(size_t(p) * n_eigs + k) * 3 + 1], s[(size_t(p) * n_eigs + k) * 3 + 2]};
    return shapes;
}
uint32_t PackModes(const ModalModes &m, float *freqs, float *t60s, float *shapes, float *orig) {
    const uint32_t k = uint32_t(m.Freqs.size());
    if (freqs) std::copy(m.Freqs.begin(), m.Freqs.end(), freqs);
    if (t60s) std::copy(m.T60s.begin(), m.T60s.end(), t60s);
    if (shapes)
        for (size_t p = 0; p < m.Shapes.size(); ++p)
            for (uint32_t j = 0; j < k; ++j) {
                shapes[(p * k + j) * 3] = m.Shapes[p][j].x;
                shapes[(p * k + j) * 3 + 1] = m.Shapes[p][j].y;
                shapes[(p * k + j) * 3 + 2] = m.Shapes[p][j].z;
            }
    if (orig) *orig = m.OriginalFundamentalFreq;
    return k;
}
} // namespace

extern "C" {
// Size of the OpenMP team the dense front kernels and front solves run on (1 = one job thread, as the reference).
void mo_set_threads(int n) { omp_set_num_threads(n < 1 ? 1 : n); }
int mo_max_threads() { return omp_get_max_threads(); }
void mo_default_config(mo_solver_config *c) {
    *c = mo_solver_config{20.f, 16000.f, 30, 45, 1e-8, 1e-4, 100, 0, 0.f};
}

mo_result *mo_mesh2modes(uint32_t n_points, const double *points_xyz, uint32_t n_tets, const uint32_t *tets, const mo_material *material,
                         uint32_t n_excite, const float *excite_xyz, const float baked_scale[3], const mo_solver_config *config,
                         const float *seed_basis, uint32_t seed_rows, uint32_t seed_cols, int keep_basis, const volatile int *cancel_flag) {
    const TetMesh mesh = MakeMesh(n_points, points_xyz, n_tets, tets);
    std::vector<vec3> ex(n_excite);
    for (uint32_t i = 0; i < n_excite; ++i) ex[i] = {excite_xyz[3 * i], excite_xyz[3 * i + 1], excite_xyz[3 * i + 2]};
    auto *out = new mo_result;
    out->r = Mesh2Modes(mesh, *material, ex, vec3{baked_scale[0], baked_scale[1], baked_scale[2]}, *config, seed_basis, seed_rows, seed_cols, keep_basis != 0, cancel_flag);
    return out;
}
void mo_result_free(mo_result *r) { delete r; }
uint32_t mo_result_num_modes(const mo_result *r) { return uint32_t(r->r.Modes.Freqs.size()); }
uint32_t mo_result_num_positions(const mo_result *r) { return uint32_t(r->r.Modes.Positions.size()); }
uint32_t mo_result_num_eigenpairs(const mo_result *r) { return uint32_t(r->r.Summary.Eigenvalues.size()); }
uint32_t mo_result_num_excitations(const mo_result *r) { return uint32_t(r->r.SamplePointOfExcitation.size()); }
// (the summary keeps its sample points also when no mode survives the band filter and Modes.Positions is empty)
uint32_t mo_result_num_summary_points(const mo_result *r) { return uint32_t(r->r.Summary.Shapes.size()); }
void mo_result_modes(const mo_result *r, float *freqs, float *t60s, float *shapes, float *positions, float *original_fundamental) {
    PackModes(r->r.Modes, freqs, t60s, shapes, original_fundamental);
    if (positions)
        for (size_t p = 0; p < r->r.Modes.Positions.size(); ++p) {
            positions[3 * p] = r->r.Modes.Positions[p].x;
            positions[3 * p + 1] = r->r.Modes.Positions[p].y;
            positions[3 * p + 2] = r->r.Modes.Positions[p].z;
        }
}
void mo_result_summary(const mo_result *r, double *eigenvalues, float *shapes) {
    const auto &s = r->r.Summary;
    if (eigenvalues) std::copy(s.Eigenvalues.begin(), s.Eigenvalues.end(), eigenvalues);
    const size_t k = s.Eigenvalues.size();
    if (shapes)
        for (size_t p = 0; p < s.Shapes.size(); ++p)
            for (size_t j = 0; j < k; ++j) {
                shapes[(p * k + j) * 3] = s.Shapes[p][j].x;
                shapes[(p * k + j) * 3 + 1] = s.Shapes[p][j].y;
                shapes[(p * k + j) * 3 + 2] = s.Shapes[p][j].z;
            }
}
void mo_result_mass_props(const mo_result *r, double *mass, float com[3], float inertia_diag[3], float quat_wxyz[4]) {
    const auto &m = r->r.MassProps;
    *mass = m.Mass;
    for (int i = 0; i < 3; ++i) {
        com[i] = m.CenterOfMass[i];
        inertia_diag[i] = m.InertiaDiagonal[i];
    }
    for (int i = 0; i < 4; ++i) quat_wxyz[i] = m.Quat[i];
}
void mo_result_profile(const mo_result *r, mo_profile *p) { *p = r->r.Profile; }
void mo_result_sample_point_of_excitation(const mo_result *r, uint32_t *out) {
    std::copy(r->r.SamplePointOfExcitation.begin(), r->r.SamplePointOfExcitation.end(), out);
}
uint32_t mo_result_basis_rows(const mo_result *r) { return r->r.BasisRows; }
uint32_t mo_result_basis_cols(const mo_result *r) { return r->r.BasisCols; }
void mo_result_basis(const mo_result *r, float *out) { std::copy(r->r.Basis.begin(), r->r.Basis.end(), out); }

mo_system *mo_assemble(uint32_t n_points, const double *points_xyz, uint32_t n_tets, const uint32_t *tets, const mo_material *material) {
    auto *s = new mo_system;
    const TetMesh in = MakeMesh(n_points, points_xyz, n_tets, tets);
    s->tets = FilterDegenerate(in, &s->kept);
    s->quad = BuildQuadMesh(s->tets);
    s->ms = AssembleQuadratic(s->tets, s->quad, *material);
    return s;
}
void mo_system_free(mo_system *s) { delete s; }
uint32_t mo_system_dofs(const mo_system *s) { return 3 * s->quad.NodeCount; }
uint32_t mo_system_node_count(const mo_system *s) { return s->quad.NodeCount; }
uint32_t mo_system_kept_tets(const mo_system *s) { return uint32_t(s->tets.Tets.size()); }
void mo_system_kept_tet_indices(const mo_system *s, uint32_t *out) { std::copy(s->kept.begin(), s->kept.end(), out); }
void mo_system_element_nodes(const mo_system *s, uint32_t *out) {
    for (size_t e = 0; e < s->quad.ElementNodes.size(); ++e)
        for (int a = 0; a < 10; ++a) out[e * 10 + a] = s->quad.ElementNodes[e][a];
}
uint64_t mo_system_nnz(const mo_system *s, int which) { return (which ? s->ms.Mass : s->ms.Stiffness).row.size(); }
void mo_system_csc(const mo_system *s, int which, int64_t *colptr, int32_t *rows, double *vals) {
    const auto &m = which ? s->ms.Mass : s->ms.Stiffness;
    std::copy(m.colptr.begin(), m.colptr.end(), colptr);
    std::copy(m.row.begin(), m.row.end(), rows);
    std::copy(m.val.begin(), m.val.end(), vals);
}
void mo_quad_basis(double *mass100, double *grad1600) {
    const auto &b = GetQuadBasis();
    std::memcpy(mass100, b.Mass, sizeof(b.Mass));
    std::memcpy(grad1600, b.Grad, sizeof(b.Grad));
}
int mo_system_eigs(const mo_system *s, uint32_t nev, uint32_t ncv, double sigma, double tol, uint32_t max_restarts, double *evals,
                   double *evecs, mo_profile *profile) {
    mo_profile local{};
    mo_profile &p = profile ? *profile : local;
    const auto start = std::chrono::steady_clock::now();
    ShiftInvertOp op{s->ms.Stiffness, s->ms.Mass, p.factorize, p.op_solve, {}, false};
    op.set_shift(sigma);
    if (!op.Ok) return 2;
    auto eig = ShiftInvertLanczos(op, s->ms.Mass, nev, ncv, sigma, tol, max_restarts);
    p.iterate = SecondsSince(start) - p.factorize;
    p.op_applications = eig.OpApplications;
    p.restarts = eig.Restarts;
    p.dofs = uint32_t(s->ms.Stiffness.n);
    p.stiffness_nonzeros = uint32_t(s->ms.Stiffness.row.size());
    if (eig.Eigenvalues.empty()) return 1;
    std::copy(eig.Eigenvalues.begin(), eig.Eigenvalues.end(), evals);
    if (evecs) std::copy(eig.Eigenvectors.begin(), eig.Eigenvectors.end(), evecs);
    return 0;
}
void mo_system_matvec(const mo_system *s, int which, const double *x, double *y) { SymMatVec(which ? s->ms.Mass : s->ms.Stiffness, x, y); }

uint32_t mo_postprocess_modes(uint32_t n_eigs, const double *eigenvalues, uint32_t n_pos, const float *shapes_in, float shape_scale,
                              const mo_material *material, const mo_solver_config *config, float *freqs, float *t60s, float *shapes_out,
                              float *original_fundamental) {
    const std::vector<double> ev(eigenvalues, eigenvalues + n_eigs);
    const auto modes = PostprocessModes(ev, UnpackShapes(n_pos, n_eigs, shapes_in), shape_scale, *material, *config, {});
    return PackModes(modes, freqs, t60s, shapes_out, original_fundamental);
}
uint32_t mo_rescale_modes(uint32_t n_eigs, const double *eigenvalues, uint32_t n_pos, const float *summary_shapes, const mo_material *solved,
                          const mo_material *edited, const mo_solver_config *config, float *freqs, float *t60s, float *shapes_out,
                          float *original_fundamental) {
    ModalEigenSummary summary;
    summary.Eigenvalues.assign(eigenvalues, eigenvalues + n_eigs);
    summary.Shapes = UnpackShapes(n_pos, n_eigs, summary_shapes);
    summary.SolvedMaterial = *solved;
    const auto modes = RescaleModes(summary, {}, *edited, *config);
    if (!modes) return UINT32_MAX;
    return PackModes(*modes, freqs, t60s, shapes_out, original_fundamental);
}
void mo_mass_properties(uint32_t n_points, const double *points_xyz, uint32_t n_tets, const uint32_t *tets, double density, const float scale[3],
                        double length_to_si, double *mass, float com[3], float inertia_diag[3], float quat_wxyz[4]) {
    const TetMesh mesh = MakeMesh(n_points, points_xyz, n_tets, tets);
    const auto m = ComputeMassProperties(mesh, density, vec3{scale[0], scale[1], scale[2]}, length_to_si);
    *mass = m.Mass;
    for (int i = 0; i < 3; ++i) {
        com[i] = m.CenterOfMass[i];
        inertia_diag[i] = m.InertiaDiagonal[i];
    }
    for (int i = 0; i < 4; ++i) quat_wxyz[i] = m.Quat[i];
}
}
