// ORACLE (test infrastructure): see dense.h.
#include "dense.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>

namespace oracle {
namespace {
// Householder reduction of a symmetric matrix to tridiagonal form, accumulating the transform.
// a is row-major n x n here (symmetric, so the caller's column-major input is the same thing).
void tridiagonalize(int n, std::vector<double> &a, std::vector<double> &d, std::vector<double> &e) {
    auto A = [&](int i, int j) -> double & { return a[size_t(i) * n + j]; };
    for (int i = n - 1; i >= 1; --i) {
        const int l = i - 1;
        double h = 0, scale = 0;
        if (l > 0) {
            for (int k = 0; k <= l; ++k) scale += std::fabs(A(i, k));
            if (scale == 0.0) {
                e[i] = A(i, l);
            } else {
                for (int k = 0; k <= l; ++k) {
                    A(i, k) /= scale;
                    h += A(i, k) * A(i, k);
                }
                double f = A(i, l);
                double g = f >= 0 ? -std::sqrt(h) : std::sqrt(h);
                e[i] = scale * g;
                h -= f * g;
                A(i, l) = f - g;
                f = 0;
                for (int j = 0; j <= l; ++j) {
                    A(j, i) = A(i, j) / h;
                    g = 0;
                    for (int k = 0; k <= j; ++k) g += A(j, k) * A(i, k);
                    for (int k = j + 1; k <= l; ++k) g += A(k, j) * A(i, k);
                    e[j] = g / h;
                    f += e[j] * A(i, j);
                }
                const double hh = f / (h + h);
                for (int j = 0; j <= l; ++j) {
                    f = A(i, j);
                    e[j] = g = e[j] - hh * f;
                    for (int k = 0; k <= j; ++k) A(j, k) -= f * e[k] + g * A(i, k);
                }
            }
        } else {
            e[i] = A(i, l);
        }
        d[i] = h;
    }
    d[0] = 0;
    e[0] = 0;
    for (int i = 0; i < n; ++i) {
        const int l = i - 1;
        if (d[i] != 0.0) {
            for (int j = 0; j <= l; ++j) {
                double g = 0;
                for (int k = 0; k <= l; ++k) g += A(i, k) * A(k, j);
                for (int k = 0; k <= l; ++k) A(k, j) -= g * A(k, i);
            }
        }
        d[i] = A(i, i);
        A(i, i) = 1;
        for (int j = 0; j <= l; ++j) A(j, i) = A(i, j) = 0;
    }
}

// Implicit-shift QL on the tridiagonal (d, e); z (row-major) accumulates the rotations.
bool ql_implicit(int n, std::vector<double> &d, std::vector<double> &e, std::vector<double> &z) {
    auto Z = [&](int i, int j) -> double & { return z[size_t(i) * n + j]; };
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0;
    const double eps = std::numeric_limits<double>::epsilon();
    for (int l = 0; l < n; ++l) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; ++m) {
                const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= eps * dd) break;
            }
            if (m != l) {
                if (iter++ == 200) return false;
                double g = (d[l + 1] - d[l]) / (2 * e[l]);
                double r = std::hypot(g, 1.0);
                g = d[m] - d[l] + e[l] / (g + (g >= 0 ? std::fabs(r) : -std::fabs(r)));
                double s = 1, c = 1, p = 0;
                int i;
                for (i = m - 1; i >= l; --i) {
                    double f = s * e[i];
                    const double b = c * e[i];
                    e[i + 1] = r = std::hypot(f, g);
                    if (r == 0.0) {
                        d[i + 1] -= p;
                        e[m] = 0;
                        break;
                    }
                    s = f / r;
                    c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s + 2 * c * b;
                    d[i + 1] = g + (p = s * r);
                    g = c * r - b;
                    for (int k = 0; k < n; ++k) {
                        f = Z(k, i + 1);
                        Z(k, i + 1) = s * Z(k, i) + c * f;
                        Z(k, i) = c * Z(k, i) - s * f;
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p;
                e[l] = g;
                e[m] = 0;
            }
        } while (m != l);
    }
    return true;
}
} // namespace

bool sym_eig(int n, const double *A, double *w, double *Z) {
    if (n == 0) return true;
    std::vector<double> a(A, A + size_t(n) * n), d(n), e(n);
    tridiagonalize(n, a, d, e);
    if (!ql_implicit(n, d, e, a)) return false;
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return d[x] < d[y]; });
    for (int j = 0; j < n; ++j) {
        w[j] = d[order[j]];
        // a is row-major with eigenvectors in columns: vector j has components a[k*n + order[j]].
        for (int k = 0; k < n; ++k) Z[size_t(j) * n + k] = a[size_t(k) * n + order[j]];
    }
    return true;
}

bool chol_lower(int n, double *A, int lda) {
    for (int j = 0; j < n; ++j) {
        double *cj = A + size_t(j) * lda;
        double djj = cj[j];
        for (int k = 0; k < j; ++k) {
            const double ljk = A[size_t(k) * lda + j];
            djj -= ljk * ljk;
        }
        if (!(djj > 0)) return false;
        djj = std::sqrt(djj);
        cj[j] = djj;
        for (int k = 0; k < j; ++k) {
            const double *ck = A + size_t(k) * lda;
            const double ljk = ck[j];
            if (ljk == 0) continue;
            for (int i = j + 1; i < n; ++i) cj[i] -= ck[i] * ljk;
        }
        const double inv = 1.0 / djj;
        for (int i = j + 1; i < n; ++i) cj[i] *= inv;
    }
    return true;
}

bool gen_sym_eig(int n, const double *A, const double *B, double *w, double *Z) {
    if (n == 0) return true;
    std::vector<double> L(B, B + size_t(n) * n);
    if (!chol_lower(n, L.data(), n)) return false;
    // C = L^-1 A L^-T.  First Y = L^-1 A (forward substitution per column), then C = Y L^-T.
    std::vector<double> C(A, A + size_t(n) * n);
    auto l = [&](int i, int j) { return L[size_t(j) * n + i]; };
    for (int col = 0; col < n; ++col) {
        double *y = C.data() + size_t(col) * n;
        for (int i = 0; i < n; ++i) {
            double s = y[i];
            for (int k = 0; k < i; ++k) s -= l(i, k) * y[k];
            y[i] = s / l(i, i);
        }
    }
    // C <- C L^-T: row-wise forward substitution, i.e. solve X L^T = C for X.
    for (int j = 0; j < n; ++j) {
        double *cj = C.data() + size_t(j) * n;
        for (int k = 0; k < j; ++k) {
            const double ljk = l(j, k);
            if (ljk == 0) continue;
            const double *ck = C.data() + size_t(k) * n;
            for (int i = 0; i < n; ++i) cj[i] -= ck[i] * ljk;
        }
        const double inv = 1.0 / l(j, j);
        for (int i = 0; i < n; ++i) cj[i] *= inv;
    }
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < j; ++i) {
            const double s = 0.5 * (C[size_t(j) * n + i] + C[size_t(i) * n + j]);
            C[size_t(j) * n + i] = C[size_t(i) * n + j] = s;
        }
    if (!sym_eig(n, C.data(), w, Z)) return false;
    // Z <- L^-T Z (back substitution per column).
    for (int col = 0; col < n; ++col) {
        double *z = Z + size_t(col) * n;
        for (int i = n - 1; i >= 0; --i) {
            double s = z[i];
            for (int k = i + 1; k < n; ++k) s -= l(k, i) * z[k];
            z[i] = s / l(i, i);
        }
    }
    return true;
}
} // namespace oracle
