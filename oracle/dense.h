// ORACLE (test infrastructure, not product code): small dense linear-algebra helpers used by the CPU
// restatement of the reference's modal path.  The reference gets these from Eigen
// (SelfAdjointEigenSolver / GeneralizedSelfAdjointEigenSolver, src/audio/mesh2modes.cpp:112,398), which is
// not vendored under /root/reference; the published algorithms (Householder tridiagonalisation + implicit
// QL, Cholesky reduction of the generalised problem) are restated here.
#pragma once
#include <vector>

namespace oracle {
// Column-major n x n symmetric A (full storage).  On return w ascending, Z column-major eigenvectors.
bool sym_eig(int n, const double *A, double *w, double *Z);
// A z = w B z with B SPD; Z is B-orthonormal (Z^T B Z = I), w ascending.
bool gen_sym_eig(int n, const double *A, const double *B, double *w, double *Z);
// In-place lower Cholesky of column-major A (upper part untouched). False if not positive definite.
bool chol_lower(int n, double *A, int lda);
} // namespace oracle
