// ORACLE (test infrastructure, not product code): sparse symmetric positive-definite direct solver.
//
// The reference's shift-invert operator factorises K - sigma*M with Apple Accelerate's sparse Cholesky
// (src/audio/CholeskyShiftInvert.cpp:26-46) and solves with SparseSolve (:48-62).  Accelerate is closed
// source and absent from /root/reference, so the published algorithm class it implements is restated:
// fill-reducing nested-dissection ordering (level-structure separators, George 1973) followed by a
// multifrontal supernodal Cholesky (Duff & Reid 1983) with dense partial factorisations per front.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace oracle {
// Lower triangle (diagonal included) of a symmetric matrix, compressed sparse column, rows ascending per column.
struct CscLower {
    int n{0};
    std::vector<int64_t> colptr;
    std::vector<int> row;
    std::vector<double> val;
};

class MultifrontalCholesky {
public:
    // False when the matrix is not positive definite (the reference throws there, CholeskyShiftInvert.cpp:44).
    bool factorize(const CscLower &a);
    // x = A^-1 b for `width` right-hand sides stored column-major with leading dimension n.
    void solve(const double *b, double *x, int width) const;
    int64_t factor_nonzeros() const { return int64_t(L.size()); }
    double factor_flops() const { return Flops; }

private:
    struct Front {
        int j0{0}, nj{0}; // pivot columns [j0, j0+nj) in the permuted numbering
        std::vector<int> urows; // update rows (permuted indices > j0+nj-1), ascending
        std::vector<int> children;
        size_t loff{0}; // offset of the (nj+nu) x nj column-major panel [L11; L21] in L
        std::vector<int> to_parent; // update row i -> index in the parent's local vector [pivots; update rows]
        size_t uoff{0}; // offset of this front's update vector in the solve workspace
    };
    int n{0};
    std::vector<int> perm, iperm; // perm[new] = old, iperm[old] = new
    std::vector<Front> fronts; // postorder
    std::vector<std::vector<int>> levels; // fronts grouped by height above the leaves: members of one level are independent
    size_t update_total{0};
    std::vector<double> L;
    double Flops{0};
};
} // namespace oracle
