// The reference's shift-invert operator concept (src/audio/CholeskyShiftInvert.h:11-30) -- rows / cols / set_shift / perform_op /
// solve_panel, y = (K - sigma M)^-1 x -- over the device path.  What is the same: the member names, argument meaning (column-major
// panels in the reference's DOF order 3 * node + component), the two time accumulators, and the error of a shift that leaves
// K - sigma M indefinite (std::runtime_error("Modal shift-invert factorization failed."), CholeskyShiftInvert.cpp:44).
// What differs, and why: (i) the reference constructs the operator from the ASSEMBLED Eigen matrices K and M; the assembly is a device
// stage here, so the operator is constructed from the tet mesh and the material the matrices come from.  (ii) There is no sparse
// factorisation on the device: set_shift builds the eigensolver's three-level hierarchy of the shift and every application is a
// preconditioned conjugate-gradient solve to `Tolerance` (relative residual, 1e-11) -- a caller that drives Spectra's Lanczos through
// it gets the reference's eigenpairs at roughly 25 cycle applications per perform_op; modal::mesh2modes (block LOBPCG on the same
// pencil) is the fast path.  One operator = one device system on the calling thread's context.
#pragma once
#include "types.hpp"

#include <cstddef>
#include <memory>

class CholeskyShiftInvert {
public:
    using Scalar = double;

    CholeskyShiftInvert(const TetMesh &mesh, const AcousticMaterialProperties &material, double &factorize_seconds, double &solve_seconds);
    ~CholeskyShiftInvert();
    CholeskyShiftInvert(const CholeskyShiftInvert &) = delete;
    CholeskyShiftInvert &operator=(const CholeskyShiftInvert &) = delete;

    std::ptrdiff_t rows() const { return Order; }
    std::ptrdiff_t cols() const { return Order; }
    void set_shift(const Scalar &sigma);
    void perform_op(const Scalar *x_in, Scalar *y_out) const;
    // Solve across a column-major panel of `width` right-hand sides (one preconditioned CG run per column, in lockstep).
    void solve_panel(const Scalar *b_in, Scalar *x_out, int width) const;

    double Tolerance{1e-11}; // relative residual of every solve
    // of the last solve: conjugate-gradient steps and the worst column's || b - (K - sigma M) x || / || b ||
    mutable unsigned LastIterations{0};
    mutable double LastResidual{0};

private:
    double &FactorizeSeconds, &SolveSeconds;
    std::ptrdiff_t Order{0};
    double Sigma{0};
    bool Shifted{false};
    struct Device;
    std::unique_ptr<Device> Dev;
};
