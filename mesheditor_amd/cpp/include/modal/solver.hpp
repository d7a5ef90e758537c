// modal::mesh2modes / PostprocessModes / RescaleModes with the reference's signatures (src/audio/mesh2modes.h:17-88),
// implemented over the C ABI of libmodalhip (include/modalhip.h): the FEM assembly and the eigensolve run on the
// MI355X, the scalar stages on the calling thread.
#pragma once
#include "types.hpp"

#include <optional>
#include <span>

namespace modal {
struct SolverConfig {
    float MinModeFreq{20}, MaxModeFreq{16'000};
    uint32_t NumModes{30}, NumFemModes{45};
    double Tolerance{1e-8}, WarmTolerance{1e-4};
    uint32_t MaxRestarts{100};
    std::optional<float> FundamentalFreq{};
};

// Seconds per stage.  Factorize = preconditioner set-up, OpSolve = preconditioner applications (the slots the
// reference fills with its Cholesky factorisation and triangular solves); Restarts = block iterations.
struct SolveProfile {
    double MassProps{}, QuadMesh{}, Assemble{}, SampleExcite{};
    double Factorize{}, Iterate{}, OpSolve{}, Extract{};
    uint32_t Dofs{}, StiffnessNonZeros{}, OpApplications{}, Restarts{};
    SolveProfile &operator+=(const SolveProfile &o) {
        MassProps += o.MassProps; QuadMesh += o.QuadMesh; Assemble += o.Assemble; SampleExcite += o.SampleExcite;
        Factorize += o.Factorize; Iterate += o.Iterate; OpSolve += o.OpSolve; Extract += o.Extract;
        Dofs += o.Dofs; StiffnessNonZeros += o.StiffnessNonZeros; OpApplications += o.OpApplications; Restarts += o.Restarts;
        return *this;
    }
};

struct ModalResult {
    ModalModes Modes;
    MassProperties MassProps;
    SolveProfile Profile;
    ModalEigenSummary Summary;
    BasisMatrixType Basis; // filled when SolveReuse::KeepBasis
    std::vector<uint32_t> SamplePointOfExcitation;
};

struct SolveReuse {
    const BasisMatrixType *SeedBasis{};
    bool KeepBasis{};
};

// Device selection for the calling thread's solves (default 0).  Each thread owns its own context and stream, so
// solves for different entities may run concurrently (reference: one std::async job per entity).
void SetDevice(int device);

// Failure or cancellation returns a default-constructed result; a non-positive-definite shifted operator throws
// std::runtime_error, as the reference's CholeskyShiftInvert::set_shift does.
ModalResult mesh2modes(const TetMesh &, const AcousticMaterialProperties &, const std::vector<vec3> &excite_positions, vec3 baked_scale,
                       SolverConfig config = {}, SolveReuse reuse = {}, JobMonitor *monitor = nullptr);
ModalModes PostprocessModes(std::span<const double> eigenvalues, const std::vector<std::vector<vec3>> &shapes, float shape_scale,
                            const AcousticMaterialProperties &, const SolverConfig &, std::vector<vec3> positions);
std::optional<ModalModes> RescaleModes(const ModalEigenSummary &, const ModalModes &current, const AcousticMaterialProperties &, SolverConfig config = {});
} // namespace modal
