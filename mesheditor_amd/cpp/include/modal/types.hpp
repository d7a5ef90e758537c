// Data contracts of the modal path, with the reference's type and field names so that code written against
// khiner/MeshEditor's headers compiles against this mirror:
//   TetMesh (src/mesh/TetMesh.h:10-13), JobMonitor (src/Job.h:13-19), AcousticMaterialProperties and the material
//   table (src/audio/AcousticMaterialProperties.h:6-16, AcousticMaterial.h:31-38), ModalModes (ModalModes.h:7-20),
//   ModalEigenSummary (ModalEigenSummary.h:12-23), MassProperties (ContactModel.h:16-23), ModalWarmStart.
#pragma once
#include "math.hpp"

#include <array>
#include <atomic>
#include <compare>
#include <cstddef>
#include <memory>
#include <string>
#include <string_view>
#include <vector>

#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#endif

struct TetMesh {
    std::vector<dvec3> Points;
    std::vector<std::array<uint32_t, 4>> Tets; // each positively oriented
};

struct JobMonitor {
    std::atomic<float> Progress{0.f};
    std::atomic<bool> CancelRequested{false};
    void RequestCancel() { CancelRequested.store(true, std::memory_order_relaxed); }
    bool Cancelled() const { return CancelRequested.load(std::memory_order_relaxed); }
};

struct AcousticMaterialProperties {
    double Density, YoungModulus, PoissonRatio;
    double Alpha, Beta;
    double Lambda() const { return (PoissonRatio * YoungModulus) / ((1 + PoissonRatio) * (1 - 2 * PoissonRatio)); }
    double Mu() const { return YoungModulus / (2 * (1 + PoissonRatio)); }
    auto operator<=>(const AcousticMaterialProperties &) const = default;
};

struct AcousticMaterial {
    std::string Name;
    AcousticMaterialProperties Properties;
    bool operator==(const AcousticMaterial &) const = default;
};

namespace materials::acoustic {
inline const AcousticMaterial Ceramic{"Ceramic", {2700, 7.2E10, 0.19, 6, 1E-7}}, Glass{"Glass", {2600, 6.2E10, 0.20, 1, 1E-7}},
    Wood{"Wood", {750, 1.1E10, 0.25, 60, 2E-6}}, Plastic{"Plastic", {1070, 1.4E9, 0.35, 30, 1E-6}}, Iron{"Iron", {8000, 2.1E11, 0.28, 5, 1E-7}},
    Polycarbonate{"Polycarbonate", {1190, 2.4E9, 0.37, 0.5, 4E-7}}, Steel{"Steel", {7850, 2.0E11, 0.29, 5, 3E-8}};
inline const std::array All{Ceramic, Glass, Wood, Plastic, Iron, Polycarbonate, Steel};
inline const AcousticMaterial *Find(std::string_view name) {
    for (const auto &m : All)
        if (m.Name == name) return &m;
    return nullptr;
}
} // namespace materials::acoustic

struct ModalModes {
    std::vector<float> Freqs, T60s;
    std::vector<std::vector<vec3>> Shapes; // [excitation position][mode], mass-normalised
    std::vector<uint32_t> Vertices;
    std::vector<vec3> Positions;
    std::vector<uint32_t> Indices;
    float OriginalFundamentalFreq{Freqs.empty() ? 0 : Freqs.front()};
    vec3 BakedScale{1.f};
    bool operator==(const ModalModes &) const = default;
};
struct ModalGain {
    float Value{1.f};
};
struct ModalTuning {
    float FundamentalFreq{0.f}, T60Scale{1.f};
};

struct ModalEigenSummary {
    std::vector<double> Eigenvalues;
    std::vector<std::vector<vec3>> Shapes; // [excitation position][eigenpair]
    AcousticMaterialProperties SolvedMaterial{};
    float SolvedMinModeFreq{20}, SolvedMaxModeFreq{16'000};
    uint32_t SolvedNumModes{30};
    size_t TetInputsHash{};
    std::vector<uint32_t> SolvedVertices;
    bool operator==(const ModalEigenSummary &) const = default;
};

struct MassProperties {
    double Mass{0};
    vec3 CenterOfMass{0};
    vec3 InertiaDiagonal{0};
    quat InertiaOrientation{1, 0, 0, 0};
    bool operator==(const MassProperties &) const = default;
};

namespace modal {
// Column-major float matrix standing in for Eigen::MatrixXf where Eigen is not installed.
class BasisMatrix {
public:
    BasisMatrix() = default;
    BasisMatrix(std::ptrdiff_t rows, std::ptrdiff_t cols) : Rows(rows), Cols(cols), Values(size_t(rows) * size_t(cols)) {}
    std::ptrdiff_t rows() const { return Rows; }
    std::ptrdiff_t cols() const { return Cols; }
    std::ptrdiff_t size() const { return Rows * Cols; }
    float *data() { return Values.data(); }
    const float *data() const { return Values.data(); }
    float &operator()(std::ptrdiff_t r, std::ptrdiff_t c) { return Values[size_t(c) * size_t(Rows) + size_t(r)]; }
    float operator()(std::ptrdiff_t r, std::ptrdiff_t c) const { return Values[size_t(c) * size_t(Rows) + size_t(r)]; }
    void resize(std::ptrdiff_t rows, std::ptrdiff_t cols) {
        Rows = rows;
        Cols = cols;
        Values.assign(size_t(rows) * size_t(cols), 0.f);
    }

private:
    std::ptrdiff_t Rows{0}, Cols{0};
    std::vector<float> Values;
};
#if __has_include(<Eigen/Core>)
using BasisMatrixType = Eigen::MatrixXf;
#else
using BasisMatrixType = BasisMatrix;
#endif
} // namespace modal

struct ModalWarmStart {
    size_t TetInputsHash{};
    std::shared_ptr<const modal::BasisMatrixType> Basis{};
};
