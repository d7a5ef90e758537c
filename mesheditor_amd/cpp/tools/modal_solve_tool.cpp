// Solve a tetrahedral mesh's modal model on the MI355X and print it as JSON on stdout: the fields of the reference's
// MeshEditorModalSolve (tests/ModalSolveTool.cpp:101-123 -- frequencies, decayRates, positions, mode-major shapes,
// indices, mass, centerOfMass, inertiaDiagonal), which its sample generator shells out to.  The reference tool starts
// from a surface .obj and tetrahedralises it; here an .obj is filled by tetra::FillStarShaped (star-shaped solids only,
// modal/tets.hpp) and its vertices are the excitation positions, or the tet mesh is given directly:
//   modal_solve <mesh.obj> [--layers k] [options]
//   modal_solve <mesh.tet> [options]        text file: "V T", V lines "x y z", T lines "a b c d" (positively oriented)
//   modal_solve --kuhn lx ly lz nx ny nz [--origin x y z] [options]
//   --young E --poisson v --density rho --alpha a --beta b   material (SI)
//   --min-freq f --max-freq f --modes n                      solve window
//   --gltf <out.gltf>                                        also write the model as a KHR_audio_rigid_bodies document
// Every boundary vertex is an excitation position; the output triangles are the mesh's boundary faces relabeled onto
// the sample points those vertices became.
#include "modal/model_io.hpp"
#include "modal/solver.hpp"
#include "modal/tets.hpp"

#include <algorithm>
#include <array>
#include <charconv>
#include <cstdio>
#include <fstream>
#include <map>
#include <numbers>
#include <string_view>

namespace {
std::optional<size_t> ArgIndex(int argc, char **argv, std::string_view name) {
    for (int i = 1; i < argc; ++i)
        if (argv[i] == name) return size_t(i);
    return std::nullopt;
}
double ArgValue(int argc, char **argv, std::string_view name, double fallback, size_t offset = 1) {
    if (const auto i = ArgIndex(argc, argv, name); i && *i + offset < size_t(argc)) {
        double value;
        const std::string_view s = argv[*i + offset];
        if (std::from_chars(s.data(), s.data() + s.size(), value).ec == std::errc{}) return value;
    }
    return fallback;
}

TetMesh KuhnBox(double lx, double ly, double lz, int nx, int ny, int nz, dvec3 origin) {
    TetMesh mesh;
    const auto vid = [&](int i, int j, int k) { return uint32_t((i * (ny + 1) + j) * (nz + 1) + k); };
    for (int i = 0; i <= nx; ++i)
        for (int j = 0; j <= ny; ++j)
            for (int k = 0; k <= nz; ++k) mesh.Points.push_back({origin.x + lx * i / nx, origin.y + ly * j / ny, origin.z + lz * k / nz});
    static constexpr int Corner[6][4]{{0, 1, 3, 7}, {0, 3, 2, 7}, {0, 2, 6, 7}, {0, 6, 4, 7}, {0, 4, 5, 7}, {0, 5, 1, 7}};
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < ny; ++j)
            for (int k = 0; k < nz; ++k) {
                const uint32_t c[8]{vid(i, j, k), vid(i + 1, j, k), vid(i, j + 1, k), vid(i + 1, j + 1, k),
                                    vid(i, j, k + 1), vid(i + 1, j, k + 1), vid(i, j + 1, k + 1), vid(i + 1, j + 1, k + 1)};
                for (const auto &t : Corner) mesh.Tets.push_back({c[t[0]], c[t[1]], c[t[2]], c[t[3]]});
            }
    return mesh;
}

std::optional<TetMesh> LoadTetFile(const char *path) {
    std::ifstream in{path};
    size_t nv = 0, nt = 0;
    if (!(in >> nv >> nt)) return std::nullopt;
    TetMesh mesh;
    mesh.Points.resize(nv);
    mesh.Tets.resize(nt);
    for (auto &p : mesh.Points)
        if (!(in >> p.x >> p.y >> p.z)) return std::nullopt;
    for (auto &t : mesh.Tets)
        if (!(in >> t[0] >> t[1] >> t[2] >> t[3])) return std::nullopt;
    return mesh;
}

// Faces that belong to exactly one tetrahedron, wound outward (for a positively oriented tet 0123 the outward faces
// are 132, 023, 031, 012).
std::vector<std::array<uint32_t, 3>> BoundaryFaces(const TetMesh &mesh) {
    static constexpr int Face[4][3]{{1, 3, 2}, {0, 2, 3}, {0, 3, 1}, {0, 1, 2}};
    std::map<std::array<uint32_t, 3>, std::pair<std::array<uint32_t, 3>, int>> seen;
    for (const auto &t : mesh.Tets)
        for (const auto &f : Face) {
            const std::array<uint32_t, 3> tri{t[f[0]], t[f[1]], t[f[2]]};
            auto key = tri;
            std::sort(key.begin(), key.end());
            auto &slot = seen[key];
            slot.first = tri;
            ++slot.second;
        }
    std::vector<std::array<uint32_t, 3>> out;
    for (const auto &[key, slot] : seen)
        if (slot.second == 1) out.push_back(slot.first);
    return out;
}

void PrintScalars(const char *key, const auto &values) {
    std::printf("  \"%s\": [", key);
    for (size_t i = 0; i < values.size(); ++i) std::printf("%s%.9g", i ? "," : "", double(values[i]));
    std::printf("],\n");
}
} // namespace

int main(int argc, char **argv) {
    std::optional<TetMesh> mesh;
    std::optional<ObjSurface> surface;
    if (const auto k = ArgIndex(argc, argv, "--kuhn"); k && *k + 6 < size_t(argc)) {
        const dvec3 origin{ArgValue(argc, argv, "--origin", 0, 1), ArgValue(argc, argv, "--origin", 0, 2), ArgValue(argc, argv, "--origin", 0, 3)};
        mesh = KuhnBox(ArgValue(argc, argv, "--kuhn", 1, 1), ArgValue(argc, argv, "--kuhn", 1, 2), ArgValue(argc, argv, "--kuhn", 1, 3), int(ArgValue(argc, argv, "--kuhn", 1, 4)),
                       int(ArgValue(argc, argv, "--kuhn", 1, 5)), int(ArgValue(argc, argv, "--kuhn", 1, 6)), origin);
    } else if (argc >= 2 && std::string_view{argv[1]}.ends_with(".obj")) {
        surface = LoadObj(argv[1]);
        if (!surface) {
            std::fprintf(stderr, "Failed to load mesh: %s\n", argv[1]);
            return 1;
        }
        auto tets = GenerateTets(surface->Positions, surface->TriangleIndices, uint32_t(ArgValue(argc, argv, "--layers", 2)));
        if (!tets) {
            std::fprintf(stderr, "Tetrahedralization failed: %s\n", tets.Error.c_str());
            return 1;
        }
        mesh = std::move(tets.Mesh);
    } else if (argc >= 2 && argv[1][0] != '-') {
        mesh = LoadTetFile(argv[1]);
    }
    if (!mesh || mesh->Tets.empty()) {
        std::fprintf(stderr, "Usage: %s <mesh.obj> [--layers k] | <mesh.tet> | --kuhn lx ly lz nx ny nz [--origin x y z]  [--young E] [--poisson v] [--density rho] [--alpha a] [--beta b] "
                             "[--min-freq f] [--max-freq f] [--modes n] [--gltf out.gltf]\n", argv[0]);
        return 1;
    }
    const AcousticMaterialProperties material{
        .Density = ArgValue(argc, argv, "--density", 2700),
        .YoungModulus = ArgValue(argc, argv, "--young", 7.2e10),
        .PoissonRatio = ArgValue(argc, argv, "--poisson", 0.19),
        .Alpha = ArgValue(argc, argv, "--alpha", 5),
        .Beta = ArgValue(argc, argv, "--beta", 2e-8),
    };
    const modal::SolverConfig config{
        .MinModeFreq = float(ArgValue(argc, argv, "--min-freq", 20)),
        .MaxModeFreq = float(ArgValue(argc, argv, "--max-freq", 16'000)),
        .NumModes = uint32_t(ArgValue(argc, argv, "--modes", 30)),
        .NumFemModes = uint32_t(ArgValue(argc, argv, "--modes", 30)) + 15,
    };

    // the surface: boundary faces, their vertices in first-use order as excitation positions
    std::vector<uint32_t> surface_of_point(mesh->Points.size(), UINT32_MAX), triangles;
    std::vector<vec3> excite;
    std::vector<std::array<uint32_t, 3>> faces;
    if (surface) { // as the reference tool: the .obj's own vertices and triangles
        excite = surface->Positions;
        triangles = surface->TriangleIndices;
    } else {
        faces = BoundaryFaces(*mesh);
    }
    for (const auto &f : faces)
        for (const auto v : f) {
            if (surface_of_point[v] == UINT32_MAX) {
                surface_of_point[v] = uint32_t(excite.size());
                excite.emplace_back(float(mesh->Points[v].x), float(mesh->Points[v].y), float(mesh->Points[v].z));
            }
            triangles.push_back(surface_of_point[v]);
        }
    const auto result = modal::mesh2modes(*mesh, material, excite, vec3{1.f}, config);
    const auto &modes = result.Modes;
    if (modes.Freqs.empty()) {
        std::fprintf(stderr, "Solve produced no modes in [%g Hz, %g Hz]\n", config.MinModeFreq, config.MaxModeFreq);
        return 1;
    }
    // triangles relabeled onto the sample points; a triangle whose corners merged has no area and is dropped
    std::vector<uint32_t> indices;
    for (size_t t = 0; t + 2 < triangles.size(); t += 3) {
        const auto a = result.SamplePointOfExcitation[triangles[t]], b = result.SamplePointOfExcitation[triangles[t + 1]], c = result.SamplePointOfExcitation[triangles[t + 2]];
        if (a == b || b == c || a == c) continue;
        indices.insert(indices.end(), {a, b, c});
    }
    static constexpr float Ln1000 = 3 * std::numbers::ln10_v<float>;
    std::vector<float> decay_rates(modes.T60s.size());
    for (size_t k = 0; k < modes.T60s.size(); ++k) decay_rates[k] = modes.T60s[k] > 0 ? Ln1000 / modes.T60s[k] : 0.f;

    std::printf("{\n");
    PrintScalars("frequencies", modes.Freqs);
    PrintScalars("decayRates", decay_rates);
    std::printf("  \"positions\": [");
    for (size_t i = 0; i < modes.Positions.size(); ++i) std::printf("%s[%.9g,%.9g,%.9g]", i ? "," : "", modes.Positions[i].x, modes.Positions[i].y, modes.Positions[i].z);
    std::printf("],\n  \"shapes\": [");
    for (size_t k = 0; k < modes.Freqs.size(); ++k) // mode-major, as the model schema
        for (size_t i = 0; i < modes.Shapes.size(); ++i) std::printf("%s[%.9g,%.9g,%.9g]", k || i ? "," : "", modes.Shapes[i][k].x, modes.Shapes[i][k].y, modes.Shapes[i][k].z);
    std::printf("],\n");
    PrintScalars("indices", indices);
    std::printf("  \"mass\": %.17g,\n", result.MassProps.Mass);
    std::printf("  \"centerOfMass\": [%.9g,%.9g,%.9g],\n", result.MassProps.CenterOfMass.x, result.MassProps.CenterOfMass.y, result.MassProps.CenterOfMass.z);
    std::printf("  \"inertiaDiagonal\": [%.9g,%.9g,%.9g]\n}\n", result.MassProps.InertiaDiagonal.x, result.MassProps.InertiaDiagonal.y, result.MassProps.InertiaDiagonal.z);

    if (const auto g = ArgIndex(argc, argv, "--gltf"); g && *g + 1 < size_t(argc)) {
        modal::io::ModalModelDocument doc;
        doc.Materials.push_back({"solved", material});
        auto stored = modes;
        stored.Indices = indices;
        doc.Models.push_back({"solved", std::move(stored), result.MassProps, 0u});
        std::ofstream out{argv[*g + 1]};
        out << modal::io::WriteGltfModalModels(doc);
        if (!out) {
            std::fprintf(stderr, "cannot write %s\n", argv[*g + 1]);
            return 1;
        }
    }
    return 0;
}
