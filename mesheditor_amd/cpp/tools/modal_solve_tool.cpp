// Solve a tetrahedral mesh's modal model on the MI355X and print it as JSON on stdout: the fields of the reference's
// MeshEditorModalSolve (tests/ModalSolveTool.cpp:101-123 -- frequencies, decayRates, positions, mode-major shapes,
// indices, mass, centerOfMass, inertiaDiagonal), which its sample generator shells out to.  The reference tool starts
// from a surface .obj and tetrahedralises it; here an .obj is filled the same way (GenerateTets,
// modal/tets.hpp) and its vertices are the excitation positions, or the tet mesh is given directly:
//   modal_solve <mesh.obj> [--quality | --layers k] [options]   general fill (tetra::Tetrahedralize), or the layered fill of a star-shaped surface
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
#include <vector>

namespace {
// "--name v1 v2 ..." options anywhere on the command line; numbers are read with from_chars (no locale, no exceptions).
class CommandLine {
public:
    CommandLine(int argc, char **argv) : Words(argv, argv + argc) {}
    std::optional<size_t> Find(std::string_view name) const {
        for (size_t i = 1; i < Words.size(); ++i)
            if (Words[i] == name) return i;
        return std::nullopt;
    }
    // the nth word after `name` as a number, or `fallback` when the option is absent or malformed
    double Number(std::string_view name, double fallback, size_t nth = 1) const {
        const auto at = Find(name);
        if (!at || *at + nth >= Words.size()) return fallback;
        const std::string_view text = Words[*at + nth];
        double parsed = 0;
        const auto [end, err] = std::from_chars(text.data(), text.data() + text.size(), parsed);
        return err == std::errc{} ? parsed : fallback;
    }
    const char *Text(std::string_view name) const {
        const auto at = Find(name);
        return at && *at + 1 < Words.size() ? Words[*at + 1].data() : nullptr;
    }
    bool HasValues(std::string_view name, size_t count) const {
        const auto at = Find(name);
        return at && *at + count < Words.size();
    }
    std::string_view Positional() const { return Words.size() > 1 && !Words[1].starts_with('-') ? Words[1] : std::string_view{}; }
    std::string_view Program() const { return Words.front(); }

private:
    std::vector<std::string_view> Words;
};

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

// Minimal JSON emitter for the flat document the sample generator consumes: one key per line, numbers with nine
// significant digits (enough to round-trip a float).
class JsonObject {
public:
    JsonObject() { std::printf("{\n"); }
    ~JsonObject() { std::printf("\n}\n"); }
    template<typename Seq> void Numbers(const char *key, const Seq &values) {
        Key(key);
        std::printf("[");
        const char *sep = "";
        for (const auto v : values) std::printf("%s%.9g", sep, double(v)), sep = ",";
        std::printf("]");
    }
    // `count` triples produced by at(i)
    template<typename At> void Triples(const char *key, size_t count, At &&at) {
        Key(key);
        std::printf("[");
        for (size_t i = 0; i < count; ++i) {
            const vec3 v = at(i);
            std::printf("%s[%.9g,%.9g,%.9g]", i ? "," : "", v.x, v.y, v.z);
        }
        std::printf("]");
    }
    void Triple(const char *key, vec3 v) {
        Key(key);
        std::printf("[%.9g,%.9g,%.9g]", v.x, v.y, v.z);
    }
    void Exact(const char *key, double v) {
        Key(key);
        std::printf("%.17g", v);
    }

private:
    bool Any{false};
    void Key(const char *key) {
        std::printf("%s  \"%s\": ", Any ? ",\n" : "", key);
        Any = true;
    }
};

// What is solved: the tets, where the body is excited, and the surface triangles over those excitation points.
struct SolveInput {
    TetMesh Mesh;
    std::vector<vec3> Excite;
    std::vector<uint32_t> Triangles;
};

// From a surface .obj (the reference tool's input): its vertices are the excitation points and its triangles the
// surface.  From bare tets: the boundary faces, their corners numbered in first-use order.
std::optional<SolveInput> ReadInput(const CommandLine &cl) {
    SolveInput in;
    const std::string_view path = cl.Positional();
    if (cl.HasValues("--kuhn", 6)) {
        const auto k = [&](size_t nth) { return cl.Number("--kuhn", 1, nth); };
        const dvec3 origin{cl.Number("--origin", 0, 1), cl.Number("--origin", 0, 2), cl.Number("--origin", 0, 3)};
        in.Mesh = KuhnBox(k(1), k(2), k(3), int(k(4)), int(k(5)), int(k(6)), origin);
    } else if (path.ends_with(".obj")) {
        auto surface = LoadObj(path.data());
        if (!surface) {
            std::fprintf(stderr, "Failed to load mesh: %s\n", path.data());
            return std::nullopt;
        }
        // as the reference's tool (tests/ModalSolveTool.cpp:72): the general fill, --quality = tetra::Options::Quality; with --layers k the
        // layered fill of a star-shaped surface instead
        auto filled = cl.Find("--layers") ? GenerateTets(surface->Positions, surface->TriangleIndices, uint32_t(cl.Number("--layers", 2)))
                                          : GenerateTets(surface->Positions, surface->TriangleIndices, {.Quality = cl.Find("--quality").has_value()});
        if (!filled) {
            std::fprintf(stderr, "Tetrahedralization failed: %s\n", filled.error().c_str());
            return std::nullopt;
        }
        in.Mesh = std::move(filled->Mesh);
        in.Excite = std::move(surface->Positions);
        in.Triangles = std::move(surface->TriangleIndices);
        return in;
    } else if (!path.empty()) {
        if (auto loaded = LoadTetFile(path.data())) in.Mesh = std::move(*loaded);
    }
    if (in.Mesh.Tets.empty()) return std::nullopt;
    std::vector<uint32_t> label(in.Mesh.Points.size(), UINT32_MAX);
    for (const auto &face : BoundaryFaces(in.Mesh))
        for (const uint32_t v : face) {
            if (label[v] == UINT32_MAX) {
                label[v] = uint32_t(in.Excite.size());
                const dvec3 &p = in.Mesh.Points[v];
                in.Excite.emplace_back(float(p.x), float(p.y), float(p.z));
            }
            in.Triangles.push_back(label[v]);
        }
    return in;
}
} // namespace

int main(int argc, char **argv) {
    const CommandLine cl(argc, argv);
    const auto input = ReadInput(cl);
    if (!input) {
        std::fprintf(stderr, "Usage: %s <mesh.obj> [--layers k] | <mesh.tet> | --kuhn lx ly lz nx ny nz [--origin x y z]  [--young E] [--poisson v] [--density rho] [--alpha a] [--beta b] "
                             "[--min-freq f] [--max-freq f] [--modes n] [--gltf out.gltf] [--write-tets out.tet [--tets-only]]\n", cl.Program().data());
        return 1;
    }
    // "--write-tets file": the tet mesh the solve would run on, in the tool's own plain format (counts, points, tets; doubles at
    // full precision); with "--tets-only" nothing is solved and no device is touched
    if (const char *tets_path = cl.Text("--write-tets")) {
        std::FILE *f = std::fopen(tets_path, "w");
        if (!f) {
            std::fprintf(stderr, "Cannot write %s\n", tets_path);
            return 1;
        }
        std::fprintf(f, "%zu %zu\n", input->Mesh.Points.size(), input->Mesh.Tets.size());
        for (const auto &q : input->Mesh.Points) std::fprintf(f, "%.17g %.17g %.17g\n", q.x, q.y, q.z);
        for (const auto &t : input->Mesh.Tets) std::fprintf(f, "%u %u %u %u\n", t[0], t[1], t[2], t[3]);
        std::fclose(f);
        if (cl.Find("--tets-only")) return 0;
    }
    // defaults as the reference tool: ceramic-like solid, the audible window, 30 kept modes out of 45 solved
    AcousticMaterialProperties material{};
    material.Density = cl.Number("--density", 2700);
    material.YoungModulus = cl.Number("--young", 7.2e10);
    material.PoissonRatio = cl.Number("--poisson", 0.19);
    material.Alpha = cl.Number("--alpha", 5);
    material.Beta = cl.Number("--beta", 2e-8);
    modal::SolverConfig config{};
    config.MinModeFreq = float(cl.Number("--min-freq", 20));
    config.MaxModeFreq = float(cl.Number("--max-freq", 16'000));
    config.NumModes = uint32_t(cl.Number("--modes", 30));
    config.NumFemModes = config.NumModes + 15;

    const auto result = modal::mesh2modes(input->Mesh, material, input->Excite, vec3{1.f}, config);
    const ModalModes &modes = result.Modes;
    if (modes.Freqs.empty()) {
        std::fprintf(stderr, "Solve produced no modes in [%g Hz, %g Hz]\n", config.MinModeFreq, config.MaxModeFreq);
        return 1;
    }
    // Surface triangles over the sample points the excitation positions became; corners that merged leave no area.
    std::vector<uint32_t> indices;
    for (size_t t = 0; t + 2 < input->Triangles.size(); t += 3) {
        std::array<uint32_t, 3> tri;
        for (int c = 0; c < 3; ++c) tri[c] = result.SamplePointOfExcitation[input->Triangles[t + c]];
        if (tri[0] != tri[1] && tri[1] != tri[2] && tri[0] != tri[2]) indices.insert(indices.end(), tri.begin(), tri.end());
    }
    // decay rate (1/s) of an amplitude that falls 60 dB in T60 seconds; an undamped mode (T60 = 0) states 0
    std::vector<float> decay_rates;
    for (const float t60 : modes.T60s) decay_rates.push_back(t60 > 0 ? 3 * std::numbers::ln10_v<float> / t60 : 0.f);

    {
        JsonObject json;
        json.Numbers("frequencies", modes.Freqs);
        json.Numbers("decayRates", decay_rates);
        json.Triples("positions", modes.Positions.size(), [&](size_t i) { return modes.Positions[i]; });
        const size_t n_points = modes.Shapes.size(); // mode-major, as the model schema: every sample point of mode 0, then of mode 1, ...
        json.Triples("shapes", modes.Freqs.size() * n_points, [&](size_t at) { return modes.Shapes[at % n_points][at / n_points]; });
        json.Numbers("indices", indices);
        json.Exact("mass", result.MassProps.Mass);
        json.Triple("centerOfMass", result.MassProps.CenterOfMass);
        json.Triple("inertiaDiagonal", result.MassProps.InertiaDiagonal);
    }
    if (const char *path = cl.Text("--gltf")) {
        modal::io::ModalModelDocument doc;
        doc.Materials.push_back({"solved", material});
        ModalModes stored = modes;
        stored.Indices = indices;
        doc.Models.push_back({"solved", std::move(stored), result.MassProps, 0u});
        std::ofstream out{path};
        out << modal::io::WriteGltfModalModels(doc);
        if (!out) {
            std::fprintf(stderr, "cannot write %s\n", path);
            return 1;
        }
    }
    return 0;
}
