// modal::SolveBatch (modal/batch.hpp) on one GPU with RCCL initialised (a one-rank communicator: ncclCommInitRank + one
// ncclAllGather per batch): every mesh's record must equal what a direct modal::mesh2modes call returns, a mesh that cannot
// be solved must come back as a failed record without keeping the others from arriving, and the deal must be the LPT rule.
#include "harness.hpp"

#include <audio/mesh2modes.h>
#include <mesh/TetMesh.h>
#include <modal/batch.hpp>

#include <array>
#include <vector>

namespace {
TetMesh Box(int n, double lx, double ly, double lz) {
    TetMesh mesh;
    const int s_j = n + 1, s_i = (n + 1) * (n + 1);
    for (int i = 0; i <= n; ++i)
        for (int j = 0; j <= n; ++j)
            for (int k = 0; k <= n; ++k) mesh.Points.push_back({lx * i / n, ly * j / n, lz * k / n});
    static constexpr int Paths[6][4]{{0, 1, 3, 7}, {0, 3, 2, 7}, {0, 2, 6, 7}, {0, 6, 4, 7}, {0, 4, 5, 7}, {0, 5, 1, 7}};
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < n; ++k) {
                const uint32_t o = uint32_t(i * s_i + j * s_j + k);
                const auto c = [&](int q) { return o + uint32_t((q & 1) * s_i + ((q >> 1) & 1) * s_j + (q >> 2)); };
                for (const auto &p : Paths) mesh.Tets.push_back({c(p[0]), c(p[1]), c(p[2]), c(p[3])});
            }
    return mesh;
}
} // namespace

CASE(deal_is_the_longest_processing_time_rule) {
    const std::vector<double> costs{5, 9, 9, 1, 4, 7};
    const auto bin = modal::DealBatch(costs, 3);
    // heaviest first, ties by index, onto the least loaded bin (lowest on ties): 9->0, 9->1, 7->2, 5->2, 4->0, 1->1
    const std::vector<uint32_t> want{2, 0, 1, 1, 0, 2};
    EXPECT(bin == want);
    EXPECT(modal::DealBatch(costs, 1) == std::vector<uint32_t>(6, 0u));
}

CASE(batch_on_one_rank_equals_the_direct_solves_and_survives_a_failing_mesh) {
    unsigned char id[modal::BatchComm::IdBytes];
    modal::BatchComm::MakeId(id);
    modal::BatchComm comm(1, 0, 0, id);
    std::vector<TetMesh> meshes{Box(5, 0.2, 0.15, 0.1), Box(4, 0.1, 0.1, 0.1), Box(6, 0.3, 0.1, 0.05)};
    TetMesh flat; // every tet degenerate: the solve returns an empty result
    flat.Points = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {1, 1, 0}};
    flat.Tets = {{0, 1, 2, 3}};
    meshes.insert(meshes.begin() + 1, flat);
    std::vector<modal::BatchItem> items;
    const auto mats = materials::acoustic::All;
    for (size_t i = 0; i < meshes.size(); ++i) {
        modal::BatchItem it;
        it.Mesh = &meshes[i];
        it.Material = mats[i % mats.size()].Properties;
        for (size_t p = 0; p < meshes[i].Points.size(); p += 7) it.ExcitePositions.push_back({float(meshes[i].Points[p].x), float(meshes[i].Points[p].y), float(meshes[i].Points[p].z)});
        it.Config.NumModes = 12;
        it.Config.NumFemModes = 24;
        it.Config.MaxModeFreq = 1e6f;
        items.push_back(std::move(it));
    }
    modal::BatchOptions options;
    options.ThreadsPerDevice = 2;
    options.MaxEigenpairs = 32;
    options.MaxPositions = 64;
    const auto records = modal::SolveBatch(items, comm, options);
    EXPECT(records.size() == items.size());
    for (size_t i = 0; i < records.size(); ++i) {
        EXPECT(records[i].Index == i);
        if (i == 1) {
            EXPECT(!records[i].Ok);
            continue;
        }
        EXPECT(records[i].Ok);
        const auto direct = modal::mesh2modes(meshes[i], items[i].Material, items[i].ExcitePositions, vec3{1, 1, 1}, items[i].Config);
        const auto &got = records[i].Result;
        EXPECT(got.Summary.Eigenvalues == direct.Summary.Eigenvalues); // the same deterministic solve: bit for bit
        EXPECT(got.Modes.Freqs == direct.Modes.Freqs);
        EXPECT(got.Modes.T60s == direct.Modes.T60s);
        EXPECT(got.Modes.Positions == direct.Modes.Positions);
        EXPECT(got.MassProps == direct.MassProps);
        EXPECT(got.Summary.Shapes == direct.Summary.Shapes);
        EXPECT(got.Profile.Dofs == direct.Profile.Dofs && got.Profile.Restarts == direct.Profile.Restarts);
    }
}

int main() { return check::run_all(); }
