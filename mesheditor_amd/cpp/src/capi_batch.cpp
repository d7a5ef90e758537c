// Flat C wrapper over modal::SolveBatch (modal/batch.hpp) for the Python binding (mesheditor_amd/batch.py) -- a library of
// its own, libmodalbatch.so: it is the only host code that links RCCL and the HIP runtime directly, and a process that merely
// wants the bank, the contact model, the tet front end or the model I/O must not have to load either.
#include "modal/batch.hpp"

#include <algorithm>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

namespace {
thread_local std::string g_error;
}

extern "C" {
const char *mhx_batch_last_error() { return g_error.c_str(); }
struct mhx_batch_item {
    const double *points;
    uint32_t n_points;
    const uint32_t *tets;
    uint32_t n_tets;
    double material[5]; // density, Young, Poisson, alpha, beta
    const float *excite;
    uint32_t n_excite;
    uint32_t num_modes, num_fem_modes;
};
void mhx_batch_make_id(unsigned char *id128) {
    unsigned char id[modal::BatchComm::IdBytes];
    try { modal::BatchComm::MakeId(id); } catch (const std::exception &e) { g_error = e.what(); std::memset(id, 0, sizeof(id)); }
    std::memcpy(id128, id, sizeof(id));
}
modal::BatchComm *mhx_batch_comm_create(int world, int rank, int device, const unsigned char *id128) {
    try {
        unsigned char id[modal::BatchComm::IdBytes];
        std::memcpy(id, id128, sizeof(id));
        return new modal::BatchComm(world, rank, device, id);
    } catch (const std::exception &e) { g_error = e.what(); return nullptr; }
}
void mhx_batch_comm_destroy(modal::BatchComm *c) { delete c; }
uint64_t mhx_batch_record_length(uint32_t nev_max, uint32_t pos_max) {
    modal::BatchOptions o;
    o.MaxEigenpairs = nev_max, o.MaxPositions = pos_max;
    return modal::BatchRecordLength(o);
}
// records_out: n_items x mhx_batch_record_length doubles, the layout of mesheditor_amd/sharding.py.  0 on success.
int mhx_solve_batch(modal::BatchComm *comm, const mhx_batch_item *items, uint32_t n_items, uint32_t threads, uint32_t nev_max, uint32_t pos_max, double *records_out) {
    try {
        std::vector<TetMesh> meshes(n_items);
        std::vector<modal::BatchItem> batch(n_items);
        for (uint32_t i = 0; i < n_items; ++i) {
            const mhx_batch_item &it = items[i];
            meshes[i].Points.resize(it.n_points);
            for (uint32_t p = 0; p < it.n_points; ++p) meshes[i].Points[p] = {it.points[3 * size_t(p)], it.points[3 * size_t(p) + 1], it.points[3 * size_t(p) + 2]};
            meshes[i].Tets.resize(it.n_tets);
            for (uint32_t t = 0; t < it.n_tets; ++t) meshes[i].Tets[t] = {it.tets[4 * size_t(t)], it.tets[4 * size_t(t) + 1], it.tets[4 * size_t(t) + 2], it.tets[4 * size_t(t) + 3]};
            batch[i].Mesh = &meshes[i];
            batch[i].Material = {it.material[0], it.material[1], it.material[2], it.material[3], it.material[4]};
            for (uint32_t e = 0; e < it.n_excite; ++e) batch[i].ExcitePositions.push_back({it.excite[3 * e], it.excite[3 * e + 1], it.excite[3 * e + 2]});
            batch[i].Config.NumModes = it.num_modes;
            batch[i].Config.NumFemModes = it.num_fem_modes;
        }
        modal::BatchOptions o;
        o.ThreadsPerDevice = threads, o.MaxEigenpairs = nev_max, o.MaxPositions = pos_max;
        const std::vector<double> raw = modal::SolveBatchRaw(batch, *comm, o);
        std::copy(raw.begin(), raw.end(), records_out);
        return 0;
    } catch (const std::exception &e) { g_error = e.what(); return 1; }
}
}
