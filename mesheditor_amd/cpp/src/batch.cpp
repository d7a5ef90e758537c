// modal::SolveBatch: LPT deal, worker threads on this rank's GPU, one ncclAllGather of fixed-size records (modal/batch.hpp).
#include "modal/batch.hpp"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <thread>

namespace modal {
namespace {
constexpr size_t HeaderWords = 8, MassWords = 11, ProfileWords = 12;
constexpr double StatusOk = 0.0, StatusFailed = 1.0;

void Check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("SolveBatch: ") + what + ": " + hipGetErrorString(e));
}
void Check(ncclResult_t r, const char *what) {
    if (r != ncclSuccess) throw std::runtime_error(std::string("SolveBatch: ") + what + ": " + ncclGetErrorString(r));
}

void Pack(double *rec, uint32_t index, const ModalResult &r, double seconds, const BatchOptions &o) {
    const size_t E = o.MaxEigenpairs, P = o.MaxPositions;
    const size_t nev = r.Summary.Eigenvalues.size(), kept = r.Modes.Freqs.size(), n_pos = r.Modes.Positions.size();
    const bool fits = nev <= E && n_pos <= P && nev > 0;
    rec[0] = index;
    rec[1] = fits ? StatusOk : StatusFailed;
    rec[5] = seconds;
    if (!fits) return;
    rec[2] = double(nev);
    rec[3] = double(kept);
    rec[4] = double(n_pos);
    rec[6] = r.Modes.OriginalFundamentalFreq;
    double *m = rec + HeaderWords;
    m[0] = r.MassProps.Mass;
    for (int i = 0; i < 3; ++i) m[1 + i] = r.MassProps.CenterOfMass[i], m[4 + i] = r.MassProps.InertiaDiagonal[i];
    m[7] = r.MassProps.InertiaOrientation.w, m[8] = r.MassProps.InertiaOrientation.x, m[9] = r.MassProps.InertiaOrientation.y, m[10] = r.MassProps.InertiaOrientation.z;
    double *p = m + MassWords;
    const SolveProfile &pr = r.Profile;
    const double prof[ProfileWords] = {pr.MassProps, pr.QuadMesh, pr.Assemble, pr.SampleExcite, pr.Factorize, pr.Iterate, pr.OpSolve, pr.Extract,
                                       double(pr.Dofs), double(pr.StiffnessNonZeros), double(pr.OpApplications), double(pr.Restarts)};
    std::copy(prof, prof + ProfileWords, p);
    double *ev = p + ProfileWords;
    std::copy(r.Summary.Eigenvalues.begin(), r.Summary.Eigenvalues.end(), ev);
    for (size_t k = 0; k < kept; ++k) ev[E + k] = r.Modes.Freqs[k], ev[2 * E + k] = r.Modes.T60s[k];
    double *pos = ev + 3 * E;
    for (size_t q = 0; q < n_pos; ++q)
        for (int i = 0; i < 3; ++i) pos[3 * q + i] = r.Modes.Positions[q][i];
    double *sh = pos + 3 * P; // [position][eigenpair][3] at the record's full extents
    for (size_t q = 0; q < std::min(n_pos, r.Summary.Shapes.size()); ++q)
        for (size_t k = 0; k < std::min(nev, r.Summary.Shapes[q].size()); ++k)
            for (int i = 0; i < 3; ++i) sh[(q * E + k) * 3 + i] = r.Summary.Shapes[q][k][i];
}

BatchRecord Unpack(const double *rec, const BatchOptions &o) {
    const size_t E = o.MaxEigenpairs, P = o.MaxPositions;
    BatchRecord out;
    out.Index = uint32_t(rec[0]);
    out.Ok = rec[1] == StatusOk;
    out.Seconds = rec[5];
    if (!out.Ok) return out;
    const size_t nev = size_t(rec[2]), kept = size_t(rec[3]), n_pos = size_t(rec[4]);
    ModalResult &r = out.Result;
    r.Modes.OriginalFundamentalFreq = float(rec[6]);
    const double *m = rec + HeaderWords;
    r.MassProps.Mass = m[0];
    for (int i = 0; i < 3; ++i) r.MassProps.CenterOfMass[i] = float(m[1 + i]), r.MassProps.InertiaDiagonal[i] = float(m[4 + i]);
    r.MassProps.InertiaOrientation = {float(m[7]), float(m[8]), float(m[9]), float(m[10])};
    const double *p = m + MassWords;
    SolveProfile &pr = r.Profile;
    pr.MassProps = p[0], pr.QuadMesh = p[1], pr.Assemble = p[2], pr.SampleExcite = p[3], pr.Factorize = p[4], pr.Iterate = p[5], pr.OpSolve = p[6], pr.Extract = p[7];
    pr.Dofs = uint32_t(p[8]), pr.StiffnessNonZeros = uint32_t(p[9]), pr.OpApplications = uint32_t(p[10]), pr.Restarts = uint32_t(p[11]);
    const double *ev = p + ProfileWords;
    r.Summary.Eigenvalues.assign(ev, ev + nev);
    r.Modes.Freqs.resize(kept);
    r.Modes.T60s.resize(kept);
    for (size_t k = 0; k < kept; ++k) r.Modes.Freqs[k] = float(ev[E + k]), r.Modes.T60s[k] = float(ev[2 * E + k]);
    const double *pos = ev + 3 * E;
    r.Modes.Positions.resize(n_pos);
    for (size_t q = 0; q < n_pos; ++q) r.Modes.Positions[q] = {float(pos[3 * q]), float(pos[3 * q + 1]), float(pos[3 * q + 2])};
    const double *sh = pos + 3 * P;
    r.Summary.Shapes.assign(n_pos, std::vector<vec3>(nev));
    for (size_t q = 0; q < n_pos; ++q)
        for (size_t k = 0; k < nev; ++k) r.Summary.Shapes[q][k] = {float(sh[(q * E + k) * 3]), float(sh[(q * E + k) * 3 + 1]), float(sh[(q * E + k) * 3 + 2])};
    return out;
}
} // namespace

void BatchComm::MakeId(unsigned char (&id)[IdBytes]) {
    static_assert(sizeof(ncclUniqueId) == IdBytes);
    ncclUniqueId u;
    Check(ncclGetUniqueId(&u), "ncclGetUniqueId");
    std::memcpy(id, &u, IdBytes);
}
BatchComm::BatchComm(int world_size, int rank, int device, const unsigned char (&id)[IdBytes]) : World(world_size), Me(rank), Dev(device) {
    Check(hipSetDevice(device), "hipSetDevice");
    (void)hipGetLastError(); // RCCL's set-up reads the thread's last HIP error: one left behind by unrelated earlier work is not its business
    ncclUniqueId u;
    std::memcpy(&u, id, IdBytes);
    ncclComm_t c{};
    Check(ncclCommInitRank(&c, world_size, u, rank), "ncclCommInitRank");
    Comm = c;
    hipStream_t s{};
    Check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
    Stream = s;
}
BatchComm::~BatchComm() {
    if (Stream) (void)hipStreamDestroy(static_cast<hipStream_t>(Stream));
    if (Comm) (void)ncclCommDestroy(static_cast<ncclComm_t>(Comm));
}
void BatchComm::AllGather(const double *send, double *recv, size_t count) {
    if (!Comm) throw std::runtime_error("SolveBatch: the communicator was aborted");
    Check(ncclAllGather(send, recv, count, ncclDouble, static_cast<ncclComm_t>(Comm), static_cast<hipStream_t>(Stream)), "ncclAllGather");
}
// Waits for the collective with a watchdog: a rank that died before joining would otherwise leave the others in the all-gather
// for ever.  The stream is polled; RCCL's asynchronous error state or the timeout abort the communicator and throw.
void BatchComm::Synchronize(double timeout_seconds) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(static_cast<hipStream_t>(Stream));
        if (q == hipSuccess) return;
        if (q != hipErrorNotReady) {
            Abort();
            Check(q, "hipStreamQuery");
        }
        ncclResult_t async = ncclSuccess;
        if (Comm && ncclCommGetAsyncError(static_cast<ncclComm_t>(Comm), &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress) {
            Abort();
            Check(async, "ncclAllGather (asynchronous error)");
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_seconds) {
            Abort();
            throw std::runtime_error("SolveBatch: the gather did not complete within " + std::to_string(int(timeout_seconds)) + " s (a rank never joined); communicator aborted");
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}
void BatchComm::Abort() {
    if (Comm) (void)ncclCommAbort(static_cast<ncclComm_t>(Comm));
    Comm = nullptr;
}

size_t BatchRecordLength(const BatchOptions &o) {
    return HeaderWords + MassWords + ProfileWords + 3 * size_t(o.MaxEigenpairs) + 3 * size_t(o.MaxPositions) + 3 * size_t(o.MaxPositions) * o.MaxEigenpairs;
}
double MeshCost(size_t tets, uint32_t eigenpairs) { return double(tets) * double(eigenpairs); }

std::vector<uint32_t> DealBatch(std::span<const double> costs, uint32_t bins) {
    std::vector<uint32_t> order(costs.size()), bin(costs.size(), 0);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return costs[a] > costs[b]; }); // heaviest first, ties by index
    std::vector<double> load(std::max(1u, bins), 0.0);
    for (const uint32_t i : order) {
        const size_t r = size_t(std::min_element(load.begin(), load.end()) - load.begin()); // least loaded, lowest bin on ties
        load[r] += costs[i];
        bin[i] = uint32_t(r);
    }
    return bin;
}

std::vector<double> SolveBatchRaw(std::span<const BatchItem> items, BatchComm &comm, const BatchOptions &options) {
    const size_t reclen = BatchRecordLength(options), n = items.size();
    const int world = comm.WorldSize(), rank = comm.Rank();
    std::vector<double> costs(n);
    for (size_t i = 0; i < n; ++i) costs[i] = items[i].Mesh ? MeshCost(items[i].Mesh->Tets.size(), items[i].Config.NumFemModes) : 0.0;
    const std::vector<uint32_t> bin = DealBatch(costs, uint32_t(world));
    std::vector<uint32_t> mine;
    std::vector<size_t> share(world, 0);
    for (size_t i = 0; i < n; ++i) {
        ++share[bin[i]];
        if (int(bin[i]) == rank) mine.push_back(uint32_t(i)); // ascending within the rank
    }
    const size_t slots = *std::max_element(share.begin(), share.end()); // every rank sends the same count: the largest share
    // [slot][1 + reclen]: word 0 says whether the slot is used
    std::vector<double> send(slots * (reclen + 1), 0.0);
    // The gather's device buffers are taken BEFORE the solves: a rank that cannot have them fails here, at once, not after minutes of
    // solving -- and once it has them nothing between here and the collective allocates on the device outside the solves' own pools,
    // so a rank whose solves all fail still joins with failed records.  (A rank that fails HERE aborts the communicator; RCCL's abort is
    // local, the peers learn of it through their watchdog: an aborted communicator must be recreated on every rank.)
    const size_t count = slots * (reclen + 1);
    // Owned from here to the end of the function, whatever is thrown in between (a thread that cannot start, a host allocation): freed on
    // every path; and a throw before the collective aborts the communicator, so that the peers' watchdogs fire instead of their waiting.
    struct DeviceBuffer {
        double *Ptr{};
        ~DeviceBuffer() { if (Ptr) (void)hipFree(Ptr); }
    } d_send, d_recv;
    struct AbortUnlessJoined {
        BatchComm &Comm;
        bool Joined{false};
        ~AbortUnlessJoined() { if (!Joined) Comm.Abort(); }
    } joining{comm};
    // the gather's host buffer too, before the solves: nothing between the solves and the collective allocates
    std::vector<double> all(size_t(world) * count, 0.0);
    if (count) {
        Check(hipSetDevice(comm.Device()), "hipSetDevice");
        Check(hipMalloc(reinterpret_cast<void **>(&d_send.Ptr), count * sizeof(double)), "hipMalloc");
        Check(hipMalloc(reinterpret_cast<void **>(&d_recv.Ptr), size_t(world) * count * sizeof(double)), "hipMalloc");
    }
    std::atomic<size_t> next{0};
    const uint32_t workers = std::max<uint32_t>(1, std::min<uint32_t>(options.ThreadsPerDevice, uint32_t(std::max<size_t>(mine.size(), 1))));
    // Nothing a worker does may keep this rank from the collective: whatever is thrown (by the device selection, a solve, the
    // packing -- std::exception or not) turns into failed records for the slots concerned, and the rank joins the gather.
    const auto work = [&] {
        bool device_ok = true;
        try {
            SetDevice(comm.Device()); // this thread's modalhip context lives on the rank's GPU
        } catch (...) { device_ok = false; }
        for (size_t k = next.fetch_add(1); k < mine.size(); k = next.fetch_add(1)) {
            const uint32_t i = mine[k];
            double *slot = send.data() + k * (reclen + 1);
            const auto t0 = std::chrono::steady_clock::now();
            ModalResult r;
            try {
                const BatchItem &it = items[i];
                if (device_ok && it.Mesh) r = mesh2modes(*it.Mesh, it.Material, it.ExcitePositions, it.BakedScale, it.Config);
            } catch (...) { // (a failed factorisation throws, as the reference's does): the record says failed
                r = ModalResult{};
            }
            try {
                Pack(slot + 1, i, r, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), options);
            } catch (...) {
                std::fill(slot + 1, slot + 1 + reclen, 0.0);
                Pack(slot + 1, i, ModalResult{}, 0.0, options); // (an empty result packs without allocating)
            }
            slot[0] = 1.0;
        }
    };
    {
        std::vector<std::thread> pool;
        try {
            for (uint32_t t = 1; t < workers; ++t) pool.emplace_back(work);
        } catch (...) { // (std::system_error: no more threads) -- the ones that started and this one do the work
        }
        work();
        for (auto &t : pool) t.join();
    }
    // the one collective: every rank's slots, on the device, over RCCL
    if (count) {
        Check(hipSetDevice(comm.Device()), "hipSetDevice");
        Check(hipMemcpy(d_send.Ptr, send.data(), count * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy"); // (a failed copy into memory we hold: the device is gone -- nothing to join with: abort)
        joining.Joined = true; // from here the collective itself reports what goes wrong (its watchdog aborts)
        comm.AllGather(d_send.Ptr, d_recv.Ptr, count);
        comm.Synchronize(options.GatherTimeoutSeconds);
        Check(hipMemcpy(all.data(), d_recv.Ptr, all.size() * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy");
    }
    joining.Joined = true;
    std::vector<double> out(n * reclen, 0.0);
    std::vector<uint8_t> seen(n, 0);
    for (size_t s = 0; s < size_t(world) * slots; ++s) {
        const double *slot = all.data() + s * (reclen + 1);
        if (slot[0] != 1.0) continue;
        const size_t i = size_t(slot[1]);
        if (i >= n) throw std::runtime_error("SolveBatch: a gathered record carries an index outside the batch");
        std::copy(slot + 1, slot + 1 + reclen, out.begin() + i * reclen);
        seen[i] = 1;
    }
    if (std::count(seen.begin(), seen.end(), uint8_t(1)) != std::ptrdiff_t(n)) throw std::runtime_error("SolveBatch: the gather did not return one record per mesh");
    return out;
}

std::vector<BatchRecord> SolveBatch(std::span<const BatchItem> items, BatchComm &comm, const BatchOptions &options) {
    const std::vector<double> raw = SolveBatchRaw(items, comm, options);
    const size_t reclen = BatchRecordLength(options);
    std::vector<BatchRecord> out;
    out.reserve(items.size());
    for (size_t i = 0; i < items.size(); ++i) out.push_back(Unpack(raw.data() + i * reclen, options));
    return out;
}
} // namespace modal
