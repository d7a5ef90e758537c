// modal::SolveBatch -- a batch of independent meshes over the GPUs of one node, host side in C++ (the
// reference solves one entity per job, several at a time: src/audio/AudioSystem.cpp:812-865).
//
// One process per GPU.  Every rank is given the whole batch, deals it with the reference's own LPT rule
// (src/audio/ModalAudio.cpp:450-458: items by descending cost onto the least loaded bin), solves its share on its GPU with a
// few host threads (one modalhip context each: the solves of different contexts iterate side by side), packs one fixed-size
// record per mesh and joins ONE ncclAllGather (RCCL over xGMI) of the records, on the device.  No other collective touches the
// analysis path.  A mesh whose solve fails travels as a record with its status: every rank still reaches the collective, and
// the failure is reported after it, on all ranks alike.
#pragma once
#include "solver.hpp"

#include <cstdint>
#include <span>
#include <string>
#include <vector>

namespace modal {
struct BatchItem {
    const TetMesh *Mesh{};
    AcousticMaterialProperties Material{};
    std::vector<vec3> ExcitePositions;
    vec3 BakedScale{1.f, 1.f, 1.f};
    SolverConfig Config{};
};

// The communicator of the ranks that share the batch: created from an id every rank was handed (ncclUniqueId, 128 bytes;
// rank 0 makes it with BatchComm::MakeId and ships it by whatever the launcher offers -- a file, a socket, a store).
class BatchComm {
public:
    static constexpr size_t IdBytes = 128;
    static void MakeId(unsigned char (&id)[IdBytes]);
    BatchComm(int world_size, int rank, int device, const unsigned char (&id)[IdBytes]); // ncclCommInitRank; world_size 1 works too
    ~BatchComm();
    BatchComm(const BatchComm &) = delete;
    BatchComm &operator=(const BatchComm &) = delete;
    int WorldSize() const { return World; }
    int Rank() const { return Me; }
    int Device() const { return Dev; }
    // every rank contributes `count` doubles from `send` (device memory) and receives world x count into `recv`
    void AllGather(const double *send, double *recv, size_t count);
    // waits for the gather; RCCL's asynchronous error state or `timeout_seconds` without completion abort the communicator and throw
    void Synchronize(double timeout_seconds = 600.0);
    void Abort(); // ncclCommAbort: the other ranks' pending collective fails instead of waiting for this one

private:
    int World{1}, Me{0}, Dev{0};
    void *Comm{};   // ncclComm_t
    void *Stream{}; // hipStream_t
};

struct BatchOptions {
    uint32_t ThreadsPerDevice{3}; // solves in flight per GPU
    uint32_t MaxEigenpairs{256}, MaxPositions{16}; // record capacity
    double GatherTimeoutSeconds{600.0}; // watchdog of the one collective (a rank that never joins must not hang the others)
};

struct BatchRecord { // what ModalResult carries per mesh, minus the optional basis
    uint32_t Index{};
    bool Ok{};
    double Seconds{};
    ModalResult Result; // Modes (freqs, t60s, shapes, positions), MassProps, Profile, Summary.Eigenvalues / Shapes
};

// Record layout (doubles), shared with mesheditor_amd/sharding.py: header {index, status, nev, kept modes, positions, seconds,
// original fundamental}, mass properties (11), profile (12), eigenvalues[E], freqs[E], t60s[E], positions[3 P],
// shapes[P][E][3] with E = MaxEigenpairs, P = MaxPositions.
size_t BatchRecordLength(const BatchOptions &);
std::vector<uint32_t> DealBatch(std::span<const double> costs, uint32_t bins); // item -> bin, the LPT rule
double MeshCost(size_t tets, uint32_t eigenpairs);

// All ranks return every mesh's record, ordered by index.  The ranks of `comm` must call it with the same batch.
std::vector<BatchRecord> SolveBatch(std::span<const BatchItem> items, BatchComm &comm, const BatchOptions &options = {});
// The same, the gathered records as raw doubles (items x BatchRecordLength): what the C wrapper hands to Python
std::vector<double> SolveBatchRaw(std::span<const BatchItem> items, BatchComm &comm, const BatchOptions &options = {});
} // namespace modal
