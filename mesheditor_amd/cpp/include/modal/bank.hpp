// Modal synthesis bank behind the reference's API surface (src/audio/ModalAudio.h): ModalEvent, ModalBank (public
// struct-of-arrays columns callers write directly), ModalAudio, and the free functions AddModalObject /
// TuneModalObject / InstallModalBank / SetModalObjectShapes / FindModalObject / EnqueueModalEvent / RenderModal,
// plus the recoil-filter helpers (ModalAudio.h:58-99).
//
// The host bank is the source of truth.  InstallModalBank mirrors it into HBM; RenderModal runs each block on the
// MI355X (mh_bank_render) with one device synchronisation per block.  There is no CPU render path.
//
// Two precisions share one implementation: ModalBank / ModalAudio are the reference's fp32 layout; ModalBank64 /
// ModalAudio64 keep every column, the impact list and the output in double (BASELINE north star: "resonator output
// sample-exact at fp64").  Declarations below follow the reference's names field for field -- that is the contract.
#pragma once
#include "types.hpp"

#include <entt/entity/fwd.hpp>

#include <array>
#include <atomic>
#include <memory>
#include <numbers>
#include <optional>
#include <span>

struct mh_bank;
struct mh_context;

enum class ModalEventKind : uint32_t { Impact, Silence };

// ModalAudio.h:28-37.  Event fields stay float in both precisions (they are produced by float strike code).
struct ModalEvent {
    ModalEventKind Kind{ModalEventKind::Impact};
    uint32_t Object{0}, ExPos{0};
    float Jx{0}, Jy{0}, Jz{0};
    float PulseStep{0}, PulseGamma{0}, AccelAmp{0};
    float ClickB0{0}, ClickA1{0}, ClickA2{0};
};

constexpr float AirDensity{1.204f}, SpeedOfSound{343.f}, ListenerDistance{1.f};
constexpr float Ln1000 = 3 * std::numbers::ln10_v<float>;

// ---- recoil radiator filters (ModalAudio.h:58-99): bilinear transforms evaluated in double, stored as float ----
struct RecoilPoles {
    double A0{0};
    float A1{0}, A2{0};
};
RecoilPoles RecoilDenominator(double wc, double kk, double beta);
struct RecoilFilter {
    float RadB0{0}, AirB0{0}, AirB1{0}, AirB2{0}, A1{0}, A2{0};
};
RecoilFilter RecoilObjectFilter(double radius, double volume, double sample_rate);
struct ClickFilter {
    float B0{0}, A1{0}, A2{0};
};
ClickFilter RecoilClickFilter(double radius, double volume, double mass, double sample_rate);

// Not in the reference: the half-open range of per-mode (or shape) entries written since the device mirror last saw
// them.  Writers widen it from any thread; the render claims it at block start and uploads exactly that range.
class ModalEditSpan {
public:
    ModalEditSpan() = default;
    ModalEditSpan(const ModalEditSpan &o) { *this = o; }
    ModalEditSpan &operator=(const ModalEditSpan &o) {
        First.store(o.First.load(std::memory_order_relaxed), std::memory_order_relaxed);
        End.store(o.End.load(std::memory_order_relaxed), std::memory_order_relaxed);
        return *this;
    }
    void Mark(uint32_t first, uint32_t end) {
        uint32_t seen = First.load(std::memory_order_relaxed);
        while (first < seen && !First.compare_exchange_weak(seen, first, std::memory_order_relaxed)) {}
        seen = End.load(std::memory_order_relaxed);
        while (end > seen && !End.compare_exchange_weak(seen, end, std::memory_order_release)) {}
    }
    // Claims the pending range; false when nothing was marked.
    bool Take(uint32_t &first, uint32_t &end) {
        end = End.exchange(0, std::memory_order_acquire);
        first = First.exchange(UINT32_MAX, std::memory_order_relaxed);
        return first < end;
    }

private:
    std::atomic<uint32_t> First{UINT32_MAX}, End{0};
};

// ModalAudio.h:103-166, columns in `Real`.
template<typename Real> struct ModalBankColumns {
    using Scalar = Real;
    // per mode
    std::vector<Real> CoeffRe, CoeffIm, StateRe, StateIm, RadiationGain, RadiationArea, DeflectionGain, OutPhaseIm, OutPhaseRe, QuadCompliance, QuadDriveScale;
    // per (object, sample position, mode): ShapeOffset[o] + p * ModeCount[o] + k
    std::vector<Real> ShapeX, ShapeY, ShapeZ;
    // per object
    std::vector<entt::entity> Entities;
    std::vector<uint32_t> ModeOffset, ModeCount, ShapeOffset, TunedModeCount, LiveModeCount;
    std::vector<Real> OutGain, ListenerGain, RadiantRadius, DeflectionScale;
    std::vector<uint8_t> Ringing;
    std::vector<Real> RigidInvMass;
    std::vector<vec3> RigidVel;
    std::vector<Real> RadiatorB0, AirB0, AirB1, AirB2, RecoilA1, RecoilA2, RadiatorZ1, RadiatorZ2, AirZ1, AirZ2;
    struct ActiveImpact {
        uint32_t Object, ExPos, SamplesLeft;
        Real Jx, Jy, Jz, PhaseRe, PhaseIm, RotRe, RotIm, Gamma, AccelAmp, ClickB0, ClickA1, ClickA2, ClickZ1, ClickZ2;
    };
    std::vector<ActiveImpact> Impacts;
    Real SampleRate{48'000};
    ModalEditSpan EditedModes, EditedShapes; // device-mirror bookkeeping (not in the reference)
};
struct ModalBank : ModalBankColumns<float> {};
struct ModalBank64 : ModalBankColumns<double> {};

// Callers that write per-mode or shape columns of a published bank directly tell the mirror which range they touched.
template<typename Real> void MarkModalColumnsEdited(ModalBankColumns<Real> &b, uint32_t first_mode, uint32_t end_mode) { b.EditedModes.Mark(first_mode, end_mode); }
template<typename Real> void MarkModalShapesEdited(ModalBankColumns<Real> &b, uint32_t first, uint32_t end) { b.EditedShapes.Mark(first, end); }

constexpr uint32_t Lanes{8};

// The reference's pool of render threads is a renderer COUNT here: it fixes the deterministic deal of objects to
// renderers and with it the summation order of the mix; the rendering itself happens on the device.
struct ModalRenderPool {
    void SetSize(uint32_t count);
    void SetWorkgroup(void *) {}
    uint32_t Size() const { return Active; }

private:
    uint32_t Active{1};
};

struct ModalDeviceMirror; // opaque: device context, device bank, per-block staging

// ModalAudio.h:255-291 over a bank type.
template<typename Bank> struct ModalAudioCore {
    using BankType = Bank;
    using Scalar = typename Bank::Scalar;
    ModalAudioCore();
    ~ModalAudioCore();
    ModalAudioCore(const ModalAudioCore &) = delete;
    ModalAudioCore &operator=(const ModalAudioCore &) = delete;

    std::unique_ptr<Bank> Live;
    std::atomic<Bank *> Published;
    std::atomic<uint64_t> ReaderSeq{0};
    std::atomic<float> ClickGain{1};
    std::atomic<uint32_t> MaxImpacts{1024}, ActiveImpacts{0}, ActiveVoices{0};
    std::atomic<double> ModalEnergy{0}, PeakModalEnergy{0};
    std::atomic<float> RenderSeconds{0}, RenderShare{0}, PeakRenderShare{0};
    uint64_t EventsDropped{0};
    static constexpr uint32_t EventCapacity{256};
    std::array<ModalEvent, EventCapacity> Events;
    std::atomic<uint32_t> EventWrite{0}, EventRead{0};
    std::atomic<bool> FlushEvents{false};
    ModalRenderPool RenderPool;
    int Device{0}; // HIP device the bank lives on
    std::unique_ptr<ModalDeviceMirror> Dev;
};
struct ModalAudio : ModalAudioCore<ModalBank> {};
struct ModalAudio64 : ModalAudioCore<ModalBank64> {};

inline ModalBank &LiveBank(ModalAudio &m) { return *m.Live; }
inline ModalBank64 &LiveBank(ModalAudio64 &m) { return *m.Live; }

// ModalAudio.h:294-315, once per precision.
uint32_t AddModalObject(ModalBank &, entt::entity, const ModalModes &);
uint32_t AddModalObject(ModalBank64 &, entt::entity, const ModalModes &);
void InstallModalBank(ModalAudio &, ModalBank &next);
void InstallModalBank(ModalAudio64 &, ModalBank64 &next);
void TuneModalObject(ModalBank &, uint32_t object, std::span<const float> freqs, std::span<const float> t60s, float radius_scale = 1.f);
void TuneModalObject(ModalBank64 &, uint32_t object, std::span<const float> freqs, std::span<const float> t60s, float radius_scale = 1.f);
bool SetModalObjectShapes(ModalBank &, uint32_t object, const ModalModes &);
bool SetModalObjectShapes(ModalBank64 &, uint32_t object, const ModalModes &);
std::optional<uint32_t> FindModalObject(const ModalBank &, entt::entity);
std::optional<uint32_t> FindModalObject(const ModalBank64 &, entt::entity);
void EnqueueModalEvent(ModalAudio &, const ModalEvent &);
void EnqueueModalEvent(ModalAudio64 &, const ModalEvent &);
// Adds frame_count mono samples into `out`.  Events take effect at the start of the block.
void RenderModal(ModalAudio &, float *out, uint32_t frame_count);
void RenderModal(ModalAudio64 &, double *out, uint32_t frame_count);
// Not in the reference: the libmodalhip context the bank's device mirror lives on (created on demand), for callers that
// time its kernels (mh_context_time_kernels / mh_context_kernel_class_stats).
mh_context *ModalDeviceContext(ModalAudio &);
mh_context *ModalDeviceContext(ModalAudio64 &);
// Not in the reference: copies the device-resident StateRe / StateIm back into the host bank for inspection.
void SyncModalState(ModalAudio &);
void SyncModalState(ModalAudio64 &);
