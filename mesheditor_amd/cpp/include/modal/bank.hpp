// Modal synthesis bank with the reference's API surface (src/audio/ModalAudio.h): ModalEvent, ModalBank (public
// struct-of-arrays columns the caller may write), ModalAudio, AddModalObject / TuneModalObject / InstallModalBank /
// SetModalObjectShapes / FindModalObject / EnqueueModalEvent / RenderModal, and the recoil filter helpers.
// The host ModalBank stays the source of truth; InstallModalBank mirrors it into HBM and RenderModal renders each
// block on the MI355X (mh_bank_render), one device synchronisation per block.
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

struct ModalEvent {
    ModalEventKind Kind{ModalEventKind::Impact};
    uint32_t Object{0}, ExPos{0};
    float Jx{0}, Jy{0}, Jz{0};
    float PulseStep{0}, PulseGamma{0}, AccelAmp{0};
    float ClickB0{0}, ClickA1{0}, ClickA2{0};
};

constexpr float AirDensity{1.204f}, SpeedOfSound{343.f}, ListenerDistance{1.f};
constexpr float Ln1000 = 3 * std::numbers::ln10_v<float>;

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

// Not in the reference: the span of per-mode (or shape) entries edited since the device mirror last saw them.
// TuneModalObject / SetModalObjectShapes mark it (from any thread); the render takes it at block start and uploads
// just that span.  Code that writes the per-mode columns directly calls MarkModalColumnsEdited.
struct ModalEditSpan {
    std::atomic<uint32_t> Lo{UINT32_MAX}, Hi{0};
    ModalEditSpan() = default;
    ModalEditSpan(const ModalEditSpan &o) : Lo{o.Lo.load()}, Hi{o.Hi.load()} {}
    ModalEditSpan &operator=(const ModalEditSpan &o) {
        Lo.store(o.Lo.load());
        Hi.store(o.Hi.load());
        return *this;
    }
    void Mark(uint32_t lo, uint32_t hi) {
        for (auto seen = Lo.load(); lo < seen && !Lo.compare_exchange_weak(seen, lo);) {}
        for (auto seen = Hi.load(); hi > seen && !Hi.compare_exchange_weak(seen, hi);) {}
    }
    bool Take(uint32_t &lo, uint32_t &hi) {
        hi = Hi.exchange(0);
        lo = Lo.exchange(UINT32_MAX);
        return lo < hi;
    }
};

struct ModalBank {
    // per mode
    std::vector<float> CoeffRe, CoeffIm, StateRe, StateIm, RadiationGain, RadiationArea, DeflectionGain, OutPhaseIm, OutPhaseRe, QuadCompliance, QuadDriveScale;
    std::vector<float> ShapeX, ShapeY, ShapeZ; // object o, position p, mode k: ShapeOffset[o] + p*ModeCount[o] + k
    // per object
    std::vector<entt::entity> Entities;
    std::vector<uint32_t> ModeOffset, ModeCount, ShapeOffset, TunedModeCount, LiveModeCount;
    std::vector<float> OutGain, ListenerGain, RadiantRadius, DeflectionScale;
    std::vector<uint8_t> Ringing;
    std::vector<float> RigidInvMass;
    std::vector<vec3> RigidVel;
    std::vector<float> RadiatorB0, AirB0, AirB1, AirB2, RecoilA1, RecoilA2, RadiatorZ1, RadiatorZ2, AirZ1, AirZ2;
    struct ActiveImpact {
        uint32_t Object, ExPos, SamplesLeft;
        float Jx, Jy, Jz, PhaseRe, PhaseIm, RotRe, RotIm, Gamma, AccelAmp, ClickB0, ClickA1, ClickA2, ClickZ1, ClickZ2;
    };
    std::vector<ActiveImpact> Impacts;
    float SampleRate{48'000};
    ModalEditSpan EditedModes, EditedShapes; // device-mirror bookkeeping (not in the reference)
};
inline void MarkModalColumnsEdited(ModalBank &b, uint32_t first_mode, uint32_t end_mode) { b.EditedModes.Mark(first_mode, end_mode); }
inline void MarkModalShapesEdited(ModalBank &b, uint32_t first, uint32_t end) { b.EditedShapes.Mark(first, end); }

constexpr uint32_t Lanes{8};

// The reference's pool of render threads becomes a renderer COUNT here: it fixes the deterministic deal of objects
// to renderers and therefore the summation order of the mix; the work itself runs on the device.
struct ModalRenderPool {
    void SetSize(uint32_t count);
    void SetWorkgroup(void *) {}
    uint32_t Size() const { return Active; }

private:
    uint32_t Active{1};
};

struct ModalAudio {
    ModalAudio();
    ~ModalAudio();
    std::unique_ptr<ModalBank> Live;
    std::atomic<ModalBank *> Published;
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

    // device mirror of the published bank (opaque)
    struct DeviceState;
    std::unique_ptr<DeviceState> Dev;
};

inline ModalBank &LiveBank(ModalAudio &m) { return *m.Live; }

uint32_t AddModalObject(ModalBank &, entt::entity, const ModalModes &);
void InstallModalBank(ModalAudio &, ModalBank &next);
void TuneModalObject(ModalBank &, uint32_t object, std::span<const float> freqs, std::span<const float> t60s, float radius_scale = 1.f);
bool SetModalObjectShapes(ModalBank &, uint32_t object, const ModalModes &);
std::optional<uint32_t> FindModalObject(const ModalBank &, entt::entity);
void EnqueueModalEvent(ModalAudio &, const ModalEvent &);
// Adds frame_count mono samples into `out`.  Events take effect at the start of the block.
void RenderModal(ModalAudio &, float *out, uint32_t frame_count);
// Not in the reference: copies the device-resident StateRe / StateIm back into the host bank for inspection.
void SyncModalState(ModalAudio &);
