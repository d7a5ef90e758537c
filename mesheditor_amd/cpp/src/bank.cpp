// Host side of the resonator bank over libmodalhip: the event queue, impact activation, object deal, tuning and
// bank publication restate the reference's control flow (src/audio/ModalAudio.cpp:28-82, 277-461, 486-590); the
// per-sample work (force curves, click filters, mode recurrences, ordered mix) runs in mh_bank_render.
#include "modal/bank.hpp"

#include "modalhip.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <stdexcept>
#include <thread>

struct ModalAudio::DeviceState {
    mh_context *ctx{nullptr};
    mh_bank *bank{nullptr};
    const ModalBank *mirrored{nullptr};
    std::vector<uint32_t> impacts_on; // active impacts per object, rebuilt once per block
    std::vector<uint32_t> deal_offset, deal_objects, render_count, tuned, live;
    std::vector<double> energy, modal_energy;
    std::vector<uint8_t> silenced;
    std::vector<mh_impact> impacts;
    std::vector<std::pair<uint64_t, uint32_t>> order;
    std::vector<uint64_t> load;
    std::vector<std::vector<uint32_t>> renderers;
    ~DeviceState() {
        mh_bank_destroy(bank);
        mh_context_destroy(ctx);
    }
};

RecoilPoles RecoilDenominator(double wc, double kk, double beta) {
    const double a0 = kk * kk + beta * wc * kk + beta * wc * wc;
    return {a0, float((2 * beta * wc * wc - 2 * kk * kk) / a0), float((kk * kk - beta * wc * kk + beta * wc * wc) / a0)};
}
RecoilFilter RecoilObjectFilter(double radius, double volume, double sample_rate) {
    if (radius <= 0 || volume <= 0) return {};
    const double wc = SpeedOfSound / radius, kk = 2 * sample_rate;
    const auto poles = RecoilDenominator(wc, kk, 2);
    const double gp = AirDensity * SpeedOfSound * radius / ListenerDistance;
    const double n2 = AirDensity * volume * wc, n1 = n2 * wc;
    return {float(gp * kk * kk / poles.A0), float((n2 * kk * kk + n1 * kk) / poles.A0), float(-2 * n2 * kk * kk / poles.A0),
            float((n2 * kk * kk - n1 * kk) / poles.A0), poles.A1, poles.A2};
}
ClickFilter RecoilClickFilter(double radius, double volume, double mass, double sample_rate) {
    if (radius <= 0 || mass <= 0) return {};
    const double wc = SpeedOfSound / radius, kk = 2 * sample_rate;
    const auto poles = RecoilDenominator(wc, kk, 2 + AirDensity * volume / mass);
    const double g = AirDensity * SpeedOfSound * radius / (ListenerDistance * mass);
    return {float(g * kk / poles.A0), poles.A1, poles.A2};
}

void ModalRenderPool::SetSize(uint32_t count) {
    const auto cores = std::max(1u, std::thread::hardware_concurrency());
    Active = std::clamp(count, 1u, cores);
}

ModalAudio::ModalAudio() : Live{std::make_unique<ModalBank>()}, Published{Live.get()}, Dev{std::make_unique<DeviceState>()} {}
ModalAudio::~ModalAudio() = default;

namespace {
constexpr float SilentEnergy{1e-12f};

void RemoveImpact(ModalBank &b, uint32_t i) {
    b.Impacts[i] = b.Impacts.back();
    b.Impacts.pop_back();
}

void ActivateImpact(const ModalAudio &m, ModalBank &b, const ModalEvent &e) {
    if (b.Impacts.size() >= m.MaxImpacts.load(std::memory_order_relaxed)) return;
    const auto theta = 2 * std::numbers::pi_v<float> * e.PulseStep;
    b.Impacts.push_back({e.Object, e.ExPos, uint32_t(std::ceil(1.f / e.PulseStep)), e.Jx, e.Jy, e.Jz, 1.f, 0.f, std::cos(theta), std::sin(theta),
                         e.PulseGamma, e.AccelAmp, e.ClickB0, e.ClickA1, e.ClickA2, 0.f, 0.f});
    b.Ringing[e.Object] = 1;
}

void SilenceObject(ModalAudio &m, ModalBank &b, uint32_t o, bool device_state_already_zero) {
    const uint32_t k0 = b.ModeOffset[o], count = b.ModeCount[o];
    std::fill_n(b.StateRe.begin() + k0, count, 0.f);
    std::fill_n(b.StateIm.begin() + k0, count, 0.f);
    if (!device_state_already_zero && m.Dev->bank && m.Dev->mirrored == &b) mh_bank_zero_state(m.Dev->bank, k0, count);
    b.Ringing[o] = 0;
    b.LiveModeCount[o] = b.TunedModeCount[o];
    for (uint32_t i = uint32_t(b.Impacts.size()); i-- > 0;)
        if (b.Impacts[i].Object == o) RemoveImpact(b, i);
}

void DrainEvents(ModalAudio &m, ModalBank &b) {
    auto read = m.EventRead.load(std::memory_order_relaxed);
    const auto write = m.EventWrite.load(std::memory_order_acquire);
    for (; read != write; ++read) {
        const auto &e = m.Events[read % ModalAudio::EventCapacity];
        if (e.Object >= b.Entities.size()) continue;
        switch (e.Kind) {
            case ModalEventKind::Impact:
                if (e.PulseStep > 0) ActivateImpact(m, b, e);
                break;
            case ModalEventKind::Silence: SilenceObject(m, b, e.Object, false); break;
        }
    }
    m.EventRead.store(read, std::memory_order_release);
}

// Deterministic deal of the ringing objects onto `count` renderers, heaviest first (ModalAudio.cpp:430-461).
void DealObjects(ModalAudio &m, const ModalBank &b, uint32_t count) {
    auto &d = *m.Dev;
    d.renderers.resize(count);
    for (auto &r : d.renderers) r.clear();
    d.order.clear();
    for (uint32_t o = 0; o < uint32_t(b.Entities.size()); ++o) {
        if (!b.Ringing[o]) continue;
        const bool excited = d.impacts_on[o] != 0;
        d.order.emplace_back(uint64_t(excited ? b.TunedModeCount[o] : b.LiveModeCount[o]), o);
    }
    if (count == 1) {
        for (const auto &[cost, o] : d.order) d.renderers.front().push_back(o);
        return;
    }
    std::sort(d.order.begin(), d.order.end(), [](const auto &a, const auto &c) { return a.first != c.first ? a.first > c.first : a.second < c.second; });
    d.load.assign(count, 0);
    for (const auto &[cost, o] : d.order) {
        const auto least = uint32_t(std::min_element(d.load.begin(), d.load.end()) - d.load.begin());
        d.load[least] += cost;
        d.renderers[least].push_back(o);
    }
    for (auto &r : d.renderers) std::sort(r.begin(), r.end());
}

void EnsureDevice(ModalAudio &m) {
    auto &d = *m.Dev;
    if (!d.ctx && mh_context_create(m.Device, &d.ctx) != MH_OK) throw std::runtime_error("modalhip: no MI355X context for the modal bank (there is no CPU fallback)");
}

// Mirror the published bank into HBM (layout + shapes + coefficients; state starts from the host's).
void MirrorBank(ModalAudio &m, ModalBank &b) {
    auto &d = *m.Dev;
    EnsureDevice(m);
    mh_bank_destroy(d.bank);
    d.bank = nullptr;
    const auto n_obj = uint32_t(b.Entities.size()), n_modes = uint32_t(b.CoeffRe.size()), n_shapes = uint32_t(b.ShapeX.size());
    if (mh_bank_create(d.ctx, 0, n_obj, n_modes, n_shapes, b.ModeOffset.data(), b.ModeCount.data(), b.ShapeOffset.data(), b.ShapeX.data(), b.ShapeY.data(),
                       b.ShapeZ.data(), &d.bank) != MH_OK)
        throw std::runtime_error(std::string("modalhip: ") + mh_last_error(d.ctx));
    mh_bank_set_coefficients(d.bank, 0, n_modes, b.CoeffRe.data(), b.CoeffIm.data(), b.RadiationGain.data(), b.OutPhaseIm.data(), b.OutPhaseRe.data());
    uint32_t lo, hi;
    b.EditedModes.Take(lo, hi);
    b.EditedShapes.Take(lo, hi);
    d.mirrored = &b;
}

// Push host-side edits made since the last block (TuneModalObject / SetModalObjectShapes mark the spans they wrote).
void SyncEdits(ModalAudio &m, ModalBank &b) {
    auto &d = *m.Dev;
    uint32_t lo, hi;
    if (b.EditedModes.Take(lo, hi)) {
        hi = std::min(hi, uint32_t(b.CoeffRe.size()));
        if (lo < hi)
            mh_bank_set_coefficients(d.bank, lo, hi - lo, b.CoeffRe.data() + lo, b.CoeffIm.data() + lo, b.RadiationGain.data() + lo, b.OutPhaseIm.data() + lo,
                                     b.OutPhaseRe.data() + lo);
    }
    if (b.EditedShapes.Take(lo, hi)) {
        hi = std::min(hi, uint32_t(b.ShapeX.size()));
        if (lo < hi) mh_bank_set_shapes(d.bank, lo, hi - lo, b.ShapeX.data() + lo, b.ShapeY.data() + lo, b.ShapeZ.data() + lo);
    }
}
} // namespace

void InstallModalBank(ModalAudio &m, ModalBank &next) {
    const auto old = std::move(m.Live);
    m.Live = std::make_unique<ModalBank>(std::move(next));
    m.FlushEvents.store(true, std::memory_order_relaxed);
    m.Published.store(m.Live.get(), std::memory_order_seq_cst);
    if (const auto seq = m.ReaderSeq.load(std::memory_order_seq_cst); seq & 1) {
        while (m.ReaderSeq.load(std::memory_order_seq_cst) == seq) std::this_thread::yield();
    }
    m.ActiveVoices.store(0, std::memory_order_relaxed);
    MirrorBank(m, *m.Live);
}

uint32_t AddModalObject(ModalBank &b, entt::entity e, const ModalModes &modes) {
    const auto count = uint32_t(modes.Freqs.size());
    const auto slot = uint32_t(b.Entities.size());
    b.Entities.push_back(e);
    b.ModeOffset.push_back(uint32_t(b.CoeffRe.size()));
    b.ModeCount.push_back(count);
    b.TunedModeCount.push_back(count);
    b.LiveModeCount.push_back(count);
    b.ShapeOffset.push_back(uint32_t(b.ShapeX.size()));
    b.Ringing.push_back(0);
    b.RigidVel.emplace_back(0.f);
    for (auto *col : {&b.OutGain, &b.RigidInvMass, &b.RadiatorB0, &b.AirB0, &b.AirB1, &b.AirB2, &b.RecoilA1, &b.RecoilA2, &b.RadiatorZ1, &b.RadiatorZ2, &b.AirZ1, &b.AirZ2})
        col->push_back(0.f);
    for (auto *col : {&b.ListenerGain, &b.DeflectionScale}) col->push_back(1.f);
    for (auto *col : {&b.CoeffRe, &b.CoeffIm, &b.StateRe, &b.StateIm, &b.RadiationGain, &b.DeflectionGain, &b.QuadCompliance, &b.QuadDriveScale})
        col->resize(col->size() + count, 0.f);
    b.OutPhaseIm.resize(b.OutPhaseIm.size() + count, 1.f);
    b.OutPhaseRe.resize(b.OutPhaseRe.size() + count, 0.f);
    for (const auto &row : modes.Shapes)
        for (const auto &shape : row) {
            b.ShapeX.push_back(shape.x);
            b.ShapeY.push_back(shape.y);
            b.ShapeZ.push_back(shape.z);
        }
    // Radiating strength per mode: centroid quadrature of the squared normal shape over the sample surface.
    const auto area_offset = b.RadiationGain.size() - count;
    float total_area = 0.f;
    b.RadiationArea.resize(area_offset + count, 0.f);
    for (size_t t = 0; t + 2 < modes.Indices.size(); t += 3) {
        const auto i = modes.Indices[t], j = modes.Indices[t + 1], l = modes.Indices[t + 2];
        const vec3 cr = cross(modes.Positions[j] - modes.Positions[i], modes.Positions[l] - modes.Positions[i]);
        const float doubled = length(cr);
        if (doubled <= 0.f) continue;
        const vec3 n = cr / doubled;
        const float area = doubled / 2;
        total_area += area;
        for (uint32_t k = 0; k < count; ++k) {
            const vec3 shape = (modes.Shapes[i][k] + modes.Shapes[j][k] + modes.Shapes[l][k]) / 3.f;
            const float normal = dot(shape, n);
            b.RadiationArea[area_offset + k] += area * normal * normal;
        }
    }
    b.RadiantRadius.push_back(std::sqrt(total_area / (4 * std::numbers::pi_v<float>)));
    return slot;
}

void TuneModalObject(ModalBank &b, uint32_t object, std::span<const float> freqs, std::span<const float> t60s, float radius_scale) {
    const auto k0 = b.ModeOffset[object];
    const auto count = std::min(b.ModeCount[object], uint32_t(std::min(freqs.size(), t60s.size())));
    const float sr = b.SampleRate;
    const float radius = b.RadiantRadius[object] * radius_scale;
    b.DeflectionScale[object] = 1.f / (radius_scale * radius_scale * radius_scale);
    for (uint32_t k = 0; k < count; ++k) {
        const float freq = freqs[k], t60 = t60s[k];
        if (!std::isfinite(freq) || !std::isfinite(t60) || freq <= 0.f || freq >= sr / 2 - 1 || t60 <= 0.f) {
            b.CoeffRe[k0 + k] = b.CoeffIm[k0 + k] = b.RadiationGain[k0 + k] = b.DeflectionGain[k0 + k] = 0.f;
            b.OutPhaseIm[k0 + k] = 1.f;
            b.OutPhaseRe[k0 + k] = 0.f;
            b.QuadCompliance[k0 + k] = b.QuadDriveScale[k0 + k] = 0.f;
            continue;
        }
        const auto omega = 2 * std::numbers::pi_v<float> * freq / sr;
        const float omega_si = 2 * std::numbers::pi_v<float> * freq;
        const float ka = omega_si * radius / SpeedOfSound;
        const float sigma = ka * ka / (1 + ka * ka);
        const float area = b.RadiationArea[k0 + k] / radius_scale;
        const float radiation_rate = AirDensity * SpeedOfSound * sigma * area * 0.5f;
        const auto decay = std::exp(-(Ln1000 / t60 + radiation_rate) / sr);
        b.CoeffRe[k0 + k] = decay * std::cos(omega);
        b.CoeffIm[k0 + k] = decay * std::sin(omega);
        const float gain = AirDensity * SpeedOfSound * std::sqrt(sigma * b.RadiationArea[k0 + k] / (4 * std::numbers::pi_v<float>)) / ListenerDistance;
        b.RadiationGain[k0 + k] = gain;
        const float spread = sigma * std::numbers::pi_v<float> * (2.f * std::fmod(0.6180339887f * float(k + 1), 1.0f) - 1.f);
        b.OutPhaseIm[k0 + k] = std::cos(spread);
        b.OutPhaseRe[k0 + k] = std::sin(spread);
        b.DeflectionGain[k0 + k] = gain > 0.f ? 1.f / (gain * omega_si) : 0.f;
        const float dt = 1.f / sr;
        const float central = dt * (1 + decay * decay + 2 * decay * std::cos(omega)) / 4;
        b.QuadCompliance[k0 + k] = central;
        b.QuadDriveScale[k0 + k] = central * omega_si / (decay * std::sin(omega));
    }
    uint32_t live = b.ModeCount[object];
    while (live > 0 && b.CoeffRe[k0 + live - 1] == 0.f && b.CoeffIm[k0 + live - 1] == 0.f) --live;
    b.TunedModeCount[object] = live;
    b.LiveModeCount[object] = live;
    b.EditedModes.Mark(k0, k0 + b.ModeCount[object]);
}

bool SetModalObjectShapes(ModalBank &b, uint32_t object, const ModalModes &modes) {
    const auto begin = b.ShapeOffset[object];
    const auto end = object + 1 < b.ShapeOffset.size() ? b.ShapeOffset[object + 1] : uint32_t(b.ShapeX.size());
    const auto count = uint32_t(modes.Freqs.size());
    if (b.ModeCount[object] != count || end - begin != count * modes.Shapes.size()) return false;
    auto i = begin;
    for (const auto &row : modes.Shapes)
        for (const auto &shape : row) {
            b.ShapeX[i] = shape.x;
            b.ShapeY[i] = shape.y;
            b.ShapeZ[i] = shape.z;
            ++i;
        }
    b.EditedShapes.Mark(begin, end);
    return true;
}

std::optional<uint32_t> FindModalObject(const ModalBank &b, entt::entity e) {
    const auto it = std::find(b.Entities.begin(), b.Entities.end(), e);
    return it != b.Entities.end() ? std::optional{uint32_t(it - b.Entities.begin())} : std::nullopt;
}

void EnqueueModalEvent(ModalAudio &m, const ModalEvent &e) {
    const auto write = m.EventWrite.load(std::memory_order_relaxed);
    if (write - m.EventRead.load(std::memory_order_acquire) >= ModalAudio::EventCapacity) {
        ++m.EventsDropped;
        return;
    }
    m.Events[write % ModalAudio::EventCapacity] = e;
    m.EventWrite.store(write + 1, std::memory_order_release);
}

void RenderModal(ModalAudio &m, float *out, uint32_t frame_count) {
    if (frame_count == 0) return;
    const auto render_start = std::chrono::steady_clock::now();
    const auto seq = m.ReaderSeq.load(std::memory_order_relaxed);
    m.ReaderSeq.store(seq + 1, std::memory_order_seq_cst);
    ModalBank &b = *m.Published.load(std::memory_order_seq_cst);
    auto &d = *m.Dev;
    if (d.mirrored != &b || !d.bank) MirrorBank(m, b); // a bank built in place (never installed) is mirrored on first use
    if (m.FlushEvents.exchange(false, std::memory_order_relaxed)) m.EventRead.store(m.EventWrite.load(std::memory_order_relaxed), std::memory_order_relaxed);
    DrainEvents(m, b);
    SyncEdits(m, b);
    const auto click_gain = m.ClickGain.load(std::memory_order_relaxed);

    d.impacts_on.assign(b.Entities.size(), 0);
    for (const auto &im : b.Impacts) ++d.impacts_on[im.Object];
    const uint32_t width = m.RenderPool.Size();
    DealObjects(m, b, width);
    d.deal_offset.assign(width + 1, 0);
    d.deal_objects.clear();
    d.render_count.clear();
    d.tuned.clear();
    for (uint32_t r = 0; r < width; ++r) {
        for (const auto o : d.renderers[r]) {
            const bool excited = d.impacts_on[o] != 0;
            d.deal_objects.push_back(o);
            d.render_count.push_back(excited ? b.TunedModeCount[o] : b.LiveModeCount[o]);
            d.tuned.push_back(b.TunedModeCount[o]);
        }
        d.deal_offset[r + 1] = uint32_t(d.deal_objects.size());
    }
    const auto n_dealt = uint32_t(d.deal_objects.size());
    d.impacts.resize(b.Impacts.size());
    for (size_t i = 0; i < b.Impacts.size(); ++i) {
        const auto &im = b.Impacts[i];
        d.impacts[i] = {im.Object, im.ExPos, im.SamplesLeft, 0, im.Jx, im.Jy, im.Jz, im.PhaseRe, im.PhaseIm, im.RotRe, im.RotIm, im.Gamma, im.AccelAmp,
                        im.ClickB0, im.ClickA1, im.ClickA2, im.ClickZ1, im.ClickZ2};
    }
    d.energy.assign(n_dealt, 0.0);
    d.live.assign(n_dealt, 0);
    d.silenced.assign(n_dealt, 0);
    d.modal_energy.assign(n_dealt, 0.0);
    if (mh_bank_render(d.bank, frame_count, click_gain, uint32_t(d.impacts.size()), d.impacts.data(), width, d.deal_offset.data(), d.deal_objects.data(),
                       d.render_count.data(), d.tuned.data(), b.OutGain.data(), b.ListenerGain.data(), out, d.energy.data(), d.live.data(), d.silenced.data(),
                       d.modal_energy.data()) != MH_OK)
        throw std::runtime_error(std::string("modalhip: ") + mh_last_error(d.ctx));
    for (size_t i = 0; i < b.Impacts.size(); ++i) {
        auto &im = b.Impacts[i];
        im.SamplesLeft = d.impacts[i].samples_left;
        im.PhaseRe = float(d.impacts[i].phase_re);
        im.PhaseIm = float(d.impacts[i].phase_im);
        im.ClickZ1 = float(d.impacts[i].click_z1);
        im.ClickZ2 = float(d.impacts[i].click_z2);
    }
    // Per-object bookkeeping of RenderObjectFast's tail (ModalAudio.cpp:141-146).
    for (uint32_t i = 0; i < n_dealt; ++i) {
        const auto o = d.deal_objects[i];
        const bool excited = d.impacts_on[o] != 0;
        if (d.silenced[i]) {
            SilenceObject(m, b, o, true);
            continue;
        }
        b.Ringing[o] = 1;
        b.LiveModeCount[o] = excited ? b.TunedModeCount[o] : d.live[i];
    }
    for (uint32_t i = uint32_t(b.Impacts.size()); i-- > 0;) {
        const auto &im = b.Impacts[i];
        if (im.SamplesLeft == 0 && std::abs(im.ClickZ1) + std::abs(im.ClickZ2) < 1e-12f) RemoveImpact(b, i);
    }
    // Modal-energy diagnostic: evaluated on the device per rendered object, summed here in object order.
    double energy = 0;
    {
        std::vector<std::pair<uint32_t, double>> per_object(n_dealt);
        for (uint32_t i = 0; i < n_dealt; ++i) per_object[i] = {d.deal_objects[i], d.modal_energy[i]};
        std::sort(per_object.begin(), per_object.end());
        for (const auto &[o, e] : per_object) energy += e;
    }
    m.ModalEnergy.store(energy, std::memory_order_relaxed);
    if (const double seen = m.PeakModalEnergy.load(std::memory_order_relaxed); energy > seen) m.PeakModalEnergy.store(energy, std::memory_order_relaxed);
    m.ActiveImpacts.store(uint32_t(b.Impacts.size()), std::memory_order_relaxed);
    m.ReaderSeq.store(seq + 2, std::memory_order_release);
    const float seconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - render_start).count();
    const float share = b.SampleRate > 0 ? seconds * b.SampleRate / float(frame_count) : 0.f;
    m.RenderSeconds.store(seconds, std::memory_order_relaxed);
    m.RenderShare.store(share, std::memory_order_relaxed);
    if (const float seen = m.PeakRenderShare.load(std::memory_order_relaxed); share > seen) m.PeakRenderShare.store(share, std::memory_order_relaxed);
}

// Pull the device-resident resonator states into the host bank's StateRe / StateIm columns (they are only needed on
// the host for inspection; the render loop never reads them back).
void SyncModalState(ModalAudio &m) {
    auto &d = *m.Dev;
    ModalBank &b = *m.Published.load(std::memory_order_seq_cst);
    if (!d.bank || d.mirrored != &b || b.StateRe.empty()) return;
    std::vector<double> re(b.StateRe.size()), im(b.StateIm.size());
    if (mh_bank_read_state(d.bank, 0, uint32_t(re.size()), re.data(), im.data()) != MH_OK) return;
    for (size_t k = 0; k < re.size(); ++k) {
        b.StateRe[k] = float(re[k]);
        b.StateIm[k] = float(im[k]);
    }
}
