// ORACLE -- TEST INFRASTRUCTURE ONLY (see modal_oracle.h).
//
// CPU restatement of the synthesis half of the reference's modal path:
//   src/audio/ModalAudio.h (ModalEvent :28-37, recoil filters :58-99, ModalBank :103-166, Lanes :169,
//     ImpactGainRow :182-188) and src/audio/ModalAudio.cpp (ActivateImpact :28-51, SilenceObject :53-64,
//     DrainEvents :66-82, RenderObjectFast :86-147, InstallModalBank :277-289, AddModalObject :291-338,
//     TuneModalObject :340-393, SetModalObjectShapes :395-410, EnqueueModalEvent :417-425, DealObjects
//     :430-461, RenderObjects :471-483, RenderModal :486-590), with the surface-contact hooks as the no-ops of
//     src/audio/SurfaceContactAbsent.cpp.
//   src/audio/ContactModel.cpp (:10-114).
// The renderer pool is restated as a sequential loop over renderers: every renderer owns a private Out buffer
// and the mix adds them in renderer order (ModalAudio.cpp:553-555), so the signal is the threaded one's exactly.
// Built with -ffp-contract=off: the expression trees below are the canonical ones the device kernels follow.
// Real = float restates the reference (fp32 bank); Real = double is the fp64 variant BASELINE.json asks for.
#include "modal_oracle_synth.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>
#include <utility>
#include <vector>

namespace {
template<typename Real> struct V3 {
    Real x{0}, y{0}, z{0};
};

constexpr float AirDensity{1.204f}, SpeedOfSound{343.f}, ListenerDistance{1.f};
constexpr double Pi = 3.14159265358979323846;
constexpr uint32_t Lanes{8};
constexpr uint32_t EventCapacity{256};

template<typename Real> struct Bank {
    std::vector<Real> CoeffRe, CoeffIm, StateRe, StateIm, RadiationGain, RadiationArea, DeflectionGain, OutPhaseIm, OutPhaseRe, QuadCompliance, QuadDriveScale;
    std::vector<Real> ShapeX, ShapeY, ShapeZ;
    std::vector<uint32_t> Entities, ModeOffset, ModeCount, ShapeOffset, TunedModeCount, LiveModeCount;
    std::vector<Real> OutGain, ListenerGain, RadiantRadius, DeflectionScale, RigidInvMass;
    std::vector<uint8_t> Ringing;
    struct ActiveImpact {
        uint32_t Object, ExPos, SamplesLeft;
        Real Jx, Jy, Jz, PhaseRe, PhaseIm, RotRe, RotIm, Gamma, AccelAmp, ClickB0, ClickA1, ClickA2, ClickZ1, ClickZ2;
    };
    std::vector<ActiveImpact> Impacts;
    Real SampleRate{48000};
};

template<typename Real> struct Renderer {
    std::vector<uint32_t> Objects, Impacts;
    std::vector<Real> Out, Gains;
};

template<typename Real> struct Audio {
    Bank<Real> Live; // the published bank
    Bank<Real> Next; // under construction until install
    Real ClickGain{1};
    uint32_t MaxImpacts{1024}, ActiveImpacts{0};
    double ModalEnergy{0}, PeakModalEnergy{0};
    uint64_t EventsDropped{0};
    std::array<mo_event, EventCapacity> Events;
    uint32_t EventWrite{0}, EventRead{0};
    bool FlushEvents{false};
    std::vector<Real> ForceScratch;
    std::vector<Renderer<Real>> Renderers;
    uint32_t PoolSize{1};
    std::vector<std::pair<uint64_t, uint32_t>> RenderOrderScratch;
    std::vector<uint64_t> RenderLoadScratch;
};

template<typename Real> constexpr Real SilentEnergy() { return Real(1e-12f); }

template<typename Real> void RemoveImpact(Bank<Real> &b, uint32_t i) {
    b.Impacts[i] = b.Impacts.back();
    b.Impacts.pop_back();
}

template<typename Real> void ActivateImpact(const Audio<Real> &m, Bank<Real> &b, const mo_event &e) {
    if (b.Impacts.size() >= m.MaxImpacts) return;
    const Real step = Real(e.pulse_step);
    const Real theta = 2 * Real(Pi) * step; // 2 * pi_v<float> * PulseStep in the reference's float
    typename Bank<Real>::ActiveImpact im{};
    im.Object = e.object;
    im.ExPos = e.ex_pos;
    im.SamplesLeft = uint32_t(std::ceil(Real(1) / step));
    im.Jx = e.jx; im.Jy = e.jy; im.Jz = e.jz;
    im.PhaseRe = 1; im.PhaseIm = 0;
    im.RotRe = std::cos(theta); im.RotIm = std::sin(theta);
    im.Gamma = e.pulse_gamma; im.AccelAmp = e.accel_amp;
    im.ClickB0 = e.click_b0; im.ClickA1 = e.click_a1; im.ClickA2 = e.click_a2;
    im.ClickZ1 = 0; im.ClickZ2 = 0;
    b.Impacts.push_back(im);
    b.Ringing[e.object] = 1;
}

template<typename Real> void SilenceObject(Bank<Real> &b, uint32_t o) {
    const uint32_t k0 = b.ModeOffset[o], count = b.ModeCount[o];
    std::fill_n(b.StateRe.begin() + k0, count, Real(0));
    std::fill_n(b.StateIm.begin() + k0, count, Real(0));
    b.Ringing[o] = 0;
    b.LiveModeCount[o] = b.TunedModeCount[o];
    for (uint32_t i = uint32_t(b.Impacts.size()); i-- > 0;)
        if (b.Impacts[i].Object == o) RemoveImpact(b, i);
}

template<typename Real> void DrainEvents(Audio<Real> &m, Bank<Real> &b) {
    auto read = m.EventRead;
    const auto write = m.EventWrite;
    for (; read != write; ++read) {
        const auto &e = m.Events[read % EventCapacity];
        if (e.object >= b.Entities.size()) continue;
        if (e.kind == 0) {
            if (e.pulse_step > 0) ActivateImpact(m, b, e);
        } else if (e.kind == 1) {
            SilenceObject(b, e.object);
        }
    }
    m.EventRead = read;
}

// ModalAudio.h:182-188
template<typename Real> void ImpactGainRow(const Bank<Real> &b, uint32_t impact, uint32_t shape0, uint32_t count, uint32_t k0, uint32_t first, uint32_t n, Real *out) {
    const auto &im = b.Impacts[impact];
    const auto base = shape0 + im.ExPos * count + first;
    for (uint32_t i = 0; i < n; ++i) out[i] = b.RadiationGain[k0 + first + i] * (b.ShapeX[base + i] * im.Jx + b.ShapeY[base + i] * im.Jy + b.ShapeZ[base + i] * im.Jz);
}

// ModalAudio.cpp:86-147
template<typename Real> void RenderObjectFast(Audio<Real> &m, Renderer<Real> &w, Bank<Real> &b, uint32_t o, const std::vector<uint32_t> &impacts, Real *out, uint32_t frame_count) {
    const auto k0 = b.ModeOffset[o], stride = b.ModeCount[o];
    const auto count = impacts.empty() ? b.LiveModeCount[o] : b.TunedModeCount[o];
    const auto shape0 = b.ShapeOffset[o];
    const Real out_gain = b.OutGain[o];
    const Real mix_gain = out_gain * b.ListenerGain[o];
    w.Gains.resize(impacts.size() * Lanes);
    Real energy = 0;
    uint32_t live = 0;
    for (uint32_t k = 0; k < count; k += Lanes) {
        const auto width = std::min(Lanes, count - k);
        Real z_re[Lanes]{}, z_im[Lanes]{}, c_re[Lanes]{}, c_im[Lanes]{}, p_re[Lanes]{}, p_im[Lanes]{};
        for (uint32_t l = 0; l < width; ++l) {
            z_re[l] = b.StateRe[k0 + k + l];
            z_im[l] = b.StateIm[k0 + k + l];
            c_re[l] = b.CoeffRe[k0 + k + l];
            c_im[l] = b.CoeffIm[k0 + k + l];
            p_im[l] = b.OutPhaseIm[k0 + k + l];
            p_re[l] = b.OutPhaseRe[k0 + k + l];
        }
        for (size_t t = 0; t < impacts.size(); ++t) {
            auto *gain = &w.Gains[t * Lanes];
            ImpactGainRow(b, impacts[t], shape0, stride, k0, k, width, gain);
            std::fill(gain + width, gain + Lanes, Real(0));
        }
        for (uint32_t s = 0; s < frame_count; ++s) {
            Real excite[Lanes]{};
            for (size_t t = 0; t < impacts.size(); ++t) {
                const auto force = m.ForceScratch[size_t(impacts[t]) * frame_count + s];
                if (force == Real(0)) continue;
                const auto *gain = &w.Gains[t * Lanes];
                for (uint32_t l = 0; l < Lanes; ++l) excite[l] += force * gain[l];
            }
            Real acc = 0;
            for (uint32_t l = 0; l < Lanes; ++l) {
                const auto re = z_re[l] * c_re[l] - z_im[l] * c_im[l] + excite[l];
                z_im[l] = z_re[l] * c_im[l] + z_im[l] * c_re[l];
                z_re[l] = re;
                acc += p_im[l] * z_im[l] + p_re[l] * re;
            }
            out[s] += acc * mix_gain;
        }
        Real chunk = 0;
        for (uint32_t l = 0; l < width; ++l) {
            b.StateRe[k0 + k + l] = z_re[l];
            b.StateIm[k0 + k + l] = z_im[l];
            chunk += z_re[l] * z_re[l] + z_im[l] * z_im[l];
        }
        energy += chunk;
        if (chunk * out_gain * out_gain >= SilentEnergy<Real>()) live = k + width;
    }
    if (impacts.empty() && energy * out_gain * out_gain < SilentEnergy<Real>()) {
        SilenceObject(b, o);
        return;
    }
    b.Ringing[o] = 1;
    b.LiveModeCount[o] = impacts.empty() ? live : b.TunedModeCount[o];
}

// ModalAudio.cpp:430-461
template<typename Real> void DealObjects(Audio<Real> &m, const Bank<Real> &b, uint32_t count) {
    m.Renderers.resize(count);
    for (auto &r : m.Renderers) r.Objects.clear();
    auto &order = m.RenderOrderScratch;
    order.clear();
    for (uint32_t o = 0; o < uint32_t(b.Entities.size()); ++o) {
        if (!b.Ringing[o]) continue;
        const uint64_t voices = 0; // SurfaceVoiceCount is 0 without the surface model
        bool excited = false;
        for (const auto &im : b.Impacts) excited = excited || im.Object == o;
        order.emplace_back(uint64_t(excited ? b.TunedModeCount[o] : b.LiveModeCount[o]) * (1 + voices), o);
    }
    if (count == 1) {
        for (const auto &[cost, o] : order) m.Renderers.front().Objects.push_back(o);
        return;
    }
    std::sort(order.begin(), order.end(), [](const auto &a, const auto &c) { return a.first != c.first ? a.first > c.first : a.second < c.second; });
    auto &load = m.RenderLoadScratch;
    load.assign(count, 0);
    for (const auto &[cost, o] : order) {
        const auto least = uint32_t(std::min_element(load.begin(), load.end()) - load.begin());
        load[least] += cost;
        m.Renderers[least].Objects.push_back(o);
    }
    for (auto &r : m.Renderers) std::sort(r.Objects.begin(), r.Objects.end());
}

// ModalAudio.cpp:486-590
template<typename Real> void RenderModal(Audio<Real> &m, Real *out, uint32_t frame_count) {
    if (frame_count == 0) return;
    Bank<Real> &b = m.Live;
    if (m.FlushEvents) {
        m.FlushEvents = false;
        m.EventRead = m.EventWrite;
    }
    DrainEvents(m, b);
    const Real click_gain = m.ClickGain;
    const auto impact_count = uint32_t(b.Impacts.size());
    m.ForceScratch.resize(size_t(impact_count) * frame_count);
    for (uint32_t i = 0; i < impact_count; ++i) {
        auto &im = b.Impacts[i];
        Real phase_re = im.PhaseRe, phase_im = im.PhaseIm;
        const auto rot_re = im.RotRe, rot_im = im.RotIm;
        const auto gamma = im.Gamma, amp = im.AccelAmp;
        const auto b0 = im.ClickB0, a1 = im.ClickA1, a2 = im.ClickA2;
        const Real impact_click_gain = click_gain * b.ListenerGain[im.Object];
        auto z1 = im.ClickZ1, z2 = im.ClickZ2;
        auto left = im.SamplesLeft;
        auto *force = &m.ForceScratch[size_t(i) * frame_count];
        for (uint32_t s = 0; s < frame_count; ++s) {
            Real cur{0};
            if (left > 0) {
                const Real re = phase_re * rot_re - phase_im * rot_im;
                phase_im = phase_re * rot_im + phase_im * rot_re;
                phase_re = re;
                cur = gamma * Real(0.5) * (Real(1) - phase_re);
                --left;
            }
            force[s] = cur;
            const Real u = amp * cur;
            const Real y = b0 * u + z1;
            z1 = -a1 * y + z2;
            z2 = -b0 * u - a2 * y;
            out[s] += y * impact_click_gain;
        }
        im.PhaseRe = phase_re;
        im.PhaseIm = phase_im;
        im.SamplesLeft = left;
        im.ClickZ1 = z1;
        im.ClickZ2 = z2;
    }
    DealObjects(m, b, m.PoolSize);
    for (auto &r : m.Renderers) r.Out.assign(frame_count, Real(0));
    // The reference's render pool: one thread per renderer, each with a private Out (ModalAudio.cpp:189-273, 471-483).
    // Renderers own disjoint objects, so the team only changes the wall time, never a sample.
#pragma omp parallel for schedule(static, 1) if (m.Renderers.size() > 1)
    for (size_t ri = 0; ri < m.Renderers.size(); ++ri) {
        auto &w = m.Renderers[ri];
        for (const auto o : w.Objects) {
            w.Impacts.clear();
            for (uint32_t i = 0; i < impact_count; ++i)
                if (b.Impacts[i].Object == o) w.Impacts.push_back(i);
            RenderObjectFast(m, w, b, o, w.Impacts, w.Out.data(), frame_count);
        }
    }
    for (const auto &r : m.Renderers)
        for (uint32_t s = 0; s < frame_count; ++s) out[s] += r.Out[s];

    for (uint32_t i = uint32_t(b.Impacts.size()); i-- > 0;) {
        const auto &im = b.Impacts[i];
        if (im.SamplesLeft == 0 && std::abs(im.ClickZ1) + std::abs(im.ClickZ2) < Real(1e-12f)) RemoveImpact(b, i);
    }
    double energy = 0;
    for (size_t o = 0; o < b.Entities.size(); ++o) {
        if (!b.Ringing[o]) continue;
        const auto k0 = b.ModeOffset[o], count = b.TunedModeCount[o];
        for (uint32_t k = 0; k < count; ++k) {
            const auto g = b.RadiationGain[k0 + k];
            if (g > 0) energy += 0.5 * (double(b.StateRe[k0 + k]) * b.StateRe[k0 + k] + double(b.StateIm[k0 + k]) * b.StateIm[k0 + k]) / (double(g) * g);
        }
    }
    m.ModalEnergy = energy;
    if (energy > m.PeakModalEnergy) m.PeakModalEnergy = energy;
    m.ActiveImpacts = uint32_t(b.Impacts.size());
}

// ModalAudio.cpp:291-338
template<typename Real> uint32_t AddModalObject(Bank<Real> &b, uint32_t entity, uint32_t count, uint32_t n_pos, const float *shapes, const float *positions, uint32_t n_idx, const uint32_t *indices) {
    const auto slot = uint32_t(b.Entities.size());
    b.Entities.push_back(entity);
    b.ModeOffset.push_back(uint32_t(b.CoeffRe.size()));
    b.ModeCount.push_back(count);
    b.TunedModeCount.push_back(count);
    b.LiveModeCount.push_back(count);
    b.ShapeOffset.push_back(uint32_t(b.ShapeX.size()));
    b.Ringing.push_back(0);
    for (auto *col : {&b.OutGain, &b.RigidInvMass}) col->push_back(Real(0));
    for (auto *col : {&b.ListenerGain, &b.DeflectionScale}) col->push_back(Real(1));
    for (auto *col : {&b.CoeffRe, &b.CoeffIm, &b.StateRe, &b.StateIm, &b.RadiationGain, &b.DeflectionGain, &b.QuadCompliance, &b.QuadDriveScale}) col->resize(col->size() + count, Real(0));
    b.OutPhaseIm.resize(b.OutPhaseIm.size() + count, Real(1));
    b.OutPhaseRe.resize(b.OutPhaseRe.size() + count, Real(0));
    auto shape = [&](uint32_t p, uint32_t k) { return V3<Real>{Real(shapes[(size_t(p) * count + k) * 3]), Real(shapes[(size_t(p) * count + k) * 3 + 1]), Real(shapes[(size_t(p) * count + k) * 3 + 2])}; };
    auto position = [&](uint32_t p) { return V3<Real>{Real(positions[3 * p]), Real(positions[3 * p + 1]), Real(positions[3 * p + 2])}; };
    for (uint32_t p = 0; p < n_pos; ++p)
        for (uint32_t k = 0; k < count; ++k) {
            const auto s = shape(p, k);
            b.ShapeX.push_back(s.x);
            b.ShapeY.push_back(s.y);
            b.ShapeZ.push_back(s.z);
        }
    const auto area_offset = b.RadiationGain.size() - count;
    Real total_area = 0;
    b.RadiationArea.resize(area_offset + count, Real(0));
    for (size_t t = 0; t + 2 < n_idx; t += 3) {
        const auto i = indices[t], j = indices[t + 1], l = indices[t + 2];
        const auto pi = position(i), pj = position(j), pl = position(l);
        const V3<Real> e1{pj.x - pi.x, pj.y - pi.y, pj.z - pi.z}, e2{pl.x - pi.x, pl.y - pi.y, pl.z - pi.z};
        const V3<Real> cr{e1.y * e2.z - e2.y * e1.z, e1.z * e2.x - e2.z * e1.x, e1.x * e2.y - e2.x * e1.y};
        const Real doubled = std::sqrt(cr.x * cr.x + cr.y * cr.y + cr.z * cr.z);
        if (doubled <= Real(0)) continue;
        const V3<Real> n{cr.x / doubled, cr.y / doubled, cr.z / doubled};
        const Real area = doubled / 2;
        total_area += area;
        for (uint32_t k = 0; k < count; ++k) {
            const auto si = shape(i, k), sj = shape(j, k), sl = shape(l, k);
            const V3<Real> mean{(si.x + sj.x + sl.x) / Real(3), (si.y + sj.y + sl.y) / Real(3), (si.z + sj.z + sl.z) / Real(3)};
            const Real normal = mean.x * n.x + mean.y * n.y + mean.z * n.z;
            b.RadiationArea[area_offset + k] += area * normal * normal;
        }
    }
    b.RadiantRadius.push_back(std::sqrt(total_area / (4 * Real(Pi))));
    return slot;
}

// ModalAudio.cpp:340-393
template<typename Real> void TuneModalObject(Bank<Real> &b, uint32_t object, uint32_t n, const float *freqs, const float *t60s, float radius_scale_f) {
    const auto k0 = b.ModeOffset[object];
    const auto count = std::min(b.ModeCount[object], n);
    const Real sr = b.SampleRate;
    const Real radius_scale = radius_scale_f;
    const Real radius = b.RadiantRadius[object] * radius_scale;
    const Real ln1000 = 3 * Real(2.302585092994045684);
    b.DeflectionScale[object] = Real(1) / (radius_scale * radius_scale * radius_scale);
    for (uint32_t k = 0; k < count; ++k) {
        const Real freq = freqs[k], t60 = t60s[k];
        if (!std::isfinite(freq) || !std::isfinite(t60) || freq <= 0 || freq >= sr / 2 - 1 || t60 <= 0) {
            b.CoeffRe[k0 + k] = 0; b.CoeffIm[k0 + k] = 0; b.RadiationGain[k0 + k] = 0; b.DeflectionGain[k0 + k] = 0;
            b.OutPhaseIm[k0 + k] = 1; b.OutPhaseRe[k0 + k] = 0; b.QuadCompliance[k0 + k] = 0; b.QuadDriveScale[k0 + k] = 0;
            continue;
        }
        const Real omega = 2 * Real(Pi) * freq / sr;
        const Real omega_si = 2 * Real(Pi) * freq;
        const Real ka = omega_si * radius / Real(SpeedOfSound);
        const Real sigma = ka * ka / (1 + ka * ka);
        const Real area = b.RadiationArea[k0 + k] / radius_scale;
        const Real radiation_rate = Real(AirDensity) * Real(SpeedOfSound) * sigma * area * Real(0.5);
        const Real decay = std::exp(-(ln1000 / t60 + radiation_rate) / sr);
        b.CoeffRe[k0 + k] = decay * std::cos(omega);
        b.CoeffIm[k0 + k] = decay * std::sin(omega);
        const Real gain = Real(AirDensity) * Real(SpeedOfSound) * std::sqrt(sigma * b.RadiationArea[k0 + k] / (4 * Real(Pi))) / Real(ListenerDistance);
        b.RadiationGain[k0 + k] = gain;
        const Real spread = sigma * Real(Pi) * (Real(2) * std::fmod(Real(0.6180339887f) * Real(k + 1), Real(1)) - Real(1));
        b.OutPhaseIm[k0 + k] = std::cos(spread);
        b.OutPhaseRe[k0 + k] = std::sin(spread);
        b.DeflectionGain[k0 + k] = gain > 0 ? Real(1) / (gain * omega_si) : Real(0);
        const Real dt = Real(1) / sr;
        const Real central = dt * (1 + decay * decay + 2 * decay * std::cos(omega)) / 4;
        b.QuadCompliance[k0 + k] = central;
        b.QuadDriveScale[k0 + k] = central * omega_si / (decay * std::sin(omega));
    }
    uint32_t live = b.ModeCount[object];
    while (live > 0 && b.CoeffRe[k0 + live - 1] == 0 && b.CoeffIm[k0 + live - 1] == 0) --live;
    b.TunedModeCount[object] = live;
    b.LiveModeCount[object] = live;
}

template<typename Real> bool SetModalObjectShapes(Bank<Real> &b, uint32_t object, uint32_t count, uint32_t n_pos, const float *shapes) {
    const auto begin = b.ShapeOffset[object];
    const auto end = object + 1 < b.ShapeOffset.size() ? b.ShapeOffset[object + 1] : uint32_t(b.ShapeX.size());
    if (b.ModeCount[object] != count || end - begin != count * n_pos) return false;
    auto i = begin;
    for (uint32_t p = 0; p < n_pos; ++p)
        for (uint32_t k = 0; k < count; ++k) {
            b.ShapeX[i] = shapes[(size_t(p) * count + k) * 3];
            b.ShapeY[i] = shapes[(size_t(p) * count + k) * 3 + 1];
            b.ShapeZ[i] = shapes[(size_t(p) * count + k) * 3 + 2];
            ++i;
        }
    return true;
}

template<typename Real> const std::vector<Real> *Column(const Bank<Real> &b, int which) {
    switch (which) {
        case 0: return &b.CoeffRe; case 1: return &b.CoeffIm; case 2: return &b.StateRe; case 3: return &b.StateIm;
        case 4: return &b.RadiationGain; case 5: return &b.RadiationArea; case 6: return &b.DeflectionGain;
        case 7: return &b.OutPhaseIm; case 8: return &b.OutPhaseRe; case 9: return &b.QuadCompliance; case 10: return &b.QuadDriveScale;
        case 11: return &b.ShapeX; case 12: return &b.ShapeY; case 13: return &b.ShapeZ;
        case 14: return &b.OutGain; case 15: return &b.ListenerGain; case 16: return &b.RadiantRadius; case 17: return &b.DeflectionScale;
        default: return nullptr;
    }
}
} // namespace

struct mo_bank {
    bool dbl;
    Audio<float> f;
    Audio<double> d;
};

#define DISPATCH(expr_f, expr_d) (bank->dbl ? (expr_d) : (expr_f))

extern "C" {
mo_bank *mo_bank_create(float sample_rate, int use_double) {
    auto *b = new mo_bank{};
    b->dbl = use_double != 0;
    b->f.Next.SampleRate = sample_rate;
    b->d.Next.SampleRate = sample_rate;
    b->f.Live.SampleRate = sample_rate;
    b->d.Live.SampleRate = sample_rate;
    return b;
}
void mo_bank_free(mo_bank *bank) { delete bank; }
uint32_t mo_bank_add_object(mo_bank *bank, uint32_t entity, uint32_t n_modes, uint32_t n_pos, const float *shapes, const float *positions, uint32_t n_indices, const uint32_t *indices) {
    return DISPATCH(AddModalObject(bank->f.Next, entity, n_modes, n_pos, shapes, positions, n_indices, indices),
                    AddModalObject(bank->d.Next, entity, n_modes, n_pos, shapes, positions, n_indices, indices));
}
void mo_bank_tune_object(mo_bank *bank, int live, uint32_t object, uint32_t n, const float *freqs, const float *t60s, float radius_scale) {
    if (bank->dbl) TuneModalObject(live ? bank->d.Live : bank->d.Next, object, n, freqs, t60s, radius_scale);
    else TuneModalObject(live ? bank->f.Live : bank->f.Next, object, n, freqs, t60s, radius_scale);
}
int mo_bank_set_shapes(mo_bank *bank, int live, uint32_t object, uint32_t n_modes, uint32_t n_pos, const float *shapes) {
    return DISPATCH(SetModalObjectShapes(live ? bank->f.Live : bank->f.Next, object, n_modes, n_pos, shapes),
                    SetModalObjectShapes(live ? bank->d.Live : bank->d.Next, object, n_modes, n_pos, shapes));
}
void mo_bank_set_gains(mo_bank *bank, int live, uint32_t object, float out_gain, float listener_gain) {
    if (bank->dbl) { auto &b = live ? bank->d.Live : bank->d.Next; b.OutGain[object] = out_gain; b.ListenerGain[object] = listener_gain; }
    else { auto &b = live ? bank->f.Live : bank->f.Next; b.OutGain[object] = out_gain; b.ListenerGain[object] = listener_gain; }
}
// InstallModalBank (ModalAudio.cpp:277-289): publish Next, flag queued events for flushing.
void mo_bank_install(mo_bank *bank) {
    if (bank->dbl) { bank->d.Live = std::move(bank->d.Next); bank->d.Next = {}; bank->d.Next.SampleRate = bank->d.Live.SampleRate; bank->d.FlushEvents = true; }
    else { bank->f.Live = std::move(bank->f.Next); bank->f.Next = {}; bank->f.Next.SampleRate = bank->f.Live.SampleRate; bank->f.FlushEvents = true; }
}
// ModalRenderPool::SetSize (ModalAudio.cpp:238-243): clamped to [1, hardware threads].
void mo_bank_set_renderers(mo_bank *bank, uint32_t count) {
    const auto cores = std::max(1u, std::thread::hardware_concurrency());
    const uint32_t w = std::clamp(count, 1u, cores);
    bank->f.PoolSize = w;
    bank->d.PoolSize = w;
}
void mo_bank_set_click_gain(mo_bank *bank, float g) { bank->f.ClickGain = g; bank->d.ClickGain = g; }
void mo_bank_set_max_impacts(mo_bank *bank, uint32_t n) { bank->f.MaxImpacts = n; bank->d.MaxImpacts = n; }
// EnqueueModalEvent (ModalAudio.cpp:417-425).  Returns 0 when the queue was full and the event dropped.
int mo_bank_enqueue(mo_bank *bank, const mo_event *e) {
    auto push = [&](auto &m) {
        if (m.EventWrite - m.EventRead >= EventCapacity) { ++m.EventsDropped; return 0; }
        m.Events[m.EventWrite % EventCapacity] = *e;
        ++m.EventWrite;
        return 1;
    };
    return bank->dbl ? push(bank->d) : push(bank->f);
}
void mo_bank_render_f32(mo_bank *bank, float *out, uint32_t frames) { RenderModal(bank->f, out, frames); }
void mo_bank_render_f64(mo_bank *bank, double *out, uint32_t frames) { RenderModal(bank->d, out, frames); }
uint32_t mo_bank_num_objects(const mo_bank *bank) { return uint32_t(DISPATCH(bank->f.Live.Entities.size(), bank->d.Live.Entities.size())); }
uint32_t mo_bank_num_modes(const mo_bank *bank) { return uint32_t(DISPATCH(bank->f.Live.CoeffRe.size(), bank->d.Live.CoeffRe.size())); }
uint32_t mo_bank_active_impacts(const mo_bank *bank) { return DISPATCH(bank->f.ActiveImpacts, bank->d.ActiveImpacts); }
double mo_bank_modal_energy(const mo_bank *bank) { return DISPATCH(bank->f.ModalEnergy, bank->d.ModalEnergy); }
uint64_t mo_bank_events_dropped(const mo_bank *bank) { return DISPATCH(bank->f.EventsDropped, bank->d.EventsDropped); }
// Copy a column of the live (or next) bank, widened to double.  Returns its length.
uint32_t mo_bank_column(const mo_bank *bank, int live, int which, double *out) {
    auto copy = [&](const auto *col) -> uint32_t {
        if (!col) return 0;
        if (out) for (size_t i = 0; i < col->size(); ++i) out[i] = double((*col)[i]);
        return uint32_t(col->size());
    };
    return bank->dbl ? copy(Column(live ? bank->d.Live : bank->d.Next, which)) : copy(Column(live ? bank->f.Live : bank->f.Next, which));
}
void mo_bank_object_state(const mo_bank *bank, uint32_t *tuned, uint32_t *live, uint8_t *ringing) {
    auto copy = [&](const auto &b) {
        for (size_t o = 0; o < b.Entities.size(); ++o) {
            if (tuned) tuned[o] = b.TunedModeCount[o];
            if (live) live[o] = b.LiveModeCount[o];
            if (ringing) ringing[o] = b.Ringing[o];
        }
    };
    if (bank->dbl) copy(bank->d.Live); else copy(bank->f.Live);
}

// ---- recoil filters, ModalAudio.h:58-99 ----
static void RecoilDenominator(double wc, double kk, double beta, double *a0, float *a1, float *a2) {
    *a0 = kk * kk + beta * wc * kk + beta * wc * wc;
    *a1 = float((2 * beta * wc * wc - 2 * kk * kk) / *a0);
    *a2 = float((kk * kk - beta * wc * kk + beta * wc * wc) / *a0);
}
void mo_recoil_object_filter(double radius, double volume, double sample_rate, float out6[6]) {
    for (int i = 0; i < 6; ++i) out6[i] = 0;
    if (radius <= 0 || volume <= 0) return;
    const double wc = SpeedOfSound / radius, kk = 2 * sample_rate;
    double a0; float a1, a2;
    RecoilDenominator(wc, kk, 2, &a0, &a1, &a2);
    const double gp = AirDensity * SpeedOfSound * radius / ListenerDistance;
    const double n2 = AirDensity * volume * wc, n1 = n2 * wc;
    out6[0] = float(gp * kk * kk / a0);
    out6[1] = float((n2 * kk * kk + n1 * kk) / a0);
    out6[2] = float(-2 * n2 * kk * kk / a0);
    out6[3] = float((n2 * kk * kk - n1 * kk) / a0);
    out6[4] = a1;
    out6[5] = a2;
}
void mo_recoil_click_filter(double radius, double volume, double mass, double sample_rate, float out3[3]) {
    out3[0] = out3[1] = out3[2] = 0;
    if (radius <= 0 || mass <= 0) return;
    const double wc = SpeedOfSound / radius, kk = 2 * sample_rate;
    double a0; float a1, a2;
    RecoilDenominator(wc, kk, 2 + AirDensity * volume / mass, &a0, &a1, &a2);
    const double g = AirDensity * SpeedOfSound * radius / (ListenerDistance * mass);
    out3[0] = float(g * kk / a0);
    out3[1] = a1;
    out3[2] = a2;
}

// ---- contact model, src/audio/ContactModel.cpp ----
double mo_striker_mass(double density, float tip_radius, float length) {
    const double r = tip_radius, l = length;
    return density * Pi * (r * r * l + 4.0 / 3.0 * r * r * r);
}
static double InvEffectiveModulus(const mo_material &a, const mo_material &b) {
    return (1 - a.poisson_ratio * a.poisson_ratio) / a.young_modulus + (1 - b.poisson_ratio * b.poisson_ratio) / b.young_modulus;
}
static double CombinedCurvature(double a, double b) { return std::max(a + b, 1e-6); }
static double ContactStiffness(double inv_mod, double curv) { return 4.0 / 3.0 / inv_mod / std::sqrt(curv); }
double mo_contact_patch_radius(double normal_force, double inv_mod, double curv) { return std::cbrt(0.75 * std::max(normal_force, 0.0) * inv_mod / curv); }
double mo_static_penetration(double normal_force, double stiffness) { return stiffness > 0 ? std::pow(std::max(normal_force, 0.0) / stiffness, 2.0 / 3.0) : 0.0; }
double mo_saturation_penetration(double curv, double area) { return area > 0 ? area * curv / Pi : std::numeric_limits<double>::infinity(); }
double mo_punch_stiffness(double inv_mod, double area) {
    if (area <= 0) return std::numeric_limits<double>::infinity();
    return 2 * std::sqrt(area / Pi) / inv_mod;
}
static double ContactWork(double pen, double hk, double sat, double pk) {
    if (pen <= 0) return 0;
    const auto hertz = [hk](double x) { return 0.4 * hk * x * x * std::sqrt(x); };
    if (pen <= sat) return hertz(pen);
    const double over = pen - sat;
    const double sat_force = hk * sat * std::sqrt(sat);
    return hertz(sat) + sat_force * over + 0.5 * pk * over * over;
}
// InverseInertiaTensor (ContactModel.cpp:18-25): quat w,x,y,z -> column-major float 3x3.
void mo_inverse_inertia_tensor(const float inertia_diag[3], const float q[4], float out9[9]) {
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    // glm::mat3_cast, r[col][row]
    float r[3][3];
    const float qxx = x * x, qyy = y * y, qzz = z * z, qxz = x * z, qxy = x * y, qyz = y * z, qwx = w * x, qwy = w * y, qwz = w * z;
    r[0][0] = 1.f - 2.f * (qyy + qzz); r[0][1] = 2.f * (qxy + qwz); r[0][2] = 2.f * (qxz - qwy);
    r[1][0] = 2.f * (qxy - qwz); r[1][1] = 1.f - 2.f * (qxx + qzz); r[1][2] = 2.f * (qyz + qwx);
    r[2][0] = 2.f * (qxz + qwy); r[2][1] = 2.f * (qyz - qwx); r[2][2] = 1.f - 2.f * (qxx + qyy);
    float inv[3];
    for (int i = 0; i < 3; ++i) inv[i] = inertia_diag[i] > 0 ? 1.f / inertia_diag[i] : 0.f;
    // r * diag(inv) * r^T
    for (int c = 0; c < 3; ++c)
        for (int rr = 0; rr < 3; ++rr) {
            float s = 0;
            for (int k = 0; k < 3; ++k) s += r[k][rr] * inv[k] * r[k][c];
            out9[c * 3 + rr] = s;
        }
}
double mo_reduced_contact_mass(double mass, const float inv_inertia9[9], const float arm[3], const float dir[3], double impactor_inv_mass) {
    if (mass <= 0) return 0;
    const float len = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    const float n[3] = {dir[0] / len, dir[1] / len, dir[2] / len};
    const float c[3] = {arm[1] * n[2] - n[1] * arm[2], arm[2] * n[0] - n[2] * arm[0], arm[0] * n[1] - n[0] * arm[1]};
    float ic[3];
    for (int r = 0; r < 3; ++r) ic[r] = inv_inertia9[0 * 3 + r] * c[0] + inv_inertia9[1 * 3 + r] * c[1] + inv_inertia9[2 * 3 + r] * c[2];
    const float d = c[0] * ic[0] + c[1] * ic[1] + c[2] * ic[2];
    const double inv_eff = 1.0 / mass + d + impactor_inv_mass;
    return 1.0 / inv_eff;
}
double mo_estimate_contact_time(double mass, const float inv_inertia9[9], const float arm[3], const float dir[3], double contact_speed,
                                const mo_material *object_material, double object_curvature, double nominal_area,
                                const mo_material *impactor_material, double impactor_curvature, double impactor_inv_mass,
                                double scale_ratio, double combined_roughness) {
    constexpr double MinContactTime = 2e-5, MaxContactTime = 5e-2;
    if (mass <= 0) return MinContactTime;
    const double effective_mass = mo_reduced_contact_mass(mass, inv_inertia9, arm, dir, impactor_inv_mass);
    const double inv_mod = InvEffectiveModulus(*object_material, *impactor_material);
    if (effective_mass <= 0 || inv_mod <= 0) return MinContactTime;
    const double curvature = CombinedCurvature(object_curvature, impactor_curvature);
    const double speed = std::max(std::abs(contact_speed), 1e-6);
    const double hk = ContactStiffness(inv_mod, curvature);
    const double sat = mo_saturation_penetration(curvature, nominal_area);
    const double pk = mo_punch_stiffness(inv_mod, nominal_area);
    const double energy = 0.5 * effective_mass * speed * speed;
    const double sat_work = std::isfinite(sat) ? ContactWork(sat, hk, sat, pk) : std::numeric_limits<double>::infinity();
    double max_pen;
    if (energy <= sat_work) {
        max_pen = std::pow(energy / (0.4 * hk), 0.4);
    } else {
        const double sat_force = hk * sat * std::sqrt(sat);
        max_pen = sat + (std::sqrt(sat_force * sat_force + 2 * pk * (energy - sat_work)) - sat_force) / pk;
    }
    constexpr int Steps = 64;
    double sum = 0;
    for (int n = 0; n < Steps; ++n) {
        const double s = (double(n) + 0.5) / Steps;
        const double left = 1 - ContactWork(max_pen * (1 - s * s), hk, sat, pk) / energy;
        if (left > 0) sum += 2 * s / std::sqrt(left);
    }
    const double bulk_time = 2 * max_pen / speed * sum / Steps * scale_ratio;
    const double u0 = 0.4 * combined_roughness;
    const double bed_time = std::sqrt(2.0) * Pi * u0 / speed;
    return std::clamp(std::sqrt(bulk_time * bulk_time + bed_time * bed_time), MinContactTime, MaxContactTime);
}
}
