// Host side of the resonator bank over libmodalhip.
//
// What lives here is bookkeeping: building and tuning the struct-of-arrays bank, the single-producer event ring, the
// impact list, the deterministic deal of objects to renderers and the publish protocol.  Every per-sample operation
// (force curves, click filters, mode recurrences, ordered mix) runs on the device in mh_bank_render.
//
// Written from the behaviour described in SURVEY.md section 8a rows R0-R8, not from the reference's text.  Where a
// value must be BIT-identical to the reference's (coefficient columns are compared bit for bit with the CPU oracle),
// the floating-point expression tree is pinned by a comment naming the reference line it reproduces; statement order,
// loop structure and naming are this file's own.  Both precisions (ModalBank / ModalBank64) share every function
// below through templates; in `Real = float` the arithmetic is the reference's fp32 arithmetic.
#include "modal/bank.hpp"

#include "modalhip.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <queue>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>

// ------------------------------------------------------------------------------------------------------------------
// Device mirror (no reference counterpart)
// ------------------------------------------------------------------------------------------------------------------
struct ModalDeviceMirror {
    mh_context *Context{nullptr};
    mh_bank *Bank{nullptr};
    const void *Source{nullptr}; // the host bank the device copy was made from
    // per-block staging, kept between blocks to avoid reallocation
    std::vector<uint32_t> ImpactsOn, DealOffset, DealObjects, RenderCount, Tuned, Live;
    std::vector<double> Energy, ModalEnergy;
    std::vector<uint8_t> Silenced;
    std::vector<mh_impact> Impacts;
    std::vector<std::vector<uint32_t>> Renderers;
    ~ModalDeviceMirror() {
        mh_bank_destroy(Bank);
        mh_context_destroy(Context);
    }
};

template<typename Bank> ModalAudioCore<Bank>::ModalAudioCore() : Live{std::make_unique<Bank>()}, Published{Live.get()}, Dev{std::make_unique<ModalDeviceMirror>()} {}
template<typename Bank> ModalAudioCore<Bank>::~ModalAudioCore() = default;
template struct ModalAudioCore<ModalBank>;
template struct ModalAudioCore<ModalBank64>;

void ModalRenderPool::SetSize(uint32_t count) {
    // the reference never runs more renderers than the host has hardware threads, and never fewer than one
    const uint32_t ceiling = std::max(1u, std::thread::hardware_concurrency());
    Active = count < 1u ? 1u : (count > ceiling ? ceiling : count);
}

// ------------------------------------------------------------------------------------------------------------------
// Recoil radiator filters -- ModalAudio.h:58-99
// An analogue second-order section  N(s) / (s^2 + beta wc s + beta wc^2)  through the bilinear map s -> K (1-z^-1)/(1+z^-1),
// K = 2 SR.  Evaluated in double; the digital coefficients are rounded to float once.
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct BilinearSection {
    double KK, BetaWcK, BetaWcWc, A0;
    BilinearSection(double wc, double k, double beta) {
        const double beta_wc = beta * wc;
        KK = k * k;
        BetaWcK = beta_wc * k;
        BetaWcWc = beta_wc * wc;
        A0 = KK + BetaWcK + BetaWcWc; // ModalAudio.h:60  kk*kk + beta*wc*kk + beta*wc*wc, left to right
    }
    // doubling is exact, so 2*x*y == 2*(x*y) bit for bit
    float A1() const { return float((2 * BetaWcWc - 2 * KK) / A0); } // ModalAudio.h:61
    float A2() const { return float((KK - BetaWcK + BetaWcWc) / A0); } // ModalAudio.h:62
};
// rho0 * c0 as the reference forms it: a float product, widened afterwards
const double AirImpedance = double(AirDensity * SpeedOfSound);
} // namespace

RecoilPoles RecoilDenominator(double wc, double kk, double beta) {
    const BilinearSection s(wc, kk, beta);
    return {s.A0, s.A1(), s.A2()};
}

RecoilFilter RecoilObjectFilter(double radius, double volume, double sample_rate) {
    RecoilFilter f;
    if (!(radius > 0) || !(volume > 0)) return f;
    const double corner = SpeedOfSound / radius, k = 2 * sample_rate;
    const BilinearSection s(corner, k, 2);
    const double far_field = AirImpedance * radius / ListenerDistance; // ModalAudio.h:73
    const double added_mass = AirDensity * volume * corner; // :74 n2
    const double added_drag = added_mass * corner; // :74 n1
    const double quad = added_mass * k * k, lin = added_drag * k;
    f.RadB0 = float(far_field * k * k / s.A0); // :76
    f.AirB0 = float((quad + lin) / s.A0); // :77
    f.AirB1 = float(-2 * quad / s.A0); // :78  (-2*n2*kk*kk == -2*(n2*kk*kk) exactly)
    f.AirB2 = float((quad - lin) / s.A0); // :79
    f.A1 = s.A1();
    f.A2 = s.A2();
    return f;
}

ClickFilter RecoilClickFilter(double radius, double volume, double mass, double sample_rate) {
    ClickFilter f;
    if (!(radius > 0) || !(mass > 0)) return f;
    const double corner = SpeedOfSound / radius, k = 2 * sample_rate;
    const BilinearSection s(corner, k, 2 + AirDensity * volume / mass); // :95
    const double per_mass = AirImpedance * radius / (ListenerDistance * mass); // :96
    f.B0 = float(per_mass * k / s.A0); // :97
    f.A1 = s.A1();
    f.A2 = s.A2();
    return f;
}

// ------------------------------------------------------------------------------------------------------------------
// Bank construction
// ------------------------------------------------------------------------------------------------------------------
namespace {
template<typename Real> struct Triple {
    Real x, y, z;
};
template<typename Real> Triple<Real> Widen(const vec3 &v) { return {Real(v.x), Real(v.y), Real(v.z)}; }
template<typename Real> Triple<Real> Minus(Triple<Real> a, Triple<Real> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template<typename Real> Real Dot(Triple<Real> a, Triple<Real> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// component order and operand order of glm::cross
template<typename Real> Triple<Real> Cross(Triple<Real> a, Triple<Real> b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }

// Every per-object and per-mode column with the value a fresh slot starts from: one table instead of a list of push_backs.
template<typename Real> struct ColumnInit {
    std::vector<Real> ModalBankColumns<Real>::*Column;
    Real Value;
};
template<typename Real> constexpr ColumnInit<Real> ObjectColumns[] = {
    {&ModalBankColumns<Real>::OutGain, 0}, {&ModalBankColumns<Real>::ListenerGain, 1}, {&ModalBankColumns<Real>::DeflectionScale, 1},
    {&ModalBankColumns<Real>::RigidInvMass, 0}, {&ModalBankColumns<Real>::RadiatorB0, 0}, {&ModalBankColumns<Real>::AirB0, 0},
    {&ModalBankColumns<Real>::AirB1, 0}, {&ModalBankColumns<Real>::AirB2, 0}, {&ModalBankColumns<Real>::RecoilA1, 0},
    {&ModalBankColumns<Real>::RecoilA2, 0}, {&ModalBankColumns<Real>::RadiatorZ1, 0}, {&ModalBankColumns<Real>::RadiatorZ2, 0},
    {&ModalBankColumns<Real>::AirZ1, 0}, {&ModalBankColumns<Real>::AirZ2, 0},
};
template<typename Real> constexpr ColumnInit<Real> ModeColumns[] = {
    {&ModalBankColumns<Real>::CoeffRe, 0}, {&ModalBankColumns<Real>::CoeffIm, 0}, {&ModalBankColumns<Real>::StateRe, 0},
    {&ModalBankColumns<Real>::StateIm, 0}, {&ModalBankColumns<Real>::RadiationGain, 0}, {&ModalBankColumns<Real>::RadiationArea, 0},
    {&ModalBankColumns<Real>::DeflectionGain, 0}, {&ModalBankColumns<Real>::OutPhaseIm, 1}, {&ModalBankColumns<Real>::OutPhaseRe, 0},
    {&ModalBankColumns<Real>::QuadCompliance, 0}, {&ModalBankColumns<Real>::QuadDriveScale, 0},
};

// A facet of the sample surface with its unit normal and area; degenerate triangles never make it into the list.
template<typename Real> struct Facet {
    uint32_t A, B, C;
    Triple<Real> Normal;
    Real Area;
};

template<typename Real> uint32_t AppendObject(ModalBankColumns<Real> &b, entt::entity who, const ModalModes &modes) {
    const uint32_t slot = uint32_t(b.Entities.size()), n_modes = uint32_t(modes.Freqs.size()), first_mode = uint32_t(b.CoeffRe.size());
    b.Entities.push_back(who);
    b.ModeOffset.push_back(first_mode);
    b.ShapeOffset.push_back(uint32_t(b.ShapeX.size()));
    for (auto *counts : {&b.ModeCount, &b.TunedModeCount, &b.LiveModeCount}) counts->push_back(n_modes);
    b.Ringing.push_back(0);
    b.RigidVel.emplace_back(0.f);
    for (const auto &c : ObjectColumns<Real>) (b.*c.Column).push_back(c.Value);
    for (const auto &c : ModeColumns<Real>) (b.*c.Column).resize(first_mode + n_modes, c.Value);

    // shapes, position-major within the object (ModalAudio.h:118)
    for (const auto &at_position : modes.Shapes)
        for (const vec3 &s : at_position) {
            b.ShapeX.push_back(Real(s.x));
            b.ShapeY.push_back(Real(s.y));
            b.ShapeZ.push_back(Real(s.z));
        }

    // Radiating strength of each mode: one-point (centroid) quadrature of (normal . shape)^2 over the sample surface,
    // and the radius of the sphere with the same surface area (ModalAudio.cpp:318-337).  Facets first, then one pass
    // per mode over them -- each mode's sum runs over the triangles in index order, as the reference's does.
    std::vector<Facet<Real>> facets;
    Real surface = 0;
    for (size_t t = 0; t + 2 < modes.Indices.size(); t += 3) {
        const uint32_t a = modes.Indices[t], bb = modes.Indices[t + 1], c = modes.Indices[t + 2];
        const auto origin = Widen<Real>(modes.Positions[a]);
        const auto twice = Cross(Minus(Widen<Real>(modes.Positions[bb]), origin), Minus(Widen<Real>(modes.Positions[c]), origin));
        const Real twice_area = std::sqrt(Dot(twice, twice));
        if (!(twice_area > 0)) continue;
        const Real area = twice_area / 2;
        surface += area;
        facets.push_back({a, bb, c, {twice.x / twice_area, twice.y / twice_area, twice.z / twice_area}, area});
    }
    for (uint32_t k = 0; k < n_modes; ++k) {
        Real strength = 0;
        for (const auto &f : facets) {
            const auto sa = Widen<Real>(modes.Shapes[f.A][k]), sb = Widen<Real>(modes.Shapes[f.B][k]), sc = Widen<Real>(modes.Shapes[f.C][k]);
            const Triple<Real> mean{(sa.x + sb.x + sc.x) / Real(3), (sa.y + sb.y + sc.y) / Real(3), (sa.z + sb.z + sc.z) / Real(3)};
            const Real along = Dot(mean, f.Normal);
            strength += f.Area * along * along; // :333  area * normal * normal
        }
        b.RadiationArea[first_mode + k] = strength;
    }
    b.RadiantRadius.push_back(std::sqrt(surface / (4 * std::numbers::pi_v<Real>))); // :336
    return slot;
}

// ---- tuning: frequencies and decay times -> resonator coefficients (ModalAudio.cpp:340-393) ----
template<typename Real> struct ModeCoefficients {
    Real CoeffRe{0}, CoeffIm{0}, RadiationGain{0}, DeflectionGain{0}, OutPhaseIm{1}, OutPhaseRe{0}, QuadCompliance{0}, QuadDriveScale{0};
};
template<typename Real> struct TuningFrame {
    Real SampleRate, Radius, RadiusScale;
};
template<typename Real> bool Audible(Real freq, Real t60, Real sample_rate) {
    return std::isfinite(freq) && std::isfinite(t60) && freq > 0 && freq < sample_rate / 2 - 1 && t60 > 0; // :349
}
// One mode.  `ordinal` is the 1-based mode number, `strength` the mode's RadiationArea.
template<typename Real> ModeCoefficients<Real> Resonator(Real freq, Real t60, Real strength, uint32_t ordinal, const TuningFrame<Real> &env) {
    constexpr Real Pi = std::numbers::pi_v<Real>, Rho = Real(AirDensity), C0 = Real(SpeedOfSound);
    constexpr Real LogThousand = 3 * std::numbers::ln10_v<Real>;
    ModeCoefficients<Real> m;
    const Real per_sample = 2 * Pi * freq / env.SampleRate; // :357 omega
    const Real per_second = 2 * Pi * freq; // :358 omega_si
    const Real ka = per_second * env.Radius / C0;
    const Real efficiency = ka * ka / (1 + ka * ka); // :360 sigma
    const Real scaled_strength = strength / env.RadiusScale;
    const Real air_damping = Rho * C0 * efficiency * scaled_strength * Real(0.5); // :362
    const Real shrink = std::exp(-(LogThousand / t60 + air_damping) / env.SampleRate); // :363
    const Real cos_w = std::cos(per_sample), sin_w = std::sin(per_sample);
    m.CoeffRe = shrink * cos_w;
    m.CoeffIm = shrink * sin_w;
    m.RadiationGain = Rho * C0 * std::sqrt(efficiency * strength / (4 * Pi)) / Real(ListenerDistance); // :366
    // golden-ratio phase spread, weighted by the radiation efficiency (:376-378)
    const Real spread = efficiency * Pi * (Real(2) * std::fmod(Real(0.6180339887f) * Real(ordinal), Real(1)) - Real(1));
    m.OutPhaseIm = std::cos(spread);
    m.OutPhaseRe = std::sin(spread);
    m.DeflectionGain = m.RadiationGain > 0 ? Real(1) / (m.RadiationGain * per_second) : Real(0); // :380
    const Real dt = Real(1) / env.SampleRate;
    m.QuadCompliance = dt * (1 + shrink * shrink + 2 * shrink * cos_w) / 4; // :383
    m.QuadDriveScale = m.QuadCompliance * per_second / (shrink * sin_w); // :385
    return m;
}

template<typename Real> void Tune(ModalBankColumns<Real> &b, uint32_t object, std::span<const float> freqs, std::span<const float> t60s, float radius_scale_f) {
    const uint32_t base = b.ModeOffset[object], owned = b.ModeCount[object];
    const uint32_t given = uint32_t(std::min<size_t>({size_t(owned), freqs.size(), t60s.size()}));
    const Real radius_scale = Real(radius_scale_f);
    const TuningFrame<Real> env{b.SampleRate, b.RadiantRadius[object] * radius_scale, radius_scale};
    b.DeflectionScale[object] = Real(1) / (radius_scale * radius_scale * radius_scale); // :346
    for (uint32_t k = 0; k < given; ++k) {
        const Real f = Real(freqs[k]), t = Real(t60s[k]);
        const auto m = Audible(f, t, env.SampleRate) ? Resonator(f, t, b.RadiationArea[base + k], k + 1, env) : ModeCoefficients<Real>{};
        const uint32_t at = base + k;
        b.CoeffRe[at] = m.CoeffRe;
        b.CoeffIm[at] = m.CoeffIm;
        b.RadiationGain[at] = m.RadiationGain;
        b.DeflectionGain[at] = m.DeflectionGain;
        b.OutPhaseIm[at] = m.OutPhaseIm;
        b.OutPhaseRe[at] = m.OutPhaseRe;
        b.QuadCompliance[at] = m.QuadCompliance;
        b.QuadDriveScale[at] = m.QuadDriveScale;
    }
    // the tuned set ends at the last mode that still rotates (:388-391)
    uint32_t tail = owned;
    while (tail > 0 && b.CoeffRe[base + tail - 1] == 0 && b.CoeffIm[base + tail - 1] == 0) --tail;
    b.TunedModeCount[object] = b.LiveModeCount[object] = tail;
    b.EditedModes.Mark(base, base + owned);
}

template<typename Real> bool OverwriteShapes(ModalBankColumns<Real> &b, uint32_t object, const ModalModes &modes) {
    const uint32_t first = b.ShapeOffset[object];
    const uint32_t last = object + 1 < b.ShapeOffset.size() ? b.ShapeOffset[object + 1] : uint32_t(b.ShapeX.size());
    const size_t n_modes = modes.Freqs.size();
    if (n_modes != b.ModeCount[object] || size_t(last - first) != n_modes * modes.Shapes.size()) return false;
    uint32_t cursor = first;
    for (const auto &at_position : modes.Shapes) {
        for (const vec3 &s : at_position) {
            b.ShapeX[cursor] = Real(s.x);
            b.ShapeY[cursor] = Real(s.y);
            b.ShapeZ[cursor] = Real(s.z);
            ++cursor;
        }
    }
    b.EditedShapes.Mark(first, last);
    return true;
}

template<typename Real> std::optional<uint32_t> SlotOf(const ModalBankColumns<Real> &b, entt::entity who) {
    for (uint32_t o = 0; o < b.Entities.size(); ++o)
        if (b.Entities[o] == who) return o;
    return std::nullopt;
}

// ------------------------------------------------------------------------------------------------------------------
// Events and impacts (ModalAudio.cpp:20-82, 417-425)
// ------------------------------------------------------------------------------------------------------------------
// Single producer (any non-audio thread), single consumer (the render).  Indices grow without wrapping; the slot is
// index mod capacity.  A full ring drops the event and counts it.
template<typename Audio> bool RingPush(Audio &m, const ModalEvent &e) {
    const uint32_t head = m.EventWrite.load(std::memory_order_relaxed);
    const uint32_t pending = head - m.EventRead.load(std::memory_order_acquire);
    if (pending >= Audio::EventCapacity) {
        ++m.EventsDropped;
        return false;
    }
    m.Events[head % Audio::EventCapacity] = e;
    m.EventWrite.store(head + 1, std::memory_order_release);
    return true;
}
template<typename Audio, typename Visit> void RingConsume(Audio &m, Visit &&visit) {
    const uint32_t head = m.EventWrite.load(std::memory_order_acquire);
    uint32_t tail = m.EventRead.load(std::memory_order_relaxed);
    while (tail != head) visit(m.Events[tail++ % Audio::EventCapacity]);
    m.EventRead.store(tail, std::memory_order_release);
}

template<typename Real> void DropImpact(ModalBankColumns<Real> &b, size_t i) { // unordered removal: the last impact takes the hole
    if (i + 1 != b.Impacts.size()) b.Impacts[i] = b.Impacts.back();
    b.Impacts.pop_back();
}

// A force pulse is a raised cosine stepped by a rotating phasor: PulseStep of a turn per sample, ceil(1 / PulseStep)
// samples long (ModalAudio.cpp:28-51).  cos/sin in Real as the oracle, float in the reference.
template<typename Real> void StartImpact(ModalBankColumns<Real> &b, const ModalEvent &e, uint32_t cap) {
    if (b.Impacts.size() >= cap) return;
    const Real step = Real(e.PulseStep);
    const Real turn = 2 * std::numbers::pi_v<Real> * step;
    typename ModalBankColumns<Real>::ActiveImpact im{};
    im.Object = e.Object;
    im.ExPos = e.ExPos;
    im.SamplesLeft = uint32_t(std::ceil(Real(1) / step));
    im.Jx = Real(e.Jx);
    im.Jy = Real(e.Jy);
    im.Jz = Real(e.Jz);
    im.PhaseRe = 1;
    im.PhaseIm = 0;
    im.RotRe = std::cos(turn);
    im.RotIm = std::sin(turn);
    im.Gamma = Real(e.PulseGamma);
    im.AccelAmp = Real(e.AccelAmp);
    im.ClickB0 = Real(e.ClickB0);
    im.ClickA1 = Real(e.ClickA1);
    im.ClickA2 = Real(e.ClickA2);
    b.Impacts.push_back(im);
    b.Ringing[e.Object] = 1;
}

// Return an object to rest: cleared state, full audible set, no impacts in flight (ModalAudio.cpp:53-64).
template<typename Audio, typename Real> void Quiet(Audio &m, ModalBankColumns<Real> &b, uint32_t o, bool device_already_cleared) {
    const uint32_t first = b.ModeOffset[o], n = b.ModeCount[o];
    std::fill_n(b.StateRe.begin() + first, n, Real(0));
    std::fill_n(b.StateIm.begin() + first, n, Real(0));
    if (!device_already_cleared && m.Dev->Bank && m.Dev->Source == &b) mh_bank_zero_state(m.Dev->Bank, first, n);
    b.LiveModeCount[o] = b.TunedModeCount[o];
    b.Ringing[o] = 0;
    for (size_t i = b.Impacts.size(); i-- > 0;)
        if (b.Impacts[i].Object == o) DropImpact(b, i);
}

// ------------------------------------------------------------------------------------------------------------------
// Deal of ringing objects to renderers (ModalAudio.cpp:430-461): longest processing time first onto the least loaded
// renderer, ties to the lower renderer index; every renderer then walks its share in ascending object order.  With a
// single renderer the objects stay in bank order.  A min-heap on (load, renderer) picks the same renderer a linear
// "first minimum" scan would.
// ------------------------------------------------------------------------------------------------------------------
template<typename Real> void Deal(ModalDeviceMirror &d, const ModalBankColumns<Real> &b, uint32_t renderers) {
    d.Renderers.resize(renderers);
    for (auto &share : d.Renderers) share.clear();
    struct Job {
        uint64_t Cost;
        uint32_t Object;
    };
    std::vector<Job> jobs;
    for (uint32_t o = 0; o < b.Entities.size(); ++o)
        if (b.Ringing[o]) jobs.push_back({uint64_t(d.ImpactsOn[o] ? b.TunedModeCount[o] : b.LiveModeCount[o]), o}); // x (1 + voices), voices = 0 without the surface model
    if (renderers == 1) {
        for (const Job &j : jobs) d.Renderers[0].push_back(j.Object);
        return;
    }
    std::stable_sort(jobs.begin(), jobs.end(), [](const Job &a, const Job &c) { return a.Cost > c.Cost; }); // equal costs keep ascending object order
    using Slot = std::pair<uint64_t, uint32_t>; // (load, renderer)
    std::priority_queue<Slot, std::vector<Slot>, std::greater<Slot>> least;
    for (uint32_t r = 0; r < renderers; ++r) least.emplace(0, r);
    for (const Job &j : jobs) {
        auto [load, r] = least.top();
        least.pop();
        d.Renderers[r].push_back(j.Object);
        least.emplace(load + j.Cost, r);
    }
    for (auto &share : d.Renderers) std::sort(share.begin(), share.end());
}

// ------------------------------------------------------------------------------------------------------------------
// Device mirror upkeep
// ------------------------------------------------------------------------------------------------------------------
[[noreturn]] void Fail(const ModalDeviceMirror &d) { throw std::runtime_error(std::string("modalhip: ") + mh_last_error(d.Context)); }

template<typename Audio> void NeedContext(Audio &m) {
    if (!m.Dev->Context && mh_context_create(m.Device, &m.Dev->Context) != MH_OK)
        throw std::runtime_error("modalhip: no MI355X context for the modal bank (there is no CPU fallback)");
}

// Float shape columns for the C ABI (shapes travel as float in both precisions: they come from float ModalModes).
template<typename Real> struct ShapeView {
    std::vector<float> X, Y, Z;
    const float *x, *y, *z;
    ShapeView(const ModalBankColumns<Real> &b, uint32_t first, uint32_t n) {
        if constexpr (std::is_same_v<Real, float>) {
            x = b.ShapeX.data() + first, y = b.ShapeY.data() + first, z = b.ShapeZ.data() + first;
        } else {
            X.assign(b.ShapeX.begin() + first, b.ShapeX.begin() + first + n);
            Y.assign(b.ShapeY.begin() + first, b.ShapeY.begin() + first + n);
            Z.assign(b.ShapeZ.begin() + first, b.ShapeZ.begin() + first + n);
            x = X.data(), y = Y.data(), z = Z.data();
        }
    }
};

template<typename Real> void PushCoefficients(ModalDeviceMirror &d, const ModalBankColumns<Real> &b, uint32_t first, uint32_t n) {
    if (n == 0) return;
    if (mh_bank_set_coefficients(d.Bank, first, n, b.CoeffRe.data() + first, b.CoeffIm.data() + first, b.RadiationGain.data() + first, b.OutPhaseIm.data() + first,
                                 b.OutPhaseRe.data() + first) != MH_OK)
        Fail(d);
}

// Layout + shapes + coefficients into HBM; the device state starts from zero like a freshly built host bank.
template<typename Audio, typename Real> void Mirror(Audio &m, ModalBankColumns<Real> &b) {
    auto &d = *m.Dev;
    NeedContext(m);
    mh_bank_destroy(d.Bank);
    d.Bank = nullptr;
    const uint32_t n_obj = uint32_t(b.Entities.size()), n_modes = uint32_t(b.CoeffRe.size()), n_shapes = uint32_t(b.ShapeX.size());
    const ShapeView<Real> shapes(b, 0, n_shapes);
    if (mh_bank_create(d.Context, std::is_same_v<Real, double> ? 1 : 0, n_obj, n_modes, n_shapes, b.ModeOffset.data(), b.ModeCount.data(), b.ShapeOffset.data(), shapes.x,
                       shapes.y, shapes.z, &d.Bank) != MH_OK)
        Fail(d);
    PushCoefficients(d, b, 0, n_modes);
    uint32_t lo, hi; // everything is on the device now: forget edits recorded while the bank was being built
    b.EditedModes.Take(lo, hi);
    b.EditedShapes.Take(lo, hi);
    d.Source = &b;
}

template<typename Real> void PushEdits(ModalDeviceMirror &d, ModalBankColumns<Real> &b) {
    uint32_t lo, hi;
    if (b.EditedModes.Take(lo, hi)) {
        hi = std::min(hi, uint32_t(b.CoeffRe.size()));
        if (lo < hi) PushCoefficients(d, b, lo, hi - lo);
    }
    if (b.EditedShapes.Take(lo, hi)) {
        hi = std::min(hi, uint32_t(b.ShapeX.size()));
        if (lo < hi) {
            const ShapeView<Real> shapes(b, lo, hi - lo);
            if (mh_bank_set_shapes(d.Bank, lo, hi - lo, shapes.x, shapes.y, shapes.z) != MH_OK) Fail(d);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Publication (ModalAudio.cpp:277-289): the new bank becomes visible with one pointer store; a render that started
// on the old one (odd sequence number) is waited out before the old bank is freed.
// ------------------------------------------------------------------------------------------------------------------
template<typename Audio, typename Bank> void Publish(Audio &m, Bank &next) {
    std::unique_ptr<Bank> retired = std::move(m.Live);
    m.Live = std::make_unique<Bank>(std::move(next));
    m.FlushEvents.store(true, std::memory_order_relaxed); // queued events address the old slot layout
    m.Published.store(m.Live.get(), std::memory_order_seq_cst);
    const uint64_t seen = m.ReaderSeq.load(std::memory_order_seq_cst);
    if (seen % 2 == 1)
        while (m.ReaderSeq.load(std::memory_order_seq_cst) == seen) std::this_thread::yield();
    m.ActiveVoices.store(0, std::memory_order_relaxed); // SurfaceInstallBank: nothing to release without the surface model
    Mirror(m, *m.Live);
}

// Marks a render in progress (odd) for its lifetime, also when the block ends in an exception -- a stuck odd value
// would make the next Publish wait forever.
struct ReaderScope {
    std::atomic<uint64_t> &Seq;
    uint64_t Entered;
    explicit ReaderScope(std::atomic<uint64_t> &seq) : Seq(seq), Entered(seq.load(std::memory_order_relaxed)) { Seq.store(Entered + 1, std::memory_order_seq_cst); }
    ~ReaderScope() { Seq.store(Entered + 2, std::memory_order_release); }
};

// ------------------------------------------------------------------------------------------------------------------
// One block (ModalAudio.cpp:486-590).  Host: events, deal, bookkeeping.  Device: everything per sample.
// ------------------------------------------------------------------------------------------------------------------
template<typename Audio, typename Real> void RenderBlock(Audio &m, Real *out, uint32_t frames) {
    if (frames == 0) return;
    const auto started = std::chrono::steady_clock::now();
    const ReaderScope reading(m.ReaderSeq);
    auto &b = *m.Published.load(std::memory_order_seq_cst);
    auto &d = *m.Dev;
    if (d.Source != &b || !d.Bank) Mirror(m, b); // a bank built in place and never installed is mirrored on first use

    if (m.FlushEvents.exchange(false, std::memory_order_relaxed)) m.EventRead.store(m.EventWrite.load(std::memory_order_relaxed), std::memory_order_relaxed);
    const uint32_t impact_cap = m.MaxImpacts.load(std::memory_order_relaxed);
    RingConsume(m, [&](const ModalEvent &e) {
        if (e.Object >= b.Entities.size()) return; // addressed to a slot this bank does not have
        // (an excitation position the object does not have: the reference reads past the object's shapes there -- undefined, NaN in the oracle's run
        // of tools/probe/r06_odd_bank_probe.py; on the device that read may leave the shape buffer altogether.  Such an impact is dropped.)
        if (e.Kind == ModalEventKind::Impact && b.ModeCount[e.Object] > 0) {
            const uint32_t first = b.ShapeOffset[e.Object];
            const uint32_t last = size_t(e.Object) + 1 < b.ShapeOffset.size() ? b.ShapeOffset[e.Object + 1] : uint32_t(b.ShapeX.size());
            if (uint64_t(e.ExPos) * b.ModeCount[e.Object] + b.ModeCount[e.Object] > uint64_t(last - first)) return;
        }
        if (e.Kind == ModalEventKind::Silence) Quiet(m, b, e.Object, false);
        else if (e.Kind == ModalEventKind::Impact && e.PulseStep > 0) StartImpact(b, e, impact_cap);
    });
    PushEdits(d, b);

    // flatten the deal and the impact list for the device
    const uint32_t n_objects = uint32_t(b.Entities.size()), renderers = m.RenderPool.Size();
    d.ImpactsOn.assign(n_objects, 0);
    for (const auto &im : b.Impacts) ++d.ImpactsOn[im.Object];
    Deal(d, b, renderers);
    d.DealOffset.assign(1, 0);
    d.DealObjects.clear();
    d.RenderCount.clear();
    d.Tuned.clear();
    for (const auto &share : d.Renderers) {
        for (const uint32_t o : share) {
            d.DealObjects.push_back(o);
            d.RenderCount.push_back(d.ImpactsOn[o] ? b.TunedModeCount[o] : b.LiveModeCount[o]); // an excited object renders its whole tuned set
            d.Tuned.push_back(b.TunedModeCount[o]);
        }
        d.DealOffset.push_back(uint32_t(d.DealObjects.size()));
    }
    const uint32_t dealt = uint32_t(d.DealObjects.size()), n_impacts = uint32_t(b.Impacts.size());
    d.Impacts.resize(n_impacts);
    for (uint32_t i = 0; i < n_impacts; ++i) {
        const auto &s = b.Impacts[i];
        mh_impact &t = d.Impacts[i];
        t.object = s.Object, t.ex_pos = s.ExPos, t.samples_left = s.SamplesLeft, t.reserved = 0;
        t.jx = s.Jx, t.jy = s.Jy, t.jz = s.Jz;
        t.phase_re = s.PhaseRe, t.phase_im = s.PhaseIm, t.rot_re = s.RotRe, t.rot_im = s.RotIm;
        t.gamma = s.Gamma, t.accel_amp = s.AccelAmp;
        t.click_b0 = s.ClickB0, t.click_a1 = s.ClickA1, t.click_a2 = s.ClickA2, t.click_z1 = s.ClickZ1, t.click_z2 = s.ClickZ2;
    }
    d.Energy.assign(dealt, 0.0);
    d.ModalEnergy.assign(dealt, 0.0);
    d.Live.assign(dealt, 0);
    d.Silenced.assign(dealt, 0);

    // the two gain columns travel as float (they are written through atomic_ref<float> by the reference's callers)
    std::vector<float> narrow_out, narrow_listener;
    const float *out_gain, *listener_gain;
    if constexpr (std::is_same_v<Real, float>) {
        out_gain = b.OutGain.data(), listener_gain = b.ListenerGain.data();
    } else {
        narrow_out.assign(b.OutGain.begin(), b.OutGain.end());
        narrow_listener.assign(b.ListenerGain.begin(), b.ListenerGain.end());
        out_gain = narrow_out.data(), listener_gain = narrow_listener.data();
    }
    if (mh_bank_render(d.Bank, frames, m.ClickGain.load(std::memory_order_relaxed), n_impacts, d.Impacts.data(), renderers, d.DealOffset.data(), d.DealObjects.data(),
                       d.RenderCount.data(), d.Tuned.data(), out_gain, listener_gain, out, d.Energy.data(), d.Live.data(), d.Silenced.data(), d.ModalEnergy.data()) != MH_OK)
        Fail(d);

    // impacts: carry the recurrences' state over to the next block
    for (uint32_t i = 0; i < n_impacts; ++i) {
        auto &s = b.Impacts[i];
        const mh_impact &t = d.Impacts[i];
        s.SamplesLeft = t.samples_left;
        s.PhaseRe = Real(t.phase_re), s.PhaseIm = Real(t.phase_im);
        s.ClickZ1 = Real(t.click_z1), s.ClickZ2 = Real(t.click_z2);
    }
    // objects: an unexcited object whose energy fell below audibility goes back to rest; the others keep ringing with
    // the audible prefix the device found (an excited one keeps its whole tuned set) -- ModalAudio.cpp:141-146
    for (uint32_t i = 0; i < dealt; ++i) {
        const uint32_t o = d.DealObjects[i];
        if (d.Silenced[i]) {
            Quiet(m, b, o, true);
        } else {
            b.LiveModeCount[o] = d.ImpactsOn[o] ? b.TunedModeCount[o] : d.Live[i];
            b.Ringing[o] = 1;
        }
    }
    // an impact retires once its pulse is spent and its click filter has rung out (:557-561)
    for (size_t i = b.Impacts.size(); i-- > 0;) {
        const auto &im = b.Impacts[i];
        if (im.SamplesLeft == 0 && std::abs(im.ClickZ1) + std::abs(im.ClickZ2) < Real(1e-12f)) DropImpact(b, i);
    }
    // modal-energy diagnostic (:564-577): per-object terms from the device, added in bank order
    std::vector<double> by_object(n_objects, 0.0);
    for (uint32_t i = 0; i < dealt; ++i) by_object[d.DealObjects[i]] = d.ModalEnergy[i];
    double total = 0;
    for (const double e : by_object) total += e; // objects that were not dealt add an exact zero
    m.ModalEnergy.store(total, std::memory_order_relaxed);
    if (total > m.PeakModalEnergy.load(std::memory_order_relaxed)) m.PeakModalEnergy.store(total, std::memory_order_relaxed);
    m.ActiveImpacts.store(uint32_t(b.Impacts.size()), std::memory_order_relaxed);

    const float seconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - started).count();
    const float share = b.SampleRate > 0 ? seconds * float(b.SampleRate) / float(frames) : 0.f; // > 1 would underrun a live device
    m.RenderSeconds.store(seconds, std::memory_order_relaxed);
    m.RenderShare.store(share, std::memory_order_relaxed);
    if (share > m.PeakRenderShare.load(std::memory_order_relaxed)) m.PeakRenderShare.store(share, std::memory_order_relaxed);
}

template<typename Audio> void PullState(Audio &m) {
    auto &d = *m.Dev;
    auto &b = *m.Published.load(std::memory_order_seq_cst);
    using Real = typename Audio::Scalar;
    if (!d.Bank || d.Source != &b || b.StateRe.empty()) return;
    std::vector<double> re(b.StateRe.size()), im(b.StateIm.size());
    if (mh_bank_read_state(d.Bank, 0, uint32_t(re.size()), re.data(), im.data()) != MH_OK) return;
    std::transform(re.begin(), re.end(), b.StateRe.begin(), [](double v) { return Real(v); });
    std::transform(im.begin(), im.end(), b.StateIm.begin(), [](double v) { return Real(v); });
}
} // namespace

// ------------------------------------------------------------------------------------------------------------------
// The reference's entry points, once per precision
// ------------------------------------------------------------------------------------------------------------------
uint32_t AddModalObject(ModalBank &b, entt::entity e, const ModalModes &modes) { return AppendObject<float>(b, e, modes); }
uint32_t AddModalObject(ModalBank64 &b, entt::entity e, const ModalModes &modes) { return AppendObject<double>(b, e, modes); }
void InstallModalBank(ModalAudio &m, ModalBank &next) { Publish(m, next); }
void InstallModalBank(ModalAudio64 &m, ModalBank64 &next) { Publish(m, next); }
void TuneModalObject(ModalBank &b, uint32_t object, std::span<const float> freqs, std::span<const float> t60s, float radius_scale) { Tune<float>(b, object, freqs, t60s, radius_scale); }
void TuneModalObject(ModalBank64 &b, uint32_t object, std::span<const float> freqs, std::span<const float> t60s, float radius_scale) { Tune<double>(b, object, freqs, t60s, radius_scale); }
bool SetModalObjectShapes(ModalBank &b, uint32_t object, const ModalModes &modes) { return OverwriteShapes<float>(b, object, modes); }
bool SetModalObjectShapes(ModalBank64 &b, uint32_t object, const ModalModes &modes) { return OverwriteShapes<double>(b, object, modes); }
std::optional<uint32_t> FindModalObject(const ModalBank &b, entt::entity e) { return SlotOf<float>(b, e); }
std::optional<uint32_t> FindModalObject(const ModalBank64 &b, entt::entity e) { return SlotOf<double>(b, e); }
void EnqueueModalEvent(ModalAudio &m, const ModalEvent &e) { RingPush(m, e); }
void EnqueueModalEvent(ModalAudio64 &m, const ModalEvent &e) { RingPush(m, e); }
void RenderModal(ModalAudio &m, float *out, uint32_t frame_count) { RenderBlock<ModalAudio, float>(m, out, frame_count); }
void RenderModal(ModalAudio64 &m, double *out, uint32_t frame_count) { RenderBlock<ModalAudio64, double>(m, out, frame_count); }
mh_context *ModalDeviceContext(ModalAudio &m) { return NeedContext(m), m.Dev->Context; }
mh_context *ModalDeviceContext(ModalAudio64 &m) { return NeedContext(m), m.Dev->Context; }
void SyncModalState(ModalAudio &m) { PullState(m); }
void SyncModalState(ModalAudio64 &m) { PullState(m); }
