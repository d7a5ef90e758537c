// Properties of the block render, whatever the force pulse looks like (the properties the reference's
// tests/ModalRenderTest.cpp states: superposition, renderer-count independence, sample-rate independent click), run
// against the device bank through the mirrored API -- in both precisions -- plus the surface-contact hooks.
#include "harness.hpp"

#include <audio/ModalAudio.h>
#include <audio/SurfaceContact.h>

#include <algorithm>
#include <array>
#include <numbers>
#include <numeric>
#include <span>

namespace {
constexpr float kRate = 48'000.f;
constexpr uint32_t kBlock = 512, kPoints = 4;

// The synthetic body every case strikes: a harmonic-ish ladder of modes (40 Hz x 1.031 x ordinal), decay times
// falling as 1/ordinal from `slowest`, smooth trigonometric shapes of amplitude 0.01 on a zig-zag strip of four sample
// points (alternating depth, so consecutive triples span triangles with area).  Same numbers as the reference harness.
ModalModes LadderModes(uint32_t n_modes, float slowest) {
    ModalModes body;
    body.Freqs.resize(n_modes);
    body.T60s.resize(n_modes);
    for (uint32_t k = 0; k < n_modes; ++k) {
        const float ordinal = float(k + 1);
        body.Freqs[k] = 40.f * ordinal * 1.031f;
        body.T60s[k] = slowest / ordinal;
    }
    for (uint32_t p = 0; p < kPoints; ++p) {
        body.Positions.push_back({float(p) * 0.01f, 0.f, (p & 1u) ? 0.02f : 0.f});
        if (p >= 2) body.Indices.insert(body.Indices.end(), {p - 2, p - 1, p});
        auto &row = body.Shapes.emplace_back(n_modes);
        for (uint32_t k = 0; k < n_modes; ++k) {
            const float phase = float(k + 1) * 0.37f + float(p);
            row[k] = vec3{std::sin(phase), std::cos(phase * 1.7f), std::sin(phase * 2.3f)} * 0.01f;
        }
    }
    return body;
}

// A strike along (1, 1/2, 0) with a raised-cosine pulse of 1 / step samples and no click.
ModalEvent Strike(uint32_t object, float impulse, uint32_t at_point = 0, float step = 1.f / 300.f) {
    ModalEvent e;
    e.Kind = ModalEventKind::Impact;
    e.Object = object;
    e.ExPos = at_point;
    e.Jx = impulse;
    e.Jy = 0.5f * impulse;
    e.PulseStep = step;
    e.PulseGamma = 20.f;
    return e;
}

// N identical bodies in one published bank, in either precision.
template<typename Audio, typename Bank, typename Sample> struct RigT {
    Audio Engine;
    std::vector<uint32_t> Slots;
    RigT(uint32_t bodies, uint32_t n_modes, float slowest, uint32_t renderers, float rate = kRate) {
        const ModalModes body = LadderModes(n_modes, slowest);
        Engine.RenderPool.SetSize(renderers);
        Bank building;
        building.SampleRate = rate;
        for (uint32_t i = 0; i < bodies; ++i) {
            const uint32_t slot = AddModalObject(building, entt::entity{i}, body);
            TuneModalObject(building, slot, body.Freqs, body.T60s);
            building.OutGain[slot] = 1; // callers write the per-object columns directly
            Slots.push_back(slot);
        }
        InstallModalBank(Engine, building);
        Run(1, kBlock); // the first block after an install discards events addressed to the previous layout
    }
    std::vector<Sample> Run(uint32_t blocks, uint32_t frames) {
        std::vector<Sample> signal(size_t(blocks) * frames, Sample(0));
        for (uint32_t i = 0; i < blocks; ++i) RenderModal(Engine, signal.data() + size_t(i) * frames, frames);
        return signal;
    }
    void StrikeAll(float impulse) {
        for (const uint32_t slot : Slots) EnqueueModalEvent(Engine, Strike(slot, impulse));
    }
};
using Rig = RigT<ModalAudio, ModalBank, float>;
using Rig64 = RigT<ModalAudio64, ModalBank64, double>;

template<typename Sample> double Loudest(const std::vector<Sample> &s) {
    return std::accumulate(s.begin(), s.end(), 0.0, [](double m, Sample v) { return std::max(m, std::abs(double(v))); });
}
template<typename Sample> double Gap(const std::vector<Sample> &a, const std::vector<Sample> &b) {
    return std::inner_product(a.begin(), a.end(), b.begin(), 0.0, [](double m, double d) { return std::max(m, d); }, [](Sample x, Sample y) { return std::abs(double(x) - double(y)); });
}
} // namespace

CASE(excitations_superpose_linearly) {
    const std::array<ModalEvent, 2> strikes{Strike(0, 1.f, 0, 1.f / 300.f), Strike(0, -0.4f, 1, 1.f / 90.f)};
    const auto heard = [&](std::initializer_list<int> which) {
        Rig rig{1, 64, 0.2f, 1};
        for (const int i : which) {
            ModalEvent e = strikes[i];
            e.Object = rig.Slots.front();
            EnqueueModalEvent(rig.Engine, e);
        }
        return rig.Run(8, kBlock);
    };
    const auto first = heard({0}), second = heard({1}), both = heard({0, 1});
    std::vector<float> sum(first.size());
    std::transform(first.begin(), first.end(), second.begin(), sum.begin(), std::plus<float>{});
    EXPECT(Loudest(first) > 0 && Loudest(second) > 0);
    EXPECT_NOTE(Gap(both, sum) <= Loudest(both) * 1e-5, std::to_string(Gap(both, sum)));
}

CASE(a_strike_does_not_depend_on_the_renderer_count) {
    const auto heard = [](uint32_t renderers) {
        Rig rig{16, 64, 0.2f, renderers};
        rig.StrikeAll(1.f);
        return rig.Run(32, kBlock);
    };
    const auto one = heard(1), four = heard(4);
    EXPECT(Loudest(one) > 0);
    EXPECT(Gap(one, four) < Loudest(one) * 1e-5);
}

CASE(the_click_does_not_depend_on_the_output_sample_rate) {
    // a 0.5 ms contact on a 5 cm, 1 kg sphere: the click's peak pressure is a property of the strike, not of the rate
    const double contact = 5e-4, radius = 0.05, mass = 1.0, impulse = 0.5;
    const double volume = 4.0 / 3.0 * std::numbers::pi * radius * radius * radius;
    const auto click_peak = [&](float rate) {
        Rig rig{1, 64, 0.2f, 1, rate};
        const ClickFilter filter = RecoilClickFilter(radius, volume, mass, rate);
        ModalEvent e = Strike(rig.Slots.front(), 0.f);
        e.Jy = 0.f;
        e.PulseStep = float(1.0 / (contact * double(rate)));
        e.PulseGamma = 2 * e.PulseStep;
        e.AccelAmp = float(impulse) * rate;
        e.ClickB0 = filter.B0, e.ClickA1 = filter.A1, e.ClickA2 = filter.A2;
        EnqueueModalEvent(rig.Engine, e);
        return Loudest(rig.Run(uint32_t(std::ceil(4 * contact * double(rate) / kBlock)), kBlock));
    };
    const double at_48k = click_peak(kRate), at_96k = click_peak(2 * kRate);
    EXPECT(at_48k > 0);
    EXPECT_NOTE(check::near(at_96k / at_48k, 1.0, 2e-2), std::to_string(at_48k) + " " + std::to_string(at_96k));
}

CASE(render_adds_into_the_output_and_events_find_objects) {
    Rig rig{2, 16, 0.2f, 1}, twin{2, 16, 0.2f, 1};
    EXPECT(FindModalObject(*rig.Engine.Live, entt::entity{1}).value_or(99) == rig.Slots[1]);
    EXPECT(!FindModalObject(*rig.Engine.Live, entt::entity{7}).has_value());
    EnqueueModalEvent(rig.Engine, Strike(rig.Slots[1], 1.f));
    EnqueueModalEvent(twin.Engine, Strike(twin.Slots[1], 1.f));
    std::vector<float> onto(kBlock, 0.25f), alone(kBlock, 0.f);
    RenderModal(rig.Engine, onto.data(), kBlock);
    RenderModal(twin.Engine, alone.data(), kBlock);
    for (float &v : alone) v += 0.25f;
    EXPECT(Gap(onto, alone) <= 1e-6);
    EXPECT(rig.Engine.ActiveImpacts.load() == 0u); // a 300-sample pulse ends inside the block and its impact retires
}

CASE(the_double_precision_bank_sounds_like_the_single_precision_one) {
    // same scene through ModalBank64 / ModalAudio64: the signals agree to single-precision rounding of the fp32 path,
    // and the fp64 render itself does not depend on the renderer count beyond double rounding
    const auto heard32 = [] {
        Rig rig{8, 48, 0.2f, 2};
        rig.StrikeAll(1.f);
        return rig.Run(6, kBlock);
    }();
    const auto heard64 = [](uint32_t renderers) {
        Rig64 rig{8, 48, 0.2f, renderers};
        rig.StrikeAll(1.f);
        return rig.Run(6, kBlock);
    };
    const auto two = heard64(2), one = heard64(1);
    std::vector<double> widened(heard32.begin(), heard32.end());
    EXPECT(Loudest(two) > 0);
    EXPECT_NOTE(Gap(two, widened) < Loudest(two) * 1e-4, std::to_string(Gap(two, widened) / Loudest(two)));
    EXPECT_NOTE(Gap(two, one) < Loudest(two) * 1e-13, std::to_string(Gap(two, one) / Loudest(two)));
}

CASE(surface_contact_hooks_are_present_and_inert) {
    // the 14 entry points + 2 deleters of the reference's SurfaceContact.h with the model absent
    Rig rig{1, 8, 0.2f, 1};
    ModalBank &bank = LiveBank(rig.Engine);
    EXPECT(MakeSurfaceAudioState() == nullptr);
    SurfaceAudioStateDelete{}(nullptr);
    SurfaceRenderScratchDelete{}(nullptr);
    SurfaceAdoptVoices(rig.Engine, bank, kBlock);
    EXPECT(SurfaceVoiceCount(rig.Engine, 0) == 0u);
    float out[4]{};
    // the opaque types are never defined in this build: hand the hooks a reference to inert storage, they only pass it on
    alignas(16) static unsigned char opaque[64];
    auto &scratch = reinterpret_cast<ModalRenderScratch &>(opaque);
    auto &registry = reinterpret_cast<entt::registry &>(opaque);
    EXPECT(!SurfaceRenderObject(rig.Engine, scratch, bank, 0, std::span<const uint32_t>{}, out, 4)); // falls to the modal-only kernel
    SurfaceSilenceObject(rig.Engine, 0);
    EXPECT(SurfaceActiveVoices(rig.Engine) == 0u);
    SurfaceInstallBank(rig.Engine);
    RegisterSurfaceContactHandlers(registry);
    SurfaceUpdateContacts(registry);
    EXPECT(SurfaceRoughnessOf(registry, entt::entity{3}) == 0.f);
    EXPECT(ContactSurfaceNode(registry, entt::entity{3}, entt::entity{5}) == entt::entity{5});
    DrawContactSurfaceControls(registry, entt::entity{0});
    DrawSurfaceSynthControls(registry, entt::entity{0});
    DrawSurfaceContactDebug(registry);
}

int main() { return check::run_all(); }
