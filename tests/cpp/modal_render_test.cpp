// Properties of the block render, whatever the force pulse looks like (the reference's tests/ModalRenderTest.cpp),
// run against the device bank through the mirrored API.
#include "harness.hpp"

#include <audio/ModalAudio.h>

#include <algorithm>
#include <array>
#include <numbers>
#include <span>

namespace {
constexpr float SampleRate = 48'000.f;
constexpr uint32_t BlockSize = 512, SamplePoints = 4;

ModalModes MakeModes(uint32_t mode_count, float longest_t60, float shape_scale = 1.f) {
    ModalModes modes;
    for (uint32_t p = 0; p < SamplePoints; ++p) modes.Positions.push_back({float(p) * 0.01f, 0.f, p % 2 ? 0.02f : 0.f});
    for (uint32_t p = 0; p + 2 < SamplePoints; ++p) modes.Indices.insert(modes.Indices.end(), {p, p + 1, p + 2});
    modes.Shapes.assign(SamplePoints, {});
    for (uint32_t k = 0; k < mode_count; ++k) {
        modes.Freqs.push_back(40.f * float(k + 1) * 1.031f);
        modes.T60s.push_back(longest_t60 / float(k + 1));
        for (uint32_t p = 0; p < SamplePoints; ++p) {
            const float a = float(k + 1) * 0.37f + float(p);
            modes.Shapes[p].push_back(vec3{std::sin(a), std::cos(a * 1.7f), std::sin(a * 2.3f)} * (0.01f * shape_scale));
        }
    }
    return modes;
}

ModalEvent ImpactEvent(uint32_t object, float impulse, uint32_t ex_pos = 0, float pulse_step = 1.f / 300.f) {
    return {.Kind = ModalEventKind::Impact, .Object = object, .ExPos = ex_pos, .Jx = impulse, .Jy = 0.5f * impulse, .Jz = 0.f, .PulseStep = pulse_step,
            .PulseGamma = 20.f, .AccelAmp = 0.f};
}

struct Scene {
    ModalAudio Audio;
    std::vector<uint32_t> Objects;
    Scene(uint32_t object_count, uint32_t mode_count, float longest_t60, uint32_t renderers, float sample_rate = SampleRate) {
        const auto modes = MakeModes(mode_count, longest_t60);
        Audio.RenderPool.SetSize(renderers);
        ModalBank next;
        next.SampleRate = sample_rate;
        for (uint32_t o = 0; o < object_count; ++o) {
            Objects.push_back(AddModalObject(next, entt::entity{o}, modes));
            TuneModalObject(next, Objects.back(), modes.Freqs, modes.T60s);
            next.OutGain[Objects.back()] = 1.f; // callers write the per-object columns directly
            next.RigidInvMass[Objects.back()] = 0.f;
        }
        InstallModalBank(Audio, next);
        std::vector<float> discard(BlockSize, 0.f);
        RenderModal(Audio, discard.data(), BlockSize); // clears the events addressed to the previous layout
    }
    std::vector<float> Render(uint32_t blocks, uint32_t frames) {
        std::vector<float> signal(size_t(blocks) * frames, 0.f);
        for (uint32_t b = 0; b < blocks; ++b) RenderModal(Audio, signal.data() + size_t(b) * frames, frames);
        return signal;
    }
};

float Peak(std::span<const float> s) {
    float p = 0;
    for (const float v : s) p = std::max(p, std::abs(v));
    return p;
}
float MaxDifference(std::span<const float> a, std::span<const float> b) {
    float w = 0;
    for (size_t i = 0; i < a.size(); ++i) w = std::max(w, std::abs(a[i] - b[i]));
    return w;
}
} // namespace

CASE(excitations_superpose_linearly) {
    const auto render = [](std::span<const ModalEvent> events) {
        Scene scene{1, 64, 0.2f, 1};
        for (auto e : events) {
            e.Object = scene.Objects.front();
            EnqueueModalEvent(scene.Audio, e);
        }
        return scene.Render(8, BlockSize);
    };
    const std::array both{ImpactEvent(0, 1.f, 0, 1.f / 300.f), ImpactEvent(0, -0.4f, 1, 1.f / 90.f)};
    const auto a = render(std::span{both}.first(1)), b = render(std::span{both}.last(1)), together = render(both);
    std::vector<float> sum(a.size());
    for (size_t i = 0; i < a.size(); ++i) sum[i] = a[i] + b[i];
    EXPECT(Peak(a) > 0.f);
    EXPECT(Peak(b) > 0.f);
    EXPECT_NOTE(MaxDifference(together, sum) <= Peak(together) * 1e-5f, std::to_string(MaxDifference(together, sum)));
}

CASE(a_strike_does_not_depend_on_the_renderer_count) {
    const auto render = [](uint32_t renderers) {
        Scene scene{16, 64, 0.2f, renderers};
        for (const auto o : scene.Objects) EnqueueModalEvent(scene.Audio, ImpactEvent(o, 1.f));
        return scene.Render(32, BlockSize);
    };
    const auto single = render(1), split = render(4);
    EXPECT(Peak(single) > 0.f);
    EXPECT(MaxDifference(single, split) < Peak(single) * 1e-5f);
}

CASE(the_click_does_not_depend_on_the_output_sample_rate) {
    constexpr double Tau{5e-4};
    constexpr double Radius{0.05}, Volume{4.0 / 3.0 * std::numbers::pi * Radius * Radius * Radius}, Mass{1.0}, Impulse{0.5};
    const auto peak_at = [](float rate) {
        Scene scene{1, 64, 0.2f, 1, rate};
        const auto step = float(1.0 / (Tau * double(rate)));
        const auto click = RecoilClickFilter(Radius, Volume, Mass, rate);
        EnqueueModalEvent(scene.Audio, {.Kind = ModalEventKind::Impact, .Object = scene.Objects.front(), .ExPos = 0, .Jx = 0.f, .Jy = 0.f, .Jz = 0.f,
                                        .PulseStep = step, .PulseGamma = 2 * step, .AccelAmp = float(Impulse) * rate, .ClickB0 = click.B0, .ClickA1 = click.A1,
                                        .ClickA2 = click.A2});
        const auto blocks = uint32_t(std::ceil(4 * Tau * double(rate) / BlockSize));
        return Peak(scene.Render(blocks, BlockSize));
    };
    const auto slow = peak_at(SampleRate), fast = peak_at(2 * SampleRate);
    EXPECT(slow > 0.f);
    EXPECT_NOTE(check::near(double(fast) / double(slow), 1.0, 2e-2), std::to_string(slow) + " " + std::to_string(fast));
}

CASE(render_adds_into_the_output_and_events_find_objects) {
    Scene scene{2, 16, 0.2f, 1};
    EXPECT(FindModalObject(*scene.Audio.Live, entt::entity{1}).value_or(99) == scene.Objects[1]);
    EXPECT(!FindModalObject(*scene.Audio.Live, entt::entity{7}).has_value());
    EnqueueModalEvent(scene.Audio, ImpactEvent(scene.Objects[1], 1.f));
    std::vector<float> out(BlockSize, 0.25f), ref(BlockSize, 0.f);
    RenderModal(scene.Audio, out.data(), BlockSize);
    Scene twin{2, 16, 0.2f, 1};
    EnqueueModalEvent(twin.Audio, ImpactEvent(twin.Objects[1], 1.f));
    RenderModal(twin.Audio, ref.data(), BlockSize);
    float worst = 0;
    for (uint32_t s = 0; s < BlockSize; ++s) worst = std::max(worst, std::abs(out[s] - (0.25f + ref[s])));
    EXPECT(worst <= 1e-6f);
    EXPECT(scene.Audio.ActiveImpacts.load() == 0u); // a 300-sample pulse ends inside the block and its impact retires
}

int main() { return check::run_all(); }
