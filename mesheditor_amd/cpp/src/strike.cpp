// Strike translation over the mirrored bank and contact model (see modal/strike.hpp).
#include "modal/strike.hpp"

#include <algorithm>
#include <cmath>
#include <numbers>

vec3 UnitOrZero(vec3 v) {
    const float len = length(v);
    return len > 0 ? v / len : vec3{0.f};
}
uint32_t NearestSamplePoint(const std::vector<vec3> &positions, vec3 local_point) {
    uint32_t best = 0;
    float best_d = INFINITY;
    for (uint32_t i = 0; i < positions.size(); ++i) {
        const vec3 d = positions[i] - local_point;
        const float d2 = dot(d, d);
        if (d2 < best_d) { // the first minimum wins, as std::ranges::min_element
            best_d = d2;
            best = i;
        }
    }
    return best;
}
float PeakModalDrive(const ModalModes &modes, uint32_t p, vec3 j) {
    if (p >= modes.Shapes.size()) return 0;
    float peak = 0;
    for (const auto &shape : modes.Shapes[p]) peak = std::max(peak, std::abs(dot(shape, j)));
    return peak;
}
double VolumeEquivalentRadius(double volume) { return std::cbrt(3.0 * volume / (4.0 * std::numbers::pi)); }
double SphereEquivalentCurvature(double density, double inv_mass) { return std::cbrt(4.0 * std::numbers::pi / 3.0 * density * inv_mass); }
double DisplacedVolume(double enclosed_volume, double mass, const AcousticMaterialProperties *props) {
    if (enclosed_volume > 0) return enclosed_volume;
    return props && props->Density > 0 && mass > 0 ? mass / props->Density : 0.0;
}

ModalEvent MakeStrikeEvent(const ModalBank &bank, uint32_t slot, uint32_t excitable_index, vec3 dir, float force, float contact_speed, const StrikeContext &sc,
                           const std::optional<PhysicsStrike> &physics, const Striker &striker) {
    double tau = 1e-4; // seconds: the default contact when the material or the dynamics are missing
    float click_amp = 0;
    ClickFilter click{};
    if (sc.Dynamics && sc.Material) {
        const Impactor imp = physics ? physics->Impactor : StrikerImpactor(striker);
        tau = EstimateContactTime(*sc.Dynamics, physics ? physics->ResultantIndex : excitable_index, dir, contact_speed, sc.Elastic, sc.Curvature,
                                  physics ? physics->NominalArea : 0.f, imp, sc.ScaleRatio, sc.Roughness);
        // The click is the recoil radiator driven by this strike's force pulse; without a volume to displace, the radius of
        // the disc holding the body's sample-surface area sets the corner.
        const double volume = DisplacedVolume(sc.EnclosedVolume, sc.Dynamics->Mass, &sc.Material->Properties);
        const double radius = volume > 0 ? VolumeEquivalentRadius(volume) : double(bank.RadiantRadius[slot] * sc.ScaleRatio);
        click = RecoilClickFilter(radius, volume, sc.Dynamics->Mass, bank.SampleRate);
        // A physics force is the true contact impulse; a manual one is nominal, from the reduced mass and the approach speed.
        const double impulse = physics ? double(force) : ReducedContactMass(*sc.Dynamics, excitable_index, dir, imp) * std::abs(double(contact_speed));
        click_amp = float(impulse * bank.SampleRate);
    }
    const auto step = float(1.0 / (tau * bank.SampleRate));
    return {.Kind = ModalEventKind::Impact, .Object = slot, .ExPos = excitable_index, .Jx = dir.x * force, .Jy = dir.y * force, .Jz = dir.z * force, .PulseStep = step,
            .PulseGamma = 2 * step, .AccelAmp = click_amp, .ClickB0 = click.B0, .ClickA1 = click.A1, .ClickA2 = click.A2};
}

bool TriggerModalStrike(ModalAudio &m, entt::entity e, const ModalModes &modes, uint32_t excitable_index, vec3 dir, float force, float contact_speed,
                        const StrikeContext &sc, const std::optional<PhysicsStrike> &physics, const Striker &striker) {
    const auto &bank = LiveBank(m);
    const auto slot = FindModalObject(bank, e);
    if (!slot) return false;
    if (excitable_index >= std::min(modes.Vertices.size(), modes.Positions.size())) return false;
    const vec3 unit = physics ? normalize(physics->Direction) : dir;
    EnqueueModalEvent(m, MakeStrikeEvent(bank, *slot, excitable_index, unit, force, contact_speed, sc, physics, striker));
    return true;
}
