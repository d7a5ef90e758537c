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

namespace {
// How long the bodies stay in contact and what strikes: the Hertz/punch estimate when the object's dynamics and material
// are known, otherwise a short default contact (1e-4 s) that radiates no click.
struct StrikeContact {
    double Seconds{1e-4};
    bool Modelled{false};
    ::Impactor Hitting{};
};
StrikeContact ResolveContact(uint32_t excitable_index, vec3 dir, float contact_speed, const StrikeContext &sc, const std::optional<PhysicsStrike> &physics, const Striker &striker) {
    StrikeContact c;
    if (!sc.Dynamics || !sc.Material) return c;
    c.Modelled = true;
    c.Hitting = physics ? physics->Impactor : StrikerImpactor(striker); // a collision brings its own impactor, a manual hit the mallet
    const uint32_t arm = physics ? physics->ResultantIndex : excitable_index;
    const double patch_limit = physics ? double(physics->NominalArea) : 0.0; // a rounded mallet tip grows its own patch
    c.Seconds = EstimateContactTime(*sc.Dynamics, arm, dir, contact_speed, sc.Elastic, sc.Curvature, patch_limit, c.Hitting, sc.ScaleRatio, sc.Roughness);
    return c;
}

// The recoil click of the strike: the radiator filter of the struck body and the force scale that drives it.
struct ClickDrive {
    ClickFilter Filter{};
    float Amplitude{0.f};
};
ClickDrive ResolveClick(const ModalBank &bank, uint32_t slot, uint32_t excitable_index, vec3 dir, float force, float contact_speed, const StrikeContext &sc,
                        const StrikeContact &contact, bool from_physics) {
    ClickDrive out;
    if (!contact.Modelled) return out;
    const double body_mass = sc.Dynamics->Mass;
    // corner of the radiator: the sphere of the displaced volume, or -- with nothing to displace -- the disc holding the
    // body's sample-surface area at its current size
    const double volume = DisplacedVolume(sc.EnclosedVolume, body_mass, &sc.Material->Properties);
    const double radius = volume > 0 ? VolumeEquivalentRadius(volume) : double(bank.RadiantRadius[slot] * sc.ScaleRatio);
    out.Filter = RecoilClickFilter(radius, volume, body_mass, bank.SampleRate);
    // The force pulse sums to one over its samples, so the filter input is impulse x sample rate (Newtons).  A collision
    // reports its true impulse; a manual hit's is nominal: reduced mass x approach speed.
    const double impulse = from_physics ? double(force) : ReducedContactMass(*sc.Dynamics, excitable_index, dir, contact.Hitting) * std::abs(double(contact_speed));
    out.Amplitude = float(impulse * bank.SampleRate);
    return out;
}
} // namespace

ModalEvent MakeStrikeEvent(const ModalBank &bank, uint32_t slot, uint32_t excitable_index, vec3 dir, float force, float contact_speed, const StrikeContext &sc,
                           const std::optional<PhysicsStrike> &physics, const Striker &striker) {
    const StrikeContact contact = ResolveContact(excitable_index, dir, contact_speed, sc, physics, striker);
    const ClickDrive click = ResolveClick(bank, slot, excitable_index, dir, force, contact_speed, sc, contact, physics.has_value());
    ModalEvent e;
    e.Kind = ModalEventKind::Impact;
    e.Object = slot;
    e.ExPos = excitable_index;
    e.Jx = dir.x * force, e.Jy = dir.y * force, e.Jz = dir.z * force; // the impulse rides in the excitation gains, not in the pulse
    e.PulseStep = float(1.0 / (contact.Seconds * bank.SampleRate)); // pulse phase advance per sample
    e.PulseGamma = 2 * e.PulseStep; // raised cosine of unit sample sum
    e.AccelAmp = click.Amplitude;
    e.ClickB0 = click.Filter.B0, e.ClickA1 = click.Filter.A1, e.ClickA2 = click.Filter.A2;
    return e;
}

bool TriggerModalStrike(ModalAudio &m, entt::entity e, const ModalModes &modes, uint32_t excitable_index, vec3 dir, float force, float contact_speed,
                        const StrikeContext &sc, const std::optional<PhysicsStrike> &physics, const Striker &striker) {
    const ModalBank &bank = LiveBank(m);
    const std::optional<uint32_t> slot = FindModalObject(bank, e);
    const size_t excitable = std::min(modes.Vertices.size(), modes.Positions.size());
    if (!slot || excitable_index >= excitable) return false;
    const vec3 along = physics ? normalize(physics->Direction) : dir;
    EnqueueModalEvent(m, MakeStrikeEvent(bank, *slot, excitable_index, along, force, contact_speed, sc, physics, striker));
    return true;
}

uint32_t StrikeContacts(ModalAudio &audio, std::span<const ContactImpact> impacts, const StrikeScene &scene, const ContactFloors &floors) {
    uint32_t queued = 0;
    for (const ContactImpact &hit : impacts) {
        if (hit.Speed < floors.MinContactSpeed) continue; // a loaded body at rest must not buzz
        const ModalModes *modes = scene.ModesOf ? scene.ModesOf(hit.Entity) : nullptr;
        if (!modes || modes->Positions.empty()) continue;
        const auto to_local = [&](const std::function<vec3(entt::entity, vec3)> &map, vec3 world) { return map ? map(hit.Entity, world) : world; };
        const vec3 at = to_local(scene.LocalPoint, hit.Point), along = to_local(scene.LocalDirection, hit.Direction);
        const uint32_t sample_point = NearestSamplePoint(modes->Positions, at);
        // audibility is decided on what the strike excites, not on the momentum behind it
        if (PeakModalDrive(*modes, sample_point, UnitOrZero(along) * hit.Impulse) < floors.MinContactExcitation) continue;
        // the other body is the impactor: its stiffness, mass and curvature shape the contact time
        PhysicsStrike physics;
        physics.Direction = along;
        physics.Impactor.Material = scene.MaterialOf ? scene.MaterialOf(hit.Other) : materials::acoustic::Ceramic.Properties;
        const std::optional<double> curvature = scene.CurvatureAt ? scene.CurvatureAt(hit.Other, hit.Point) : std::nullopt;
        physics.Impactor.Curvature = curvature.value_or(SphereEquivalentCurvature(physics.Impactor.Material.Density, hit.OtherInvMass));
        physics.Impactor.InvMass = hit.OtherInvMass;
        physics.NominalArea = hit.NominalArea;
        physics.ResultantIndex = NearestSamplePoint(modes->Positions, to_local(scene.LocalPoint, hit.ResultantPoint));
        StrikeContext struck = scene.StruckBody ? scene.StruckBody(hit.Entity, hit.Point) : StrikeContext{};
        const double own = scene.RoughnessOf ? scene.RoughnessOf(hit.Entity) : 0.0, other = scene.RoughnessOf ? scene.RoughnessOf(hit.Other) : 0.0;
        struck.Roughness = std::hypot(own, other); // the pair's combined RMS roughness
        queued += TriggerModalStrike(audio, hit.Entity, *modes, sample_point, along, hit.Impulse, hit.Speed, struck, physics) ? 1u : 0u;
    }
    return queued;
}
