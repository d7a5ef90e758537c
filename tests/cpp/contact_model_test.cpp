// Known answers of the contact-time model (the reference's tests/ContactModelTest.cpp:55-125) against the mirrored
// host implementation: the Hertz and flat-punch closed forms, their speed laws, the clamps and the striker's mass.
#include "harness.hpp"

#include <audio/ContactModel.h>
#include <modal/strike.hpp>

#include <numbers>

namespace {
constexpr AcousticMaterialProperties Polymer{.Density = 1000, .YoungModulus = 1e9, .PoissonRatio = 0.3, .Alpha = 0, .Beta = 0};
constexpr AcousticMaterialProperties Ceramic{.Density = 2700, .YoungModulus = 7.2e10, .PoissonRatio = 0.19, .Alpha = 0, .Beta = 0};

// An impactor that contributes nothing: rigid, flat and immovable, so the closed forms of the struck body alone apply.
Striker Immovable() {
    Striker s;
    s.Material = {"immovable", {.Density = 1e6, .YoungModulus = 1e30, .PoissonRatio = 0, .Alpha = 0, .Beta = 0}};
    s.TipRadius = s.Length = 1e6f;
    return s;
}

// One head-on collision along +z; every case varies a field or two and asks for the contact time.
struct Collision {
    double Mass{1};
    float RotationalCompliance{1}; // the inverse inertia tensor is this times the identity
    vec3 Arm{0.f};
    AcousticMaterialProperties Solid{Polymer};
    double Curvature{100}, Area{0}, Speed{1}, Scale{1};
    Striker With{Immovable()};

    ContactDynamics Dynamics() const {
        ContactDynamics d;
        d.Mass = Mass;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) d.InverseInertia[i][j] = i == j ? RotationalCompliance : 0.f;
        d.ContactArm.push_back(Arm);
        return d;
    }
    double Seconds() const { return EstimateContactTime(Dynamics(), 0, vec3{0, 0, 1}, Speed, Solid, Curvature, Area, StrikerImpactor(With), Scale); }
    Collision At(double speed) const { auto c = *this; c.Speed = speed; return c; }
    Collision Over(double area) const { auto c = *this; c.Area = area; return c; }
};
using check::near;
} // namespace

CASE(inverse_inertia_undoes_a_principal_decomposition) {
    // I = R diag(2, 5, 9) R^T for an arbitrary unit quaternion; I * InverseInertiaTensor must be the identity
    MassProperties mp;
    mp.Mass = 1.0;
    mp.InertiaDiagonal = {2.f, 5.f, 9.f};
    const float raw[4]{0.3f, 0.1f, -0.5f, 0.8f};
    const float norm = std::sqrt(raw[0] * raw[0] + raw[1] * raw[1] + raw[2] * raw[2] + raw[3] * raw[3]);
    mp.InertiaOrientation = {raw[0] / norm, raw[1] / norm, raw[2] / norm, raw[3] / norm};
    // rotate the basis vectors by the quaternion: v' = v + 2 w (u x v) + 2 u x (u x v)
    const double w = mp.InertiaOrientation.w, u[3]{mp.InertiaOrientation.x, mp.InertiaOrientation.y, mp.InertiaOrientation.z};
    double axis[3][3]; // axis[k] = image of basis vector k
    for (int k = 0; k < 3; ++k) {
        double v[3]{0, 0, 0};
        v[k] = 1;
        const double c1[3]{u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
        const double c2[3]{u[1] * c1[2] - u[2] * c1[1], u[2] * c1[0] - u[0] * c1[2], u[0] * c1[1] - u[1] * c1[0]};
        for (int i = 0; i < 3; ++i) axis[k][i] = v[i] + 2 * w * c1[i] + 2 * c2[i];
    }
    const mat3 inverse = InverseInertiaTensor(mp);
    for (int row = 0; row < 3; ++row)
        for (int col = 0; col < 3; ++col) {
            double product = 0; // (I * I^-1)[row][col]
            for (int mid = 0; mid < 3; ++mid) {
                double inertia = 0;
                for (int k = 0; k < 3; ++k) inertia += axis[k][row] * double(mp.InertiaDiagonal[k]) * axis[k][mid];
                product += inertia * double(inverse[col][mid]);
            }
            EXPECT(std::abs(product - (row == col ? 1.0 : 0.0)) < 1e-4);
        }
}

CASE(contact_time_matches_the_hertz_formula) {
    const Collision centred;
    EXPECT_NOTE(near(centred.Seconds(), 1.744e-3, 2e-2), std::to_string(centred.Seconds()));
    Collision levered;
    levered.Arm = vec3{0.2f, 0, 0};
    EXPECT(levered.Seconds() < centred.Seconds()); // a lever arm lowers the effective mass
}

CASE(scale_ratio_and_clamps) {
    const auto scaled = [](double scale) {
        Collision c;
        c.Scale = scale;
        return c.Seconds();
    };
    EXPECT(near(scaled(2.0), 2 * scaled(1.0), 1e-6)); // contact time grows linearly with the body's size
    EXPECT(near(scaled(100.0), MaxContactTime, 1e-12));
    EXPECT(near(scaled(1e-6), MinContactTime, 1e-12));
}

CASE(the_contact_time_reaches_both_limits) {
    Collision sphere; // no rotational give: the pure translational closed forms
    sphere.RotationalCompliance = 0;
    Collision flat = sphere;
    flat.Curvature = 0;
    flat.Area = 1e-4;
    const double compliance = 0.91 / 1e9; // (1 - nu^2) / E of the polymer
    // Hertz: tau = 2.868 (m^2 / (E*^2 R v))^(1/5); flat punch: half a period of the mass on the punch stiffness
    EXPECT(near(sphere.Seconds(), 2.868 * std::pow(compliance * compliance * sphere.Curvature, 0.2), 1e-3));
    EXPECT(near(flat.Seconds(), std::numbers::pi * std::sqrt(compliance / (2 * std::sqrt(flat.Area / std::numbers::pi))), 1e-3));
    EXPECT(near(sphere.At(32).Seconds() / sphere.Seconds(), std::pow(32.0, -0.2), 1e-3)); // Hertz shortens as v^(-1/5)
    EXPECT(near(flat.At(32).Seconds() / flat.Seconds(), 1.0, 1e-3)); // a linear spring does not care
}

CASE(filling_the_patch_stops_the_contact_stiffening) {
    Collision base;
    base.Mass = 0.5;
    base.RotationalCompliance = 0;
    base.Solid = Ceramic;
    base.Curvature = 10;
    const double patch = 1e-5;
    EXPECT(near(SaturationPenetration(base.Curvature, patch), 3.183e-5, 1e-3));
    EXPECT(near(base.Over(patch).At(0.1).Seconds(), base.At(0.1).Seconds(), 1e-6)); // a gentle hit never fills the patch
    EXPECT(near(base.Over(1.0).At(3).Seconds(), base.At(3).Seconds(), 1e-6)); // nor does a hard one fill a huge patch
    EXPECT(base.Over(patch).At(3).Seconds() > base.At(3).Seconds());
    EXPECT(base.Over(patch).At(3).Seconds() > std::numbers::pi * std::sqrt(0.5 / PunchStiffness(0.91 / 7.2e10, patch)));
    EXPECT(near(base.Over(1.7e-5).Seconds(), base.Over(1.5e-5).Seconds(), 1e-3)); // continuous across the fill depth
    const double free_law = base.At(3).Seconds() / base.At(0.1).Seconds();
    const double filled_law = base.Over(patch).At(3).Seconds() / base.Over(patch).At(0.1).Seconds();
    EXPECT(near(free_law, std::pow(30.0, -0.2), 1e-3));
    EXPECT(filled_law > free_law && filled_law < 1.0);
}

CASE(a_lighter_striker_shortens_the_contact) {
    Collision c;
    c.Mass = 1000;
    c.RotationalCompliance = 0;
    c.Solid = Ceramic;
    c.Curvature = 5;
    Collision heavy = c;
    c.With = Striker{};
    c.With.Length = 0.05f;
    heavy.With = Striker{};
    heavy.With.Length = 5.f;
    EXPECT(c.Seconds() < heavy.Seconds());
    EXPECT(StrikerMass(heavy.With) > StrikerMass(c.With));
}

// ---- strike translation (the reference's TriggerModalStrike, src/audio/AudioSystem.cpp:400-465) ------------
namespace {
ModalModes TwoPointModes() {
    ModalModes m;
    m.Freqs = {440.f, 880.f};
    m.T60s = {0.5f, 0.25f};
    m.Positions = {{0, 0, 0}, {0.1f, 0, 0}, {0, 0.1f, 0}};
    m.Vertices = {0, 1, 2};
    m.Indices = {0, 1, 2};
    m.Shapes = {{{1, 0, 0}, {0, 2, 0}}, {{0, 1, 0}, {3, 0, 0}}, {{0, 0, 1}, {0, 0, 1}}};
    return m;
}
} // namespace

CASE(a_strike_becomes_the_event_the_bank_consumes) {
    const auto modes = TwoPointModes();
    ModalAudio audio; // bank bookkeeping only: nothing is rendered, so no device is touched
    ModalBank &bank = LiveBank(audio);
    bank.SampleRate = 48'000.f;
    const auto slot = AddModalObject(bank, entt::entity{7}, modes);
    // helpers of the scene look-ups
    EXPECT(NearestSamplePoint(modes.Positions, vec3{0.09f, 0.01f, 0}) == 1u);
    EXPECT(check::near(PeakModalDrive(modes, 1, vec3{2.f, 0, 0}), 6.0, 1e-6)); // mode 1 at point 1: (3,0,0) . (2,0,0)
    EXPECT(length(UnitOrZero(vec3{0.f})) == 0.f && check::near(length(UnitOrZero(vec3{3, 4, 0})), 1.0, 1e-6));
    EXPECT(check::near(VolumeEquivalentRadius(4.0 / 3.0 * std::numbers::pi * 0.027), 0.3, 1e-12));
    // without dynamics or material: the default 0.1 ms contact, no click
    StrikeContext bare;
    const auto plain = MakeStrikeEvent(bank, slot, 1, vec3{0, 0, 1}, 2.f, 1.f, bare);
    EXPECT(plain.Object == slot && plain.ExPos == 1u && plain.Jz == 2.f && plain.Jx == 0.f);
    EXPECT(check::near(plain.PulseStep, 1.0 / (1e-4 * 48000.0), 1e-6) && plain.PulseGamma == 2 * plain.PulseStep && plain.AccelAmp == 0.f && plain.ClickB0 == 0.f);
    // with them: the contact time of the model, the recoil click of the displaced volume, a nominal impulse for a mallet
    Collision struck;
    struck.Mass = 0.5;
    struck.RotationalCompliance = 0;
    ContactDynamics dyn = struck.Dynamics();
    dyn.ContactArm = {vec3{0.f}, vec3{0.f}, vec3{0.f}};
    const AcousticMaterial ceramic{"c", Ceramic};
    StrikeContext sc{.Dynamics = &dyn, .Material = &ceramic, .Elastic = Ceramic, .Curvature = 10, .EnclosedVolume = 0, .ScaleRatio = 1.f, .Roughness = 0};
    const Striker mallet{};
    const auto hit = MakeStrikeEvent(bank, slot, 1, vec3{0, 0, 1}, 1.f, 2.f, sc, std::nullopt, mallet);
    const double tau = EstimateContactTime(dyn, 1, vec3{0, 0, 1}, 2.0, Ceramic, 10, 0, StrikerImpactor(mallet), 1.0, 0.0);
    EXPECT(check::near(hit.PulseStep, 1.0 / (tau * 48000.0), 1e-6));
    const double volume = 0.5 / 2700.0;
    const auto click = RecoilClickFilter(VolumeEquivalentRadius(volume), volume, 0.5, 48000.0);
    EXPECT(hit.ClickB0 == click.B0 && hit.ClickA1 == click.A1 && hit.ClickA2 == click.A2);
    EXPECT(check::near(hit.AccelAmp, ReducedContactMass(dyn, 1, vec3{0, 0, 1}, StrikerImpactor(mallet)) * 2.0 * 48000.0, 1e-6));
    // a collision carries its own impactor and a true impulse
    PhysicsStrike phys{.Direction = vec3{0, 0, 2}, .Impactor = Impactor{Polymer, 5.0, 1.0}, .NominalArea = 1e-5f, .ResultantIndex = 2};
    const auto crash = MakeStrikeEvent(bank, slot, 1, vec3{0, 0, 1}, 0.03f, 1.5f, sc, phys);
    EXPECT(check::near(crash.AccelAmp, 0.03 * 48000.0, 1e-6));
    const double tau_c = EstimateContactTime(dyn, 2, vec3{0, 0, 1}, 1.5, Ceramic, 10, 1e-5f, phys.Impactor, 1.0, 0.0);
    EXPECT(check::near(crash.PulseStep, 1.0 / (tau_c * 48000.0), 1e-6));
    // queued through the entity look-up; unknown entities and out-of-range points are ignored
    EXPECT(TriggerModalStrike(audio, entt::entity{7}, modes, 1, vec3{0, 0, 1}, 1.f, 1.f, sc));
    EXPECT(!TriggerModalStrike(audio, entt::entity{8}, modes, 1, vec3{0, 0, 1}, 1.f, 1.f, sc));
    EXPECT(!TriggerModalStrike(audio, entt::entity{7}, modes, 5, vec3{0, 0, 1}, 1.f, 1.f, sc));
    EXPECT(audio.EventWrite.load() - audio.EventRead.load() == 1u);
}

CASE(collisions_strike_the_objects_they_hit) {
    // The physics event source (reference: PhysicsContact.h ContactImpact, drained by AudioSystem.cpp:1007-1037): speed and
    // excitation floors, nearest sample point, the other body as impactor, one queued strike per audible contact point.
    const auto modes = TwoPointModes();
    ModalAudio audio;
    ModalBank &bank = LiveBank(audio);
    bank.SampleRate = 48'000.f;
    AddModalObject(bank, entt::entity{7}, modes);
    Collision struck_body;
    struck_body.Mass = 0.5;
    struck_body.RotationalCompliance = 0;
    ContactDynamics dyn = struck_body.Dynamics();
    dyn.ContactArm = {vec3{0.f}, vec3{0.f}, vec3{0.f}};
    const AcousticMaterial ceramic{"c", Ceramic};
    int context_calls = 0;
    StrikeScene scene;
    scene.ModesOf = [&](entt::entity e) { return e == entt::entity{7} ? &modes : nullptr; };
    scene.LocalPoint = [](entt::entity, vec3 world) { return world - vec3{1.f, 0.f, 0.f}; }; // the body sits at x = 1
    scene.StruckBody = [&](entt::entity, vec3) {
        ++context_calls;
        return StrikeContext{.Dynamics = &dyn, .Material = &ceramic, .Elastic = Ceramic, .Curvature = 10};
    };
    scene.MaterialOf = [](entt::entity) { return Polymer; };
    scene.RoughnessOf = [](entt::entity e) { return e == entt::entity{7} ? 3e-6f : 4e-6f; };
    ContactImpact hit;
    hit.Entity = entt::entity{7}, hit.Other = entt::entity{9};
    hit.Point = vec3{1.09f, 0.01f, 0.f}, hit.ResultantPoint = vec3{1.f, 0.09f, 0.f}; // nearest sample points 1 and 2
    hit.Direction = vec3{1.f, 0.f, 0.f};
    hit.Impulse = 0.02f, hit.Speed = 1.2f, hit.OtherInvMass = 4.f, hit.NominalArea = 0.f;
    ContactImpact slow = hit;
    slow.Speed = 0.001f; // below the speed floor
    ContactImpact mute = hit;
    mute.Direction = vec3{0.f, 0.f, 1.f}; // point 1's shapes have no z component: nothing is excited
    mute.Impulse = 1e-9f;
    ContactImpact stranger = hit;
    stranger.Entity = entt::entity{8}; // not a sounding object
    const ContactImpact all[] = {slow, mute, stranger, hit};
    EXPECT(StrikeContacts(audio, all, scene) == 1u && context_calls == 1);
    EXPECT(audio.EventWrite.load() - audio.EventRead.load() == 1u);
    // the queued event is the one MakeStrikeEvent builds for that contact
    const ModalEvent &queued = audio.Events[audio.EventRead.load() % audio.Events.size()];
    PhysicsStrike phys{.Direction = vec3{1.f, 0.f, 0.f}, .Impactor = Impactor{Polymer, SphereEquivalentCurvature(Polymer.Density, 4.0), 4.0}, .NominalArea = 0.f, .ResultantIndex = 2};
    StrikeContext sc{.Dynamics = &dyn, .Material = &ceramic, .Elastic = Ceramic, .Curvature = 10};
    sc.Roughness = 5e-6; // hypot(3, 4) um
    const ModalEvent want = MakeStrikeEvent(bank, 0, 1, vec3{1.f, 0.f, 0.f}, 0.02f, 1.2f, sc, phys);
    EXPECT(queued.Object == want.Object && queued.ExPos == 1u && queued.Jx == want.Jx && queued.Jy == 0.f);
    EXPECT(check::near(queued.PulseStep, want.PulseStep, 1e-6) && queued.AccelAmp == want.AccelAmp && queued.ClickB0 == want.ClickB0);
}

int main() { return check::run_all(); }
