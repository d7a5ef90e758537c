// Known answers of the contact-time model (the reference's tests/ContactModelTest.cpp:55-125) against the mirrored
// host implementation: the Hertz and flat-punch closed forms, their speed laws, the clamps and the striker's mass.
#include "harness.hpp"

#include <audio/ContactModel.h>
#include <modal/strike.hpp>

#include <numbers>

namespace {
const Striker NullStriker{.Material = {.Name = "null", .Properties = {.Density = 1e6, .YoungModulus = 1e30, .PoissonRatio = 0, .Alpha = 0, .Beta = 0}},
                          .TipRadius = 1e6f, .Length = 1e6f};
constexpr AcousticMaterialProperties Polymer{.Density = 1000, .YoungModulus = 1e9, .PoissonRatio = 0.3, .Alpha = 0, .Beta = 0};
constexpr AcousticMaterialProperties Ceramic{.Density = 2700, .YoungModulus = 7.2e10, .PoissonRatio = 0.19, .Alpha = 0, .Beta = 0};

mat3 Scaled(float s) {
    mat3 m;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) m[c][r] = c == r ? s : 0.f;
    return m;
}
ContactDynamics Body(double mass, mat3 inverse_inertia, vec3 arm = vec3{0}) {
    ContactDynamics d;
    d.Mass = mass;
    d.InverseInertia = inverse_inertia;
    d.ContactArm = {arm};
    return d;
}
double ContactTime(const ContactDynamics &d, const AcousticMaterialProperties &material, double curvature, double area = 0, double speed = 1, double scale = 1,
                   const Striker &striker = NullStriker) {
    return EstimateContactTime(d, 0, vec3{0, 0, 1}, speed, material, curvature, area, StrikerImpactor(striker), scale);
}
using check::near;
} // namespace

CASE(inverse_inertia_undoes_a_principal_decomposition) {
    MassProperties mp;
    mp.Mass = 1.0;
    mp.InertiaDiagonal = {2.f, 5.f, 9.f};
    const float qn = std::sqrt(0.3f * 0.3f + 0.1f * 0.1f + 0.5f * 0.5f + 0.8f * 0.8f);
    mp.InertiaOrientation = {0.3f / qn, 0.1f / qn, -0.5f / qn, 0.8f / qn};
    const auto q = mp.InertiaOrientation;
    // rotation matrix of the unit quaternion, column-major m[col][row]
    const float R[3][3]{{1 - 2 * (q.y * q.y + q.z * q.z), 2 * (q.x * q.y + q.w * q.z), 2 * (q.x * q.z - q.w * q.y)},
                        {2 * (q.x * q.y - q.w * q.z), 1 - 2 * (q.x * q.x + q.z * q.z), 2 * (q.y * q.z + q.w * q.x)},
                        {2 * (q.x * q.z + q.w * q.y), 2 * (q.y * q.z - q.w * q.x), 1 - 2 * (q.x * q.x + q.y * q.y)}};
    const float diag[3]{2, 5, 9};
    double inertia[3][3]{}; // [col][row] of R diag R^T
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k) inertia[c][r] += double(R[k][r]) * diag[k] * R[k][c];
    const auto inv = InverseInertiaTensor(mp);
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) {
            double v = 0;
            for (int k = 0; k < 3; ++k) v += inertia[k][r] * double(inv[c][k]);
            EXPECT(std::abs(v - (c == r ? 1.0 : 0.0)) < 1e-4);
        }
}

CASE(contact_time_matches_the_hertz_formula) {
    const double tau = ContactTime(Body(1.0, Scaled(1.f)), Polymer, 100);
    EXPECT_NOTE(near(tau, 1.744e-3, 2e-2), std::to_string(tau));
    const double off_centre = ContactTime(Body(1.0, Scaled(1.f), vec3{0.2f, 0, 0}), Polymer, 100);
    EXPECT(off_centre < tau); // a lever arm lowers the effective mass
}

CASE(scale_ratio_and_clamps) {
    const auto d = Body(1.0, Scaled(1.f));
    const auto tau = [&d](double scale) { return ContactTime(d, Polymer, 100, 0, 1, scale); };
    EXPECT(near(tau(2.0), 2 * tau(1.0), 1e-6));
    EXPECT(near(tau(100.0), MaxContactTime, 1e-12));
    EXPECT(near(tau(1e-6), MinContactTime, 1e-12));
}

CASE(the_contact_time_reaches_both_limits) {
    const auto d = Body(1.0, Scaled(0.f));
    constexpr double InvModulus = 0.91 / 1e9;
    const auto tau = [&d](double curvature, double area, double speed) { return ContactTime(d, Polymer, curvature, area, speed); };
    constexpr double Curvature = 100;
    const double hertz = 2.868 * std::pow(std::pow(InvModulus, 2) * Curvature, 0.2);
    EXPECT(near(tau(Curvature, 0.0, 1.0), hertz, 1e-3));
    constexpr double Area = 1e-4;
    const double punch = std::numbers::pi * std::sqrt(InvModulus / (2 * std::sqrt(Area / std::numbers::pi)));
    EXPECT(near(tau(0.0, Area, 1.0), punch, 1e-3));
    EXPECT(near(tau(Curvature, 0.0, 32.0) / tau(Curvature, 0.0, 1.0), std::pow(32.0, -0.2), 1e-3));
    EXPECT(near(tau(0.0, Area, 32.0) / tau(0.0, Area, 1.0), 1.0, 1e-3));
}

CASE(filling_the_patch_stops_the_contact_stiffening) {
    const auto d = Body(0.5, Scaled(0.f));
    constexpr double Curvature = 10, Area = 1e-5;
    const auto tau = [&d](double area, double speed) { return ContactTime(d, Ceramic, Curvature, area, speed); };
    EXPECT(near(SaturationPenetration(Curvature, Area), 3.183e-5, 1e-3));
    EXPECT(near(tau(Area, 0.1), tau(0.0, 0.1), 1e-6));
    EXPECT(near(tau(1.0, 3.0), tau(0.0, 3.0), 1e-6));
    EXPECT(tau(Area, 3.0) > tau(0.0, 3.0));
    EXPECT(tau(Area, 3.0) > std::numbers::pi * std::sqrt(0.5 / PunchStiffness(0.91 / 7.2e10, Area)));
    EXPECT(near(tau(1.7e-5, 1.0), tau(1.5e-5, 1.0), 1e-3));
    const double hertz_ratio = tau(0.0, 3.0) / tau(0.0, 0.1), saturating_ratio = tau(Area, 3.0) / tau(Area, 0.1);
    EXPECT(near(hertz_ratio, std::pow(30.0, -0.2), 1e-3));
    EXPECT(saturating_ratio > hertz_ratio);
    EXPECT(saturating_ratio < 1.0);
}

CASE(a_lighter_striker_shortens_the_contact) {
    const auto d = Body(1000.0, Scaled(0.f));
    Striker light;
    light.Length = 0.05f;
    Striker heavy = light;
    heavy.Length = 5.f;
    EXPECT(ContactTime(d, Ceramic, 5, 0, 1, 1, light) < ContactTime(d, Ceramic, 5, 0, 1, 1, heavy));
    EXPECT(StrikerMass(heavy) > StrikerMass(light));
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
    ContactDynamics dyn = Body(0.5, Scaled(0.f));
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

int main() { return check::run_all(); }
