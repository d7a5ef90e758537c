// Contact-time model behind the reference's API (src/audio/ContactModel.h:25-98; behaviour: SURVEY.md section 8a row R9).
// Scalar host code, evaluated once per strike; it produces the pulse length of a ModalEvent.
//
// Physics: two elastic bodies meet with approach speed v.  The normal force follows Hertz (F = kH x^(3/2)) until the
// contact patch has grown to the nominal area the geometry allows, then a flat punch (constant stiffness kP) on top of
// the load Hertz had reached.  The collision lasts twice the time the reduced mass needs to come to rest against that
// force law, plus -- in quadrature -- the arrest time on the asperity cushion of a rough interface.
// Written from that description; the structure (a force-law object and a generic quadrature) is this file's own.
#include "modal/contact.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <numbers>

namespace {
constexpr double Inf = std::numeric_limits<double>::infinity();
constexpr double Pi = std::numbers::pi;

// Force law of the contact as a function of penetration x (m).
class ContactLaw {
public:
    ContactLaw(double inv_modulus, double curvature, double nominal_area)
        : Hertz(ContactStiffness(inv_modulus, curvature)), Punch(PunchStiffness(inv_modulus, nominal_area)), Fill(SaturationPenetration(curvature, nominal_area)) {}

    // Work done pressing the contact to depth x, J.
    double Work(double x) const {
        if (!(x > 0)) return 0;
        if (x <= Fill) return HertzWork(x);
        const double beyond = x - Fill;
        return HertzWork(Fill) + (FillForce() + 0.5 * Punch * beyond) * beyond;
    }
    // The depth at which the contact has absorbed `energy`: the turning point of the approach.
    double DepthAt(double energy) const {
        const double at_fill = std::isfinite(Fill) ? HertzWork(Fill) : Inf;
        if (energy <= at_fill) return std::pow(2.5 * energy / Hertz, 0.4); // invert (2/5) kH x^(5/2)
        // beyond the fill depth: (1/2) kP y^2 + F_fill y = energy - at_fill, positive root
        const double f = FillForce();
        return Fill + (std::sqrt(f * f + 2 * Punch * (energy - at_fill)) - f) / Punch;
    }

private:
    double Hertz, Punch, Fill;
    double HertzWork(double x) const { return 0.4 * Hertz * x * x * std::sqrt(x); }
    double FillForce() const { return Hertz * Fill * std::sqrt(Fill); }
};

// Midpoint rule on [0, 1] with N cells.
template<int N, typename F> double MidpointUnit(F &&f) {
    double total = 0;
    for (int cell = 0; cell < N; ++cell) total += f((cell + 0.5) / N);
    return total / N;
}
} // namespace

double StrikerMass(const Striker &s) {
    // a cylinder of the tip radius closed by two hemispherical caps
    const double r = s.TipRadius, l = s.Length;
    const double cylinder = r * r * l, caps = 4.0 / 3.0 * r * r * r;
    return s.Material.Properties.Density * Pi * (cylinder + caps);
}

Impactor StrikerImpactor(const Striker &s) {
    Impactor out;
    out.Material = s.Material.Properties;
    out.Curvature = 1.0 / s.TipRadius;
    out.InvMass = 1.0 / StrikerMass(s);
    return out;
}

// World-frame inverse inertia R diag(1/I) R^T, accumulated as a sum of outer products of the principal axes.
mat3 InverseInertiaTensor(const MassProperties &mp) {
    const quat &q = mp.InertiaOrientation;
    // principal axes = columns of the rotation matrix of q
    const float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
    const float xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z, wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
    const vec3 axis[3] = {
        {1.f - 2.f * (yy + zz), 2.f * (xy + wz), 2.f * (xz - wy)},
        {2.f * (xy - wz), 1.f - 2.f * (xx + zz), 2.f * (yz + wx)},
        {2.f * (xz + wy), 2.f * (yz - wx), 1.f - 2.f * (xx + yy)},
    };
    mat3 out;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) out[c][r] = 0.f;
    for (int k = 0; k < 3; ++k) {
        const float moment = mp.InertiaDiagonal[k];
        if (!(moment > 0)) continue; // a vanishing principal moment contributes no compliance
        const float compliance = 1.f / moment;
        for (int c = 0; c < 3; ++c)
            for (int r = 0; r < 3; ++r) out[c][r] += axis[k][r] * compliance * axis[k][c];
    }
    return out;
}

// 1 / (1/m + (r x n) . I^-1 (r x n) + 1/m_impactor): translation, the leverage of an off-centre impulse, and the impactor.
double ReducedContactMass(const ContactDynamics &d, uint32_t i, vec3 impact_direction, const Impactor &impactor) {
    if (i >= d.ContactArm.size() || !(d.Mass > 0)) return 0;
    const vec3 lever = cross(d.ContactArm[i], normalize(impact_direction));
    vec3 turned{0.f};
    for (int c = 0; c < 3; ++c) // column-major matrix times vector
        for (int r = 0; r < 3; ++r) turned[r] += d.InverseInertia[c][r] * lever[c];
    const double compliance = 1.0 / d.Mass + double(dot(lever, turned)) + impactor.InvMass;
    return 1.0 / compliance;
}

// Plane-strain moduli in series: 1/E* = (1 - nu_a^2)/E_a + (1 - nu_b^2)/E_b
double InvEffectiveModulus(const AcousticMaterialProperties &a, const AcousticMaterialProperties &b) {
    const auto one_side = [](const AcousticMaterialProperties &m) { return (1 - m.PoissonRatio * m.PoissonRatio) / m.YoungModulus; };
    return one_side(a) + one_side(b);
}
double CombinedCurvature(double a, double b) { return std::max(a + b, 1e-6); }
// Hertz: F = (4/3) E* sqrt(R) x^(3/2), R = 1 / combined curvature
double ContactStiffness(double inv_modulus, double curvature) { return (4.0 / 3.0) / inv_modulus / std::sqrt(curvature); }
// Hertz patch radius a^3 = 3 F R / (4 E*)
double ContactPatchRadius(double force, double inv_modulus, double curvature) { return std::cbrt(0.75 * std::max(force, 0.0) * inv_modulus / curvature); }
double StaticPenetration(double force, double stiffness) { return stiffness > 0 ? std::pow(std::max(force, 0.0) / stiffness, 2.0 / 3.0) : 0.0; }
// The patch (pi a^2, a^2 = R x) fills the nominal area at x = A kappa / pi; no area limit: never.
double SaturationPenetration(double curvature, double area) { return area > 0 ? area * curvature / Pi : Inf; }
// Flat circular punch of the nominal area: k = 2 E* a, a = sqrt(A / pi)
double PunchStiffness(double inv_modulus, double area) { return area > 0 ? 2 * std::sqrt(area / Pi) / inv_modulus : Inf; }

double EstimateContactTime(const ContactDynamics &d, uint32_t i, vec3 impact_direction, double contact_speed, const AcousticMaterialProperties &m,
                           double object_curvature, double nominal_area, const Impactor &impactor, double scale_ratio, double combined_roughness) {
    if (i >= d.ContactArm.size() || !(d.Mass > 0)) return MinContactTime;
    const double mass = ReducedContactMass(d, i, impact_direction, impactor);
    const double inv_modulus = InvEffectiveModulus(m, impactor.Material);
    if (!(mass > 0) || !(inv_modulus > 0)) return MinContactTime;

    const double speed = std::max(std::abs(contact_speed), 1e-6);
    const double energy = 0.5 * mass * speed * speed;
    const ContactLaw law(inv_modulus, CombinedCurvature(object_curvature, impactor.Curvature), nominal_area);
    const double deepest = law.DepthAt(energy);

    // Approach time = integral of dx / v(x) with v(x) = v sqrt(1 - W(x)/E).  Substituting x = deepest (1 - s^2) removes the
    // inverse-square-root singularity at the turning point; 64 midpoints resolve what is left.  The rebound mirrors it.
    const double shape = MidpointUnit<64>([&](double s) {
        const double remaining = 1 - law.Work(deepest * (1 - s * s)) / energy;
        return remaining > 0 ? 2 * s / std::sqrt(remaining) : 0.0;
    });
    const double bulk = 2 * deepest / speed * shape * scale_ratio;
    // Rough surfaces meet on an exponential asperity cushion of decay length u0 = 0.4 x the combined rms roughness; the
    // arrest on it takes pi sqrt(2) u0 / v.  Compliances in series add their contact times in quadrature.
    const double cushion = std::numbers::sqrt2 * Pi * (0.4 * combined_roughness) / speed;
    return std::clamp(std::hypot(bulk, cushion), MinContactTime, MaxContactTime);
}
