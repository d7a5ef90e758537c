// Contact-time model (reference src/audio/ContactModel.cpp:10-114), scalar host code.
#include "modal/contact.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <numbers>

double StrikerMass(const Striker &s) {
    const double r = s.TipRadius, l = s.Length;
    return s.Material.Properties.Density * std::numbers::pi * (r * r * l + 4.0 / 3.0 * r * r * r);
}
Impactor StrikerImpactor(const Striker &s) { return {s.Material.Properties, 1.0 / s.TipRadius, 1.0 / StrikerMass(s)}; }

mat3 InverseInertiaTensor(const MassProperties &mp) {
    const auto &q = mp.InertiaOrientation;
    mat3 r; // rotation from the quaternion, column-major
    const float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z, xz = q.x * q.z, xy = q.x * q.y, yz = q.y * q.z, wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
    r[0][0] = 1.f - 2.f * (yy + zz); r[0][1] = 2.f * (xy + wz); r[0][2] = 2.f * (xz - wy);
    r[1][0] = 2.f * (xy - wz); r[1][1] = 1.f - 2.f * (xx + zz); r[1][2] = 2.f * (yz + wx);
    r[2][0] = 2.f * (xz + wy); r[2][1] = 2.f * (yz - wx); r[2][2] = 1.f - 2.f * (xx + yy);
    float inv[3];
    for (int i = 0; i < 3; ++i) inv[i] = mp.InertiaDiagonal[i] > 0 ? 1.f / mp.InertiaDiagonal[i] : 0.f;
    mat3 out;
    for (int c = 0; c < 3; ++c)
        for (int row = 0; row < 3; ++row) {
            float s = 0;
            for (int k = 0; k < 3; ++k) s += r[k][row] * inv[k] * r[k][c];
            out[c][row] = s;
        }
    return out;
}

double ReducedContactMass(const ContactDynamics &d, uint32_t i, vec3 impact_direction, const Impactor &impactor) {
    if (i >= d.ContactArm.size() || d.Mass <= 0) return 0;
    const vec3 n = normalize(impact_direction);
    const vec3 c = cross(d.ContactArm[i], n);
    const auto &I = d.InverseInertia;
    const vec3 ic{I[0][0] * c.x + I[1][0] * c.y + I[2][0] * c.z, I[0][1] * c.x + I[1][1] * c.y + I[2][1] * c.z, I[0][2] * c.x + I[1][2] * c.y + I[2][2] * c.z};
    const double inv_effective_mass = 1.0 / d.Mass + dot(c, ic) + impactor.InvMass;
    return 1.0 / inv_effective_mass;
}

double InvEffectiveModulus(const AcousticMaterialProperties &a, const AcousticMaterialProperties &b) {
    return (1 - a.PoissonRatio * a.PoissonRatio) / a.YoungModulus + (1 - b.PoissonRatio * b.PoissonRatio) / b.YoungModulus;
}
double CombinedCurvature(double a, double b) { return std::max(a + b, 1e-6); }
double ContactStiffness(double inv_modulus, double curvature) { return 4.0 / 3.0 / inv_modulus / std::sqrt(curvature); }
double ContactPatchRadius(double force, double inv_modulus, double curvature) { return std::cbrt(0.75 * std::max(force, 0.0) * inv_modulus / curvature); }
double StaticPenetration(double force, double stiffness) { return stiffness > 0 ? std::pow(std::max(force, 0.0) / stiffness, 2.0 / 3.0) : 0.0; }
double SaturationPenetration(double curvature, double area) { return area > 0 ? area * curvature / std::numbers::pi : std::numeric_limits<double>::infinity(); }
double PunchStiffness(double inv_modulus, double area) {
    if (area <= 0) return std::numeric_limits<double>::infinity();
    return 2 * std::sqrt(area / std::numbers::pi) / inv_modulus;
}

namespace {
// Work against the contact up to `penetration`: Hertz below saturation, constant stiffness above it.
double ContactWork(double penetration, double hertz_k, double sat, double punch_k) {
    if (penetration <= 0) return 0;
    const auto hertz = [hertz_k](double x) { return 0.4 * hertz_k * x * x * std::sqrt(x); };
    if (penetration <= sat) return hertz(penetration);
    const double over = penetration - sat;
    const double sat_force = hertz_k * sat * std::sqrt(sat);
    return hertz(sat) + sat_force * over + 0.5 * punch_k * over * over;
}
} // namespace

double EstimateContactTime(const ContactDynamics &d, uint32_t i, vec3 impact_direction, double contact_speed, const AcousticMaterialProperties &m,
                           double object_curvature, double nominal_area, const Impactor &impactor, double scale_ratio, double combined_roughness) {
    if (i >= d.ContactArm.size() || d.Mass <= 0) return MinContactTime;
    const double effective_mass = ReducedContactMass(d, i, impact_direction, impactor);
    const double inv_modulus = InvEffectiveModulus(m, impactor.Material);
    if (effective_mass <= 0 || inv_modulus <= 0) return MinContactTime;
    const double curvature = CombinedCurvature(object_curvature, impactor.Curvature);
    const double speed = std::max(std::abs(contact_speed), 1e-6);
    const double hertz_k = ContactStiffness(inv_modulus, curvature);
    const double sat = SaturationPenetration(curvature, nominal_area);
    const double punch_k = PunchStiffness(inv_modulus, nominal_area);
    const double energy = 0.5 * effective_mass * speed * speed;
    const double sat_work = std::isfinite(sat) ? ContactWork(sat, hertz_k, sat, punch_k) : std::numeric_limits<double>::infinity();
    double max_pen;
    if (energy <= sat_work) {
        max_pen = std::pow(energy / (0.4 * hertz_k), 0.4);
    } else {
        const double sat_force = hertz_k * sat * std::sqrt(sat);
        max_pen = sat + (std::sqrt(sat_force * sat_force + 2 * punch_k * (energy - sat_work)) - sat_force) / punch_k;
    }
    constexpr int Steps = 64; // midpoint rule in s with x = max*(1 - s^2)
    double sum = 0;
    for (int n = 0; n < Steps; ++n) {
        const double s = (double(n) + 0.5) / Steps;
        const double left = 1 - ContactWork(max_pen * (1 - s * s), hertz_k, sat, punch_k) / energy;
        if (left > 0) sum += 2 * s / std::sqrt(left);
    }
    const double bulk_time = 2 * max_pen / speed * sum / Steps * scale_ratio;
    const double u0 = 0.4 * combined_roughness;
    const double bed_time = std::numbers::sqrt2 * std::numbers::pi * u0 / speed;
    return std::clamp(std::sqrt(bulk_time * bulk_time + bed_time * bed_time), MinContactTime, MaxContactTime);
}
