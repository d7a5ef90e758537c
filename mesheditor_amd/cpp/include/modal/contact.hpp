// Hertz / flat-punch contact-time model with the reference's API (src/audio/ContactModel.h:25-98).  Scalar host code:
// it runs once per strike on the calling thread and produces ModalEvent fields.
#pragma once
#include "types.hpp"

struct ContactDynamics {
    double Mass{0};
    mat3 InverseInertia{};
    std::vector<vec3> ContactArm;
};
struct Striker {
    AcousticMaterial Material{materials::acoustic::Steel};
    float TipRadius{0.01f}, Length{0.19f};
};
struct Impactor {
    AcousticMaterialProperties Material{};
    double Curvature{0}, InvMass{0};
};

double StrikerMass(const Striker &);
Impactor StrikerImpactor(const Striker &);
mat3 InverseInertiaTensor(const MassProperties &);
double ReducedContactMass(const ContactDynamics &, uint32_t excitable_index, vec3 impact_direction, const Impactor &);
double InvEffectiveModulus(const AcousticMaterialProperties &, const AcousticMaterialProperties &);
double CombinedCurvature(double curvature_a, double curvature_b);
double ContactStiffness(double inv_effective_modulus, double combined_curvature);
double ContactPatchRadius(double normal_force, double inv_effective_modulus, double combined_curvature);
double StaticPenetration(double normal_force, double stiffness);
double SaturationPenetration(double combined_curvature, double nominal_area);
double PunchStiffness(double inv_effective_modulus, double nominal_area);
double EstimateContactTime(const ContactDynamics &, uint32_t excitable_index, vec3 impact_direction, double contact_speed,
                           const AcousticMaterialProperties &object_material, double object_curvature, double nominal_area, const Impactor &,
                           double scale_ratio, double combined_roughness = 0);
inline constexpr double MinContactTime = 2e-5, MaxContactTime = 5e-2;
