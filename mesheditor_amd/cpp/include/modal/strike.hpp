// Strike translation: a contact -- a mallet hit or a physics collision -- becomes the
// ModalEvent the bank consumes.  The reference does this inside its ECS (src/audio/AudioSystem.cpp:400-465,
// TriggerModalStrike; :1007-1037 for collisions; scene-lookup helpers of src/audio/ContactScene.h:97-122); here the
// registry look-ups are replaced by an explicit StrikeContext the caller fills, the arithmetic is the reference's.
#pragma once
#include "bank.hpp"
#include "contact.hpp"

#include <functional>
#include <optional>
#include <span>

// What the reference reads off the scene for the struck object.
struct StrikeContext {
    const ContactDynamics *Dynamics{nullptr}; // mass, inverse inertia, contact arms per excitable point
    const AcousticMaterial *Material{nullptr}; // the struck object's material (for the displaced volume)
    AcousticMaterialProperties Elastic{materials::acoustic::Ceramic.Properties}; // elastic constants of the struck surface
    double Curvature{0}; // 1/m at the strike point (0: flat)
    double EnclosedVolume{0}; // m^3 the mesh encloses at its world scale; 0: take mass / density
    float ScaleRatio{1.f}; // current size over the solved size (UniformScaleRatio, ContactScene.h:97)
    double Roughness{0}; // RMS roughness of the contact pair, m
};
// The collision-only inputs (PhysicsStrike of the reference); a mallet hit leaves them out.
struct PhysicsStrike {
    vec3 Direction{0.f, 0.f, 1.f};
    ::Impactor Impactor{};
    float NominalArea{0.f};
    uint32_t ResultantIndex{0};
};

vec3 UnitOrZero(vec3 v);
uint32_t NearestSamplePoint(const std::vector<vec3> &positions, vec3 local_point);
float PeakModalDrive(const ModalModes &, uint32_t sample_point, vec3 impulse);
double VolumeEquivalentRadius(double volume);
double SphereEquivalentCurvature(double density, double inv_mass);
double DisplacedVolume(double enclosed_volume, double mass, const AcousticMaterialProperties *props);

// The event of one strike on bank slot `slot` at excitable point `excitable_index`, along unit direction `dir`, with
// impulse `force` (N s) and approach speed `contact_speed` (m/s).  Without dynamics or a material the contact is a short
// default one with no click (tau = 1e-4 s), as in the reference.  `striker` is the mallet of a manual strike.
ModalEvent MakeStrikeEvent(const ModalBank &, uint32_t slot, uint32_t excitable_index, vec3 dir, float force, float contact_speed, const StrikeContext &,
                           const std::optional<PhysicsStrike> &physics = std::nullopt, const Striker &striker = {});
// Looks the object up, checks the excitable point, builds the event and queues it.  False when the entity has no slot
// or the point is out of range (the reference returns silently).
bool TriggerModalStrike(ModalAudio &, entt::entity, const ModalModes &, uint32_t excitable_index, vec3 dir, float force, float contact_speed, const StrikeContext &,
                        const std::optional<PhysicsStrike> &physics = std::nullopt, const Striker &striker = {});

// ---- the physics event source ---------------------------------------------------------------------------------------------
// One discrete impact on one rigid body, as the physics step reports it: the fields of the reference's ContactImpact
// (src/physics/PhysicsContact.h:16-40) that the audio side reads, collider nodes already resolved to the sounding body.
// World space at the impact frame; one event per contact point per body of a pair.
struct ContactImpact {
    entt::entity Entity{}, Other{}; // the struck body, the body that struck it
    vec3 Point{0.f};                // contact point
    vec3 ResultantPoint{0.f};       // load-weighted centre of the point's manifold (where the collision's duration is decided)
    vec3 Direction{0.f};            // unit impulse direction into the struck body
    float Impulse{0.f};             // kg m / s, in excess of the settled support
    float Speed{0.f};               // normal approach speed the strike arrested, m / s
    float OtherInvMass{0.f};        // 1 / kg of the other body; 0: immovable
    float NominalArea{0.f};         // m^2 of the manifold's contact polygon; 0 where the patch grows with load
};
// The floors of the reference's ModalSoundControls (src/audio/AudioTypes.h:25-35): a collision sounds only when its
// approach speed and the modal excitation it produces clear them.
struct ContactFloors {
    float MinContactExcitation{1e-7f}, MinContactSpeed{0.01f};
};
// What the reference looks up in its registry for the two bodies of a contact (AudioSystem.cpp:1007-1037); the caller
// supplies the look-ups, any of which may be left empty (the stated default then applies).
struct StrikeScene {
    std::function<const ModalModes *(entt::entity)> ModesOf;                    // null: the body is not a modal sounding object
    std::function<vec3(entt::entity, vec3)> LocalPoint, LocalDirection;           // world -> the frame the modes are defined in (default: identity)
    std::function<StrikeContext(entt::entity, vec3 world_point)> StruckBody;      // dynamics, material, curvature, volume, scale of the struck body
    std::function<AcousticMaterialProperties(entt::entity)> MaterialOf;           // of the striking body (default: ceramic)
    std::function<std::optional<double>(entt::entity, vec3 world_point)> CurvatureAt; // of the striking body; none: a solid sphere of its mass
    std::function<float(entt::entity)> RoughnessOf;                               // RMS roughness of a body's surface, m (default 0)
};
// Last step's collisions strike the objects they hit, once per contact point: the loop of AudioSystem.cpp:1007-1037 over an
// explicit scene.  Returns the number of strikes queued.
uint32_t StrikeContacts(ModalAudio &, std::span<const ContactImpact>, const StrikeScene &, const ContactFloors & = {});
