// The surface-contact hooks with the model absent (reference behaviour: src/audio/SurfaceContactAbsent.cpp:6-25): no state,
// no voices, no rendering, zero roughness, and a contact reads the body's own surface.  Every hook is a deliberate no-op;
// the arguments are named so that each definition says what it ignores.
#include "modal/surface.hpp"

namespace {
template<typename... Ignored> inline void NotUsed(const Ignored &...) {}
} // namespace

// ---- ownership: there is nothing to own
void SurfaceAudioStateDelete::operator()(SurfaceAudioState *state) const { NotUsed(state); }
void SurfaceRenderScratchDelete::operator()(SurfaceRenderScratch *scratch) const { NotUsed(scratch); }
SurfaceAudioStatePtr MakeSurfaceAudioState() {
    return {}; // a null state: the bank asks SurfaceVoiceCount before it would ever dereference it
}

// ---- render side (the reference's audio thread)
void SurfaceAdoptVoices(ModalAudio &audio, ModalBank &bank, uint32_t frame_count) { NotUsed(audio, bank, frame_count); }
uint32_t SurfaceVoiceCount(const ModalAudio &audio, uint32_t object) {
    NotUsed(audio, object);
    return 0; // no sustained voices: every object takes the collision-only kernel
}
bool SurfaceRenderObject(ModalAudio &audio, ModalRenderScratch &scratch, ModalBank &bank, uint32_t object, std::span<const uint32_t> impacts, float *out,
                         uint32_t frame_count) {
    NotUsed(audio, scratch, bank, object, impacts, out, frame_count);
    return false; // "not rendered here": the caller falls through to its own renderer
}
void SurfaceSilenceObject(ModalAudio &audio, uint32_t object) { NotUsed(audio, object); }
uint32_t SurfaceActiveVoices(const ModalAudio &audio) {
    NotUsed(audio);
    return 0;
}

// ---- scene side (the reference's main thread)
void SurfaceInstallBank(ModalAudio &audio) { NotUsed(audio); }
void RegisterSurfaceContactHandlers(entt::registry &registry) { NotUsed(registry); }
void SurfaceUpdateContacts(entt::registry &registry) { NotUsed(registry); }
float SurfaceRoughnessOf(const entt::registry &registry, entt::entity node) {
    NotUsed(registry, node);
    return 0.0f; // perfectly smooth: the strike translation's pair roughness is then zero as well
}
entt::entity ContactSurfaceNode(const entt::registry &registry, entt::entity collider, entt::entity body) {
    NotUsed(registry, collider);
    return body; // the body speaks for all of its colliders
}

// ---- user interface: nothing to draw
void DrawContactSurfaceControls(entt::registry &registry, entt::entity sound_entity) { NotUsed(registry, sound_entity); }
void DrawSurfaceSynthControls(entt::registry &registry, entt::entity viewport) { NotUsed(registry, viewport); }
void DrawSurfaceContactDebug(const entt::registry &registry) { NotUsed(registry); }
