// The surface-contact model's entry points as the modal core sees them (reference: src/audio/SurfaceContact.h:22-73).
// The model itself (src/audio/surface/, an optional build flag in the reference, off by default) is out of scope; this
// build always links the "absent" implementation (cpp/src/surface_absent.cpp, reference SurfaceContactAbsent.cpp:6-25):
// every hook does nothing and reports nothing, so objects sound by collision alone.  The declarations are the
// reference's, so code that includes <audio/SurfaceContact.h> and calls them compiles and links unchanged.
#pragma once
#include "bank.hpp"

#include <entt/entity/fwd.hpp>

#include <cstdint>
#include <memory>
#include <span>

struct ModalRenderScratch; // a renderer's scratch in the reference; opaque here (the device owns all render scratch)

// Owning handles of the model's two opaque objects (defined by the model, which this build does not have).
struct SurfaceAudioState;
struct SurfaceAudioStateDelete {
    void operator()(SurfaceAudioState *state) const;
};
struct SurfaceRenderScratch;
struct SurfaceRenderScratchDelete {
    void operator()(SurfaceRenderScratch *scratch) const;
};
using SurfaceAudioStatePtr = std::unique_ptr<SurfaceAudioState, SurfaceAudioStateDelete>;
using SurfaceRenderScratchPtr = std::unique_ptr<SurfaceRenderScratch, SurfaceRenderScratchDelete>;

SurfaceAudioStatePtr MakeSurfaceAudioState(); // null without the model

// ---- called from RenderModal (the reference's audio thread)
void SurfaceAdoptVoices(ModalAudio &audio, ModalBank &bank, uint32_t frame_count);
uint32_t SurfaceVoiceCount(const ModalAudio &audio, uint32_t object);
bool SurfaceRenderObject(ModalAudio &audio, ModalRenderScratch &scratch, ModalBank &bank, uint32_t object, std::span<const uint32_t> impacts, float *out,
                         uint32_t frame_count);
void SurfaceSilenceObject(ModalAudio &audio, uint32_t object);
uint32_t SurfaceActiveVoices(const ModalAudio &audio);

// ---- called from the scene update (the reference's main thread)
void SurfaceInstallBank(ModalAudio &audio);
void RegisterSurfaceContactHandlers(entt::registry &registry);
void SurfaceUpdateContacts(entt::registry &registry);
float SurfaceRoughnessOf(const entt::registry &registry, entt::entity node);
entt::entity ContactSurfaceNode(const entt::registry &registry, entt::entity collider, entt::entity body);

// ---- called from the editor's panels
void DrawContactSurfaceControls(entt::registry &registry, entt::entity sound_entity);
void DrawSurfaceSynthControls(entt::registry &registry, entt::entity viewport);
void DrawSurfaceContactDebug(const entt::registry &registry);
