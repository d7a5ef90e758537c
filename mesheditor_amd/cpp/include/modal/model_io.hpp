// The wire formats either side of the path, host code only:
//  * KHR_audio_rigid_bodies modal models of a glTF 2.0 document <-> ModalModes / MassProperties, with the reference's
//    import rules (src/gltf/GltfScene.cpp:2455-2508: four accessors, one decay rate per mode, a mode-major M*P shape
//    block, every value finite, f > 0, d >= 0, optional whole-triangle sample surface) and its export conventions
//    (:4519-4562: d = ln 1000 / T60 with T60 == 0 the undamped sentinel, shapes mode-major, inertiaOrientation x,y,z,w);
//  * the write-once, content-addressed `.modal` store of solved models (src/audio/ModalModelFile.h:12-31).
#pragma once
#include "types.hpp"

#include <filesystem>
#include <optional>
#include <string>
#include <string_view>
#include <vector>

namespace modal::io {
struct ModalModelRecord {
    std::string Name;
    ModalModes Modes; // empty when the model failed validation: the slot stays so indices line up with the document
    std::optional<MassProperties> Mass;
    std::optional<uint32_t> Material; // index into ModalModelDocument::Materials
};
struct ModalModelDocument {
    std::vector<AcousticMaterial> Materials;
    std::vector<ModalModelRecord> Models;
    std::vector<std::string> Warnings;
};

// Reads the extension's acoustic materials and modal models.  Buffers are base64 data URIs or files relative to
// base_dir.  nullopt when the text is not a JSON object; a document without the extension reads back empty.
std::optional<ModalModelDocument> ReadGltfModalModels(std::string_view gltf_json, const std::filesystem::path &base_dir = {});
// A self-contained glTF 2.0 document (one embedded buffer) carrying the models; invalid (empty) models are skipped.
std::string WriteGltfModalModels(const ModalModelDocument &);
} // namespace modal::io

// ---- `.modal` store ------------------------------------------------------------------------------------------------
struct TetMeshData { // src/mesh/TetMeshData.h:8-13
    std::vector<vec3> Positions;
    std::vector<uint32_t> EdgeIndices;
    bool operator==(const TetMeshData &) const = default;
};
struct ModalModelData { // src/audio/ModalModelFile.h:14-21
    ModalModes Modes;
    MassProperties Mass;
    TetMeshData Tets;
    ModalEigenSummary Summary;
    bool operator==(const ModalModelData &) const = default;
};
// The reference keeps the store under its project directory; here the directory is the caller's.
// Save: serialises, names the file by the 64-bit FNV-1a hash of the bytes ("%016x.modal", "-N" suffix on a collision with
// different content), reuses an existing identical file, and returns the name relative to `dir` (empty on IO failure).
std::filesystem::path SaveModalModelFile(const std::filesystem::path &dir, const ModalModelData &);
std::optional<ModalModelData> LoadModalModelFile(const std::filesystem::path &file);
// The byte image itself (little-endian, members in declaration order, sequences as u32 count + elements, optionals as
// a presence byte + value -- zpp::bits' documented defaults, which the reference serialises with; compatibility with
// files it wrote is unverified: neither the library nor a sample file is available to this build).
std::vector<std::byte> SerializeModalModel(const ModalModelData &);
std::optional<ModalModelData> DeserializeModalModel(const std::vector<std::byte> &);
