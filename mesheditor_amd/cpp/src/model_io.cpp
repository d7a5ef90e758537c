// glTF KHR_audio_rigid_bodies modal models and the `.modal` store (see modal/model_io.hpp).  Self-contained: a small
// JSON reader/writer and base64 codec stand in for the reference's fastgltf (absent from this build).
#include "modal/model_io.hpp"

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstring>
#include <fstream>
#include <numbers>
#include <variant>

namespace {
namespace fs = std::filesystem;
constexpr double Ln1000 = 3 * std::numbers::ln10;

// ---- JSON ----------------------------------------------------------------------------------------------------------
struct Json;
using JsonObject = std::vector<std::pair<std::string, Json>>;
using JsonArray = std::vector<Json>;
struct Json {
    std::variant<std::monostate, bool, double, std::string, JsonArray, JsonObject> v;
    const Json *get(std::string_view key) const {
        if (const auto *o = std::get_if<JsonObject>(&v))
            for (const auto &[k, val] : *o)
                if (k == key) return &val;
        return nullptr;
    }
    const JsonArray *array() const { return std::get_if<JsonArray>(&v); }
    std::optional<double> number() const {
        if (const auto *d = std::get_if<double>(&v)) return *d;
        return std::nullopt;
    }
    std::optional<std::string> string() const {
        if (const auto *s = std::get_if<std::string>(&v)) return *s;
        return std::nullopt;
    }
};

struct JsonParser {
    std::string_view s;
    size_t i{0};
    bool ok{true};
    void ws() {
        while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) ++i;
    }
    bool eat(char c) {
        ws();
        if (i < s.size() && s[i] == c) {
            ++i;
            return true;
        }
        return false;
    }
    static void utf8(std::string &out, uint32_t cp) {
        if (cp < 0x80) out += char(cp);
        else if (cp < 0x800) { out += char(0xc0 | (cp >> 6)); out += char(0x80 | (cp & 0x3f)); }
        else if (cp < 0x10000) { out += char(0xe0 | (cp >> 12)); out += char(0x80 | ((cp >> 6) & 0x3f)); out += char(0x80 | (cp & 0x3f)); }
        else { out += char(0xf0 | (cp >> 18)); out += char(0x80 | ((cp >> 12) & 0x3f)); out += char(0x80 | ((cp >> 6) & 0x3f)); out += char(0x80 | (cp & 0x3f)); }
    }
    std::string str() {
        std::string out;
        while (i < s.size() && s[i] != '"') {
            char c = s[i++];
            if (c != '\\') { out += c; continue; }
            if (i >= s.size()) { ok = false; break; }
            c = s[i++];
            switch (c) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': {
                    if (i + 4 > s.size()) { ok = false; break; }
                    uint32_t cp = 0;
                    std::from_chars(s.data() + i, s.data() + i + 4, cp, 16);
                    i += 4;
                    utf8(out, cp);
                    break;
                }
                default: out += c;
            }
        }
        if (i < s.size()) ++i; else ok = false;
        return out;
    }
    Json value(int depth = 0) {
        Json j;
        ws();
        if (i >= s.size() || depth > 64) { ok = false; return j; }
        const char c = s[i];
        if (c == '{') {
            ++i;
            JsonObject o;
            if (!eat('}')) {
                do {
                    ws();
                    if (i >= s.size() || s[i] != '"') { ok = false; break; }
                    ++i;
                    auto key = str();
                    if (!eat(':')) { ok = false; break; }
                    o.emplace_back(std::move(key), value(depth + 1));
                } while (ok && eat(','));
                if (!eat('}')) ok = false;
            }
            j.v = std::move(o);
        } else if (c == '[') {
            ++i;
            JsonArray a;
            if (!eat(']')) {
                do a.push_back(value(depth + 1));
                while (ok && eat(','));
                if (!eat(']')) ok = false;
            }
            j.v = std::move(a);
        } else if (c == '"') {
            ++i;
            j.v = str();
        } else if (s.compare(i, 4, "true") == 0) { i += 4; j.v = true; }
        else if (s.compare(i, 5, "false") == 0) { i += 5; j.v = false; }
        else if (s.compare(i, 4, "null") == 0) { i += 4; }
        else {
            double d = 0;
            const auto r = std::from_chars(s.data() + i, s.data() + s.size(), d);
            if (r.ec != std::errc{}) { ok = false; return j; }
            i = size_t(r.ptr - s.data());
            j.v = d;
        }
        return j;
    }
};

// ---- base64 --------------------------------------------------------------------------------------------------------
std::vector<std::byte> Base64Decode(std::string_view in) {
    static const auto table = [] {
        std::array<int8_t, 256> t;
        t.fill(-1);
        const char *abc = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int k = 0; k < 64; ++k) t[uint8_t(abc[k])] = int8_t(k);
        return t;
    }();
    std::vector<std::byte> out;
    out.reserve(in.size() * 3 / 4);
    uint32_t acc = 0;
    int bits = 0;
    for (const char c : in) {
        const int v = table[uint8_t(c)];
        if (v < 0) continue; // padding, whitespace
        acc = (acc << 6) | uint32_t(v);
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out.push_back(std::byte((acc >> bits) & 0xff));
        }
    }
    return out;
}
std::string Base64Encode(const std::vector<std::byte> &in) {
    static const char *abc = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    std::string out;
    out.reserve((in.size() + 2) / 3 * 4);
    for (size_t k = 0; k < in.size(); k += 3) {
        const uint32_t b0 = uint8_t(in[k]), b1 = k + 1 < in.size() ? uint8_t(in[k + 1]) : 0, b2 = k + 2 < in.size() ? uint8_t(in[k + 2]) : 0;
        const uint32_t w = (b0 << 16) | (b1 << 8) | b2;
        out += abc[(w >> 18) & 63];
        out += abc[(w >> 12) & 63];
        out += k + 1 < in.size() ? abc[(w >> 6) & 63] : '=';
        out += k + 2 < in.size() ? abc[w & 63] : '=';
    }
    return out;
}

// ---- accessors -----------------------------------------------------------------------------------------------------
struct Document {
    const Json &root;
    std::vector<std::vector<std::byte>> buffers;
    size_t accessor_count() const {
        const auto *a = root.get("accessors");
        return a && a->array() ? a->array()->size() : 0;
    }
    // Reads `components` numbers per element as double; empty on any structural problem.
    std::vector<double> read(size_t index, int components) const {
        const auto *accs = root.get("accessors"), *views = root.get("bufferViews");
        if (!accs || !accs->array() || index >= accs->array()->size() || !views || !views->array()) return {};
        const Json &acc = (*accs->array())[index];
        const auto type = acc.get("type") ? acc.get("type")->string() : std::nullopt;
        const auto count = acc.get("count") ? acc.get("count")->number() : std::nullopt;
        const auto ctype = acc.get("componentType") ? acc.get("componentType")->number() : std::nullopt;
        const auto view_i = acc.get("bufferView") ? acc.get("bufferView")->number() : std::nullopt;
        if (!type || !count || !ctype || !view_i || *view_i < 0 || size_t(*view_i) >= views->array()->size()) return {};
        if ((components == 1 && *type != "SCALAR") || (components == 3 && *type != "VEC3")) return {};
        const Json &view = (*views->array())[size_t(*view_i)];
        const auto buf_i = view.get("buffer") ? view.get("buffer")->number() : std::nullopt;
        if (!buf_i || *buf_i < 0 || size_t(*buf_i) >= buffers.size()) return {};
        const auto num = [](const Json *j) { return j && j->number() ? size_t(*j->number()) : size_t(0); };
        size_t csize = 0;
        switch (int(*ctype)) {
            case 5126: case 5125: csize = 4; break;
            case 5123: csize = 2; break;
            case 5121: csize = 1; break;
            default: return {};
        }
        const size_t offset = num(view.get("byteOffset")) + num(acc.get("byteOffset"));
        size_t stride = num(view.get("byteStride"));
        if (stride == 0) stride = csize * size_t(components);
        const auto &bytes = buffers[size_t(*buf_i)];
        const size_t n = size_t(*count);
        if (n == 0 || offset + (n - 1) * stride + csize * size_t(components) > bytes.size()) return {};
        std::vector<double> out(n * size_t(components));
        for (size_t e = 0; e < n; ++e)
            for (int c = 0; c < components; ++c) {
                const std::byte *p = bytes.data() + offset + e * stride + size_t(c) * csize;
                double v = 0;
                if (int(*ctype) == 5126) { float f; std::memcpy(&f, p, 4); v = f; }
                else if (int(*ctype) == 5125) { uint32_t u; std::memcpy(&u, p, 4); v = u; }
                else if (int(*ctype) == 5123) { uint16_t u; std::memcpy(&u, p, 2); v = u; }
                else v = double(uint8_t(*p));
                out[e * size_t(components) + size_t(c)] = v;
            }
        return out;
    }
};

std::optional<size_t> AccessorIndex(const Json &model, std::string_view key, size_t accessors) {
    const auto *j = model.get(key);
    if (!j || !j->number()) return std::nullopt;
    const double d = *j->number();
    if (d < 0 || d >= double(accessors) || d != std::floor(d)) return std::nullopt;
    return size_t(d);
}

std::optional<std::vector<std::byte>> ReadFile(const fs::path &p) {
    std::ifstream in{p, std::ios::binary};
    if (!in) return std::nullopt;
    std::vector<char> raw{std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>()};
    std::vector<std::byte> out(raw.size());
    std::memcpy(out.data(), raw.data(), raw.size());
    return out;
}

void AppendNumber(std::string &out, double v, bool single_precision) {
    char buf[40];
    std::snprintf(buf, sizeof(buf), single_precision ? "%.9g" : "%.17g", v);
    out += buf;
}
void AppendString(std::string &out, std::string_view s) {
    out += '"';
    for (const char c : s) {
        if (c == '"' || c == '\\') { out += '\\'; out += c; }
        else if (c == '\n') out += "\\n";
        else if (uint8_t(c) < 0x20) { char b[8]; std::snprintf(b, sizeof(b), "\\u%04x", c); out += b; }
        else out += c;
    }
    out += '"';
}
} // namespace

namespace modal::io {
std::optional<ModalModelDocument> ReadGltfModalModels(std::string_view gltf_json, const fs::path &base_dir) {
    JsonParser parser{gltf_json};
    const Json root = parser.value();
    if (!parser.ok || !std::holds_alternative<JsonObject>(root.v)) return std::nullopt;
    ModalModelDocument out;
    const Json *exts = root.get("extensions");
    const Json *ext = exts ? exts->get("KHR_audio_rigid_bodies") : nullptr;
    if (!ext) return out;

    Document doc{root, {}};
    if (const auto *bufs = root.get("buffers"); bufs && bufs->array()) {
        for (const Json &b : *bufs->array()) {
            std::vector<std::byte> bytes;
            if (const auto uri = b.get("uri") ? b.get("uri")->string() : std::nullopt) {
                if (uri->rfind("data:", 0) == 0) {
                    const auto comma = uri->find(',');
                    if (comma != std::string::npos) bytes = Base64Decode(std::string_view{*uri}.substr(comma + 1));
                } else if (auto file = ReadFile(base_dir / *uri)) bytes = std::move(*file);
            }
            doc.buffers.push_back(std::move(bytes));
        }
    }

    // materials: an out-of-range value reads back as the default with a warning (GltfScene.cpp:2430-2452)
    constexpr AcousticMaterialProperties Defaults{2700, 7.2e10, 0.19, 5, 2e-8};
    if (const auto *mats = ext->get("acousticMaterials"); mats && mats->array()) {
        for (const Json &m : *mats->array()) {
            const std::string name = m.get("name") && m.get("name")->string() ? *m.get("name")->string() : std::string{};
            const auto pick = [&](std::string_view key, double fallback, auto &&valid) {
                const auto *j = m.get(key);
                if (!j || !j->number()) return fallback;
                if (std::isfinite(*j->number()) && valid(*j->number())) return *j->number();
                out.Warnings.push_back("acoustic material '" + name + "': " + std::string{key} + " out of range; using the default");
                return fallback;
            };
            const auto positive = [](double v) { return v > 0; };
            const auto non_negative = [](double v) { return v >= 0; };
            out.Materials.push_back({name,
                                     {pick("density", Defaults.Density, positive), pick("youngsModulus", Defaults.YoungModulus, positive),
                                      pick("poissonRatio", Defaults.PoissonRatio, [](double v) { return v > -1 && v < 0.5; }), pick("alpha", Defaults.Alpha, non_negative),
                                      pick("beta", Defaults.Beta, non_negative)}});
        }
    }

    const size_t accessors = doc.accessor_count();
    const auto all_finite = [](const std::vector<double> &v) { return std::all_of(v.begin(), v.end(), [](double x) { return std::isfinite(x); }); };
    if (const auto *models = ext->get("modalModels"); models && models->array()) {
        for (const Json &m : *models->array()) {
            ModalModelRecord rec;
            rec.Name = m.get("name") && m.get("name")->string() ? *m.get("name")->string() : std::string{};
            const auto read_model = [&]() -> ModalModes {
                const auto freqs_i = AccessorIndex(m, "frequencies", accessors), decays_i = AccessorIndex(m, "decayRates", accessors);
                const auto positions_i = AccessorIndex(m, "positions", accessors), shapes_i = AccessorIndex(m, "shapes", accessors);
                if (!freqs_i || !decays_i || !positions_i || !shapes_i) return {};
                const auto freqs = doc.read(*freqs_i, 1), decays = doc.read(*decays_i, 1), positions = doc.read(*positions_i, 3), shapes = doc.read(*shapes_i, 3);
                const size_t n_modes = freqs.size(), n_points = positions.size() / 3;
                if (n_modes == 0 || n_points == 0 || decays.size() != n_modes || shapes.size() != 3 * n_modes * n_points) return {};
                if (!all_finite(freqs) || !all_finite(decays) || !all_finite(positions) || !all_finite(shapes)) return {};
                if (std::any_of(freqs.begin(), freqs.end(), [](double f) { return f <= 0; })) return {};
                if (std::any_of(decays.begin(), decays.end(), [](double d) { return d < 0; })) return {};
                ModalModes modes;
                modes.Freqs.assign(freqs.begin(), freqs.end());
                modes.T60s.resize(n_modes);
                for (size_t k = 0; k < n_modes; ++k) modes.T60s[k] = decays[k] > 0 ? float(Ln1000 / double(float(decays[k]))) : 0.f;
                for (size_t p = 0; p < n_points; ++p) modes.Positions.push_back({float(positions[3 * p]), float(positions[3 * p + 1]), float(positions[3 * p + 2])});
                // mode-major on the wire (element mode*P + point), position-major in memory
                modes.Shapes.assign(n_points, std::vector<vec3>(n_modes));
                for (size_t mode = 0; mode < n_modes; ++mode)
                    for (size_t p = 0; p < n_points; ++p) {
                        const double *s = shapes.data() + 3 * (mode * n_points + p);
                        modes.Shapes[p][mode] = {float(s[0]), float(s[1]), float(s[2])};
                    }
                if (const auto indices_i = AccessorIndex(m, "indices", accessors)) {
                    const auto tris = doc.read(*indices_i, 1);
                    if (!tris.empty() && tris.size() % 3 == 0 && std::all_of(tris.begin(), tris.end(), [&](double t) { return t >= 0 && t < double(n_points); }))
                        modes.Indices.assign(tris.begin(), tris.end());
                    else out.Warnings.push_back("modal model '" + rec.Name + "': sample surface indices outside its sample points; ignoring them");
                }
                modes.OriginalFundamentalFreq = modes.Freqs.front();
                return modes;
            };
            rec.Modes = read_model();
            if (rec.Modes.Freqs.empty())
                out.Warnings.push_back("modal model '" + rec.Name + "': accessors do not match, or a frequency at or below zero, or a negative decay rate; ignoring it");
            if (const auto *mat = m.get("material"); mat && mat->number() && *mat->number() >= 0 && size_t(*mat->number()) < out.Materials.size())
                rec.Material = uint32_t(*mat->number());
            if (const auto *mp = m.get("massProperties"); mp && mp->get("mass") && mp->get("mass")->number()) {
                MassProperties mass;
                mass.Mass = *mp->get("mass")->number();
                const auto vec = [&](std::string_view key, size_t n, float *dst) {
                    const auto *j = mp->get(key);
                    if (!j || !j->array() || j->array()->size() != n) return false;
                    for (size_t k = 0; k < n; ++k) dst[k] = float((*j->array())[k].number().value_or(0.0));
                    return true;
                };
                float c[3], d[3], q[4];
                if (vec("centerOfMass", 3, c)) mass.CenterOfMass = {c[0], c[1], c[2]};
                if (vec("inertiaDiagonal", 3, d)) mass.InertiaDiagonal = {d[0], d[1], d[2]};
                if (vec("inertiaOrientation", 4, q)) mass.InertiaOrientation = {q[3], q[0], q[1], q[2]}; // x,y,z,w on the wire
                rec.Mass = mass;
            }
            out.Models.push_back(std::move(rec));
        }
    }
    return out;
}

std::string WriteGltfModalModels(const ModalModelDocument &doc) {
    std::vector<std::byte> blob;
    std::string accessors, views;
    size_t n_accessors = 0;
    const auto add = [&](const void *data, size_t count, int components, bool is_index) {
        while (blob.size() % 4) blob.push_back(std::byte{0});
        const size_t offset = blob.size(), bytes = count * size_t(components) * 4;
        blob.resize(offset + bytes);
        std::memcpy(blob.data() + offset, data, bytes);
        if (n_accessors) { accessors += ","; views += ","; }
        views += "{\"buffer\":0,\"byteOffset\":" + std::to_string(offset) + ",\"byteLength\":" + std::to_string(bytes) + "}";
        accessors += "{\"bufferView\":" + std::to_string(n_accessors) + ",\"componentType\":" + (is_index ? "5125" : "5126") + ",\"count\":" + std::to_string(count) +
                     ",\"type\":\"" + (components == 3 ? "VEC3" : "SCALAR") + "\"}";
        return n_accessors++;
    };
    std::string models;
    bool first = true;
    for (const auto &rec : doc.Models) {
        const auto &modes = rec.Modes;
        const size_t n_modes = modes.Freqs.size(), n_points = modes.Positions.size();
        if (n_modes == 0 || n_points == 0 || modes.T60s.size() != n_modes || modes.Shapes.size() != n_points) continue;
        std::vector<float> decay(n_modes);
        for (size_t k = 0; k < n_modes; ++k) decay[k] = modes.T60s[k] > 0 ? float(Ln1000 / modes.T60s[k]) : 0.f;
        std::vector<vec3> shapes(n_modes * n_points);
        for (size_t k = 0; k < n_modes; ++k)
            for (size_t p = 0; p < n_points; ++p) shapes[k * n_points + p] = modes.Shapes[p][k];
        const size_t f = add(modes.Freqs.data(), n_modes, 1, false), d = add(decay.data(), n_modes, 1, false);
        const size_t p = add(modes.Positions.data(), n_points, 3, false), s = add(shapes.data(), shapes.size(), 3, false);
        if (!first) models += ",";
        first = false;
        models += "{\"name\":";
        AppendString(models, rec.Name);
        models += ",\"frequencies\":" + std::to_string(f) + ",\"decayRates\":" + std::to_string(d) + ",\"positions\":" + std::to_string(p) + ",\"shapes\":" + std::to_string(s);
        if (!modes.Indices.empty()) models += ",\"indices\":" + std::to_string(add(modes.Indices.data(), modes.Indices.size(), 1, true));
        if (rec.Material && *rec.Material < doc.Materials.size()) models += ",\"material\":" + std::to_string(*rec.Material);
        if (rec.Mass) {
            const auto &mp = *rec.Mass;
            models += ",\"massProperties\":{\"mass\":";
            AppendNumber(models, mp.Mass, false);
            const auto vec = [&](const char *key, std::initializer_list<float> v) {
                models += std::string{",\""} + key + "\":[";
                bool f0 = true;
                for (const float x : v) {
                    if (!f0) models += ",";
                    f0 = false;
                    AppendNumber(models, x, true);
                }
                models += "]";
            };
            vec("centerOfMass", {mp.CenterOfMass.x, mp.CenterOfMass.y, mp.CenterOfMass.z});
            vec("inertiaDiagonal", {mp.InertiaDiagonal.x, mp.InertiaDiagonal.y, mp.InertiaDiagonal.z});
            vec("inertiaOrientation", {mp.InertiaOrientation.x, mp.InertiaOrientation.y, mp.InertiaOrientation.z, mp.InertiaOrientation.w});
            models += "}";
        }
        models += "}";
    }
    std::string materials;
    for (size_t k = 0; k < doc.Materials.size(); ++k) {
        const auto &m = doc.Materials[k];
        if (k) materials += ",";
        materials += "{\"name\":";
        AppendString(materials, m.Name);
        const auto field = [&](const char *key, double v) {
            materials += std::string{",\""} + key + "\":";
            AppendNumber(materials, v, false);
        };
        field("density", m.Properties.Density);
        field("youngsModulus", m.Properties.YoungModulus);
        field("poissonRatio", m.Properties.PoissonRatio);
        field("alpha", m.Properties.Alpha);
        field("beta", m.Properties.Beta);
        materials += "}";
    }
    std::string out = "{\"asset\":{\"version\":\"2.0\",\"generator\":\"modal-hip\"},\"extensionsUsed\":[\"KHR_audio_rigid_bodies\"],";
    out += "\"buffers\":[{\"byteLength\":" + std::to_string(blob.size()) + ",\"uri\":\"data:application/octet-stream;base64," + Base64Encode(blob) + "\"}],";
    out += "\"bufferViews\":[" + views + "],\"accessors\":[" + accessors + "],";
    out += "\"extensions\":{\"KHR_audio_rigid_bodies\":{\"acousticMaterials\":[" + materials + "],\"modalModels\":[" + models + "]}}}";
    return out;
}
} // namespace modal::io

// ---- `.modal` store ------------------------------------------------------------------------------------------------
namespace {
struct Writer {
    std::vector<std::byte> bytes;
    template<typename T> void pod(const T &v) {
        const auto *p = reinterpret_cast<const std::byte *>(&v);
        bytes.insert(bytes.end(), p, p + sizeof(T));
    }
    void vec(const vec3 &v) { pod(v.x); pod(v.y); pod(v.z); }
    template<typename T, typename F> void seq(const std::vector<T> &v, F &&each) {
        pod(uint32_t(v.size()));
        for (const auto &e : v) each(e);
    }
};
struct Reader {
    const std::vector<std::byte> &bytes;
    size_t at{0};
    bool ok{true};
    template<typename T> T pod() {
        T v{};
        if (at + sizeof(T) > bytes.size()) { ok = false; return v; }
        std::memcpy(&v, bytes.data() + at, sizeof(T));
        at += sizeof(T);
        return v;
    }
    vec3 vec() { const float x = pod<float>(), y = pod<float>(), z = pod<float>(); return {x, y, z}; }
    template<typename T, typename F> std::vector<T> seq(size_t element_bytes, F &&each) {
        const uint32_t n = pod<uint32_t>();
        std::vector<T> v;
        if (!ok || size_t(n) * element_bytes > bytes.size() - at) { ok = false; return v; }
        v.reserve(n);
        for (uint32_t k = 0; k < n && ok; ++k) v.push_back(each());
        return v;
    }
};
} // namespace

std::vector<std::byte> SerializeModalModel(const ModalModelData &d) {
    Writer w;
    const auto floats = [&](const std::vector<float> &v) { w.seq(v, [&](float x) { w.pod(x); }); };
    const auto u32s = [&](const std::vector<uint32_t> &v) { w.seq(v, [&](uint32_t x) { w.pod(x); }); };
    const auto vecs = [&](const std::vector<vec3> &v) { w.seq(v, [&](const vec3 &x) { w.vec(x); }); };
    const auto rows = [&](const std::vector<std::vector<vec3>> &v) { w.seq(v, [&](const std::vector<vec3> &r) { vecs(r); }); };
    // ModalModes
    floats(d.Modes.Freqs); floats(d.Modes.T60s); rows(d.Modes.Shapes); u32s(d.Modes.Vertices); vecs(d.Modes.Positions); u32s(d.Modes.Indices);
    w.pod(d.Modes.OriginalFundamentalFreq); w.vec(d.Modes.BakedScale);
    // MassProperties (quaternion in glm's storage order x, y, z, w)
    w.pod(d.Mass.Mass); w.vec(d.Mass.CenterOfMass); w.vec(d.Mass.InertiaDiagonal);
    w.pod(d.Mass.InertiaOrientation.x); w.pod(d.Mass.InertiaOrientation.y); w.pod(d.Mass.InertiaOrientation.z); w.pod(d.Mass.InertiaOrientation.w);
    // TetMeshData
    vecs(d.Tets.Positions); u32s(d.Tets.EdgeIndices);
    // ModalEigenSummary
    w.seq(d.Summary.Eigenvalues, [&](double x) { w.pod(x); });
    rows(d.Summary.Shapes);
    const auto &m = d.Summary.SolvedMaterial;
    w.pod(m.Density); w.pod(m.YoungModulus); w.pod(m.PoissonRatio); w.pod(m.Alpha); w.pod(m.Beta);
    w.pod(d.Summary.SolvedMinModeFreq); w.pod(d.Summary.SolvedMaxModeFreq); w.pod(d.Summary.SolvedNumModes);
    w.pod(uint64_t(d.Summary.TetInputsHash));
    u32s(d.Summary.SolvedVertices);
    return w.bytes;
}

std::optional<ModalModelData> DeserializeModalModel(const std::vector<std::byte> &bytes) {
    Reader r{bytes};
    const auto floats = [&] { return r.seq<float>(4, [&] { return r.pod<float>(); }); };
    const auto u32s = [&] { return r.seq<uint32_t>(4, [&] { return r.pod<uint32_t>(); }); };
    const auto vecs = [&] { return r.seq<vec3>(12, [&] { return r.vec(); }); };
    const auto rows = [&] { return r.seq<std::vector<vec3>>(4, [&] { return vecs(); }); };
    ModalModelData d;
    d.Modes.Freqs = floats(); d.Modes.T60s = floats(); d.Modes.Shapes = rows(); d.Modes.Vertices = u32s(); d.Modes.Positions = vecs(); d.Modes.Indices = u32s();
    d.Modes.OriginalFundamentalFreq = r.pod<float>(); d.Modes.BakedScale = r.vec();
    d.Mass.Mass = r.pod<double>(); d.Mass.CenterOfMass = r.vec(); d.Mass.InertiaDiagonal = r.vec();
    d.Mass.InertiaOrientation.x = r.pod<float>(); d.Mass.InertiaOrientation.y = r.pod<float>(); d.Mass.InertiaOrientation.z = r.pod<float>();
    d.Mass.InertiaOrientation.w = r.pod<float>();
    d.Tets.Positions = vecs(); d.Tets.EdgeIndices = u32s();
    d.Summary.Eigenvalues = r.seq<double>(8, [&] { return r.pod<double>(); });
    d.Summary.Shapes = rows();
    auto &m = d.Summary.SolvedMaterial;
    m.Density = r.pod<double>(); m.YoungModulus = r.pod<double>(); m.PoissonRatio = r.pod<double>(); m.Alpha = r.pod<double>(); m.Beta = r.pod<double>();
    d.Summary.SolvedMinModeFreq = r.pod<float>(); d.Summary.SolvedMaxModeFreq = r.pod<float>(); d.Summary.SolvedNumModes = r.pod<uint32_t>();
    d.Summary.TetInputsHash = size_t(r.pod<uint64_t>());
    d.Summary.SolvedVertices = u32s();
    if (!r.ok || r.at != bytes.size()) return std::nullopt;
    return d;
}

std::filesystem::path SaveModalModelFile(const std::filesystem::path &dir, const ModalModelData &data) {
    const auto bytes = SerializeModalModel(data);
    if (bytes.empty()) return {};
    std::error_code ec;
    fs::create_directories(dir, ec);
    if (ec) return {};
    uint64_t hash = 0xcbf29ce484222325ull; // FNV-1a: the name must not depend on the standard library's std::hash
    for (const auto b : bytes) hash = (hash ^ uint64_t(uint8_t(b))) * 0x100000001b3ull;
    for (uint32_t suffix = 0;; ++suffix) {
        char name[48];
        if (suffix == 0) std::snprintf(name, sizeof(name), "%016llx.modal", static_cast<unsigned long long>(hash));
        else std::snprintf(name, sizeof(name), "%016llx-%u.modal", static_cast<unsigned long long>(hash), suffix);
        const auto path = dir / name;
        if (fs::exists(path)) {
            if (const auto existing = ReadFile(path); existing && *existing == bytes) return name; // write-once: identical content reuses the file
            continue;
        }
        std::ofstream out{path, std::ios::binary};
        out.write(reinterpret_cast<const char *>(bytes.data()), std::streamsize(bytes.size()));
        return out ? fs::path{name} : fs::path{};
    }
}

std::optional<ModalModelData> LoadModalModelFile(const std::filesystem::path &file) {
    const auto bytes = ReadFile(file);
    if (!bytes || bytes->empty()) return std::nullopt;
    return DeserializeModalModel(*bytes);
}
