// Flat C wrapper over the C++ host mirror, for the Python test-suite (ctypes cannot call C++ directly).
// Every function forwards to the reference-named API in modal/*.hpp.
#include "modal/bank.hpp"
#include "modal/contact.hpp"
#include "modal/solver.hpp"
#include "modal/tets.hpp"

#include "modalhip.h"

#include <cstring>
#include <span>
#include <vector>
#include <type_traits>
#include <exception>
#include <string>

namespace {
thread_local std::string g_error;
ModalModes MakeModes(uint32_t n_modes, uint32_t n_pos, const float *freqs, const float *t60s, const float *shapes, const float *positions, uint32_t n_idx, const uint32_t *idx) {
    ModalModes m;
    if (freqs) m.Freqs.assign(freqs, freqs + n_modes); else m.Freqs.assign(n_modes, 0.f);
    if (t60s) m.T60s.assign(t60s, t60s + n_modes); else m.T60s.assign(n_modes, 0.f);
    m.Shapes.assign(n_pos, std::vector<vec3>(n_modes));
    for (uint32_t p = 0; p < n_pos; ++p)
        for (uint32_t k = 0; k < n_modes; ++k) m.Shapes[p][k] = {shapes[(size_t(p) * n_modes + k) * 3], shapes[(size_t(p) * n_modes + k) * 3 + 1], shapes[(size_t(p) * n_modes + k) * 3 + 2]};
    if (positions)
        for (uint32_t p = 0; p < n_pos; ++p) m.Positions.push_back({positions[3 * p], positions[3 * p + 1], positions[3 * p + 2]});
    if (idx) m.Indices.assign(idx, idx + n_idx);
    return m;
}
} // namespace

// One scene in either precision.  `dbl` picks which pair is live; every entry point dispatches on it.
struct mhx_scene {
    bool dbl{false};
    ModalAudio audio;
    ModalBank next;
    ModalAudio64 audio64;
    ModalBank64 next64;
};

namespace {
// f(audio, bank under construction) in the scene's precision
template<typename F> auto Dispatch(mhx_scene *s, F &&f) { return s->dbl ? f(s->audio64, s->next64) : f(s->audio, s->next); }
template<typename Audio, typename Bank> Bank &Pick(Audio &a, Bank &next, int live) { return live ? static_cast<Bank &>(LiveBank(a)) : next; }
} // namespace

extern "C" {
const char *mhx_last_error() { return g_error.c_str(); }

static mhx_scene *NewScene(float sample_rate, int device, bool dbl) {
    auto *s = new mhx_scene;
    s->dbl = dbl;
    s->next.SampleRate = sample_rate;
    s->next64.SampleRate = sample_rate;
    s->audio.Device = s->audio64.Device = device;
    return s;
}
mhx_scene *mhx_scene_create(float sample_rate, int device) { return NewScene(sample_rate, device, false); }
// The fp64 bank behind the same API (ModalBank64 / ModalAudio64): columns, impacts and output in double.
mhx_scene *mhx_scene_create_f64(float sample_rate, int device) { return NewScene(sample_rate, device, true); }
void mhx_scene_destroy(mhx_scene *s) { delete s; }
int mhx_scene_is_f64(mhx_scene *s) { return s->dbl ? 1 : 0; }
uint32_t mhx_add_object(mhx_scene *s, uint32_t entity, uint32_t n_modes, uint32_t n_pos, const float *shapes, const float *positions, uint32_t n_idx, const uint32_t *idx) {
    const auto modes = MakeModes(n_modes, n_pos, nullptr, nullptr, shapes, positions, n_idx, idx);
    return Dispatch(s, [&](auto &, auto &next) { return AddModalObject(next, entt::entity{entity}, modes); });
}
void mhx_tune_object(mhx_scene *s, int live, uint32_t object, uint32_t n, const float *freqs, const float *t60s, float radius_scale) {
    Dispatch(s, [&](auto &a, auto &next) { TuneModalObject(Pick(a, next, live), object, std::span<const float>(freqs, n), std::span<const float>(t60s, n), radius_scale); return 0; });
}
int mhx_set_shapes(mhx_scene *s, int live, uint32_t object, uint32_t n_modes, uint32_t n_pos, const float *shapes) {
    const auto modes = MakeModes(n_modes, n_pos, nullptr, nullptr, shapes, nullptr, 0, nullptr);
    return Dispatch(s, [&](auto &a, auto &next) { return SetModalObjectShapes(Pick(a, next, live), object, modes) ? 1 : 0; });
}
void mhx_set_gains(mhx_scene *s, int live, uint32_t object, float out_gain, float listener_gain) {
    Dispatch(s, [&](auto &a, auto &next) {
        auto &b = Pick(a, next, live);
        b.OutGain[object] = out_gain;
        b.ListenerGain[object] = listener_gain;
        return 0;
    });
}
int mhx_install(mhx_scene *s) {
    try {
        return Dispatch(s, [&](auto &a, auto &next) {
            InstallModalBank(a, next);
            next = std::remove_reference_t<decltype(next)>{};
            next.SampleRate = LiveBank(a).SampleRate;
            return 0;
        });
    } catch (const std::exception &e) { g_error = e.what(); return 1; }
}
void mhx_set_renderers(mhx_scene *s, uint32_t n) { s->audio.RenderPool.SetSize(n); s->audio64.RenderPool.SetSize(n); }
void mhx_set_click_gain(mhx_scene *s, float g) { s->audio.ClickGain.store(g); s->audio64.ClickGain.store(g); }
void mhx_set_max_impacts(mhx_scene *s, uint32_t n) { s->audio.MaxImpacts.store(n); s->audio64.MaxImpacts.store(n); }
int mhx_enqueue(mhx_scene *s, const ModalEvent *e) {
    return Dispatch(s, [&](auto &a, auto &) {
        const auto before = a.EventsDropped;
        EnqueueModalEvent(a, *e);
        return a.EventsDropped == before ? 1 : 0;
    });
}
// `out`: float samples for an fp32 scene, double samples for an fp64 scene
int mhx_render(mhx_scene *s, void *out, uint32_t frames) {
    try {
        if (s->dbl) RenderModal(s->audio64, static_cast<double *>(out), frames);
        else RenderModal(s->audio, static_cast<float *>(out), frames);
        return 0;
    } catch (const std::exception &e) { g_error = e.what(); return 1; }
}
// Kernel timing of the scene's device context: enable / read one class (modalhip.h MH_KERNEL_*)
int mhx_time_kernels(mhx_scene *s, int enable) {
    try {
        return Dispatch(s, [&](auto &a, auto &) { return mh_context_time_kernels(ModalDeviceContext(a), enable); });
    } catch (const std::exception &e) { g_error = e.what(); return 1; }
}
int mhx_kernel_class_stats(mhx_scene *s, int kernel_class, uint64_t *launches, double *total_ms, double *total_work) {
    try {
        return Dispatch(s, [&](auto &a, auto &) { return mh_context_kernel_class_stats(ModalDeviceContext(a), kernel_class, launches, total_ms, total_work); });
    } catch (const std::exception &e) { g_error = e.what(); return 1; }
}
uint32_t mhx_num_objects(mhx_scene *s) { return Dispatch(s, [](auto &a, auto &) { return uint32_t(LiveBank(a).Entities.size()); }); }
uint32_t mhx_active_impacts(mhx_scene *s) { return Dispatch(s, [](auto &a, auto &) { return a.ActiveImpacts.load(); }); }
double mhx_modal_energy(mhx_scene *s) { return Dispatch(s, [](auto &a, auto &) { return a.ModalEnergy.load(); }); }
float mhx_render_share(mhx_scene *s) { return Dispatch(s, [](auto &a, auto &) { return a.RenderShare.load(); }); }
int mhx_find_object(mhx_scene *s, uint32_t entity) {
    return Dispatch(s, [&](auto &a, auto &) {
        const auto o = FindModalObject(LiveBank(a), entt::entity{entity});
        return o ? int(*o) : -1;
    });
}
// which: as oracle mo_bank_column
uint32_t mhx_column(mhx_scene *s, int live, int which, double *out) {
    if (which < 0 || which >= 18) return 0;
    return Dispatch(s, [&](auto &a, auto &next) {
        if (live && (which == 2 || which == 3)) SyncModalState(a);
        const auto &b = Pick(a, next, live);
        using Col = std::remove_reference_t<decltype(b.CoeffRe)>;
        const Col *cols[] = {&b.CoeffRe, &b.CoeffIm, &b.StateRe, &b.StateIm, &b.RadiationGain, &b.RadiationArea, &b.DeflectionGain, &b.OutPhaseIm, &b.OutPhaseRe,
                             &b.QuadCompliance, &b.QuadDriveScale, &b.ShapeX, &b.ShapeY, &b.ShapeZ, &b.OutGain, &b.ListenerGain, &b.RadiantRadius, &b.DeflectionScale};
        const auto &c = *cols[which];
        if (out) for (size_t i = 0; i < c.size(); ++i) out[i] = double(c[i]);
        return uint32_t(c.size());
    });
}
void mhx_object_state(mhx_scene *s, uint32_t *tuned, uint32_t *live, uint8_t *ringing) {
    Dispatch(s, [&](auto &a, auto &) {
        const auto &b = LiveBank(a);
        for (size_t o = 0; o < b.Entities.size(); ++o) {
            if (tuned) tuned[o] = b.TunedModeCount[o];
            if (live) live[o] = b.LiveModeCount[o];
            if (ringing) ringing[o] = b.Ringing[o];
        }
        return 0;
    });
}
void mhx_recoil_click_filter(double radius, double volume, double mass, double sample_rate, float out3[3]) {
    const auto f = RecoilClickFilter(radius, volume, mass, sample_rate);
    out3[0] = f.B0; out3[1] = f.A1; out3[2] = f.A2;
}
void mhx_recoil_object_filter(double radius, double volume, double sample_rate, float out6[6]) {
    const auto f = RecoilObjectFilter(radius, volume, sample_rate);
    out6[0] = f.RadB0; out6[1] = f.AirB0; out6[2] = f.AirB1; out6[3] = f.AirB2; out6[4] = f.A1; out6[5] = f.A2;
}
double mhx_estimate_contact_time(double mass, const float inv_inertia9[9], const float arm[3], const float dir[3], double speed, const double object_mat[5],
                                 double object_curvature, double area, const double impactor_mat[5], double impactor_curvature, double impactor_inv_mass,
                                 double scale, double roughness) {
    ContactDynamics d;
    d.Mass = mass;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) d.InverseInertia[c][r] = inv_inertia9[c * 3 + r];
    d.ContactArm = {vec3{arm[0], arm[1], arm[2]}};
    const AcousticMaterialProperties om{object_mat[0], object_mat[1], object_mat[2], object_mat[3], object_mat[4]};
    const Impactor imp{{impactor_mat[0], impactor_mat[1], impactor_mat[2], impactor_mat[3], impactor_mat[4]}, impactor_curvature, impactor_inv_mass};
    return EstimateContactTime(d, 0, vec3{dir[0], dir[1], dir[2]}, speed, om, object_curvature, area, imp, scale, roughness);
}
double mhx_striker_mass(double density, float tip_radius, float length) {
    Striker s;
    s.Material.Properties.Density = density;
    s.TipRadius = tip_radius;
    s.Length = length;
    return StrikerMass(s);
}
void mhx_inverse_inertia_tensor(const float diag[3], const float q[4], float out9[9]) {
    MassProperties mp;
    mp.InertiaDiagonal = {diag[0], diag[1], diag[2]};
    mp.InertiaOrientation = {q[0], q[1], q[2], q[3]};
    const auto m = InverseInertiaTensor(mp);
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) out9[c * 3 + r] = m[c][r];
}
double mhx_saturation_penetration(double curvature, double area) { return SaturationPenetration(curvature, area); }
double mhx_punch_stiffness(double inv_modulus, double area) { return PunchStiffness(inv_modulus, area); }
// ---- tet-generation front end (modal/tets.hpp): surface in, tet mesh out; host code only -----------------------------------
struct mhx_tets {
    tetra::Result Result;
    std::string Error;
};
mhx_tets *mhx_tetrahedralize2(const double *points, uint32_t n_points, const uint32_t *triangles, uint32_t n_triangles, uint64_t max_steiner, int flags, double max_volume) {
    auto *h = new mhx_tets;
    try {
        std::vector<dvec3> pts(n_points);
        for (uint32_t i = 0; i < n_points; ++i) pts[i] = {points[3 * size_t(i)], points[3 * size_t(i) + 1], points[3 * size_t(i) + 2]};
        tetra::Options options;
        options.MaxSteinerPoints = size_t(max_steiner);
        options.InteriorSteiner = (flags & 1) != 0; // bit 0: points moved off the surface, bit 1: sliver repair
        options.RepairSlivers = (flags & 2) != 0;
        options.InteriorShell = (flags & 4) ? tetra::Options::Shell::Never : (flags & 8) ? tetra::Options::Shell::Always : tetra::Options::Shell::WhenFlat; // bits 2, 3
        options.Quality = (flags & 16) != 0; // bit 4: the reference's Options::Quality; max_volume: its Options::MaxVolume
        options.MaxVolume = max_volume;
        options.BreakFlatCells = (flags & 32) == 0; // bit 5: leave flat cells as the other repairs leave them (tests)
        auto filled = tetra::Tetrahedralize(pts, std::span<const uint32_t>(triangles, size_t(n_triangles) * 3), options);
        if (filled) h->Result = std::move(*filled);
        else h->Error = filled.error();
    } catch (const std::exception &e) { h->Error = e.what(); }
    return h;
}
mhx_tets *mhx_tetrahedralize(const double *points, uint32_t n_points, const uint32_t *triangles, uint32_t n_triangles, uint64_t max_steiner, int interior_steiner) {
    return mhx_tetrahedralize2(points, n_points, triangles, n_triangles, max_steiner, interior_steiner, 0.0);
}
const char *mhx_tets_error(const mhx_tets *h) { return h->Error.c_str(); }
uint32_t mhx_tets_num_points(const mhx_tets *h) { return uint32_t(h->Result.Mesh.Points.size()); }
uint32_t mhx_tets_num_tets(const mhx_tets *h) { return uint32_t(h->Result.Mesh.Tets.size()); }
uint32_t mhx_tets_boundary_steiner(const mhx_tets *h) { return h->Result.Profile.BdrySteinerCount; }
// tetra::Profile as 16 doubles: the seven stage times, then the counters in declaration order (TetCount ... Builds), then the interior points by origin
void mhx_tets_profile(const mhx_tets *h, double *out) {
    const auto &p = h->Result.Profile;
    const double v[] = {p.DelaunaySeconds, p.RecoverSeconds, p.CarveSeconds, p.RefineSeconds, p.SegmentSeconds, p.FaceSeconds, p.SuppressSeconds, double(p.TetCount), double(p.SteinerCount), double(p.DelaunayTetCount), double(p.BdrySteinerCount), double(p.VolSteinerCount), double(p.FlipCount), double(p.SplitCount), double(p.MissingEdgeCount), double(p.MissingFaceCount), double(p.Builds), double(p.ShellPointCount), double(p.QualityPointCount), double(p.FlatCellPointCount), double(p.SliverExchangeCount)};
    for (size_t i = 0; i < sizeof(v) / sizeof(v[0]); ++i) out[i] = v[i];
}
void mhx_tets_copy(const mhx_tets *h, double *points, uint32_t *tets) {
    for (size_t i = 0; i < h->Result.Mesh.Points.size(); ++i)
        for (int k = 0; k < 3; ++k) points[3 * i + k] = h->Result.Mesh.Points[i][k];
    for (size_t t = 0; t < h->Result.Mesh.Tets.size(); ++t)
        for (int k = 0; k < 4; ++k) tets[4 * t + k] = h->Result.Mesh.Tets[t][k];
}
void mhx_tets_free(mhx_tets *h) { delete h; }
// SimplifySurface (modal/tets.hpp): positions and triangles in, the coarsened surface out (counts through the pointers; the
// arrays are overwritten in place, their leading parts valid).
void mhx_simplify_surface(float *positions, uint32_t *n_positions, uint32_t *triangles, uint32_t *n_triangles, float ratio) {
    std::vector<vec3> pos(*n_positions);
    for (uint32_t i = 0; i < *n_positions; ++i) pos[i] = {positions[3 * size_t(i)], positions[3 * size_t(i) + 1], positions[3 * size_t(i) + 2]};
    std::vector<uint32_t> tri(triangles, triangles + size_t(*n_triangles) * 3);
    SimplifySurface(pos, tri, ratio);
    for (size_t i = 0; i < pos.size(); ++i)
        for (int k = 0; k < 3; ++k) positions[3 * i + k] = pos[i][k];
    std::copy(tri.begin(), tri.end(), triangles);
    *n_positions = uint32_t(pos.size());
    *n_triangles = uint32_t(tri.size() / 3);
}
}
