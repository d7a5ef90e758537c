// Host-side scalar stages of the path (they run on the calling thread in the reference as well):
//   ComputeMassProperties  src/audio/mesh2modes.cpp:73-126
//   modal::PostprocessModes :515-588, modal::RescaleModes :590-603
#include "../../include/modalhip.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace {
struct D3 {
    double x, y, z;
};
inline D3 sub(const D3 &a, const D3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline double dot(const D3 &a, const D3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline D3 cross(const D3 &a, const D3 &b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }

// Cyclic Jacobi for a symmetric 3x3 (a[r][c]); eigenvalues ascending in w, eigenvectors in the columns of v.
void jacobi3(double a[3][3], double w[3], double v[3][3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) v[i][j] = i == j;
    for (int sweep = 0; sweep < 64; ++sweep) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        const double diag = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (off <= 1e-32 * diag || off == 0) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0) continue;
                const double tau = (a[q][q] - a[p][p]) / (2 * a[p][q]);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (std::abs(tau) + std::sqrt(1 + tau * tau));
                const double c = 1 / std::sqrt(1 + t * t), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int order[3] = {0, 1, 2};
    std::sort(order, order + 3, [&](int x, int y) { return a[x][x] < a[y][y]; });
    double vv[3][3];
    for (int j = 0; j < 3; ++j) {
        w[j] = a[order[j]][order[j]];
        for (int k = 0; k < 3; ++k) vv[k][j] = v[k][order[j]];
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) v[i][j] = vv[i][j];
}

struct Modes {
    std::vector<float> freqs, t60s;
    uint32_t lowest{0};
    float original{0};
    bool empty{true};
};

Modes postprocess(uint32_t n, const double *eigenvalues, const mh_material &mat, const mh_solver_config &cfg) {
    Modes out;
    std::vector<float> f(n), t60(n);
    std::vector<double> omega(n);
    const double lambda_eps = std::pow(2 * M_PI * cfg.min_mode_freq, 2) * 1e-10;
    for (uint32_t k = 0; k < n; ++k) omega[k] = eigenvalues[k] > lambda_eps ? std::sqrt(eigenvalues[k]) : 0;
    const auto damping = [&](double w) { return mat.alpha + mat.beta * (w * w); };
    const auto damped_hz = [](double w, double c) {
        const double wd2 = w * w - 0.25 * c * c;
        return wd2 > 0 ? std::sqrt(wd2) / (2 * M_PI) : 0;
    };
    uint32_t lowest = n;
    float lowest_freq = 0;
    for (uint32_t k = 0; k < n; ++k) {
        if (omega[k] <= 0) {
            f[k] = t60[k] = 0.f;
            continue;
        }
        f[k] = float(damped_hz(omega[k], damping(omega[k])));
        if (lowest == n && f[k] >= cfg.min_mode_freq) {
            lowest = k;
            lowest_freq = f[k];
        }
    }
    if (lowest == n) return out;
    static const double ln_1000 = std::log(1000);
    const float freq_scale = cfg.has_fundamental ? cfg.fundamental_freq / lowest_freq : 1.f;
    for (uint32_t k = lowest; k < n; ++k) {
        const double ws = omega[k] * freq_scale;
        const double c = damping(ws);
        f[k] = float(damped_hz(ws, c));
        t60[k] = float(c > 0 ? (2 * ln_1000) / c : 0);
    }
    const float max_freq = cfg.max_mode_freq * std::max(1.f, freq_scale);
    uint32_t highest = n;
    while (highest > lowest && f[highest - 1] > max_freq) --highest;
    const uint32_t kept = std::min({cfg.num_modes, n, highest - lowest});
    out.freqs.assign(f.begin() + lowest, f.begin() + lowest + kept);
    out.t60s.assign(t60.begin() + lowest, t60.begin() + lowest + kept);
    out.lowest = lowest;
    out.original = lowest_freq;
    out.empty = false;
    return out;
}

void emit(const Modes &m, uint32_t n_eigs, uint32_t n_pos, const float *shapes, float shape_scale, uint32_t *n_modes, float *freqs, float *t60s,
          float *shapes_out, float *original) {
    const uint32_t k = m.empty ? 0 : uint32_t(m.freqs.size());
    if (n_modes) *n_modes = k;
    if (original) *original = m.empty ? 0.f : m.original;
    for (uint32_t j = 0; j < k; ++j) {
        if (freqs) freqs[j] = m.freqs[j];
        if (t60s) t60s[j] = m.t60s[j];
    }
    if (shapes_out && shapes)
        for (uint32_t p = 0; p < n_pos; ++p)
            for (uint32_t j = 0; j < k; ++j)
                for (int c = 0; c < 3; ++c) shapes_out[(size_t(p) * k + j) * 3 + c] = shapes[(size_t(p) * n_eigs + j + m.lowest) * 3 + c] * shape_scale;
}
} // namespace

extern "C" {
int mh_compute_mass_properties(uint32_t n_points, const double *points_xyz, uint32_t n_tets, const uint32_t *tets, double density, const float baked_scale[3],
                       double length_to_si, mh_mass_props *out) {
    if (!out || (n_points && !points_xyz) || (n_tets && !tets) || !baked_scale) return MH_EINVAL;
    *out = mh_mass_props{0, {0, 0, 0}, {0, 0, 0}, {1, 0, 0, 0}};
    const double inv[3] = {1.0 / baked_scale[0], 1.0 / baked_scale[1], 1.0 / baked_scale[2]};
    std::vector<D3> pos(n_points);
    for (uint32_t i = 0; i < n_points; ++i) pos[i] = {points_xyz[3 * size_t(i)] * inv[0], points_xyz[3 * size_t(i) + 1] * inv[1], points_xyz[3 * size_t(i) + 2] * inv[2]};
    std::vector<double> vol(n_points, 0.0);
    for (uint32_t t = 0; t < n_tets; ++t) {
        const uint32_t *v = tets + 4 * size_t(t);
        // FilterDegenerate runs first in the reference (mesh2modes.cpp:606); a degenerate tet adds (near) nothing here,
        // but apply the same predicate so the lumped volumes match bit for bit.
        const D3 &a = pos[v[0]];
        const D3 r0 = sub(pos[v[1]], a), r1 = sub(pos[v[2]], a), r2 = sub(pos[v[3]], a);
        {
            const D3 &pa = D3{points_xyz[3 * size_t(v[0])], points_xyz[3 * size_t(v[0]) + 1], points_xyz[3 * size_t(v[0]) + 2]};
            D3 p[4] = {pa, {}, {}, {}};
            for (int k = 1; k < 4; ++k) p[k] = {points_xyz[3 * size_t(v[k])], points_xyz[3 * size_t(v[k]) + 1], points_xyz[3 * size_t(v[k]) + 2]};
            const double det = std::abs(dot(sub(p[1], p[0]), cross(sub(p[2], p[0]), sub(p[3], p[0]))));
            double lmax_sq = 0;
            for (int i = 0; i < 4; ++i)
                for (int j = i + 1; j < 4; ++j) {
                    const D3 d = sub(p[i], p[j]);
                    lmax_sq = std::max(lmax_sq, dot(d, d));
                }
            if (!(det > 1e-12 * lmax_sq * std::sqrt(lmax_sq))) continue;
        }
        // GetTetDeterminant(a,b,c,d) = dot(d - a, cross(b - a, c - a)); volume uses the float constant 1.f/6.f (:67-69)
        const double det = dot(r2, cross(r0, r1));
        const double quarter = (1.f / 6.f) * std::fabs(det) * 0.25;
        for (int c = 0; c < 4; ++c) vol[v[c]] += quarter;
    }
    double total = 0;
    D3 com{0, 0, 0};
    for (uint32_t i = 0; i < n_points; ++i) {
        total += vol[i];
        com.x += vol[i] * pos[i].x;
        com.y += vol[i] * pos[i].y;
        com.z += vol[i] * pos[i].z;
    }
    if (total <= 0) return MH_OK;
    com = {com.x / total, com.y / total, com.z / total};
    double I[3][3] = {};
    for (uint32_t i = 0; i < n_points; ++i) {
        const D3 r = sub(pos[i], com);
        const double rr = dot(r, r);
        I[0][0] += vol[i] * (rr - r.x * r.x);
        I[1][1] += vol[i] * (rr - r.y * r.y);
        I[2][2] += vol[i] * (rr - r.z * r.z);
        I[0][1] -= vol[i] * r.x * r.y;
        I[0][2] -= vol[i] * r.x * r.z;
        I[1][2] -= vol[i] * r.y * r.z;
    }
    I[1][0] = I[0][1];
    I[2][0] = I[0][2];
    I[2][1] = I[1][2];
    const double s = length_to_si, k = density * s * s * s * s * s;
    for (auto &row : I)
        for (double &v : row) v *= k;
    double w[3], ev[3][3];
    jacobi3(I, w, ev);
    float m[3][3]; // m[col][row], as glm
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) m[c][r] = float(ev[r][c]);
    const float det = m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2]) - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2]) +
        m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]);
    if (det < 0)
        for (int r = 0; r < 3; ++r) m[0][r] = -m[0][r];
    // rotation matrix -> quaternion, largest component first, then normalise
    const float tr[4] = {m[0][0] + m[1][1] + m[2][2], m[0][0] - m[1][1] - m[2][2], m[1][1] - m[0][0] - m[2][2], m[2][2] - m[0][0] - m[1][1]};
    int big = 0;
    for (int i = 1; i < 4; ++i)
        if (tr[i] > tr[big]) big = i;
    const float bv = std::sqrt(tr[big] + 1.f) * 0.5f, mult = 0.25f / bv;
    float q[4];
    switch (big) {
        case 0: q[0] = bv; q[1] = (m[1][2] - m[2][1]) * mult; q[2] = (m[2][0] - m[0][2]) * mult; q[3] = (m[0][1] - m[1][0]) * mult; break;
        case 1: q[0] = (m[1][2] - m[2][1]) * mult; q[1] = bv; q[2] = (m[0][1] + m[1][0]) * mult; q[3] = (m[2][0] + m[0][2]) * mult; break;
        case 2: q[0] = (m[2][0] - m[0][2]) * mult; q[1] = (m[0][1] + m[1][0]) * mult; q[2] = bv; q[3] = (m[1][2] + m[2][1]) * mult; break;
        default: q[0] = (m[0][1] - m[1][0]) * mult; q[1] = (m[2][0] + m[0][2]) * mult; q[2] = (m[1][2] + m[2][1]) * mult; q[3] = bv; break;
    }
    const float qn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    out->mass = density * total * s * s * s;
    out->center_of_mass[0] = float(com.x); out->center_of_mass[1] = float(com.y); out->center_of_mass[2] = float(com.z);
    for (int i = 0; i < 3; ++i) out->inertia_diagonal[i] = float(w[i]);
    for (int i = 0; i < 4; ++i) out->inertia_orientation_wxyz[i] = qn > 0 ? q[i] / qn : (i == 0 ? 1.f : 0.f);
    return MH_OK;
}

int mh_postprocess_modes(uint32_t n_eigs, const double *eigenvalues, uint32_t n_pos, const float *shapes, float shape_scale, const mh_material *mat,
                         const mh_solver_config *cfg, uint32_t *n_modes, float *freqs, float *t60s, float *shapes_out, float *original_fundamental) {
    if (!eigenvalues || !mat || !cfg || !n_modes) return MH_EINVAL;
    emit(postprocess(n_eigs, eigenvalues, *mat, *cfg), n_eigs, n_pos, shapes, shape_scale, n_modes, freqs, t60s, shapes_out, original_fundamental);
    return MH_OK;
}

int mh_rescale_modes(uint32_t n_eigs, const double *eigenvalues, uint32_t n_pos, const float *summary_shapes, const mh_material *solved,
                     const mh_material *edited, const mh_solver_config *cfg, int *scalable, uint32_t *n_modes, float *freqs, float *t60s, float *shapes_out,
                     float *original_fundamental) {
    if (!eigenvalues || !solved || !edited || !cfg || !scalable || !n_modes) return MH_EINVAL;
    *n_modes = 0;
    *scalable = !(n_eigs == 0 || edited->poisson_ratio != solved->poisson_ratio);
    if (!*scalable) return MH_OK;
    const double rho_ratio = edited->density / solved->density;
    const double scale = (edited->young_modulus / solved->young_modulus) / rho_ratio;
    std::vector<double> ev(eigenvalues, eigenvalues + n_eigs);
    for (double &v : ev) v *= scale;
    emit(postprocess(n_eigs, ev.data(), *edited, *cfg), n_eigs, n_pos, summary_shapes, float(1 / std::sqrt(rho_ratio)), n_modes, freqs, t60s, shapes_out,
         original_fundamental);
    return MH_OK;
}
}
