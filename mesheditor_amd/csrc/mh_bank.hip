// Resonator bank on the device (gfx950): one thread per mode, coupled-form complex one-pole
//     z <- z*c + excitation,   out += p_im*Im z + p_re*Re z
// restating RenderModal / RenderObjectFast of the reference (src/audio/ModalAudio.cpp:86-147, 486-555).
//
// Bit-exactness contract: this file is compiled with -ffp-contract=off and follows the reference's expression trees
// and summation order exactly -- 8-mode chunks summed lane 0..7, chunks accumulated in ascending order into the
// renderer's buffer, objects in the renderer's deal order, renderers mixed in renderer order, click filters first --
// so the signal equals the CPU restatement's sample for sample, for fp32 (the reference's bank) and fp64 alike.
// The sequential part (ordered accumulation) is a separate pass over per-chunk partial signals staged in HBM.
#include "mh_common.h"

#include <algorithm>
#include <memory>

namespace {
constexpr int LANES = 8; // ModalAudio.h:169
constexpr int WAVE = 64;
constexpr int CHUNKS_PER_WAVE = WAVE / LANES;

template<typename Real> struct ImpactDev {
    uint32_t object, ex_pos, samples_left, pad;
    Real jx, jy, jz, phase_re, phase_im, rot_re, rot_im, gamma, accel_amp, b0, a1, a2, z1, z2;
};

struct WaveDesc {
    uint32_t dealt; // index into the flattened deal
    uint32_t first_mode; // first mode of this wave inside the object (multiple of 64)
};

// Force curve + click filter per impact (ModalAudio.cpp:504-538).  force/click: [impact][frames].
template<typename Real>
__global__ void k_bank_forces(ImpactDev<Real> *__restrict__ impacts, uint32_t n_impacts, const Real *__restrict__ listener_gain, Real click_gain,
                              uint32_t frames, Real *__restrict__ force, Real *__restrict__ click) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_impacts) return;
    ImpactDev<Real> im = impacts[i];
    Real phase_re = im.phase_re, phase_im = im.phase_im;
    const Real rot_re = im.rot_re, rot_im = im.rot_im, gamma = im.gamma, amp = im.accel_amp, b0 = im.b0, a1 = im.a1, a2 = im.a2;
    const Real impact_click_gain = click_gain * listener_gain[im.object];
    Real z1 = im.z1, z2 = im.z2;
    uint32_t left = im.samples_left;
    Real *f = force + size_t(i) * frames, *ck = click + size_t(i) * frames;
    for (uint32_t s = 0; s < frames; ++s) {
        Real cur = 0;
        if (left > 0) {
            const Real re = phase_re * rot_re - phase_im * rot_im;
            phase_im = phase_re * rot_im + phase_im * rot_re;
            phase_re = re;
            cur = gamma * Real(0.5) * (Real(1) - phase_re);
            --left;
        }
        f[s] = cur;
        const Real u = amp * cur;
        const Real y = b0 * u + z1;
        z1 = -a1 * y + z2;
        z2 = -b0 * u - a2 * y;
        ck[s] = y * impact_click_gain;
    }
    im.phase_re = phase_re;
    im.phase_im = phase_im;
    im.samples_left = left;
    im.z1 = z1;
    im.z2 = z2;
    impacts[i] = im;
}

template<typename Real> struct BankCols {
    Real *coeff_re, *coeff_im, *state_re, *state_im, *rad_gain, *phase_im, *phase_re, *shape_x, *shape_y, *shape_z;
    const uint32_t *mode_offset, *mode_count, *shape_offset;
};

// One wave = 64 consecutive modes of one dealt object = 8 chunks.  partial: [global chunk][frames].
template<typename Real>
__global__ void __launch_bounds__(WAVE) k_bank_modes(BankCols<Real> b, const WaveDesc *__restrict__ waves, const uint32_t *__restrict__ deal_objects,
                                                    const uint32_t *__restrict__ render_count, const uint32_t *__restrict__ chunk_base,
                                                    const uint32_t *__restrict__ imp_ptr, const uint32_t *__restrict__ imp_idx,
                                                    const ImpactDev<Real> *__restrict__ impacts, const Real *__restrict__ force,
                                                    const Real *__restrict__ out_gain, const Real *__restrict__ listener_gain, uint32_t frames,
                                                    Real *__restrict__ partial, Real *__restrict__ chunk_energy, Real *__restrict__ gain_scratch, uint32_t max_imp) {
    __shared__ Real s_tile[CHUNKS_PER_WAVE][WAVE];
    const WaveDesc wd = waves[blockIdx.x];
    const uint32_t lane = threadIdx.x;
    const uint32_t o = deal_objects[wd.dealt];
    const uint32_t count = render_count[wd.dealt];
    const uint32_t k0 = b.mode_offset[o], stride = b.mode_count[o], shape0 = b.shape_offset[o];
    const uint32_t k = wd.first_mode + lane;
    const bool live = k < count;
    const uint32_t chunk0 = chunk_base[wd.dealt] + wd.first_mode / LANES; // global index of this wave's first chunk
    const uint32_t chunks_here = min(uint32_t(CHUNKS_PER_WAVE), (count - wd.first_mode + LANES - 1) / LANES);
    Real z_re = 0, z_im = 0, c_re = 0, c_im = 0, p_re = 0, p_im = 0;
    if (live) {
        z_re = b.state_re[k0 + k]; z_im = b.state_im[k0 + k];
        c_re = b.coeff_re[k0 + k]; c_im = b.coeff_im[k0 + k];
        p_im = b.phase_im[k0 + k]; p_re = b.phase_re[k0 + k];
    }
    const uint32_t i0 = imp_ptr[wd.dealt], n_imp = imp_ptr[wd.dealt + 1] - i0;
    // Hoisted impact gains (ImpactGainRow, ModalAudio.h:182-188); zero on padded lanes.
    Real g_reg[4] = {0, 0, 0, 0};
    Real *g_mem = gain_scratch + size_t(blockIdx.x) * max_imp * WAVE;
    for (uint32_t t = 0; t < n_imp; ++t) {
        Real g = 0;
        if (live) {
            const ImpactDev<Real> &im = impacts[imp_idx[i0 + t]];
            const uint32_t base = shape0 + im.ex_pos * stride + k;
            g = b.rad_gain[k0 + k] * (b.shape_x[base] * im.jx + b.shape_y[base] * im.jy + b.shape_z[base] * im.jz);
        }
        if (t < 4) g_reg[t] = g;
        else g_mem[size_t(t) * WAVE + lane] = g;
    }
    const Real mix_gain = out_gain[o] * listener_gain[o];
    const uint32_t chunk_lane0 = lane & ~uint32_t(LANES - 1);
    for (uint32_t s0 = 0; s0 < frames; s0 += WAVE) {
        const uint32_t sn = min(uint32_t(WAVE), frames - s0);
        for (uint32_t ds = 0; ds < sn; ++ds) {
            const uint32_t s = s0 + ds;
            Real excite = 0;
            for (uint32_t t = 0; t < n_imp; ++t) {
                const Real f = force[size_t(imp_idx[i0 + t]) * frames + s];
                if (f == Real(0)) continue;
                const Real g = t < 4 ? g_reg[t] : g_mem[size_t(t) * WAVE + lane];
                excite += f * g;
            }
            const Real re = z_re * c_re - z_im * c_im + excite;
            z_im = z_re * c_im + z_im * c_re;
            z_re = re;
            const Real term = p_im * z_im + p_re * re;
            // lanes 0..7 of the chunk, in order
            Real acc = 0;
            for (int l = 0; l < LANES; ++l) acc += __shfl(term, int(chunk_lane0) + l, WAVE);
            if ((lane & (LANES - 1)) == 0) s_tile[lane / LANES][ds] = acc * mix_gain;
        }
        __syncthreads();
        for (uint32_t c = 0; c < chunks_here; ++c)
            if (lane < sn) partial[size_t(chunk0 + c) * frames + s0 + lane] = s_tile[c][lane];
        __syncthreads();
    }
    if (live) {
        b.state_re[k0 + k] = z_re;
        b.state_im[k0 + k] = z_im;
    }
    // chunk energy: sum over the chunk's valid lanes in order (padded lanes hold zero state)
    const Real e = z_re * z_re + z_im * z_im;
    Real chunk = 0;
    for (int l = 0; l < LANES; ++l) {
        const Real el = __shfl(e, int(chunk_lane0) + l, WAVE);
        if (wd.first_mode + chunk_lane0 + l < count) chunk += el;
    }
    if ((lane & (LANES - 1)) == 0 && lane / LANES < chunks_here) chunk_energy[chunk0 + lane / LANES] = chunk;
}

// Per dealt object: energy, audible prefix, whole-object silence (ModalAudio.cpp:132-146).
template<typename Real>
__global__ void k_bank_objects(BankCols<Real> b, const uint32_t *__restrict__ deal_objects, const uint32_t *__restrict__ render_count,
                               const uint32_t *__restrict__ chunk_base, const uint32_t *__restrict__ imp_ptr, const Real *__restrict__ out_gain,
                               const Real *__restrict__ chunk_energy, uint32_t n_dealt, double *__restrict__ energy_out, uint32_t *__restrict__ live_out,
                               uint8_t *__restrict__ silenced, const uint32_t *__restrict__ tuned_count, double *__restrict__ modal_energy) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_dealt) return;
    const uint32_t o = deal_objects[d], count = render_count[d];
    const Real og = out_gain[o];
    Real energy = 0;
    uint32_t live = 0;
    const uint32_t nchunks = (count + LANES - 1) / LANES;
    for (uint32_t c = 0; c < nchunks; ++c) {
        const Real chunk = chunk_energy[chunk_base[d] + c];
        energy += chunk;
        if (chunk * og * og >= Real(1e-12f)) live = min(count, (c + 1) * LANES);
    }
    const bool no_impacts = imp_ptr[d + 1] == imp_ptr[d];
    const bool silent = no_impacts && energy * og * og < Real(1e-12f);
    if (silent) {
        const uint32_t k0 = b.mode_offset[o], n = b.mode_count[o];
        for (uint32_t k = 0; k < n; ++k) {
            b.state_re[k0 + k] = 0;
            b.state_im[k0 + k] = 0;
        }
    }
    energy_out[d] = double(energy);
    live_out[d] = live;
    silenced[d] = silent ? 1 : 0;
    // Mechanical energy behind the pressure-unit states (the diagnostic of ModalAudio.cpp:564-577), in double.
    double me = 0;
    if (!silent) {
        const uint32_t k0 = b.mode_offset[o], n = tuned_count[d];
        for (uint32_t k = 0; k < n; ++k) {
            const double g = double(b.rad_gain[k0 + k]);
            if (g > 0) {
                const double re = double(b.state_re[k0 + k]), im = double(b.state_im[k0 + k]);
                me += 0.5 * (re * re + im * im) / (g * g);
            }
        }
    }
    modal_energy[d] = me;
}

// Renderer r's private buffer: its chunks' partial signals added in order (ModalAudio.cpp:130).
template<typename Real>
__global__ void k_bank_renderer_sum(const Real *__restrict__ partial, const uint32_t *__restrict__ renderer_chunk_ptr, uint32_t frames, Real *__restrict__ rout) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t r = blockIdx.y;
    if (s >= frames) return;
    Real acc = 0;
    for (uint32_t c = renderer_chunk_ptr[r]; c < renderer_chunk_ptr[r + 1]; ++c) acc += partial[size_t(c) * frames + s];
    rout[size_t(r) * frames + s] = acc;
}
// out[s] += clicks in impact order, then the renderers' buffers in renderer order (ModalAudio.cpp:531,553-555).
template<typename Real>
__global__ void k_bank_mix(const Real *__restrict__ click, uint32_t n_impacts, const Real *__restrict__ rout, uint32_t n_renderers, uint32_t frames, Real *__restrict__ out) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= frames) return;
    Real acc = out[s];
    for (uint32_t i = 0; i < n_impacts; ++i) acc += click[size_t(i) * frames + s];
    for (uint32_t r = 0; r < n_renderers; ++r) acc += rout[size_t(r) * frames + s];
    out[s] = acc;
}

template<typename Real> struct BankImpl {
    mh_context *ctx;
    uint32_t n_objects, n_modes, n_shapes;
    DevArray<Real> coeff_re, coeff_im, state_re, state_im, rad_gain, phase_im, phase_re, shape_x, shape_y, shape_z;
    DevArray<uint32_t> mode_offset, mode_count, shape_offset;
    std::vector<uint32_t> h_mode_count;
    // per-block scratch
    DevArray<ImpactDev<Real>> d_impacts;
    DevArray<Real> force, click, partial, chunk_energy, gain_scratch, rout, d_out, d_out_gain, d_listener_gain;
    DevArray<WaveDesc> d_waves;
    DevArray<uint32_t> d_deal_objects, d_render_count, d_chunk_base, d_imp_ptr, d_imp_idx, d_renderer_chunk_ptr, d_live;
    DevArray<double> d_energy, d_modal_energy;
    DevArray<uint32_t> d_tuned;
    DevArray<uint8_t> d_silenced;
    BankCols<Real> cols() {
        return {coeff_re, coeff_im, state_re, state_im, rad_gain, phase_im, phase_re, shape_x, shape_y, shape_z, mode_offset, mode_count, shape_offset};
    }
};

template<typename T> void ensure(mh_context *ctx, DevArray<T> &a, size_t n) {
    if (a.count < n) a.reset(ctx, n + n / 4 + 16);
}

template<typename Real, typename Src>
void upload_converted(mh_context *ctx, DevArray<Real> &dst, size_t offset, const Src *src, size_t n) {
    if (!n) return;
    std::vector<Real> tmp(src, src + n);
    HIP_CHECK(hipMemcpyAsync(dst.get() + offset, tmp.data(), n * sizeof(Real), hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
}

template<typename Real>
void render_impl(BankImpl<Real> &B, uint32_t frames, float click_gain, uint32_t n_impacts, mh_impact *impacts, uint32_t n_renderers, const uint32_t *deal_offset,
                 const uint32_t *deal_objects, const uint32_t *render_count, const uint32_t *tuned_count, const float *out_gain, const float *listener_gain,
                 void *out_v, double *object_energy, uint32_t *object_live, uint8_t *object_silenced, double *object_modal_energy) {
    mh_context *ctx = B.ctx;
    hipStream_t st = ctx->stream;
    Real *out = static_cast<Real *>(out_v);
    const uint32_t n_dealt = n_renderers ? deal_offset[n_renderers] : 0;
    // host-side descriptors: waves, chunk bases, per-object impact lists
    std::vector<WaveDesc> waves;
    std::vector<uint32_t> chunk_base(n_dealt + 1, 0), imp_ptr(n_dealt + 1, 0), imp_idx, renderer_chunk_ptr(n_renderers + 1, 0);
    uint32_t max_imp = 1;
    {
        uint32_t r = 0;
        for (uint32_t d = 0; d < n_dealt; ++d) {
            while (r < n_renderers && d >= deal_offset[r + 1]) ++r;
            if (d == deal_offset[r]) renderer_chunk_ptr[r] = chunk_base[d];
            const uint32_t count = render_count[d];
            chunk_base[d + 1] = chunk_base[d] + (count + LANES - 1) / LANES;
            for (uint32_t k = 0; k < count; k += WAVE) waves.push_back({d, k});
            for (uint32_t i = 0; i < n_impacts; ++i)
                if (impacts[i].object == deal_objects[d]) imp_idx.push_back(i);
            imp_ptr[d + 1] = uint32_t(imp_idx.size());
            max_imp = std::max(max_imp, imp_ptr[d + 1] - imp_ptr[d]);
        }
        // renderers with no objects: empty ranges
        for (uint32_t q = 0; q <= n_renderers; ++q) {
            const uint32_t d0 = q < n_renderers ? deal_offset[q] : n_dealt;
            renderer_chunk_ptr[q] = chunk_base[std::min(d0, n_dealt)];
        }
    }
    const uint32_t n_chunks = chunk_base[n_dealt], n_waves = uint32_t(waves.size());
    // uploads
    ensure(ctx, B.d_out, frames);
    HIP_CHECK(hipMemcpyAsync(B.d_out.get(), out, frames * sizeof(Real), hipMemcpyHostToDevice, st));
    {
        std::vector<Real> og(out_gain, out_gain + B.n_objects), lg(listener_gain, listener_gain + B.n_objects);
        ensure(ctx, B.d_out_gain, B.n_objects);
        ensure(ctx, B.d_listener_gain, B.n_objects);
        HIP_CHECK(hipMemcpyAsync(B.d_out_gain.get(), og.data(), og.size() * sizeof(Real), hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(B.d_listener_gain.get(), lg.data(), lg.size() * sizeof(Real), hipMemcpyHostToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));
    }
    std::vector<ImpactDev<Real>> himp(n_impacts);
    for (uint32_t i = 0; i < n_impacts; ++i) {
        const mh_impact &m = impacts[i];
        himp[i] = {m.object, m.ex_pos, m.samples_left, 0, Real(m.jx), Real(m.jy), Real(m.jz), Real(m.phase_re), Real(m.phase_im), Real(m.rot_re), Real(m.rot_im),
                   Real(m.gamma), Real(m.accel_amp), Real(m.click_b0), Real(m.click_a1), Real(m.click_a2), Real(m.click_z1), Real(m.click_z2)};
    }
    ensure(ctx, B.d_impacts, std::max<uint32_t>(n_impacts, 1));
    ensure(ctx, B.force, size_t(std::max<uint32_t>(n_impacts, 1)) * frames);
    ensure(ctx, B.click, size_t(std::max<uint32_t>(n_impacts, 1)) * frames);
    if (n_impacts) {
        HIP_CHECK(hipMemcpyAsync(B.d_impacts.get(), himp.data(), n_impacts * sizeof(ImpactDev<Real>), hipMemcpyHostToDevice, st));
        k_bank_forces<Real><<<div_up(n_impacts, 64), 64, 0, st>>>(B.d_impacts, n_impacts, B.d_listener_gain, Real(click_gain), frames, B.force, B.click);
        KERNEL_CHECK();
    }
    ensure(ctx, B.rout, size_t(std::max<uint32_t>(n_renderers, 1)) * frames);
    if (n_dealt) {
        ensure(ctx, B.d_waves, n_waves + 1);
        ensure(ctx, B.d_deal_objects, n_dealt);
        ensure(ctx, B.d_render_count, n_dealt);
        ensure(ctx, B.d_chunk_base, n_dealt + 1);
        ensure(ctx, B.d_imp_ptr, n_dealt + 1);
        ensure(ctx, B.d_imp_idx, imp_idx.size() + 1);
        ensure(ctx, B.d_renderer_chunk_ptr, n_renderers + 1);
        ensure(ctx, B.partial, size_t(n_chunks + 1) * frames);
        ensure(ctx, B.chunk_energy, n_chunks + 1);
        ensure(ctx, B.gain_scratch, size_t(n_waves + 1) * max_imp * WAVE);
        ensure(ctx, B.d_energy, n_dealt);
        ensure(ctx, B.d_modal_energy, n_dealt);
        ensure(ctx, B.d_tuned, n_dealt);
        HIP_CHECK(hipMemcpyAsync(B.d_tuned.get(), tuned_count, n_dealt * 4, hipMemcpyHostToDevice, st));
        ensure(ctx, B.d_live, n_dealt);
        ensure(ctx, B.d_silenced, n_dealt);
        if (n_waves) HIP_CHECK(hipMemcpyAsync(B.d_waves.get(), waves.data(), n_waves * sizeof(WaveDesc), hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(B.d_deal_objects.get(), deal_objects, n_dealt * 4, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(B.d_render_count.get(), render_count, n_dealt * 4, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(B.d_chunk_base.get(), chunk_base.data(), (n_dealt + 1) * 4, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(B.d_imp_ptr.get(), imp_ptr.data(), (n_dealt + 1) * 4, hipMemcpyHostToDevice, st));
        if (!imp_idx.empty()) HIP_CHECK(hipMemcpyAsync(B.d_imp_idx.get(), imp_idx.data(), imp_idx.size() * 4, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipMemcpyAsync(B.d_renderer_chunk_ptr.get(), renderer_chunk_ptr.data(), (n_renderers + 1) * 4, hipMemcpyHostToDevice, st));
        if (n_waves) {
            k_bank_modes<Real><<<n_waves, WAVE, 0, st>>>(B.cols(), B.d_waves, B.d_deal_objects, B.d_render_count, B.d_chunk_base, B.d_imp_ptr, B.d_imp_idx,
                                                         B.d_impacts, B.force, B.d_out_gain, B.d_listener_gain, frames, B.partial, B.chunk_energy,
                                                         B.gain_scratch, max_imp);
            KERNEL_CHECK();
        }
        k_bank_objects<Real><<<div_up(n_dealt, 64), 64, 0, st>>>(B.cols(), B.d_deal_objects, B.d_render_count, B.d_chunk_base, B.d_imp_ptr, B.d_out_gain,
                                                                  B.chunk_energy, n_dealt, B.d_energy, B.d_live, B.d_silenced, B.d_tuned, B.d_modal_energy);
        KERNEL_CHECK();
        dim3 grid(div_up(frames, 64), n_renderers);
        k_bank_renderer_sum<Real><<<grid, 64, 0, st>>>(B.partial, B.d_renderer_chunk_ptr, frames, B.rout);
        KERNEL_CHECK();
    } else if (n_renderers) {
        HIP_CHECK(hipMemsetAsync(B.rout.get(), 0, size_t(n_renderers) * frames * sizeof(Real), st));
    }
    k_bank_mix<Real><<<div_up(frames, 64), 64, 0, st>>>(B.click, n_impacts, B.rout, n_renderers, frames, B.d_out);
    KERNEL_CHECK();
    // downloads
    HIP_CHECK(hipMemcpyAsync(out, B.d_out.get(), frames * sizeof(Real), hipMemcpyDeviceToHost, st));
    if (n_impacts) HIP_CHECK(hipMemcpyAsync(himp.data(), B.d_impacts.get(), n_impacts * sizeof(ImpactDev<Real>), hipMemcpyDeviceToHost, st));
    if (n_dealt) {
        HIP_CHECK(hipMemcpyAsync(object_energy, B.d_energy.get(), n_dealt * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(object_live, B.d_live.get(), n_dealt * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(object_silenced, B.d_silenced.get(), n_dealt, hipMemcpyDeviceToHost, st));
        if (object_modal_energy) HIP_CHECK(hipMemcpyAsync(object_modal_energy, B.d_modal_energy.get(), n_dealt * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    HIP_CHECK(hipStreamSynchronize(st));
    for (uint32_t i = 0; i < n_impacts; ++i) {
        mh_impact &m = impacts[i];
        m.samples_left = himp[i].samples_left;
        m.phase_re = double(himp[i].phase_re);
        m.phase_im = double(himp[i].phase_im);
        m.click_z1 = double(himp[i].z1);
        m.click_z2 = double(himp[i].z2);
    }
}
} // namespace

struct mh_bank {
    mh_context *ctx;
    bool dbl;
    std::unique_ptr<BankImpl<float>> f;
    std::unique_ptr<BankImpl<double>> d;
};

template<typename Real>
static std::unique_ptr<BankImpl<Real>> make_bank(mh_context *ctx, uint32_t n_objects, uint32_t n_modes, uint32_t n_shapes, const uint32_t *mode_offset,
                                                 const uint32_t *mode_count, const uint32_t *shape_offset, const float *sx, const float *sy, const float *sz) {
    auto B = std::make_unique<BankImpl<Real>>();
    B->ctx = ctx;
    B->n_objects = n_objects;
    B->n_modes = n_modes;
    B->n_shapes = n_shapes;
    for (auto *col : {&B->coeff_re, &B->coeff_im, &B->state_re, &B->state_im, &B->rad_gain, &B->phase_im, &B->phase_re}) {
        col->reset(ctx, std::max<uint32_t>(n_modes, 1));
        col->zero();
    }
    B->shape_x.reset(ctx, std::max<uint32_t>(n_shapes, 1));
    B->shape_y.reset(ctx, std::max<uint32_t>(n_shapes, 1));
    B->shape_z.reset(ctx, std::max<uint32_t>(n_shapes, 1));
    upload_converted(ctx, B->shape_x, 0, sx, n_shapes);
    upload_converted(ctx, B->shape_y, 0, sy, n_shapes);
    upload_converted(ctx, B->shape_z, 0, sz, n_shapes);
    B->mode_offset.reset(ctx, std::max<uint32_t>(n_objects, 1));
    B->mode_count.reset(ctx, std::max<uint32_t>(n_objects, 1));
    B->shape_offset.reset(ctx, std::max<uint32_t>(n_objects, 1));
    if (n_objects) {
        B->mode_offset.upload(mode_offset, n_objects);
        B->mode_count.upload(mode_count, n_objects);
        B->shape_offset.upload(shape_offset, n_objects);
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    B->h_mode_count.assign(mode_count, mode_count + n_objects);
    return B;
}

extern "C" {
int mh_bank_create(mh_context *ctx, int use_double, uint32_t n_objects, uint32_t n_modes, uint32_t n_shapes, const uint32_t *mode_offset,
                   const uint32_t *mode_count, const uint32_t *shape_offset, const float *shape_x, const float *shape_y, const float *shape_z, mh_bank **out) {
    if (!ctx || !out) return MH_EINVAL;
    *out = nullptr;
    try {
        HIP_CHECK(hipSetDevice(ctx->device));
        auto bank = std::make_unique<mh_bank>();
        bank->ctx = ctx;
        bank->dbl = use_double != 0;
        if (bank->dbl) bank->d = make_bank<double>(ctx, n_objects, n_modes, n_shapes, mode_offset, mode_count, shape_offset, shape_x, shape_y, shape_z);
        else bank->f = make_bank<float>(ctx, n_objects, n_modes, n_shapes, mode_offset, mode_count, shape_offset, shape_x, shape_y, shape_z);
        *out = bank.release();
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}
void mh_bank_destroy(mh_bank *b) { delete b; }

int mh_bank_set_coefficients(mh_bank *bank, uint32_t first, uint32_t count, const void *coeff_re, const void *coeff_im, const void *radiation_gain,
                             const void *out_phase_im, const void *out_phase_re) {
    if (!bank || (count && (!coeff_re || !coeff_im || !radiation_gain || !out_phase_im || !out_phase_re))) return MH_EINVAL;
    try {
        HIP_CHECK(hipSetDevice(bank->ctx->device));
        auto go = [&](auto &B, auto tag) {
            using Real = decltype(tag);
            if (size_t(first) + count > B.n_modes) mh_throw(MH_EINVAL, "mode range [%u, %u) outside the bank's %u modes", first, first + count, B.n_modes);
            upload_converted(bank->ctx, B.coeff_re, first, static_cast<const Real *>(coeff_re), count);
            upload_converted(bank->ctx, B.coeff_im, first, static_cast<const Real *>(coeff_im), count);
            upload_converted(bank->ctx, B.rad_gain, first, static_cast<const Real *>(radiation_gain), count);
            upload_converted(bank->ctx, B.phase_im, first, static_cast<const Real *>(out_phase_im), count);
            upload_converted(bank->ctx, B.phase_re, first, static_cast<const Real *>(out_phase_re), count);
        };
        if (bank->dbl) go(*bank->d, double{}); else go(*bank->f, float{});
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(bank->ctx, e); }
}
int mh_bank_set_shapes(mh_bank *bank, uint32_t first, uint32_t count, const float *x, const float *y, const float *z) {
    if (!bank || (count && (!x || !y || !z))) return MH_EINVAL;
    try {
        HIP_CHECK(hipSetDevice(bank->ctx->device));
        auto go = [&](auto &B) {
            if (size_t(first) + count > B.n_shapes) mh_throw(MH_EINVAL, "shape range outside the bank");
            upload_converted(bank->ctx, B.shape_x, first, x, count);
            upload_converted(bank->ctx, B.shape_y, first, y, count);
            upload_converted(bank->ctx, B.shape_z, first, z, count);
        };
        if (bank->dbl) go(*bank->d); else go(*bank->f);
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(bank->ctx, e); }
}
int mh_bank_zero_state(mh_bank *bank, uint32_t first, uint32_t count) {
    if (!bank) return MH_EINVAL;
    try {
        HIP_CHECK(hipSetDevice(bank->ctx->device));
        auto go = [&](auto &B) {
            using Real = std::remove_pointer_t<decltype(B.state_re.get())>;
            if (size_t(first) + count > B.n_modes) mh_throw(MH_EINVAL, "mode range outside the bank");
            if (!count) return;
            HIP_CHECK(hipMemsetAsync(B.state_re.get() + first, 0, count * sizeof(Real), bank->ctx->stream));
            HIP_CHECK(hipMemsetAsync(B.state_im.get() + first, 0, count * sizeof(Real), bank->ctx->stream));
            HIP_CHECK(hipStreamSynchronize(bank->ctx->stream));
        };
        if (bank->dbl) go(*bank->d); else go(*bank->f);
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(bank->ctx, e); }
}
int mh_bank_render(mh_bank *bank, uint32_t frames, float click_gain, uint32_t n_impacts, mh_impact *impacts, uint32_t n_renderers, const uint32_t *deal_offset,
                   const uint32_t *deal_objects, const uint32_t *render_count, const uint32_t *tuned_count, const float *out_gain, const float *listener_gain,
                   void *out, double *object_energy, uint32_t *object_live, uint8_t *object_silenced, double *object_modal_energy) {
    if (!bank || !out || (n_impacts && !impacts) || (n_renderers && !deal_offset) || !out_gain || !listener_gain) return MH_EINVAL;
    if (n_renderers && deal_offset[n_renderers] && (!deal_objects || !render_count || !tuned_count || !object_energy || !object_live || !object_silenced)) return MH_EINVAL;
    if (frames == 0) return MH_OK;
    try {
        HIP_CHECK(hipSetDevice(bank->ctx->device));
        if (bank->dbl) render_impl(*bank->d, frames, click_gain, n_impacts, impacts, n_renderers, deal_offset, deal_objects, render_count, tuned_count, out_gain, listener_gain, out, object_energy, object_live, object_silenced, object_modal_energy);
        else render_impl(*bank->f, frames, click_gain, n_impacts, impacts, n_renderers, deal_offset, deal_objects, render_count, tuned_count, out_gain, listener_gain, out, object_energy, object_live, object_silenced, object_modal_energy);
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(bank->ctx, e); }
}
int mh_bank_read_state(const mh_bank *bank, uint32_t first, uint32_t count, double *state_re, double *state_im) {
    if (!bank || (count && (!state_re || !state_im))) return MH_EINVAL;
    try {
        HIP_CHECK(hipSetDevice(bank->ctx->device));
        auto go = [&](auto &B) {
            using Real = std::remove_pointer_t<decltype(B.state_re.get())>;
            if (size_t(first) + count > B.n_modes) mh_throw(MH_EINVAL, "mode range outside the bank");
            std::vector<Real> re(count), im(count);
            if (!count) return;
            HIP_CHECK(hipMemcpyAsync(re.data(), B.state_re.get() + first, count * sizeof(Real), hipMemcpyDeviceToHost, bank->ctx->stream));
            HIP_CHECK(hipMemcpyAsync(im.data(), B.state_im.get() + first, count * sizeof(Real), hipMemcpyDeviceToHost, bank->ctx->stream));
            HIP_CHECK(hipStreamSynchronize(bank->ctx->stream));
            for (uint32_t i = 0; i < count; ++i) { state_re[i] = double(re[i]); state_im[i] = double(im[i]); }
        };
        if (bank->dbl) go(*bank->d); else go(*bank->f);
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(bank->ctx, e); }
}
}
