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

template<typename Real> struct ImpactDev {
    uint32_t object, ex_pos, samples_left, pad;
    Real jx, jy, jz, phase_re, phase_im, rot_re, rot_im, gamma, accel_amp, b0, a1, a2, z1, z2;
};

template<typename Real> struct ImpactBack { // an impact's state after the block, as the host reads it back
    uint32_t samples_left;
    Real phase_re, phase_im, z1, z2;
};

struct WaveDesc {
    uint32_t dealt; // index into the flattened deal
    uint32_t first_mode; // first mode of this wave inside the object (multiple of MODES_PER_WAVE)
};

// Force curve + click filter per impact (ModalAudio.cpp:504-538).  force/click: [impact][frames].
template<typename Real>
__global__ void k_bank_forces(ImpactDev<Real> *__restrict__ impacts, uint32_t n_impacts, const Real *__restrict__ listener_gain, Real click_gain,
                              uint32_t frames, Real *__restrict__ force, Real *__restrict__ click, ImpactBack<Real> *__restrict__ back) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_impacts) return;
    ImpactDev<Real> im = impacts[i];
    Real phase_re = im.phase_re, phase_im = im.phase_im;
    const Real rot_re = im.rot_re, rot_im = im.rot_im, gamma = im.gamma, amp = im.accel_amp, b0 = im.b0, a1 = im.a1, a2 = im.a2;
    const Real impact_click_gain = click_gain * listener_gain[im.object];
    Real z1 = im.z1, z2 = im.z2;
    uint32_t left = im.samples_left;
    Real *f = force + size_t(i) * frames, *ck = click + size_t(i) * frames;
    auto sample = [&](Real &force_out, Real &click_out) {
        Real cur = 0;
        if (left > 0) {
            const Real re = phase_re * rot_re - phase_im * rot_im;
            phase_im = phase_re * rot_im + phase_im * rot_re;
            phase_re = re;
            cur = gamma * Real(0.5) * (Real(1) - phase_re);
            --left;
        }
        force_out = cur;
        const Real u = amp * cur;
        const Real y = b0 * u + z1;
        z1 = -a1 * y + z2;
        z2 = -b0 * u - a2 * y;
        click_out = y * impact_click_gain;
    };
    // every thread writes its own two rows: four samples per store (a wave's store touches 64 rows either way, and the loop was
    // bound by issuing those stores, not by its recurrences)
    typedef Real Quad __attribute__((ext_vector_type(4)));
    uint32_t s = 0;
    if ((frames & 3u) == 0) {
        for (; s < frames; s += 4) {
            Real fv[4], cv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) sample(fv[u], cv[u]);
            *reinterpret_cast<Quad *>(f + s) = Quad{fv[0], fv[1], fv[2], fv[3]};
            *reinterpret_cast<Quad *>(ck + s) = Quad{cv[0], cv[1], cv[2], cv[3]};
        }
    }
    for (; s < frames; ++s) sample(f[s], ck[s]);
    im.phase_re = phase_re;
    im.phase_im = phase_im;
    im.samples_left = left;
    im.z1 = z1;
    im.z2 = z2;
    impacts[i] = im;
    back[i] = {left, phase_re, phase_im, z1, z2}; // what the host takes back, written where it reads it (pinned memory)
}

template<typename Real> struct BankCols {
    Real *coeff_re, *coeff_im, *state_re, *state_im, *rad_gain, *phase_im, *phase_re, *shape_x, *shape_y, *shape_z;
    const uint32_t *mode_offset, *mode_count, *shape_offset;
};

// Cross-lane moves that stay on the VALU (no LDS crossbar): lane i reads lane i+N of its 16-lane row, and a
// wave-uniform lane broadcast.
template<int N> __device__ __forceinline__ float row_shl(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + N, 0xf, 0xf, true));
}
template<int N> __device__ __forceinline__ double row_shl(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, int(b), 0x100 + N, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, int(b >> 32), 0x100 + N, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ float lane_bcast(float v, uint32_t l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), int(l))); }
__device__ __forceinline__ double lane_bcast(double v, uint32_t l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane(int(b), int(l)), hi = __builtin_amdgcn_readlane(int(b >> 32), int(l));
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
// Sum of the 8 lanes of a chunk, lane 0..7 in order, valid in the chunk's first lane (ModalAudio.cpp:121-128).
template<typename Real> __device__ __forceinline__ Real chunk_sum_in_order(Real term) {
    Real acc = Real(0) + term;
    acc += row_shl<1>(term);
    acc += row_shl<2>(term);
    acc += row_shl<3>(term);
    acc += row_shl<4>(term);
    acc += row_shl<5>(term);
    acc += row_shl<6>(term);
    acc += row_shl<7>(term);
    return acc;
}

// One wave = 128 consecutive modes of one dealt object = 16 chunks, two adjacent modes per lane: the resonator arithmetic is
// issue bound (every operation its own instruction -- no contraction, the reference's expression tree), and on a pair of fp32
// values one packed instruction does the work of two (the fp64 bank runs the same code unpacked).  partial: [global chunk][frames].
// Samples run in tiles of TS: during a tile every lane advances its two resonators sample by sample and drops their output terms
// into an LDS tile [sample][mode]; after the tile the wave turns around -- lane = (chunk group, sample) -- and adds each chunk's 8
// terms in mode order 0..7, which is the reference's summation order at two LDS reads per mode-sample instead of a cross-lane
// chain per sample, and makes the partial-signal stores contiguous in the sample.
constexpr int MODES_PER_WAVE = 2 * WAVE;
template<typename Real>
__global__ void __launch_bounds__(WAVE) k_bank_modes(BankCols<Real> b, const WaveDesc *__restrict__ waves, const uint32_t *__restrict__ deal_objects,
                                                    const uint32_t *__restrict__ render_count, const uint32_t *__restrict__ chunk_base,
                                                    const uint32_t *__restrict__ imp_ptr, const uint32_t *__restrict__ imp_idx,
                                                    const ImpactDev<Real> *__restrict__ impacts, const Real *__restrict__ force,
                                                    const Real *__restrict__ out_gain, const Real *__restrict__ listener_gain, uint32_t frames,
                                                    Real *__restrict__ partial, Real *__restrict__ chunk_energy, Real *__restrict__ gain_scratch, uint32_t max_imp) {
    typedef Real Pair __attribute__((ext_vector_type(2)));
    constexpr uint32_t TS = 32, PITCH = MODES_PER_WAVE + 2, CHUNKS = MODES_PER_WAVE / LANES;
    __shared__ __attribute__((aligned(16))) Real s_term[TS * PITCH];
    const WaveDesc wd = waves[blockIdx.x];
    const uint32_t lane = threadIdx.x;
    const uint32_t o = deal_objects[wd.dealt];
    const uint32_t count = render_count[wd.dealt];
    const uint32_t k0 = b.mode_offset[o], stride = b.mode_count[o], shape0 = b.shape_offset[o];
    const uint32_t k = wd.first_mode + 2 * lane; // this lane's modes: k, k + 1
    const bool live[2] = {k < count, k + 1 < count};
    const uint32_t chunk0 = chunk_base[wd.dealt] + wd.first_mode / LANES; // global index of this wave's first chunk
    const uint32_t chunks_here = min(CHUNKS, (count - wd.first_mode + LANES - 1) / LANES);
    Pair z_re = {0, 0}, z_im = {0, 0}, c_re = {0, 0}, c_im = {0, 0}, p_re = {0, 0}, p_im = {0, 0};
#pragma unroll
    for (int h = 0; h < 2; ++h)
        if (live[h]) {
            z_re[h] = b.state_re[k0 + k + h]; z_im[h] = b.state_im[k0 + k + h];
            c_re[h] = b.coeff_re[k0 + k + h]; c_im[h] = b.coeff_im[k0 + k + h];
            p_im[h] = b.phase_im[k0 + k + h]; p_re[h] = b.phase_re[k0 + k + h];
        }
    const uint32_t i0 = imp_ptr[wd.dealt], n_imp = imp_ptr[wd.dealt + 1] - i0;
    // Hoisted impact gains (ImpactGainRow, ModalAudio.h:182-188); zero on padded modes.  The first IMP_REG impacts of
    // the object live in registers, further ones (rare) in a scratch row.
    constexpr uint32_t IMP_REG = 2;
    Pair g_reg[IMP_REG] = {};
    uint32_t f_row[IMP_REG] = {};
    Real *g_mem = gain_scratch + size_t(blockIdx.x) * max_imp * MODES_PER_WAVE;
    for (uint32_t t = 0; t < n_imp; ++t) {
        Pair g = {0, 0};
        const uint32_t ii = imp_idx[i0 + t];
        const ImpactDev<Real> &im = impacts[ii];
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (live[h]) {
                const uint32_t base = shape0 + im.ex_pos * stride + k + h;
                g[h] = b.rad_gain[k0 + k + h] * (b.shape_x[base] * im.jx + b.shape_y[base] * im.jy + b.shape_z[base] * im.jz);
            }
        if (t < IMP_REG) { g_reg[t] = g; f_row[t] = ii; }
        else *reinterpret_cast<Pair *>(g_mem + size_t(t) * MODES_PER_WAVE + 2 * lane) = g;
    }
    const Real mix_gain = out_gain[o] * listener_gain[o];
    const uint32_t half = lane / TS, ts = lane % TS; // turn-around mapping: chunks 8*half .. 8*half+7 of sample ts

    // NR = impacts held in registers (0, 1 or 2); EXTRA = the object has more than IMP_REG impacts.
    auto run = [&](auto nr_tag, auto extra_tag) {
        constexpr uint32_t NR = decltype(nr_tag)::value;
        constexpr bool EXTRA = decltype(extra_tag)::value;
        for (uint32_t s0 = 0; s0 < frames; s0 += TS) {
            const uint32_t sn = min(TS, frames - s0);
            Real f_tile[NR > 0 ? NR : 1] = {};
#pragma unroll
            for (uint32_t t = 0; t < NR; ++t)
                if (lane < sn) f_tile[t] = force[size_t(f_row[t]) * frames + s0 + lane];
            auto sample = [&](uint32_t ds) {
                Pair excite = {0, 0};
#pragma unroll
                for (uint32_t t = 0; t < NR; ++t) {
                    // a zero force sample is skipped by the reference; adding its product instead is the same bits: the product is a
                    // zero of either sign (the gains are finite), and excite -- never a negative zero, it starts at +0 -- is unchanged
                    // by one (a select here was two more vector instructions per impact and sample in an issue-bound loop)
                    const Real f = lane_bcast(f_tile[t], ds);
                    excite += f * g_reg[t];
                }
                if (EXTRA) {
                    for (uint32_t t = IMP_REG; t < n_imp; ++t) {
                        const Real f = force[size_t(imp_idx[i0 + t]) * frames + s0 + ds];
                        if (f == Real(0)) continue;
                        excite += f * *reinterpret_cast<const Pair *>(g_mem + size_t(t) * MODES_PER_WAVE + 2 * lane);
                    }
                }
                const Pair re = z_re * c_re - z_im * c_im + excite;
                z_im = z_re * c_im + z_im * c_re;
                z_re = re;
                *reinterpret_cast<Pair *>(s_term + ds * PITCH + 2 * lane) = p_im * z_im + p_re * re;
            };
            if (sn == TS && !EXTRA) {
#pragma unroll
                for (uint32_t ds = 0; ds < TS; ++ds) sample(ds);
            } else {
                for (uint32_t ds = 0; ds < sn; ++ds) sample(ds);
            }
            __syncthreads();
            if (ts < sn) {
                const Real *row = s_term + ts * PITCH + half * (CHUNKS / 2 * LANES);
#pragma unroll
                for (uint32_t c = 0; c < CHUNKS / 2; ++c) {
                    Real acc = 0;
#pragma unroll
                    for (uint32_t l = 0; l < LANES; l += 2) {
                        const Pair v = *reinterpret_cast<const Pair *>(row + c * LANES + l);
                        acc += v.x;
                        acc += v.y;
                    }
                    const uint32_t chunk = CHUNKS / 2 * half + c;
                    if (chunk < chunks_here) partial[size_t(chunk0 + chunk) * frames + s0 + ts] = acc * mix_gain;
                }
            }
            __syncthreads();
        }
    };
    using T0 = std::integral_constant<uint32_t, 0>;
    using T1 = std::integral_constant<uint32_t, 1>;
    using T2 = std::integral_constant<uint32_t, 2>;
    if (n_imp == 0) run(T0{}, std::false_type{});
    else if (n_imp == 1) run(T1{}, std::false_type{});
    else if (n_imp == 2) run(T2{}, std::false_type{});
    else run(T2{}, std::true_type{});
#pragma unroll
    for (int h = 0; h < 2; ++h)
        if (live[h]) {
            b.state_re[k0 + k + h] = z_re[h];
            b.state_im[k0 + k + h] = z_im[h];
        }
    // chunk energy: sum over the chunk's valid modes in order (padded modes hold zero state); a chunk is four lanes' pairs
    const Pair e = z_re * z_re + z_im * z_im;
    const Real e0 = live[0] ? e.x : Real(0), e1 = live[1] ? e.y : Real(0);
    Real chunk = Real(0) + e0;
    chunk += e1;
    chunk += row_shl<1>(e0);
    chunk += row_shl<1>(e1);
    chunk += row_shl<2>(e0);
    chunk += row_shl<2>(e1);
    chunk += row_shl<3>(e0);
    chunk += row_shl<3>(e1);
    if ((lane & (LANES / 2 - 1)) == 0 && lane / (LANES / 2) < chunks_here) chunk_energy[chunk0 + lane / (LANES / 2)] = chunk;
}

// Per dealt object (one wave each): energy, audible prefix, whole-object silence (ModalAudio.cpp:132-146).  Loads are
// lane-parallel; every sum runs in the reference's order through wave-uniform lane broadcasts.
struct PerDeviceOnceBank { // hipFuncSetAttribute once per (kernel, device)
    std::once_flag flag[64];
    template<typename F> void run(int device, F &&f) { std::call_once(flag[device >= 0 && device < 64 ? device : 0], std::forward<F>(f)); }
};
template<typename Real> struct ObjectPassArgs {
    BankCols<Real> b;
    const uint32_t *deal_objects, *render_count, *chunk_base, *imp_ptr;
    const Real *out_gain, *chunk_energy;
    uint32_t n_dealt;
    double *energy_out;
    uint32_t *live_out;
    uint8_t *silenced;
    const uint32_t *tuned_count;
    double *modal_energy;
};
template<typename Real> __device__ void bank_object_pass(const ObjectPassArgs<Real> &a, uint32_t d, uint32_t lane) {
    const BankCols<Real> &b = a.b;
    const uint32_t *deal_objects = a.deal_objects, *render_count = a.render_count, *chunk_base = a.chunk_base, *imp_ptr = a.imp_ptr, *tuned_count = a.tuned_count;
    const Real *out_gain = a.out_gain, *chunk_energy = a.chunk_energy;
    double *energy_out = a.energy_out, *modal_energy = a.modal_energy;
    uint32_t *live_out = a.live_out;
    uint8_t *silenced = a.silenced;
    if (d >= a.n_dealt) return;
    const uint32_t o = deal_objects[d], count = render_count[d];
    const Real og = out_gain[o];
    Real energy = 0;
    uint32_t live = 0;
    const uint32_t nchunks = (count + LANES - 1) / LANES, cb = chunk_base[d];
    for (uint32_t c0 = 0; c0 < nchunks; c0 += WAVE) {
        const uint32_t m = min(uint32_t(WAVE), nchunks - c0);
        const Real mine = lane < m ? chunk_energy[cb + c0 + lane] : Real(0);
        for (uint32_t l = 0; l < m; ++l) {
            const Real chunk = lane_bcast(mine, l);
            energy += chunk;
            if (chunk * og * og >= Real(1e-12f)) live = min(count, (c0 + l + 1) * LANES);
        }
    }
    const bool no_impacts = imp_ptr[d + 1] == imp_ptr[d];
    const bool silent = no_impacts && energy * og * og < Real(1e-12f);
    const uint32_t k0 = b.mode_offset[o];
    if (silent) {
        const uint32_t n = b.mode_count[o];
        for (uint32_t k = lane; k < n; k += WAVE) {
            b.state_re[k0 + k] = 0;
            b.state_im[k0 + k] = 0;
        }
    }
    // Mechanical energy behind the pressure-unit states (the diagnostic of ModalAudio.cpp:564-577), in double.
    double me = 0;
    if (!silent) {
        const uint32_t n = tuned_count[d];
        for (uint32_t q0 = 0; q0 < n; q0 += WAVE) {
            const uint32_t m = min(uint32_t(WAVE), n - q0);
            double term = 0;
            if (lane < m) {
                const double g = double(b.rad_gain[k0 + q0 + lane]);
                if (g > 0) {
                    const double re = double(b.state_re[k0 + q0 + lane]), im = double(b.state_im[k0 + q0 + lane]);
                    term = 0.5 * (re * re + im * im) / (g * g);
                }
            }
            for (uint32_t l = 0; l < m; ++l) me += lane_bcast(term, l);
        }
    }
    if (lane == 0) {
        energy_out[d] = double(energy);
        live_out[d] = live;
        silenced[d] = silent ? 1 : 0;
        modal_energy[d] = me;
    }
}

// Renderer r's private buffer: its chunks' partial signals added in chunk order (ModalAudio.cpp:130).  The chain of
// adds per sample is sequential by contract, so the work is a latency problem: one 1024-thread workgroup per
// (SW-sample strip, renderer) streams tiles of RPT*1024/SW chunk rows through LDS with all 16 waves loading (next tile
// in registers while the current one is consumed) and its first wave runs the ordered chain out of LDS.
// The passes after the resonators depend on them (or on the forces) but not on one another, and each is a latency chain that
// fills a few CUs: ONE launch runs them side by side, told apart by blockIdx.y --
//   y < n_renderers                 renderer y's ordered chunk sum
//   y == n_renderers (click_rows)   the impacts' click rows added, in impact order, to what the caller left in the block
//                                   (click_inout: a workgroup reads its own strip's start values before it writes them)
//   above                           the per-object pass, one wave per dealt object
template<typename Real, int SW>
__global__ void __launch_bounds__(1024) k_bank_post(const Real *__restrict__ partial, const uint32_t *__restrict__ renderer_chunk_ptr, uint32_t n_renderers, uint32_t frames,
                                                   Real *__restrict__ rout, const Real *__restrict__ click, uint32_t click_rows, Real *__restrict__ click_inout,
                                                   ObjectPassArgs<Real> objects) {
    // An ordered sum: the adds of one sample are a dependent chain (8 192 rows per renderer with every mode live), run by wave 0
    // out of LDS; the block's 16 waves stream tiles of rows in, transposed, so that a chain lane reads four consecutive rows with
    // one 16-byte LDS read.  Two tile buffers, one barrier per tile: the next tile lands while this one is added.
    constexpr int RPT = 32 / sizeof(Real), ROUND = 1024 / SW, TILE = RPT * ROUND, PITCH = TILE + 16 / sizeof(Real);
    extern __shared__ __attribute__((aligned(16))) unsigned char post_lds[];
    Real *xs = reinterpret_cast<Real *>(post_lds); // [2][SW][PITCH]
    const uint32_t tid = threadIdx.x, col = tid % SW, rr = tid / SW;
    const uint32_t sum_slices = n_renderers + (click_rows ? 1u : 0u);
    if (blockIdx.y >= sum_slices) {
        bank_object_pass<Real>(objects, ((blockIdx.y - sum_slices) * gridDim.x + blockIdx.x) * (1024 / WAVE) + tid / WAVE, tid % WAVE);
        return;
    }
    const uint32_t r = blockIdx.y;
    const bool clicks = r >= n_renderers;
    if (clicks) partial = click;
    const Real *start = clicks ? click_inout : nullptr;
    const uint32_t c_begin = clicks ? 0u : renderer_chunk_ptr[r], c_end = clicks ? click_rows : renderer_chunk_ptr[r + 1];
    const uint32_t s = blockIdx.x * SW + col;
    const bool in_range = s < frames;
    const uint32_t sc = in_range ? s : frames - 1; // loads of a strip's missing samples read the last one (never stored)
    // wave 0 runs the chain, a pure latency problem; the others (and whatever else shares the CU in this launch) only feed it: it
    // goes first whenever it can issue
    if (tid < WAVE) __builtin_amdgcn_s_setprio(3);
    Real pre[RPT];
    // rows past the end read the last row: the chain stops at the row count, nothing is zero-filled
    auto fetch = [&](uint32_t base) {
        if (base + TILE <= c_end) {
            const Real *p = partial + size_t(base + rr) * frames + sc;
#pragma unroll
            for (int j = 0; j < RPT; ++j) pre[j] = p[size_t(j) * ROUND * frames];
        } else {
#pragma unroll
            for (int j = 0; j < RPT; ++j) pre[j] = partial[size_t(min(base + j * ROUND + rr, c_end - 1)) * frames + sc];
        }
    };
    auto commit = [&](int buf) {
        Real *dst = xs + (size_t(buf) * SW + col) * PITCH + rr;
#pragma unroll
        for (int j = 0; j < RPT; ++j) dst[j * ROUND] = pre[j];
    };
    Real acc = (start && tid < SW && in_range) ? start[s] : Real(0);
    // (this load must have landed on every path into the tile loop: left pending on one of them, the compiler guards the chain's
    // first add with a wait for ALL outstanding loads -- which inside the loop are the next tile's, so the chain would start only
    // after its own prefetch had returned)
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    if (c_begin < c_end) {
        fetch(c_begin);
        commit(0);
    }
    __syncthreads();
    typedef Real Quad __attribute__((ext_vector_type(4)));
    int buf = 0;
    for (uint32_t base = c_begin; base < c_end; base += TILE, buf ^= 1) {
        const bool more = base + TILE < c_end;
        if (more) fetch(base + TILE);
        if (tid < SW) {
            const uint32_t cnt = min(uint32_t(TILE), c_end - base);
            const Real *row = xs + (size_t(buf) * SW + col) * PITCH;
            // the adds are one dependent chain; the LDS reads are not: the next sixteen rows are on their way while these sixteen are
            // added (a read's latency is about sixteen dependent adds)
            uint32_t q = 0;
            auto quad = [&](uint32_t at) { return *reinterpret_cast<const Quad *>(row + at); };
            auto add4 = [&](const Quad &v) { acc += v.x, acc += v.y, acc += v.z, acc += v.w; };
            if (cnt >= 16) {
                Quad a0 = quad(0), a1 = quad(4), a2 = quad(8), a3 = quad(12), b0, b1, b2, b3;
                for (; q + 48 <= cnt; q += 32) {
                    b0 = quad(q + 16), b1 = quad(q + 20), b2 = quad(q + 24), b3 = quad(q + 28);
                    add4(a0), add4(a1), add4(a2), add4(a3);
                    a0 = quad(q + 32), a1 = quad(q + 36), a2 = quad(q + 40), a3 = quad(q + 44);
                    add4(b0), add4(b1), add4(b2), add4(b3);
                }
                add4(a0), add4(a1), add4(a2), add4(a3);
                q += 16;
            }
            for (; q < cnt; ++q) acc += row[q];
        }
        if (more) commit(buf ^ 1); // the other buffer: its chain ended before the last barrier
        __syncthreads();
    }
    if (tid < SW && in_range) (clicks ? click_inout[s] : rout[size_t(r) * frames + s]) = acc;
}
template<typename Real, int SW> size_t bank_post_lds() { return size_t(2) * SW * ((32 / sizeof(Real)) * (1024 / SW) + 16 / sizeof(Real)) * sizeof(Real); }
// Host-pinned staging buffer mirrored by a device buffer: every small per-block array travels to the device in ONE copy.  Nothing
// is copied back: the kernels that produce what the host reads (impact states, per-object energies, the block's samples) write it
// into the pinned arena themselves, and the block's last kernel then raises a sequence number the host spins on -- a blit +
// hipStreamSynchronize cost the copy engine's completion signal and a thread wake-up (~40 us of a 0.34 ms block), a copy kernel for
// the ~120 KB region 15 us at the end of the chain; the spin sees the results ~2 us after the last kernel.
// That last kernel is the mix: out[s] += clicks in impact order (when they did not go through the streaming sum), then the
// renderers' buffers in renderer order (ModalAudio.cpp:531,553-555).
template<typename Real>
__global__ void __launch_bounds__(256) k_bank_mix(const Real *__restrict__ click, uint32_t n_impacts, const Real *__restrict__ rout, uint32_t n_renderers, uint32_t frames,
                                                  const Real *__restrict__ out_dev, Real *__restrict__ out_host, volatile uint32_t *flag_host, uint32_t seq) {
    for (uint32_t s = threadIdx.x; s < frames; s += 256) {
        Real acc = out_dev[s];
        for (uint32_t i = 0; i < n_impacts; ++i) acc += click[size_t(i) * frames + s];
        for (uint32_t r = 0; r < n_renderers; ++r) acc += rout[size_t(r) * frames + s];
        out_host[s] = acc;
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        *flag_host = seq;
        __threadfence_system();
    }
}

struct Arena {
    char *host{nullptr}, *dev{nullptr}, *host_seen_by_device{nullptr};
    uint32_t *flag{nullptr}, *flag_seen_by_device{nullptr}; // pinned word the download kernel raises
    uint32_t seq{0};
    size_t cap{0}, used{0};
    void reserve(size_t n) {
        if (n <= cap) return;
        release();
        cap = n + n / 2 + 4096;
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&host), cap, hipHostMallocMapped));
        HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&host_seen_by_device), host, 0));
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&flag), 64, hipHostMallocMapped));
        HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&flag_seen_by_device), flag, 0));
        *flag = 0;
        seq = 0;
        HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&dev), cap));
    }
    void release() {
        if (host) (void)hipHostFree(host);
        if (flag) (void)hipHostFree(flag);
        if (dev) (void)hipFree(dev);
        host = dev = host_seen_by_device = nullptr;
        flag = flag_seen_by_device = nullptr;
        cap = 0;
    }
    // the block's mix into the pinned arena, then wait for it: spin on the flag, with the stream's own synchronisation as the way
    // out of a stall (and the place where an asynchronous error would surface)
    template<typename Real>
    void mix_and_wait(hipStream_t st, const Real *click, uint32_t n_impacts, const Real *rout, uint32_t n_renderers, uint32_t frames, size_t out_off) {
        ++seq;
        k_bank_mix<Real><<<1, 256, 0, st>>>(click, n_impacts, rout, n_renderers, frames, d<Real>(out_off), hd<Real>(out_off), flag_seen_by_device, seq);
        HIP_CHECK(hipGetLastError());
        const volatile uint32_t *f = flag;
        for (uint64_t spin = 0; *f != seq; ++spin) {
            __builtin_ia32_pause();
            if (spin > (1ull << 22)) { // ~10 ms: not a normal block any more
                HIP_CHECK(hipStreamSynchronize(st));
                break;
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    size_t take(size_t bytes) {
        const size_t o = used;
        used += (bytes + 63) & ~size_t(63);
        return o;
    }
    template<typename T> T *h(size_t off) const { return reinterpret_cast<T *>(host + off); }
    template<typename T> T *d(size_t off) const { return reinterpret_cast<T *>(dev + off); }
    template<typename T> T *hd(size_t off) const { return reinterpret_cast<T *>(host_seen_by_device + off); } // the pinned arena as kernels address it
    ~Arena() { release(); }
};

template<typename Real> struct BankImpl {
    mh_context *ctx;
    uint32_t n_objects, n_modes, n_shapes;
    DevArray<Real> coeff_re, coeff_im, state_re, state_im, rad_gain, phase_im, phase_re, shape_x, shape_y, shape_z;
    DevArray<uint32_t> mode_offset, mode_count, shape_offset;
    std::vector<uint32_t> h_mode_count;
    // per-block scratch
    Arena arena;
    DevArray<Real> force, click, partial, chunk_energy, gain_scratch, rout;
    std::vector<int32_t> dealt_of_object;
    std::vector<uint32_t> imp_fill;
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
    uint32_t n_waves = 0;
    for (uint32_t d = 0; d < n_dealt; ++d) n_waves += (render_count[d] + MODES_PER_WAVE - 1) / MODES_PER_WAVE;
    // ---- arena layout: [upload only | both ways | download only] ----
    Arena &A = B.arena;
    A.used = 0;
    const size_t o_out_gain = A.take(B.n_objects * sizeof(Real)), o_listener = A.take(B.n_objects * sizeof(Real));
    const size_t o_waves = A.take((n_waves + 1) * sizeof(WaveDesc)), o_deal = A.take((n_dealt + 1) * 4), o_count = A.take((n_dealt + 1) * 4);
    const size_t o_tuned = A.take((n_dealt + 1) * 4), o_chunk_base = A.take((n_dealt + 1) * 4), o_imp_ptr = A.take((n_dealt + 1) * 4);
    const size_t o_imp_idx = A.take((n_impacts + 1) * 4), o_rcp = A.take((n_renderers + 1) * 4);
    const size_t o_out = A.take(frames * sizeof(Real)), o_impacts = A.take((n_impacts + 1) * sizeof(ImpactDev<Real>));
    const size_t both_end = A.used;
    // (written by the kernels straight into the pinned arena -- no copy back: the host sees them once the block's last kernel has
    // raised the sequence number)
    const size_t o_energy = A.take((n_dealt + 1) * 8), o_modal = A.take((n_dealt + 1) * 8), o_live = A.take((n_dealt + 1) * 4), o_silenced = A.take(n_dealt + 1);
    const size_t o_back = A.take((n_impacts + 1) * sizeof(ImpactBack<Real>));
    const size_t total = A.used;
    if (total > A.cap) {
        HIP_CHECK(hipStreamSynchronize(st));
        A.reserve(total);
    }
    // ---- host-side descriptors: waves, chunk bases, per-object impact lists (impact order kept) ----
    WaveDesc *waves = A.h<WaveDesc>(o_waves);
    uint32_t *chunk_base = A.h<uint32_t>(o_chunk_base), *imp_ptr = A.h<uint32_t>(o_imp_ptr), *imp_idx = A.h<uint32_t>(o_imp_idx), *rcp = A.h<uint32_t>(o_rcp);
    B.dealt_of_object.assign(B.n_objects, -1);
    chunk_base[0] = 0;
    {
        uint32_t wv = 0;
        for (uint32_t d = 0; d < n_dealt; ++d) {
            const uint32_t count = render_count[d];
            chunk_base[d + 1] = chunk_base[d] + (count + LANES - 1) / LANES;
            for (uint32_t k = 0; k < count; k += MODES_PER_WAVE) waves[wv++] = {d, k};
            if (deal_objects[d] < B.n_objects) B.dealt_of_object[deal_objects[d]] = int32_t(d);
            imp_ptr[d + 1] = 0;
        }
        imp_ptr[0] = 0;
    }
    uint32_t max_imp = 1;
    if (n_dealt) {
        for (uint32_t i = 0; i < n_impacts; ++i) {
            const int32_t d = impacts[i].object < B.n_objects ? B.dealt_of_object[impacts[i].object] : -1;
            if (d >= 0) ++imp_ptr[d + 1];
        }
        for (uint32_t d = 0; d < n_dealt; ++d) {
            max_imp = std::max(max_imp, imp_ptr[d + 1]);
            imp_ptr[d + 1] += imp_ptr[d];
        }
        B.imp_fill.assign(imp_ptr, imp_ptr + n_dealt);
        for (uint32_t i = 0; i < n_impacts; ++i) {
            const int32_t d = impacts[i].object < B.n_objects ? B.dealt_of_object[impacts[i].object] : -1;
            if (d >= 0) imp_idx[B.imp_fill[d]++] = i;
        }
    }
    for (uint32_t q = 0; q <= n_renderers; ++q) {
        const uint32_t d0 = q < n_renderers ? deal_offset[q] : n_dealt;
        rcp[q] = chunk_base[std::min(d0, n_dealt)];
    }
    const uint32_t n_chunks = chunk_base[n_dealt];
    std::copy(out_gain, out_gain + B.n_objects, A.h<Real>(o_out_gain));
    std::copy(listener_gain, listener_gain + B.n_objects, A.h<Real>(o_listener));
    std::copy(deal_objects, deal_objects + n_dealt, A.h<uint32_t>(o_deal));
    std::copy(render_count, render_count + n_dealt, A.h<uint32_t>(o_count));
    std::copy(tuned_count, tuned_count + n_dealt, A.h<uint32_t>(o_tuned));
    std::copy(out, out + frames, A.h<Real>(o_out));
    ImpactDev<Real> *himp = A.h<ImpactDev<Real>>(o_impacts);
    for (uint32_t i = 0; i < n_impacts; ++i) {
        const mh_impact &m = impacts[i];
        himp[i] = {m.object, m.ex_pos, m.samples_left, 0, Real(m.jx), Real(m.jy), Real(m.jz), Real(m.phase_re), Real(m.phase_im), Real(m.rot_re), Real(m.rot_im),
                   Real(m.gamma), Real(m.accel_amp), Real(m.click_b0), Real(m.click_a1), Real(m.click_a2), Real(m.click_z1), Real(m.click_z2)};
    }
    HIP_CHECK(hipMemcpyAsync(A.dev, A.host, both_end, hipMemcpyHostToDevice, st));
    // ---- device passes ----
    Real *d_out_gain = A.d<Real>(o_out_gain), *d_listener = A.d<Real>(o_listener), *d_out = A.d<Real>(o_out);
    ImpactDev<Real> *d_impacts = A.d<ImpactDev<Real>>(o_impacts);
    ensure(ctx, B.force, size_t(std::max<uint32_t>(n_impacts, 1)) * frames);
    ensure(ctx, B.click, size_t(std::max<uint32_t>(n_impacts, 1)) * frames);
    if (n_impacts) {
        k_bank_forces<Real><<<div_up(n_impacts, 64), 64, 0, st>>>(d_impacts, n_impacts, d_listener, Real(click_gain), frames, B.force, B.click, A.hd<ImpactBack<Real>>(o_back));
        KERNEL_CHECK();
    }
    ensure(ctx, B.rout, size_t(std::max<uint32_t>(n_renderers, 1)) * frames);
    if (n_dealt) {
        ensure(ctx, B.partial, size_t(n_chunks + 1) * frames);
        ensure(ctx, B.chunk_energy, n_chunks + 1);
        ensure(ctx, B.gain_scratch, size_t(n_waves + 1) * max_imp * MODES_PER_WAVE);
        const uint32_t *d_deal = A.d<uint32_t>(o_deal), *d_count = A.d<uint32_t>(o_count), *d_chunk_base = A.d<uint32_t>(o_chunk_base), *d_imp_ptr = A.d<uint32_t>(o_imp_ptr);
        if (n_waves) {
            uint64_t rendered_modes = 0;
            for (uint32_t d = 0; d < n_dealt; ++d) rendered_modes += render_count[d];
            TimedLaunch timed(ctx, MH_KERNEL_BANK, 11.0 * double(rendered_modes) * double(frames)); // ~11 flop per mode-sample
            k_bank_modes<Real><<<n_waves, WAVE, 0, st>>>(B.cols(), A.d<WaveDesc>(o_waves), d_deal, d_count, d_chunk_base, d_imp_ptr, A.d<uint32_t>(o_imp_idx), d_impacts,
                                                         B.force, d_out_gain, d_listener, frames, B.partial, B.chunk_energy, B.gain_scratch, max_imp);
            KERNEL_CHECK();
        }
    } else if (n_renderers) {
        HIP_CHECK(hipMemsetAsync(B.rout.get(), 0, size_t(n_renderers) * frames * sizeof(Real), st));
    }
    // out[s] += clicks in impact order, then the renderers in order.  With many impacts in flight the click chain is the
    // same latency problem as a renderer's chunks (1 024 impacts: 100 us as a per-sample loop): it goes through the streaming
    // sum, continuing from what the caller left in the block, and the mix then starts from its result.  That sum, the renderers'
    // sums and the per-object pass are one launch (k_bank_post).
    const uint32_t streamed_clicks = n_impacts > 64 ? n_impacts : 0;
    if (n_dealt || streamed_clicks) {
        constexpr int SW = 16;
        const uint32_t strips = div_up(frames, SW), sum_slices = (n_dealt ? n_renderers : 0) + (streamed_clicks ? 1 : 0);
        const uint32_t object_slices = n_dealt ? div_up(n_dealt, strips * (1024 / WAVE)) : 0;
        ObjectPassArgs<Real> objects{B.cols(), A.d<uint32_t>(o_deal), A.d<uint32_t>(o_count), A.d<uint32_t>(o_chunk_base), A.d<uint32_t>(o_imp_ptr), d_out_gain, B.chunk_energy, n_dealt,
                                     A.hd<double>(o_energy), A.hd<uint32_t>(o_live), A.hd<uint8_t>(o_silenced), A.d<uint32_t>(o_tuned), A.hd<double>(o_modal)};
        static PerDeviceOnceBank attr;
        attr.run(ctx->device, [] { HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bank_post<Real, SW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); });
        k_bank_post<Real, SW><<<dim3(strips, sum_slices + object_slices), 1024, bank_post_lds<Real, SW>(), st>>>(B.partial, A.d<uint32_t>(o_rcp), n_dealt ? n_renderers : 0, frames, B.rout, B.click, streamed_clicks,
                                                                                        d_out, objects);
        KERNEL_CHECK();
    }
    const uint32_t clicks_in_mix = streamed_clicks ? 0 : n_impacts;
    // ---- the mix, written to the host with the sequence number the host waits for ----
    A.template mix_and_wait<Real>(st, B.click, clicks_in_mix, B.rout, n_renderers, frames, o_out);
    std::copy(A.h<Real>(o_out), A.h<Real>(o_out) + frames, out);
    if (n_dealt) {
        std::copy(A.h<double>(o_energy), A.h<double>(o_energy) + n_dealt, object_energy);
        std::copy(A.h<uint32_t>(o_live), A.h<uint32_t>(o_live) + n_dealt, object_live);
        std::copy(A.h<uint8_t>(o_silenced), A.h<uint8_t>(o_silenced) + n_dealt, object_silenced);
        if (object_modal_energy) std::copy(A.h<double>(o_modal), A.h<double>(o_modal) + n_dealt, object_modal_energy);
    }
    const ImpactBack<Real> *back = A.h<ImpactBack<Real>>(o_back);
    for (uint32_t i = 0; i < n_impacts; ++i) {
        mh_impact &m = impacts[i];
        m.samples_left = back[i].samples_left;
        m.phase_re = double(back[i].phase_re);
        m.phase_im = double(back[i].phase_im);
        m.click_z1 = double(back[i].z1);
        m.click_z2 = double(back[i].z2);
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
        MhSharedPhase not_during_a_factorisation(bank->ctx->device); // a solve's dense coarse factorisation runs alone on the device (mh_eigs.hip)
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
