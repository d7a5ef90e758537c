// Lab soak of the multi-workgroup tridiagonalisation's tagged exchange (round 5): the product kernel's exchange
// (mh_dense.hip: k_sytrd_multi) restated with its transport as a template policy, plus a RECORD / VERIFY mode -- a clean run logs every
// value every workgroup collected (and what it loaded of A); soak runs compare each collected value with that log and record the first
// deviations: launch, workgroup, step, lane, kind, the bits received, the bits expected, the slot re-read afterwards, the XCC ids of the
// participating workgroups.  That tells a wrong value with a right tag (transport) from a wrong local computation (publisher) and shows
// what the wrong bits are (zero, an older step's value, another launch's).  Not part of the ABI.
#include "../mh_common.h"
#include "modalhip_lab.h"

#include <algorithm>
#include <thread>

namespace {
constexpr int SOAK_LD = 272;
typedef unsigned soak_granule_pair __attribute__((ext_vector_type(4)));

__device__ inline double soak_wave_sum(const double *buf, int count, int lane) {
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = lane + 64 * q;
        s += i < count ? buf[i] : 0.0;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}

// ACC: 0 = one 16-byte sc1 store / load per value (the product's form); 1 = two 8-byte relaxed agent-scope atomics per value;
//      2 = 16-byte sc0 sc1 (system scope); 3 = form 0 with a buffer_inv sc1 ahead of every poll round; 4 = form 0 with the store's
//      string ending in s_nop 1 (the > 64-bit store-data hazard hipcc does not pad inside asm); 5 = form 0 with s_waitcnt vmcnt(0) after the store;
//      6 = form 0 with an explicit s_waitcnt lgkmcnt(0) ahead of the barrier at the top of the column loop (the fix)
template<int ACC> __device__ __forceinline__ void soak_publish(unsigned long long *slot, double value, unsigned tag) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(value);
    if constexpr (ACC == 1) {
        __hip_atomic_store(slot, (bits & 0xffffffffull) | ((unsigned long long)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(slot + 1, (bits >> 32) | ((unsigned long long)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        const soak_granule_pair g = {unsigned(bits), tag, unsigned(bits >> 32), tag};
        if constexpr (ACC == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(slot), "v"(g) : "memory");
        else if constexpr (ACC == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(slot), "v"(g) : "memory");
        else if constexpr (ACC == 5) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(slot), "v"(g) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(slot), "v"(g) : "memory");
    }
}
template<int ACC> __device__ __forceinline__ void soak_load2(const unsigned long long *slot_a, const unsigned long long *slot_b, soak_granule_pair &ga, soak_granule_pair &gb) {
    if constexpr (ACC == 1) {
        const unsigned long long a0 = __hip_atomic_load(slot_a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a1 = __hip_atomic_load(slot_a + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b0 = __hip_atomic_load(slot_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b1 = __hip_atomic_load(slot_b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ga = {unsigned(a0), unsigned(a0 >> 32), unsigned(a1), unsigned(a1 >> 32)};
        gb = {unsigned(b0), unsigned(b0 >> 32), unsigned(b1), unsigned(b1 >> 32)};
    } else if constexpr (ACC == 2) {
        asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %3, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(ga), "=&v"(gb) : "v"(slot_a), "v"(slot_b) : "memory");
    } else if constexpr (ACC == 3) {
        asm volatile("buffer_inv sc1\n\tglobal_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(ga), "=&v"(gb) : "v"(slot_a), "v"(slot_b) : "memory");
    } else {
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(ga), "=&v"(gb) : "v"(slot_a), "v"(slot_b) : "memory");
    }
}
__device__ __forceinline__ double soak_value(const soak_granule_pair &g) { return __longlong_as_double((long long)((unsigned long long)g.x | ((unsigned long long)g.z << 32))); }
template<int ACC>
__device__ __forceinline__ bool soak_collect2(const unsigned long long *slot_a, const unsigned long long *slot_b, unsigned tag, double &a, double &b, unsigned &spins) {
    for (int spin = 0; spin < (1 << 22); ++spin) {
        soak_granule_pair ga, gb;
        soak_load2<ACC>(slot_a, slot_b, ga, gb);
        if (ga.y == tag && ga.w == tag && gb.y == tag && gb.w == tag) {
            a = soak_value(ga), b = soak_value(gb);
            spins = unsigned(spin);
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

struct SoakDeviation { // one record per deviating collected value (or loaded matrix entry)
    unsigned launch, group, step, lane, kind, spins, xcc, pad; // kind 0: p value, 1: column value, 2: an entry of A loaded at the start
    unsigned long long got, expected, reread_lo, reread_hi;     // reread: the slot's two granules, read again after the deviation was seen
};
struct SoakShared {
    unsigned n_deviations, n_bad_launches, first_bad_launch, first_bad_index, n_gave_up, n_split_xcc, n_bad_split, launches;
    unsigned xcc_of[16];
    unsigned bad_xcc_sets[16][16]; // the XCC ids of the first 16 bad launches
    // per workgroup, of the launch just finished: longest pause between two steps (100 MHz ticks), the step it ended at, HW_ID at the start,
    // HW_ID at the end, how often HW_ID changed between steps, the first step it had changed at
    unsigned long long wg_trace[16][6];
    unsigned long long bad_traces[16][16][6]; // the same of the first 16 bad launches
    unsigned long long longest_gap, longest_gap_launch, launches_with_moves;
    unsigned n_records, pad2;
    SoakDeviation dev[256];
};

// mode 0: plain; 1: record the collected values into log; 2: verify them against log
template<int G, int STRIDE, int ACC, bool STATIC_LDS>
__global__ void __launch_bounds__(256) k_soak_sytrd(double *__restrict__ A, const double *__restrict__ A0, int m, double *__restrict__ D, double *__restrict__ E, double *__restrict__ TAU,
                                                    unsigned long long *__restrict__ xch, unsigned epoch, int *__restrict__ gave_up, int mode, double *__restrict__ log, SoakShared *__restrict__ sh, unsigned lds_bytes, unsigned lds_lo) {
    if (blockIdx.x % 8) return;
    const int g = blockIdx.x / 8, tid = threadIdx.x, lane = tid & 63;
    // STATIC_LDS: the product kernel's own static arrays (49 288 bytes).  Otherwise ONE dynamic allocation, laid out by hand: n_lo canary words,
    // the working arrays, canary words up to the end; lds_bytes is what the launch asked for (the size of the allocation is a parameter of
    // the soak: odd sizes against multiples of the allocation granule), lds_lo the canary words in front.
    __shared__ double st_a[STATIC_LDS ? 16 * SOAK_LD : 1];
    __shared__ double st_v[STATIC_LDS ? 256 : 1], st_vp[STATIC_LDS ? 256 : 1], st_wp[STATIC_LDS ? 256 : 1], st_xs[STATIC_LDS ? 264 : 1], st_sq[STATIC_LDS ? 264 : 1], st_pq[STATIC_LDS ? 256 : 1],
        st_prs[STATIC_LDS ? 256 : 1];
    __shared__ int st_flags[2];
    extern __shared__ __attribute__((aligned(16))) double soak_lds[];
    const int n_lo = STATIC_LDS ? 0 : int(lds_lo);
    double *canary_lo = soak_lds, *a = STATIC_LDS ? st_a : soak_lds + n_lo, *v = STATIC_LDS ? st_v : a + 16 * SOAK_LD, *vp = STATIC_LDS ? st_vp : v + 256, *wp = STATIC_LDS ? st_wp : vp + 256,
           *xs = STATIC_LDS ? st_xs : wp + 256, *sq = STATIC_LDS ? st_sq : xs + 264, *pq = STATIC_LDS ? st_pq : sq + 264, *prs = STATIC_LDS ? st_prs : pq + 256;
    int *s_flags = STATIC_LDS ? st_flags : reinterpret_cast<int *>(prs + 256);
    int &s_fail = s_flags[0], &s_reported = s_flags[1];
    double *canary_hi = prs + 256 + 2;
    const int n_hi = STATIC_LDS ? 0 : int(lds_bytes / 8) - int(canary_hi - soak_lds);
    auto canary_word = [](int i) { return __longlong_as_double(0x7ff4a5a500000000ll | (long long)i); }; // (signalling-NaN patterns: never a computed value)
    for (int i = threadIdx.x; i < n_lo; i += 256) canary_lo[i] = canary_word(i);
    for (int i = threadIdx.x; i < n_hi; i += 256) canary_hi[i] = canary_word(1000 + i);
    auto slot = [&](int parity, int kind, int index) { return xch + ((size_t(parity) * 2 + kind) * 256 + index) * STRIDE; };
    auto ack_slot = [&](int parity) { return xch + size_t(2 * 2 * 256) * STRIDE + size_t(parity) * STRIDE; };
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf; // HW_REG_XCC_ID (20), bits 3:0
    if (tid == 0) sh->xcc_of[g] = xcc, s_reported = 0;
    __syncthreads();
    const unsigned hwid0 = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    unsigned hwid_last = hwid0, moves = 0, first_move = 0xffffffffu, gap_step = 0;
    const unsigned long long t_start = wall_clock64();
    unsigned long long t_prev = t_start, gap_max = 0;
    auto deviation = [&](unsigned step, unsigned kind, unsigned spins, double got, double expected, const unsigned long long *s) {
        const unsigned at = atomicAdd(&sh->n_deviations, 1u);
        if (atomicAdd(&s_reported, 1) >= 2) return; // two records per workgroup and launch: the first ones in time
        const unsigned rec = atomicAdd(&sh->n_records, 1u);
        if (rec < 256) {
            (void)at;
            SoakDeviation &d = sh->dev[rec];
            d.launch = epoch, d.group = g, d.step = step, d.lane = tid, d.kind = kind, d.spins = spins, d.xcc = xcc, d.pad = unsigned(wall_clock64() - t_start);
            d.got = (unsigned long long)__double_as_longlong(got), d.expected = (unsigned long long)__double_as_longlong(expected);
            if (s) {
                soak_granule_pair ga, gb;
                soak_load2<ACC>(s, s, ga, gb);
                d.reread_lo = (unsigned long long)ga.x | ((unsigned long long)ga.y << 32), d.reread_hi = (unsigned long long)ga.z | ((unsigned long long)ga.w << 32);
            }
        }
    };
    auto trace_out = [&](int k) {
        if (tid == 0) {
            sh->wg_trace[g][0] = gap_max, sh->wg_trace[g][1] = gap_step, sh->wg_trace[g][2] = hwid0, sh->wg_trace[g][3] = hwid_last, sh->wg_trace[g][4] = moves, sh->wg_trace[g][5] = first_move;
        }
        (void)k;
    };
    const int jl = tid >> 4, t16 = tid & 15, cl = g + jl * G;
    const int last_col = g + G * ((m - 1 - g) / G);
    for (int j = 0; j < 16; ++j) {
        const int c = g + j * G;
        if (c < m && tid < m) {
            const double x = A[size_t(c) * m + tid];
            a[j * SOAK_LD + tid] = x;
            if (mode == 2 && __double_as_longlong(x) != __double_as_longlong(A0[size_t(c) * m + tid])) deviation(unsigned(c), 2, 0, x, A0[size_t(c) * m + tid], nullptr);
        }
    }
    if (tid < m) {
        const double x = A[tid];
        xs[tid] = x;
        sq[tid] = tid >= 2 ? x * x : 0.0;
        if (mode == 2 && __double_as_longlong(x) != __double_as_longlong(A0[tid])) deviation(0, 2, 0, x, A0[tid], nullptr);
    }
    v[tid] = 0.0, vp[tid] = 0.0, wp[tid] = 0.0;
    if (tid == 0) s_fail = 0;
    double *mylog = log ? log + size_t(g) * 256 * 256 * 2 : nullptr;
    for (int k = 0; k + 1 < m; ++k) {
        const int l = m - k - 1;
        const unsigned tag = (epoch << 9) | unsigned(k + 1);
        const int parity = k & 1;
        const bool mine = g == k % G;
        // ACC 6: the wait hipcc drops here -- the release fence of __syncthreads() loses its s_waitcnt lgkmcnt(0) at this loop header, whose
        // back edge carries the ds_writes of vp, wp, xs and sq (tools/check_barrier_waits.py finds it in the assembly)
        if constexpr (ACC == 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        {
            const unsigned long long now = wall_clock64();
            if (now - t_prev > gap_max) gap_max = now - t_prev, gap_step = unsigned(k);
            t_prev = now;
            const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
            if (hw != hwid_last) {
                if (!moves) first_move = unsigned(k);
                ++moves, hwid_last = hw;
            }
        }
        if (mode == 2) { // the canaries around the working arrays: a foreign write into this workgroup's LDS shows here
            for (int i = tid; i < n_lo; i += 256)
                if (__double_as_longlong(canary_lo[i]) != __double_as_longlong(canary_word(i))) deviation(unsigned(k), 3, unsigned(i), canary_lo[i], canary_word(i), nullptr);
            for (int i = tid; i < n_hi; i += 256)
                if (__double_as_longlong(canary_hi[i]) != __double_as_longlong(canary_word(1000 + i))) deviation(unsigned(k), 4, unsigned(i), canary_hi[i], canary_word(1000 + i), nullptr);
        }
        const double xnorm2 = soak_wave_sum(sq, l + 1, lane);
        const double alpha = xs[1];
        double tau = 0.0, beta = alpha, scale = 0.0;
        if (xnorm2 > 0.0) {
            beta = -copysign(sqrt(alpha * alpha + xnorm2), alpha);
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        if (mine && tid == 0) {
            D[k] = xs[0];
            E[k] = beta;
            TAU[k] = tau;
        }
        double vi = 0.0;
        if (tid < l) {
            vi = tau == 0.0 ? 0.0 : (tid == 0 ? 1.0 : xs[tid + 1] * scale);
            v[k + 1 + tid] = vi;
            if (mine && k > 0) A[size_t(k) * m + k + 1 + tid] = tid == 0 ? beta : vi;
        }
        const bool someone_leaves = k >= m - G;
        if (last_col <= k) {
            if (last_col == k && tid == 0) soak_publish<ACC>(ack_slot(parity), 1.0, tag);
            trace_out(k);
            return;
        }
        __syncthreads();
        if (cl > k && cl < m) {
            const double vpc = vp[cl], wpc = wp[cl];
            double *col = a + jl * SOAK_LD;
            double acc = 0.0;
            for (int r = k + 1 + t16; r < m; r += 16) {
                const double aa = col[r] - (vp[r] * wpc + wp[r] * vpc);
                col[r] = aa;
                acc += aa * v[r];
                if (cl == k + 1) soak_publish<ACC>(slot(parity, 1, r), aa, tag);
            }
            for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 16);
            if (t16 == 0) soak_publish<ACC>(slot(parity, 0, cl), acc, tag);
        }
        double pr = 0.0, xn = 0.0;
        bool ok = true;
        if (tid < l) {
            unsigned spins = 0;
            ok = soak_collect2<ACC>(slot(parity, 0, k + 1 + tid), slot(parity, 1, k + 1 + tid), tag, pr, xn, spins);
            if (ok && mylog) {
                double *rec = mylog + (size_t(k) * 256 + tid) * 2;
                if (mode == 1) rec[0] = pr, rec[1] = xn;
                else if (mode == 2) {
                    if (__double_as_longlong(pr) != __double_as_longlong(rec[0])) deviation(unsigned(k), 0, spins, pr, rec[0], slot(parity, 0, k + 1 + tid));
                    if (__double_as_longlong(xn) != __double_as_longlong(rec[1])) deviation(unsigned(k), 1, spins, xn, rec[1], slot(parity, 1, k + 1 + tid));
                }
            }
            pr *= tau;
            prs[tid] = pr;
            pq[tid] = pr * vi;
        } else if (someone_leaves && tid == 255) {
            double a0, a1;
            unsigned spins = 0;
            ok = soak_collect2<ACC>(ack_slot(parity), ack_slot(parity), tag, a0, a1, spins);
        }
        if (!ok) s_fail = 1;
        __syncthreads();
        if (s_fail) {
            if (tid == 0) *gave_up = 1;
            trace_out(k);
            return;
        }
        const double pv = soak_wave_sum(pq, l, lane);
        if (tid < l) {
            const double wi = pr - 0.5 * tau * pv * vi;
            const double v0 = v[k + 1], w0 = prs[0] - 0.5 * tau * pv * v0;
            const double xc = xn - (vi * w0 + wi * v0);
            if (mine && k == 0) A[size_t(k) * m + k + 1 + tid] = tid == 0 ? beta : vi;
            vp[k + 1 + tid] = vi;
            wp[k + 1 + tid] = wi;
            xs[tid] = xc;
            sq[tid] = tid >= 2 ? xc * xc : 0.0;
        }
    }
    __syncthreads();
    trace_out(m);
    if (tid == 0) {
        D[m - 1] = xs[0];
        TAU[m - 1] = 0.0;
    }
}

// outputs of one launch against the recorded ones, bit for bit; bookkeeping of the launch's XCC placement
__global__ void __launch_bounds__(256) k_soak_compare(const double *__restrict__ A, const double *__restrict__ D, const double *__restrict__ E, const double *__restrict__ TAU, const double *__restrict__ Ar,
                                                      const double *__restrict__ Dr, const double *__restrict__ Er, const double *__restrict__ TAUr, int m, unsigned launch, const int *__restrict__ gave_up,
                                                      int groups, SoakShared *__restrict__ sh) {
    __shared__ unsigned bad, first;
    const int tid = threadIdx.x;
    if (tid == 0) bad = 0, first = 0xffffffffu;
    __syncthreads();
    auto differ = [](double x, double y) { return __double_as_longlong(x) != __double_as_longlong(y); };
    unsigned mybad = 0, myfirst = 0xffffffffu;
    for (int i = tid; i < m; i += 256) {
        bool b = differ(D[i], Dr[i]) || differ(TAU[i], TAUr[i]);
        if (i + 1 < m) b = b || differ(E[i], Er[i]);
        if (b) ++mybad, myfirst = min(myfirst, unsigned(i));
    }
    for (size_t i = tid; i < size_t(m) * m; i += 256)
        if (i % m > i / m + 1 && differ(A[i], Ar[i])) ++mybad; // the reflector tails (strictly below the subdiagonal)
    if (mybad) atomicAdd(&bad, mybad), atomicMin(&first, myfirst);
    __syncthreads();
    if (tid == 0) {
        bool split = false;
        for (int g = 1; g < groups; ++g) split = split || sh->xcc_of[g] != sh->xcc_of[0];
        sh->launches += 1;
        bool moved = false;
        for (int g = 0; g < groups; ++g) {
            if (sh->wg_trace[g][0] > sh->longest_gap) sh->longest_gap = sh->wg_trace[g][0], sh->longest_gap_launch = launch;
            moved = moved || sh->wg_trace[g][4] != 0;
        }
        if (moved) sh->launches_with_moves += 1;
        if (split) sh->n_split_xcc += 1;
        if (*gave_up) sh->n_gave_up += 1;
        if (bad) {
            if (sh->n_bad_launches < 16)
                for (int g = 0; g < 16; ++g) {
                    sh->bad_xcc_sets[sh->n_bad_launches][g] = g < groups ? sh->xcc_of[g] : 0xffu;
                    for (int q = 0; q < 6; ++q) sh->bad_traces[sh->n_bad_launches][g][q] = sh->wg_trace[g][q];
                }
            if (sh->n_bad_launches == 0) sh->first_bad_launch = launch, sh->first_bad_index = first;
            sh->n_bad_launches += 1;
            if (split) sh->n_bad_split += 1;
        }
    }
}

unsigned g_soak_lds_bytes = 0, g_soak_lds_lo = 0;
template<int STRIDE, int ACC>
void soak_launch(hipStream_t stream, double *a, const double *a0, int m, double *d, double *e, double *tau, unsigned long long *xch, unsigned epoch, int *flag, int mode, double *log, SoakShared *sh) {
    if (g_soak_lds_bytes == 0) k_soak_sytrd<16, STRIDE, ACC, true><<<128, 256, 0, stream>>>(a, a0, m, d, e, tau, xch, epoch, flag, mode, log, sh, 0, 0);
    else k_soak_sytrd<16, STRIDE, ACC, false><<<128, 256, g_soak_lds_bytes, stream>>>(a, a0, m, d, e, tau, xch, epoch, flag, mode, log, sh, g_soak_lds_bytes, g_soak_lds_lo);
}
void soak_dispatch(int stride16, int acc, hipStream_t stream, double *a, const double *a0, int m, double *d, double *e, double *tau, unsigned long long *xch, unsigned epoch, int *flag, int mode, double *log,
                   SoakShared *sh) {
#define SOAK_CASE(S, C) \
    if (stride16 == (S == 16) && acc == C) return soak_launch<S, C>(stream, a, a0, m, d, e, tau, xch, epoch, flag, mode, log, sh);
    SOAK_CASE(2, 0) SOAK_CASE(2, 1) SOAK_CASE(2, 2) SOAK_CASE(2, 3) SOAK_CASE(2, 4) SOAK_CASE(2, 5) SOAK_CASE(2, 6) SOAK_CASE(16, 0) SOAK_CASE(16, 1) SOAK_CASE(16, 2) SOAK_CASE(16, 3)
#undef SOAK_CASE
    mh_throw(MH_EINVAL, "soak: no such transport %d / %d", stride16, acc);
}
} // namespace

extern "C" {
// `launches` tridiagonalisations of one fixed random symmetric matrix of order m on the context's stream, each compared bit for bit with a
// recorded first run (made before *go is awaited: the caller starts its disturbing work after this function has set *recorded).
// transport: bit 0 one slot per 128-byte line; bits 1-3 access form (0 16-byte sc1, 1 8-byte agent atomics, 2 16-byte sc0 sc1, 3 sc1 +
// buffer_inv, 4 sc1 + s_nop 1, 5 sc1 + vmcnt(0)); bits 6-9 the LDS kind (0 static as the product's; 1.. dynamic with canaries, see lds_kinds);
// bit 10: the PRODUCT's own kernel (mh_sytrd_small, variant 1) instead of the restated one -- only its outputs are compared; bits 4-5 memory of the slots (0 hipMalloc, 1 uncached, 2 fine-grained).  out: a SoakShared image (modalhip_lab.h gives its size).
int mhl_sytrd_soak(mh_context *ctx, uint32_t transport, uint32_t m, uint32_t launches, volatile int *recorded, volatile int *go, void *out, uint64_t out_bytes) {
    if (!ctx || !out || m < 32 || m > 256 || out_bytes < sizeof(SoakShared)) return MH_EINVAL;
    try {
        HIP_CHECK(hipSetDevice(ctx->device));
        const bool product = (transport >> 10) & 1;
        const int stride16 = transport & 1, acc = (transport >> 1) & 7, memkind = (transport >> 4) & 3, ldskind = (transport >> 6) & 15;
        // the working arrays take 49 296 bytes.  LDS kinds: 0 the product kernel's static arrays (49 288 bytes, no canaries); then dynamic ones as
        // {canary words in front, bytes of the allocation}
        static const unsigned lds_kinds[16][2] = {{0, 0}, {128, 51336}, {128, 53760}, {128, 65536}, {128, 64000}, {0, 49296}, {0, 50320}, {128, 50320}, {0, 53760}, {0, 49664}, {0, 49920}, {0, 51200}, {64, 49808}, {0, 0}, {0, 0}, {0, 0}};
        g_soak_lds_lo = lds_kinds[ldskind][0], g_soak_lds_bytes = lds_kinds[ldskind][1];
        const size_t stride = stride16 ? 16 : 2, xch_bytes = (size_t(2 * 2 * 256) * stride + 2 * stride) * 8 + 256;
        unsigned long long *xch = nullptr;
        if (memkind == 0) HIP_CHECK(hipMalloc(&xch, xch_bytes));
        else HIP_CHECK(hipExtMallocWithFlags(reinterpret_cast<void **>(&xch), xch_bytes, memkind == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained));
        HIP_CHECK(hipMemsetAsync(xch, 0, xch_bytes, ctx->stream));
        int *flag = reinterpret_cast<int *>(reinterpret_cast<char *>(xch) + xch_bytes - 128);
        std::vector<double> h(size_t(m) * m);
        uint64_t s = 0x9e3779b97f4a7c15ull;
        auto rnd = [&] {
            s ^= s << 13, s ^= s >> 7, s ^= s << 17;
            return double(s >> 11) * (1.0 / 9007199254740992.0) - 0.5;
        };
        for (uint32_t c = 0; c < m; ++c)
            for (uint32_t r = c; r < m; ++r) h[size_t(c) * m + r] = h[size_t(r) * m + c] = rnd() + (r == c ? 4.0 : 0.0);
        DevArray<double> a0(ctx, size_t(m) * m), a(ctx, size_t(m) * m), ar(ctx, size_t(m) * m), d(ctx, m), e(ctx, m), tau(ctx, m), dr(ctx, m), er(ctx, m), taur(ctx, m);
        DevArray<double> log(ctx, size_t(16) * 256 * 256 * 2);
        DevArray<char> shared(ctx, sizeof(SoakShared));
        SoakShared *sh = reinterpret_cast<SoakShared *>(shared.get());
        a0.upload(h.data(), h.size());
        HIP_CHECK(hipMemsetAsync(sh, 0, sizeof(SoakShared), ctx->stream));
        HIP_CHECK(hipMemsetAsync(log.get(), 0, log.count * sizeof(double), ctx->stream));
        unsigned epoch = 1;
        // the recorded run
        HIP_CHECK(hipMemcpyAsync(ar.get(), a0.get(), size_t(m) * m * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        if (product) mh_sytrd_small(ctx, ar, m, dr, er, taur, 1);
        else soak_dispatch(stride16, acc, ctx->stream, ar, a0, int(m), dr, er, taur, xch, epoch, flag, 1, log, sh);
        KERNEL_CHECK();
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        // two more undisturbed runs must agree with it (the exchange is deterministic)
        for (int warm = 0; warm < 2; ++warm) {
            ++epoch;
            HIP_CHECK(hipMemcpyAsync(a.get(), a0.get(), size_t(m) * m * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            if (product) {
                mh_sytrd_small(ctx, a, m, d, e, tau, 1);
                k_soak_compare<<<1, 256, 0, ctx->stream>>>(a, d, e, tau, ar, dr, er, taur, int(m), epoch, ctx->sytrd_flag, 0, sh);
            } else {
                soak_dispatch(stride16, acc, ctx->stream, a, a0, int(m), d, e, tau, xch, epoch, flag, 2, log, sh);
                k_soak_compare<<<1, 256, 0, ctx->stream>>>(a, d, e, tau, ar, dr, er, taur, int(m), epoch, flag, 16, sh);
            }
            KERNEL_CHECK();
        }
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (recorded) *recorded = 1;
        while (go && !*go) std::this_thread::yield();
        for (uint32_t it = 0; it < launches; ++it) {
            ++epoch;
            HIP_CHECK(hipMemcpyAsync(a.get(), a0.get(), size_t(m) * m * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
            if (product) { // the PRODUCT's kernel (mh_sytrd_small, several workgroups): outputs compared, no collect-level record
                mh_sytrd_small(ctx, a, m, d, e, tau, 1);
                k_soak_compare<<<1, 256, 0, ctx->stream>>>(a, d, e, tau, ar, dr, er, taur, int(m), epoch, ctx->sytrd_flag, 0, sh);
            } else {
                soak_dispatch(stride16, acc, ctx->stream, a, a0, int(m), d, e, tau, xch, epoch, flag, 2, log, sh);
                k_soak_compare<<<1, 256, 0, ctx->stream>>>(a, d, e, tau, ar, dr, er, taur, int(m), epoch, flag, 16, sh);
            }
            KERNEL_CHECK();
            if ((it & 31) == 31) HIP_CHECK(hipStreamSynchronize(ctx->stream));
        }
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        HIP_CHECK(hipMemcpy(out, sh, sizeof(SoakShared), hipMemcpyDeviceToHost));
        (void)hipFree(xch);
        return MH_OK;
    } catch (const std::exception &ex) { return mh_guard(ctx, ex); }
}
uint64_t mhl_sytrd_soak_bytes(void) { return sizeof(SoakShared); }

// One kind of work launched over and over on the context's stream until *stop: the pieces a solve with blocks wider than 128 columns runs
// and a narrower one does not, each alone.  kind: 1 rocblas_dgemm of the basis update's shape (n x 408 times 408 x 136); 2 the Gram
// blocks' strided-batched dgemm (128 slabs of 136^T x 136); 3 mh_sytrd_wide at order 408; 4 mh_potrf at order 408; 5 mh_gram 136 x 136;
// 6 mh_apply_q at order 408; 7 mh_small_gemm 408^3; 8 mh_sytrd_wide at order 720; 9 mh_gram 240 x 240; 10 mh_combine 720 -> 240 columns;
// 11 mh_combine 408 -> 136; 12 k_sytrd_multi of another context (order 240).
int mhl_soak_aggressor(mh_context *ctx, int kind, volatile int *stop, uint64_t *count) {
    if (!ctx || !stop) return MH_EINVAL;
    try {
        HIP_CHECK(hipSetDevice(ctx->device));
        const size_t n = 170000;
        const uint32_t wbig = kind == 8 || kind == 9 || kind == 10 ? 240 : 136, m = 3 * wbig;
        DevArray<double> x(ctx, n * m), y(ctx, n * wbig), c(ctx, size_t(m) * m), c2(ctx, size_t(m) * m), d(ctx, m), e(ctx, m), tau(ctx, m), g(ctx, size_t(m) * m);
        DevArray<int> info(ctx, 4);
        std::vector<double> h(size_t(m) * m);
        uint64_t s = 0x12345678abcdefull;
        auto rnd = [&] {
            s ^= s << 13, s ^= s >> 7, s ^= s << 17;
            return double(s >> 11) * (1.0 / 9007199254740992.0) - 0.5;
        };
        for (uint32_t cc = 0; cc < m; ++cc)
            for (uint32_t r = cc; r < m; ++r) h[size_t(cc) * m + r] = h[size_t(r) * m + cc] = 0.01 * rnd() + (r == cc ? 4.0 : 0.0);
        c.upload(h.data(), h.size());
        {
            std::vector<double> hx(n * 8);
            for (auto &v : hx) v = rnd();
            for (size_t off = 0; off < n * m; off += hx.size()) HIP_CHECK(hipMemcpyAsync(x.get() + off, hx.data(), std::min(hx.size(), n * m - off) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
        }
        const double one = 1, zero = 0;
        uint64_t done = 0;
        while (!*stop) {
            for (int rep = 0; rep < 8; ++rep) {
                switch (kind) {
                    case 1: ROCBLAS_CHECK(rocblas_dgemm(ctx->blas, rocblas_operation_none, rocblas_operation_none, wbig, int(n), m, &one, c, wbig, x, m, &zero, y, wbig)); break;
                    case 2:
                        ROCBLAS_CHECK(rocblas_dgemm_strided_batched(ctx->blas, rocblas_operation_none, rocblas_operation_transpose, wbig, wbig, int(n / 128), &one, x, m, size_t(n / 128) * m, x + wbig, m,
                                                                    size_t(n / 128) * m, &zero, y, wbig, size_t(wbig) * wbig, 128));
                        break;
                    case 3:
                    case 8:
                        HIP_CHECK(hipMemcpyAsync(c2.get(), c.get(), size_t(m) * m * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                        mh_sytrd_wide(ctx, c2, m, d, e, tau);
                        break;
                    case 4:
                        HIP_CHECK(hipMemcpyAsync(c2.get(), c.get(), size_t(m) * m * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                        mh_potrf(ctx, c2, m, m, info);
                        break;
                    case 5:
                    case 9: mh_gram(ctx, n, x, wbig, x + wbig, wbig, g, wbig, m); break;
                    case 6: mh_apply_q(ctx, c, d, m, c2, m, wbig); break; // (any numbers do: the reflector loads and the register file are what matter)
                    case 7: mh_small_gemm(ctx, false, false, m, m, m, 1.0, c, m, c, m, 0.0, c2, m); break;
                    case 10:
                    case 11: mh_combine(ctx, n, x, wbig, x + wbig, wbig, x + 2 * wbig, wbig, c, wbig, y, wbig, nullptr, false, m, nullptr, wbig, nullptr, 0, 0); break;
                    case 12:
                        HIP_CHECK(hipMemcpyAsync(c2.get(), c.get(), size_t(240) * 240 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                        mh_sytrd_small(ctx, c2, 240, d, e, tau, 1);
                        break;
                    default: mh_throw(MH_EINVAL, "aggressor: no kind %d", kind);
                }
                ++done;
            }
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
        }
        if (count) *count = done;
        return MH_OK;
    } catch (const std::exception &ex) { return mh_guard(ctx, ex); }
}
}
