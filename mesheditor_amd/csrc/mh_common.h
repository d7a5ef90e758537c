// Shared internals of libmodalhip: context (stream, library handles, device memory pool), error plumbing,
// device-side data structures of the assembled system.
#pragma once
#include "../../include/modalhip.h"

#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

struct MhError : std::runtime_error {
    int code;
    MhError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void mh_throw(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw MhError(code, buf);
}

#define HIP_CHECK(expr)                                                                                         \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) mh_throw(MH_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define ROCBLAS_CHECK(expr)                                                                                     \
    do {                                                                                                        \
        rocblas_status s_ = (expr);                                                                             \
        if (s_ != rocblas_status_success) mh_throw(MH_EHIP, "%s failed: rocblas status %d (%s:%d)", #expr, int(s_), __FILE__, __LINE__); \
    } while (0)
#define KERNEL_CHECK() HIP_CHECK(hipGetLastError())

// Size-bucketed caching allocator: device buffers are recycled across solves so a steady-state solve performs no
// hipMalloc/hipFree (which synchronise the device).  The cache of idle blocks is capped (MH_POOL_CAP_MB; default: the larger
// of 16 GiB and an eighth of the device's memory: 36 GB here, twice the working set of a 215-pair solve of 540 k unknowns): a release that takes it over the cap frees the blocks that have been idle
// LONGEST first, so a long-lived process that once solved a huge mesh does not sit on that memory for ever -- and a solve
// whose own working set is near the cap keeps it from one solve to the next (freeing the largest blocks first, as this did
// before, threw away exactly the panels the next solve of the same mesh asks for: 1.2 s became 1.7 s for a 215-pair solve of
// 540 k unknowns after any other workload had left its blocks in the cache).
// An explicit wait for this wave's LDS stores, to be placed in front of a __syncthreads() that hipcc leaves without one.
// Found in round 5 (tools/check_barrier_waits.py): __syncthreads() is a workgroup release fence + s_barrier, and the
// fence's `s_waitcnt lgkmcnt(0)` is a "soft" wait that hipcc's wait-count pass (ROCm 7.2, gfx950) deletes when its scoreboard shows nothing
// pending.  At the header of a loop whose BACK EDGE carries ds_write instructions -- and whose body holds inline asm -- the pass decided that
// on the loop's first visit and never put the wait back: the waves of k_sytrd_multi / k_sytrd_wide reached the barrier at the top of the
// column loop with the stores of vp, wp, xs and sq still queued.  Alone on a CU those land before any other wave's next ds_read; beside an
// LDS-bound kernel on the same SIMDs (rocBLAS's dsymm: sixteen waves per workgroup, 32 LDS reads per 8 FMAs) the queue is long enough that
// another wave passes the barrier and reads the previous step's values: one workgroup publishes slightly wrong numbers, nothing times out.
// Measured with lab/mh_soak.hip beside 215-pair solves: 23-37 wrong results in 20-30 thousand launches without this wait, 0 in 120 000 with it.
// build() runs tools/check_barrier_waits.py over every kernel's assembly: a barrier reachable with an LDS store in flight fails the build.
#if defined(__HIPCC__)
__device__ __forceinline__ void mh_lds_writes_landed() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Sums across lanes on the vector ALU's data-parallel primitives (DPP).  __shfl_xor compiles to ds_bpermute, which goes through the LDS
// pipeline: about a hundred cycles per stage behind the other waves' LDS traffic, six stages for a wave -- the longest serial stretch of
// a Householder column in every tridiagonalisation kernel here (measured with cycle stamps, tools/probe/sytrd_regs_probe.py).  A DPP move
// is an ordinary ALU instruction.  Every lane receives the same bits (each stage adds the same two numbers on both sides).
template <int CTRL> __device__ __forceinline__ double mh_dpp_move(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
constexpr int MH_DPP_QUAD_XOR1 = 0xB1, MH_DPP_QUAD_XOR2 = 0x4E, MH_DPP_ROW_HALF_MIRROR = 0x141, MH_DPP_ROW_MIRROR = 0x140;
__device__ __forceinline__ double mh_quad_sum(double x) { // over each aligned group of four lanes
    x += mh_dpp_move<MH_DPP_QUAD_XOR1>(x);
    x += mh_dpp_move<MH_DPP_QUAD_XOR2>(x);
    return x;
}
__device__ __forceinline__ double mh_row_sum(double x) { // over each aligned group of sixteen lanes
    x = mh_quad_sum(x);
    x += mh_dpp_move<MH_DPP_ROW_HALF_MIRROR>(x); // (quads already uniform: the mirror pairs each quad with the other one of its half)
    x += mh_dpp_move<MH_DPP_ROW_MIRROR>(x);
    return x;
}
__device__ __forceinline__ double mh_lane_value(double x, int lane) { // lane: the same in every lane
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}
__device__ __forceinline__ double mh_wave_sum(double x) { // over the 64 lanes
    x = mh_row_sum(x);
    return (mh_lane_value(x, 0) + mh_lane_value(x, 16)) + (mh_lane_value(x, 32) + mh_lane_value(x, 48));
}
// 1 / x and (sqrt x, 1 / sqrt x) from the hardware's estimates (v_rcp_f64, v_rsq_f64: ~26 bits) and two Newton / Goldschmidt steps: a dozen
// dependent instructions where the IEEE division and square root sequences take thirty to forty each.  For the serial stretches of the
// one-workgroup dense kernels (a pivot, a Householder reflector, a Cholesky column per step), where that latency is the step's time.
// Within an ulp or two of the correctly rounded values; outside 1e-290 < |x| < 1e290 the IEEE operations are used (a uniform branch there).
__device__ __forceinline__ double mh_fast_rcp(double x) {
    if (!(fabs(x) > 1e-290 && fabs(x) < 1e290)) return 1.0 / x;
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ void mh_fast_sqrt_rsqrt(double x, double &root, double &inverse_root) { // x > 0
    if (!(x > 1e-290 && x < 1e290)) {
        root = sqrt(x), inverse_root = 1.0 / root;
        return;
    }
    const double r = __builtin_amdgcn_rsq(x);
    double g = x * r, h = 0.5 * r; // g -> sqrt x, h -> 1 / (2 sqrt x)
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g), h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    g = fma(g, e, g), h = fma(h, e, h);
    g = fma(fma(-g, g, x), h, g); // one more correction of the root from its residual x - g^2
    root = g, inverse_root = 2.0 * h;
}
__device__ __forceinline__ double mh_half_wave_sum(double x) { // over lanes 0-31 and over lanes 32-63
    x = mh_row_sum(x);
    const double low = mh_lane_value(x, 0) + mh_lane_value(x, 16), high = mh_lane_value(x, 32) + mh_lane_value(x, 48);
    return (threadIdx.x & 32) ? high : low;
}
#endif

struct DevicePool {
    struct Idle { void *p; unsigned long long stamp; };
    std::multimap<size_t, Idle> free_blocks;
    std::map<void *, size_t> live;
    size_t bytes_reserved{0}, bytes_idle{0};
    unsigned long long clock{0};
    size_t cap{size_t(16) << 30};
    void set_cap_for_device(size_t device_bytes) {
        if (const char *c = getenv("MH_POOL_CAP_MB")) cap = size_t(std::max(0, atoi(c))) << 20;
        else cap = std::max(size_t(16) << 30, device_bytes / 8);
    }
    static size_t round_up(size_t n) {
        if (n < 256) return 256;
        if (n < (1u << 20)) return (n + 4095) & ~size_t(4095);
        return (n + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);
    }
    // MH_TEST=redzone (test mode): every block carries 64 KB guard zones before and after it, filled with a pattern at allocation and
    // checked at release (a blocking copy: slow); a kernel that writes outside its array is reported with the array's size.  Written for
    // the hunt of DESIGN 11.1d (nothing within 64 KB of any array, in single-thread solves of 65 / 120 / 215 pairs).
    static constexpr size_t RZ = 64 << 10;
    static bool redzone() {
        static const bool on = getenv("MH_TEST") && strstr(getenv("MH_TEST"), "redzone");
        return on;
    }
    std::map<void *, size_t> user_bytes; // (redzone mode) user pointer -> bytes asked for
    static size_t farzone() { // (hunt aid) MH_TEST=farzone: every array sits in the middle of 2 x 32 MB of slack of its own, never checked
        static const size_t z = getenv("MH_TEST") && strstr(getenv("MH_TEST"), "farzone") ? size_t(32) << 20 : 0;
        return z;
    }
    void *alloc(size_t n_user) {
        if (farzone()) {
            char *base = static_cast<char *>(alloc_plain(n_user + 2 * farzone()));
            user_bytes[base + farzone()] = n_user;
            return base + farzone();
        }
        if (redzone()) {
            char *base = static_cast<char *>(alloc_plain(n_user + 2 * RZ));
            (void)hipDeviceSynchronize();
            (void)hipMemset(base, 0xA5, RZ);
            (void)hipMemset(base + RZ + n_user, 0xA5, RZ);
            (void)hipDeviceSynchronize();
            user_bytes[base + RZ] = n_user;
            return base + RZ;
        }
        return alloc_plain(n_user);
    }
    void release(void *p) {
        if (farzone() && p) {
            auto it = user_bytes.find(p);
            if (it != user_bytes.end()) {
                user_bytes.erase(it);
                release_plain(static_cast<char *>(p) - farzone());
                return;
            }
        }
        if (redzone() && p) {
            auto it = user_bytes.find(p);
            if (it != user_bytes.end()) {
                const size_t n_user = it->second;
                char *base = static_cast<char *>(p) - RZ;
                std::vector<unsigned char> z(2 * RZ);
                (void)hipDeviceSynchronize();
                (void)hipMemcpy(z.data(), base, RZ, hipMemcpyDeviceToHost);
                (void)hipMemcpy(z.data() + RZ, base + RZ + n_user, RZ, hipMemcpyDeviceToHost);
                size_t before = 0, after = 0, first_after = RZ;
                for (size_t i = 0; i < RZ; ++i) before += z[i] != 0xA5;
                for (size_t i = 0; i < RZ; ++i)
                    if (z[RZ + i] != 0xA5) { ++after; first_after = std::min(first_after, i); }
                if (before || after)
                    fprintf(stderr, "[redzone] array of %zu bytes: %zu guard bytes changed BEFORE it, %zu AFTER it (first at +%zu)\n", n_user, before, after, first_after);
                user_bytes.erase(it);
                release_plain(base);
                return;
            }
        }
        release_plain(p);
    }
    void *alloc_plain(size_t n) {
        const size_t r = round_up(n);
        auto it = free_blocks.lower_bound(r);
        if (it != free_blocks.end() && it->first <= r + r / 4 + (size_t(1) << 20)) {
            void *p = it->second.p;
            live[p] = it->first;
            bytes_idle -= it->first;
            free_blocks.erase(it);
            return p;
        }
        void *p = nullptr;
        if (hipMalloc(&p, r) != hipSuccess) { // out of device memory: give the idle cache back and try once more
            (void)hipGetLastError();
            trim();
            HIP_CHECK(hipMalloc(&p, r));
        }
        bytes_reserved += r;
        live[p] = r;
        return p;
    }
    void release_plain(void *p) {
        if (!p) return;
        auto it = live.find(p);
        if (it == live.end()) return;
        free_blocks.emplace(it->second, Idle{p, ++clock});
        bytes_idle += it->second;
        live.erase(it);
        while (bytes_idle > cap && !free_blocks.empty()) { // the block idle longest goes first
            auto oldest = free_blocks.begin();
            for (auto f = free_blocks.begin(); f != free_blocks.end(); ++f)
                if (f->second.stamp < oldest->second.stamp) oldest = f;
            (void)hipFree(oldest->second.p);
            bytes_idle -= oldest->first;
            bytes_reserved -= oldest->first;
            free_blocks.erase(oldest);
        }
    }
    void trim() {
        for (auto &kv : free_blocks) (void)hipFree(kv.second.p), bytes_reserved -= kv.first;
        free_blocks.clear();
        bytes_idle = 0;
    }
    ~DevicePool() {
        trim();
        for (auto &kv : live) (void)hipFree(kv.first);
    }
};

struct mh_context {
    int device{0};
    int cu_count{256}; // compute units of the device (persistent grids are sized by it)
    hipStream_t stream{nullptr};
    hipStream_t aux_stream{nullptr}; // second stream of the context: the coarse elimination beside the smoothers' set-up (created on first use)
    bool aux_stream_ready() {
        if (!aux_stream && hipStreamCreateWithFlags(&aux_stream, hipStreamNonBlocking) != hipSuccess) aux_stream = nullptr;
        if (aux_stream && !blas_aux) {
            if (rocblas_create_handle(&blas_aux) != rocblas_status_success) blas_aux = nullptr;
            else if (rocblas_set_stream(blas_aux, aux_stream) != rocblas_status_success) {
                (void)rocblas_destroy_handle(blas_aux);
                blas_aux = nullptr;
            }
        }
        return aux_stream != nullptr && blas_aux != nullptr;
    }
    hipStream_t aux2_stream{nullptr}; // third stream: the next pivot block's inverse beside the coarse elimination's rank update (our kernel only, no library handle)
    std::vector<hipEvent_t> ahead_ev; // the look-ahead's events, two per elimination step (pivot block ready, its inverse ready), never shared between steps; they live as long as the context
    bool aux2_stream_ready(size_t events = 0) {
        // (a stream of the highest priority for the pivot inverses changed nothing: round 5, notebook section 12)
        if (!aux2_stream && hipStreamCreateWithFlags(&aux2_stream, hipStreamNonBlocking) != hipSuccess) aux2_stream = nullptr, (void)hipGetLastError();
        while (aux2_stream && ahead_ev.size() < events) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
            ahead_ev.push_back(e);
        }
        return aux2_stream != nullptr;
    }
    rocblas_handle blas{nullptr};
    rocblas_handle blas_aux{nullptr}; // the second stream's own handle (a handle's workspace must not serve two streams at once), created with the stream
    DevicePool pool;
    std::string last_error;
    hipEvent_t ev0{nullptr}, ev1{nullptr};
    void *gram_ws{nullptr}; // partial-Gram workspace of mh_gram, grown on demand
    size_t gram_ws_bytes{0};
    uint32_t *iota{nullptr}; // 0, 1, 2, ... (1 024 entries): the identity column map, created on first use
    unsigned long long *sytrd_xch{nullptr}; // exchange slots of the multi-workgroup tridiagonalisation (+ its give-up flag), created on first use
    uint32_t sytrd_epoch{0};                // launch counter folded into the slots' tags
    int *sytrd_flag{nullptr};               // set by a workgroup that gave up waiting (mh_sytrd_gave_up)
    unsigned long long *sytrd_xch_wide{nullptr}; // the same for orders 257 .. 768 (mh_sytrd_wide)
    uint32_t sytrd_epoch_wide{0};
    bool exchange_disabled{false};          // set after a failed Rayleigh-Ritz self-check: orders 257-768 then go to the library's syevd instead of the tagged-exchange kernel (k_sytrd_wide), and the solve is redone once
    uint32_t sytrd_redos{0};                // Rayleigh-Ritz steps redone by a fall-back because a workgroup of the exchange gave up (co-resident work stalled it past the poll bound); mh_profile reports the count per solve
    unsigned long long *rr_check{nullptr};  // device word: worst sampled residual of the Rayleigh-Ritz steps' self-check (k_rr_selfcheck), double bits folded by atomicMax
    // Optional per-launch timing of the path's named kernels (measurement aid for bench.py's roofline objects): HIP
    // events on this stream around every launch of a kernel class, resolved lazily.  `work` is the class's algorithmic
    // unit: bytes for the HBM-bound classes, flops for the resonator bank.
    bool time_kernels{false};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timer_events;
    std::vector<double> timer_work;
    std::vector<int> timer_class;
    size_t timer_used{0};
    struct ClassTotals {
        double ms{0}, work{0};
        uint64_t launches{0};
    } totals[MH_KERNEL_CLASSES];
};
void mh_timer_flush(mh_context *ctx); // mh_spmm.hip
// HIP events around one launch on the context's stream, booked under a kernel class; resolved by mh_timer_flush.
struct TimedLaunch {
    mh_context *ctx;
    bool on;
    size_t slot{0};
    TimedLaunch(mh_context *c, int kernel_class, double work) : ctx(c), on(c->time_kernels) {
        if (!on) return;
        if (ctx->timer_used == ctx->timer_events.size()) {
            hipEvent_t a, b;
            HIP_CHECK(hipEventCreate(&a));
            HIP_CHECK(hipEventCreate(&b));
            ctx->timer_events.emplace_back(a, b);
            ctx->timer_work.push_back(0);
            ctx->timer_class.push_back(0);
        }
        slot = ctx->timer_used++;
        ctx->timer_work[slot] = work;
        ctx->timer_class[slot] = kernel_class;
        HIP_CHECK(hipEventRecord(ctx->timer_events[slot].first, ctx->stream));
    }
    ~TimedLaunch() {
        if (on) (void)hipEventRecord(ctx->timer_events[slot].second, ctx->stream);
    }
    TimedLaunch(const TimedLaunch &) = delete;
    TimedLaunch &operator=(const TimedLaunch &) = delete;
};

// RAII device array bound to a context's pool.
template<typename T> struct DevArray {
    mh_context *ctx{nullptr};
    T *ptr{nullptr};
    size_t count{0};
    DevArray() = default;
    DevArray(mh_context *c, size_t n) { reset(c, n); }
    DevArray(const DevArray &) = delete;
    DevArray &operator=(const DevArray &) = delete;
    DevArray(DevArray &&o) noexcept : ctx(o.ctx), ptr(o.ptr), count(o.count) { o.ptr = nullptr; o.count = 0; }
    DevArray &operator=(DevArray &&o) noexcept {
        if (this != &o) {
            free();
            ctx = o.ctx; ptr = o.ptr; count = o.count;
            o.ptr = nullptr; o.count = 0;
        }
        return *this;
    }
    ~DevArray() { free(); }
    void reset(mh_context *c, size_t n) {
        free();
        ctx = c;
        count = n;
        ptr = n ? static_cast<T *>(c->pool.alloc(n * sizeof(T))) : nullptr;
        // MH_TEST=poison (test mode): every fresh array starts as all-ones bits (NaN for floating point), so a read of
        // memory the code never wrote shows up as a wrong result instead of depending on what the pool handed back
        static const bool poison = getenv("MH_TEST") && strstr(getenv("MH_TEST"), "poison");
        if (poison && ptr) (void)hipMemsetAsync(ptr, 0xff, n * sizeof(T), c->stream);
    }
    void free() {
        if (ptr && ctx) ctx->pool.release(ptr);
        ptr = nullptr;
        count = 0;
    }
    T *get() const { return ptr; }
    operator T *() const { return ptr; }
    void zero() const { if (ptr) HIP_CHECK(hipMemsetAsync(ptr, 0, count * sizeof(T), ctx->stream)); }
    void upload(const T *src, size_t n) const { HIP_CHECK(hipMemcpyAsync(ptr, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream)); }
    void download(T *dst, size_t n) const {
        HIP_CHECK(hipMemcpyAsync(dst, ptr, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    std::vector<T> to_host() const {
        std::vector<T> h(count);
        if (count) download(h.data(), count);
        return h;
    }
};

struct mh_mesh {
    mh_context *ctx;
    uint32_t n_points{0}, n_tets{0};
    DevArray<double> points; // n_points x 3
    DevArray<uint32_t> tets; // n_tets x 4
};

// One level of the operator hierarchy: symmetric matrix in 3x3 node blocks (BSR), rows and columns in the level's
// internal (Morton) numbering.
struct BsrLevel {
    int id{0}; // 2 = P2 operator, 1 = P1 operator
    uint32_t n_nodes{0};
    uint64_t n_blocks{0};
    DevArray<uint32_t> row_ptr; // n_nodes + 1
    DevArray<uint32_t> col; // n_blocks
    DevArray<double> kval; // n_blocks x 9, row-major 3x3 (stiffness K)
    DevArray<double> mval; // n_blocks (mass scalar of the node block)
    DevArray<double> aval; // n_blocks x 9: K - sigma*M, filled by the eigensolver set-up
    DevArray<double> dinv; // 3*n_nodes: 1 / diag(A)
    DevArray<float> aval32, dinv32; // single-precision copies for the preconditioner's smoothers
    double lmax{0}; // spectral radius estimate of D^-1 A
};

// Exact inverses of a level's operator on the nodes of badly shaped elements (mh_patch.hip): the patch part of the smoothers
struct PatchSet {
    uint32_t npe{0}, n_patches{0}, n_touched{0}; // nodes per element (10: P2 level, 4: P1 level)
    DevArray<uint32_t> nodes;                      // n_patches x npe, the level's node ids
    DevArray<double> inv64;                        // n_patches x (3 npe)^2, row-major (weighted)
    DevArray<double> weight;                       // per patch: (the largest number of patches sharing one of its nodes)^-0.35 (exponent -1 lost)
    DevArray<int> dropped;                         // [0]: patches whose block was not safely positive definite (they contribute nothing), [1]: one of them + 1
    DevArray<float> inv32;
    DevArray<uint32_t> touched, t_ptr, t_patch, t_local; // nodes in some patch; per touched node its (patch, local node) entries, CSR
    // CLUSTERS (round 6): badly shaped elements that share nodes (a pole fan's needles, stacked caps: connected components of the bad
    // elements, two or more of them) get ONE exact inverse on the union of their nodes instead of overlapping element patches -- measured
    // on a 96 x 48 UV sphere's fill with 172 cells flat to 1e-8 (two-grid bound, exact P1 level; tools/proto/cluster_patches.py): condition
    // number 786 with the weighted element patches, 339 / 124 / 75 with clusters cut at 64 / 128 / 256 nodes, 9 with whole clusters (818 nodes);
    // a well-shaped sphere: 5.  Node sets are pairwise disjoint (a component larger than the cap is cut into node-disjoint pieces).
    uint32_t n_clusters{0}, cluster_rows{0}, cluster_tiles{0}; // clusters; the sum of their orders 3 n_c; 64-row tiles over all of them
    uint32_t n_bad_elements{0};                                   // elements below the shape threshold, in element patches and clusters together
    uint32_t largest_cluster{0};                                  // nodes of the largest one
    std::vector<uint32_t> h_cluster_ptr;                          // host copy of the row offsets (n_clusters + 1), in rows (3 per node)
    DevArray<uint32_t> cluster_row;                               // cluster_rows: the level's dof row (3 node + k) of every cluster row
    DevArray<uint32_t> cluster_ptr;                               // n_clusters + 1: first row of each cluster in cluster_row
    DevArray<uint64_t> cluster_inv_ptr;                           // n_clusters + 1: offset of each cluster's inverse (order^2 values, row-major)
    DevArray<uint32_t> tile_cluster, tile_row0;                   // per 64-row tile: its cluster and first row within it
    DevArray<double> cinv64;
    DevArray<float> cinv32;
    bool any() const { return n_patches || n_clusters; }
    static constexpr int kClusterSlices = 8;                          // K slices of the cluster product when its row tiles alone are too few workgroups
    size_t scratch_rows() const { return size_t(n_patches) * 3 * npe + size_t(cluster_rows) * kClusterSlices; } // rows of the work panel mh_apply_patches needs
};

struct mh_system {
    mh_context *ctx;
    uint32_t dropped_patches[2]{0, 0}; // sliver patches dropped at the last hierarchy build (P2 level, P1 level)
    double plain_residual{-1.0}; // last solve: worst 2-norm relative residual of the returned elastic pairs when they were accepted in the Jacobi-scaled norm, else -1
    mh_material material{};
    uint32_t n_points{0}, kept_tets{0}, n_nodes{0}, n_edges{0};
    bool lmax_widened{false}; // a solve stalled and was redone with the smoothers' spectral bounds widened by a quarter (mh_eigs.hip, eigs_impl)
    DevArray<double> points; // copy of mesh points (P1 node coordinates), reference numbering
    DevArray<uint32_t> elem_nodes_ref; // kept_tets x 10, reference numbering
    DevArray<uint32_t> elem_nodes; // kept_tets x 10, internal numbering
    DevArray<double> elem_basis; // kept_tets x 16: four (gradient xyz, volume) quadruples = one 128-byte line per tet
    DevArray<uint32_t> perm; // internal -> reference node id
    DevArray<uint32_t> inv_perm; // reference -> internal
    DevArray<double> node_xyz; // n_nodes x 3, internal numbering
    // P2 <- P1 interpolation: internal P2 node -> its one (corner) or two (midside) P1 parents, internal P1 ids
    DevArray<uint32_t> parent_a, parent_b;
    // P1 node -> the midside P2 nodes of its edges (CSR), for the transposed interpolation
    DevArray<uint32_t> p1_edge_ptr, p1_edge_mid;
    DevArray<uint32_t> p1_corner; // P1 internal id -> internal P2 id of the same point
    DevArray<double> p1_xyz; // n_points x 3, P1 internal numbering
    BsrLevel L2, L1;
    // level 0: rigid-body modes of aggregates of P1 nodes -- connected node sets grown on the P1 operator's graph (root +
    // neighbours, then pairwise merging; mh_pipeline.hip), given as CSR lists -- with a dense, explicitly inverted operator
    uint32_t agg_target{16}, n_agg{0}; // agg_target: merging goes on while the mean aggregate is well below it (MH_AGG)
    DevArray<uint32_t> agg_of, agg_ptr, agg_nodes; // P1 node -> aggregate; aggregate -> its nodes (ascending), CSR
    DevArray<double> agg_t; // n_points x 18: the 3x6 tentative-prolongator block of each P1 node (row-major)
    DevArray<double> a0; // (6 n_agg)^2 column-major: the coarse operator, replaced by its explicit inverse at set-up
    // Connected bodies of the mesh (components of the P1 graph): each is a free body with six rigid-body modes of its own.  A scan
    // with stray fragments has several; the eigensolver seeds the exact modes of every one (mh_eigs.hip: start).
    uint32_t n_components{1};
    uint32_t unreferenced_points{0}; // mesh points no kept tetrahedron uses: their rows of K and M are empty, the shifted operator is singular
    DevArray<uint32_t> node_component; // n_nodes (P2, internal numbering)
    DevArray<double> component_centroid; // n_components x 3
    PatchSet patches2, patches1; // sliver patches of the two smoothed levels (empty on well-shaped meshes)
    DevArray<uint32_t> elem_p1;  // kept_tets x 4, internal P1 numbering
    float worst_quality{1.f};    // smallest element shape measure (1 = regular tetrahedron)
    double sigma_built{0};
    bool hierarchy_ready{false};
    // The coarse inverse's elimination runs on the context's second stream and may still be running when mh_build_hierarchy
    // returns (defer = true): its workspaces live here until mh_finish_hierarchy has made the main stream wait for it -- which
    // the first application of the preconditioner does -- so the eigensolver's start-up (first block, its images, the first
    // Rayleigh-Ritz step) runs beside the elimination's one-workgroup kernels.
    hipEvent_t coarse_done{nullptr};
    bool coarse_pending{false};
    double coarse_lift{0}; // relative lift of the coarse operator's diagonal beyond the default (raised by eigs_impl when the elimination met a non-positive pivot)
    DevArray<double> coarse_ws[5];
    DevArray<int> coarse_info;
    ~mh_system() {
        if (coarse_pending && coarse_done) (void)hipEventSynchronize(coarse_done);
        if (coarse_done) (void)hipEventDestroy(coarse_done);
    }
    // eigensolver result (internal numbering, row-major n x ncols)
    DevArray<double> evecs;
    uint32_t evec_cols{0};
    mh_profile profile{};
};

inline int mh_guard(mh_context *ctx, const std::exception &e) {
    if (ctx) ctx->last_error = e.what();
    if (auto *m = dynamic_cast<const MhError *>(&e)) return m->code;
    return MH_EHIP;
}

#define MH_TRY(ctx_expr, body)                         \
    mh_context *ctx_guard_ = (ctx_expr);               \
    try {                                              \
        body;                                          \
        return MH_OK;                                  \
    } catch (const std::exception &e) {                \
        return mh_guard(ctx_guard_, e);                \
    }

inline unsigned div_up(size_t a, size_t b) { return unsigned((a + b - 1) / b); }

// ---- stage entry points implemented across the .hip files ----
void mh_phase_shared_lock(int device);   // mh_eigs.hip: device work of other entry points keeps out of an exclusive factorisation phase
void mh_phase_shared_unlock(int device); // ON THE SAME DEVICE (one lock per device; no-ops with MH_CONCURRENT_SOLVES=0)
struct MhSharedPhase {
    const int device;
    explicit MhSharedPhase(int dev) : device(dev) { mh_phase_shared_lock(device); }
    ~MhSharedPhase() { mh_phase_shared_unlock(device); }
    MhSharedPhase(const MhSharedPhase &) = delete;
    MhSharedPhase &operator=(const MhSharedPhase &) = delete;
};
std::mutex &mh_solve_mutex(); // mh_eigs.hip: one eigensolve (or Gram benchmark) at a time per process, see there
void mh_build_system(mh_context *ctx, const mh_mesh *mesh, const mh_material &mat, mh_system *sys); // mh_pipeline.hip
void mh_build_hierarchy(mh_system *sys, double sigma, bool defer = false);
void mh_finish_hierarchy(mh_system *sys);
uint32_t mh_shift_invert_panel(mh_system *sys, double sigma, const double *b, double *x, uint32_t w, double rel_tol, uint32_t max_iters, double *worst_rel); // mh_eigs.hip: x = (K - sigma M)^-1 b by preconditioned CG
 // no-op unless a deferred elimination is pending // mh_pipeline.hip: A = K - sigma M on both levels, dense coarse factor
uint32_t mh_graph_aggregates(const std::vector<uint32_t> &row_ptr, const std::vector<uint32_t> &col, uint32_t n, uint32_t target, uint32_t max_order, std::vector<uint32_t> &agg_of); // mh_pipeline.hip
void mh_select_patches(mh_system *sys, float threshold);                                  // mh_patch.hip: elements whose shape measure is below the threshold
void mh_build_patch_inverses(mh_context *ctx, const BsrLevel &lvl, PatchSet &ps);         // mh_patch.hip: (A_ee)^-1 of every patch from lvl.aval
// d += s, x += s (either may be null) or z (double, pitch wz) += s with s = coef * sum_e R_e^T (A_ee)^-1 R_e (in - minus), e over element patches and clusters; scratch: ps.scratch_rows() x w
template<typename T>
void mh_apply_patches(mh_context *ctx, const PatchSet &ps, const T *in, const T *minus, uint32_t w, T coef, T *d, T *x, double *z, uint32_t wz, T *scratch);
// y (n x w row-major, ld = w) = A x with A given by 9-value blocks; optionally y2 = M x from the scalar blocks.
// G (wa x wb, column-major, ld) = X^T Y for row-major panels (fp64 MFMA, deterministic two-stage reduction).  mh_dense.hip
// Optional column map on Y: logical column j of Y is physical column ymap[j] of a panel of pitch ldy (0 = wb, no map).
void mh_gram(mh_context *ctx, size_t n, const double *x, uint32_t wa, const double *y, uint32_t wb, double *g, uint32_t ld, uint32_t ldy = 0, const uint32_t *ymap = nullptr);
// [X | W | P] * Ct -> out1 (first n1 columns), out2 (the rest); Ct row-major m x nc.  mh_dense.hip
void mh_combine(mh_context *ctx, size_t n, const double *x, uint32_t wx, const double *w, uint32_t ww, const double *p, uint32_t wp, const double *ct, uint32_t nc,
                double *out1, uint32_t n1, double *out2, bool accumulate = false, uint32_t ldx = 0, const uint32_t *xmap = nullptr, uint32_t ld1 = 0,
                const uint32_t *omap = nullptr, uint32_t col_begin = 0, uint32_t col_count = 0); // optional column maps on X (read) and out1 (write), pitches
                                                                                                // ldx / ld1; col_count > 0: only columns [col_begin, +col_count) of Ct
bool mh_tridiag_lowest_wide(mh_context *ctx, const double *d, const double *e, uint32_t m, uint32_t k, double *w, double *z, uint32_t ldz, double *work, int *info2,
                            double *quality_host); // orders 257 .. 768 (mh_dense.hip)
void mh_apply_q(mh_context *ctx, const double *a, const double *tau, uint32_t m, double *z, uint32_t ldz, uint32_t ncols); // mh_dense.hip: Z <- Q Z after mh_sytrd_small / mh_sytrd_wide (order <= 768)
bool mh_tridiag_lowest(mh_context *ctx, const double *d, const double *e, uint32_t m, uint32_t k, double *w, double *z, uint32_t ldz, double *ufac,
                       double *quality, double *lam_scratch); // mh_dense.hip: k lowest eigenpairs of a tridiagonal matrix (quality: 8 doubles, lam_scratch: k)
void mh_spd_inverse_small(mh_context *ctx, const double *a, uint32_t lda, uint32_t w, double *out, uint32_t ldo, int *info); // mh_dense.hip
void mh_potrf(mh_context *ctx, double *a, uint32_t ld, uint32_t w, int *info); // mh_dense.hip: lower Cholesky of any order without rocSOLVER (info: two ints)
void mh_potrf_small(mh_context *ctx, double *a, uint32_t w, int *info); // mh_dense.hip: lower Cholesky, order <= 128, one workgroup
void mh_potrf_small_inverse(mh_context *ctx, double *a, uint32_t w, int *info, const double *dscale, double *linv); // the same, then a <- diag(1 / dscale) a and linv <- a^-1, one launch
bool mh_sytrd_gave_up(mh_context *ctx);
void mh_sytrd_wide(mh_context *ctx, double *a, uint32_t m, double *d, double *e, double *tau); // orders up to 768, 48 workgroups over all XCDs (mh_dense.hip)
void mh_sytrd_small(mh_context *ctx, double *a, uint32_t m, double *d, double *e, double *tau, int variant = -1); // variant: -1 = the process default, 0 = one workgroup, 1 = several; mh_dense.hip: A (column-major, ld m, symmetric, full) -> D, E, tau, reflectors
// C = alpha op(A) op(B) + beta C, column-major, orders up to ~1000 (the Rayleigh-Ritz step's small matrices)
void mh_small_gemm(mh_context *ctx, bool ta, bool tb, uint32_t M, uint32_t N, uint32_t K, double alpha, const double *a, uint32_t lda, const double *b, uint32_t ldb, double beta, double *c,
                   uint32_t ldc);
void mh_short_product(mh_context *ctx, size_t n, const double *a, uint32_t m, const double *ct, uint32_t nc, double *out, double *partial, uint32_t slices);
void mh_pack_stacked(mh_context *ctx, const double *c1, uint32_t r1, const double *c2, uint32_t r2, uint32_t cols, double scale, double *ct);
void mh_pack_coefficients(mh_context *ctx, const double *c1, uint32_t n1, const double *c2, uint32_t n2, uint32_t m, uint32_t ld, double *ct);
void mh_spmm_mapped(mh_context *ctx, const BsrLevel &lvl, const double *vals9, const double *x, double *y, const double *mscal, double *y2, uint32_t w, uint32_t ldy,
                    uint32_t wreal, const uint32_t *omap, const double *res_theta = nullptr, double *res_out = nullptr, double *res_partial = nullptr,
                    const double *res_dinv = nullptr); // mh_spmm.hip: results into mapped columns of wider panels; optional residual epilogue (w <= 128):
                                                       // res_out (n x w) = A x - theta M x, res_partial (n_nodes x 2 x w) = per-node sums of r^2 and (M x)^2 (weighted by res_dinv)
bool mh_spmm_f32_cheb_step(mh_context *ctx, const BsrLevel &lvl, const float *d_in, float *d_out, float *r, float *x, const float *dinv, float c1, float c2,
                           uint32_t w); // mh_spmm.hip: product + Chebyshev step in one launch
void mh_spmm_f32(mh_context *ctx, const BsrLevel &lvl, const float *x, float *y, uint32_t w); // mh_spmm.hip
void mh_spmm_mixed(mh_context *ctx, const BsrLevel &lvl, const float *x, double *y, uint32_t w); // double A x of a float panel
const uint32_t *mh_identity_map(mh_context *ctx); // 0, 1, 2, ... (1 024 entries) on the device
void mh_spmm(mh_context *ctx, const BsrLevel &lvl, const double *vals9, const double *x, double *y, const double *mscal, double *y2, uint32_t w); // mh_spmm.hip
