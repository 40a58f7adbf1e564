// crc32_api.hip -- host side of the C ABI declared in include/sift_amd.h:
// per-device context (tables, per-stream scratch, staging), launch geometry, and
// the Digest / compute entry points (reference: sift-persist/src/crc.rs).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/sift_amd.h"
#include "crc32_device.h"
#include "gf2.h"

namespace sift {


namespace {

thread_local std::string g_last_error;

// Finish inside the tile kernel (last group to arrive combines) instead of a
// second launch.  sift_tune(SIFT_TUNE_SINGLE_LAUNCH, 0) selects the two-launch
// form for A/B measurement.
std::atomic<bool> g_single_launch{true};
// Tuning knobs of the batch entry points (sift_tune).
std::atomic<int> g_batch_cut{1};          // 0: every item stays on one wave (round-1 behaviour, A/B only)
std::atomic<int> g_batch_wave_max{0};     // bytes a single wave keeps; 0 = chosen from n (run_batch)
std::atomic<int> g_batch_piece_log2{kPieceMinLog2};
std::atomic<int> g_batch_plan{1};
std::atomic<int> g_host_copy_threads{4};  // threads that fill the pinned bounce ring (pageable host input)         // 0: always file through the wave kernel, even for few items

int fail_hip(hipError_t e, const char* what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return SIFT_ERR_HIP;
}

#define SIFT_HIP(call)                                  \
    do {                                                \
        hipError_t e__ = (call);                        \
        if (e__ != hipSuccess) return fail_hip(e__, #call); \
    } while (0)

// Workspace slot, in u32 words: residues (two-launch form) | pad | exchange
// words of the single-launch form: a top word and one word per shard of 32
// groups, each u64 on its own 128-B line.
constexpr size_t kExchangeWord = kMaxGroups + 32;
constexpr size_t kExchangeWords = 32 * (1 + kMaxGroups / 32);
constexpr size_t kSlotWords = kExchangeWord + kExchangeWords;
constexpr uint64_t kStagePiece = 32ull << 20;    // host->device staging piece
constexpr unsigned kResultSlots = 1024;          // pinned result words of the synchronous entry points
// A stream's small scratch block: compute workspace | accumulators of the piece
// kernel's self-planned form (kPlanMax records), all zero between calls.
constexpr size_t kPlanRecsOffset = (kSlotWords * sizeof(uint32_t) + 255) / 256 * 256;
constexpr size_t kStreamBlockBytes = kPlanRecsOffset + kPlanMax * sizeof(CutRecord);

// Scratch of sift_crc32_compute_device_static, the entry point that allocates
// nothing: one workspace slot in the code object's own device memory.  Each GPU
// that loads the library holds its own copy at its own address, until the process
// ends.  In host code the NAME below is only a handle for hipGetSymbolAddress
// (HIP_SYMBOL); the memory is reached through DeviceCtx::static_scratch, the
// per-device address that get_ctx() asks the runtime for.
constexpr size_t kStaticScratchWords = (kSlotWords + 63) / 64 * 64;  // whole 256-B lines
static __device__ __align__(256) uint32_t scratch[kStaticScratchWords];

const X2nTable& x2n() {
    static const X2nTable t = make_x2n_table();
    return t;
}

// Pageable host input (sift_crc32_compute_host): a few host threads copy the
// caller's bytes into a ring of pinned slots while the tile kernel reads the
// slots filled before IN PLACE over PCIe (pinned memory is device-addressable),
// the running state chained on the device.  Each byte crosses PCIe once, and no
// staging copy runs on the GPU; several memcpy threads are needed because one
// core copies slower than the link carries.
constexpr int kBounceSlots = 4;
constexpr size_t kBounceSlotBytes = 8ull << 20;
#ifndef SIFT_BOUNCE_MIN_KIB
#define SIFT_BOUNCE_MIN_KIB 512
#endif
#ifndef SIFT_SMALL_HOST_MAX_KIB
#define SIFT_SMALL_HOST_MAX_KIB 512
#endif
constexpr uint64_t kBounceMin = (uint64_t)SIFT_BOUNCE_MIN_KIB << 10;  // above the small path: host threads fill the pinned ring (the staged-copy path remains for SIFT_TUNE_HOST_COPY_THREADS = 0)
constexpr uint64_t kSmallHostMax = (uint64_t)SIFT_SMALL_HOST_MAX_KIB << 10;  // up to this the caller's thread copies into a pinned slot itself

class CopyCrew {
   public:
    explicit CopyCrew(int helpers) {
        for (int i = 0; i < helpers; ++i) threads_.emplace_back([this, i] { work(i + 1); });
    }
    ~CopyCrew() {
        {
            std::lock_guard<std::mutex> lock(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    // memcpy(dst, src, n) split over the helper threads; returns at once, wait()
    // returns when the copy is complete.  The caller is free in between (it
    // launches the kernel over the slot filled before).
    void begin(uint8_t* dst, const uint8_t* src, size_t n) {
        if (threads_.empty()) {
            std::memcpy(dst, src, n);
            return;
        }
        {
            std::lock_guard<std::mutex> lock(m_);
            dst_ = dst;
            src_ = src;
            n_ = n;
            pending_ = (int)threads_.size();
            ++generation_;
        }
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lock(m_);
        done_.wait(lock, [this] { return pending_ == 0; });
    }

   private:
    void share(int part, int parts) {  // part 1..parts
        --part;
        const size_t chunk = ((n_ + parts - 1) / parts + 4095) & ~size_t(4095);  // whole pages, never 0
        const size_t lo = std::min(n_, chunk * part), hi = std::min(n_, lo + chunk);
        if (hi > lo) std::memcpy(dst_ + lo, src_ + lo, hi - lo);
    }
    void work(int part) {
        unsigned seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lock(m_);
            cv_.wait(lock, [&] { return stop_ || generation_ != seen; });
            if (stop_) return;
            seen = generation_;
            lock.unlock();
            share(part, (int)threads_.size());
            lock.lock();
            if (--pending_ == 0) done_.notify_one();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    uint8_t* dst_ = nullptr;
    const uint8_t* src_ = nullptr;
    size_t n_ = 0;
    int pending_ = 0;
    unsigned generation_ = 0;
    bool stop_ = false;
};

// Device scratch of the calls issued on one stream.  Calls on one stream run in
// order on the device, so they can share a block; calls on different streams
// may overlap, so every stream gets blocks of its own (allocated on its first
// call, kept until sift_shutdown).  Nothing is handed out round-robin: there is
// no state that a call still in flight on another stream could be sharing.
struct StreamScratch {
    std::mutex m;                    // one call at a time issues its launch sequence on this stream's scratch
    std::atomic<uint32_t*> compute_ws{nullptr};  // kStreamBlockBytes, zero between calls (sift_crc32_compute_* with workspace NULL)
    uint8_t* batch = nullptr;        // batch entry points: head (kCutHeadBytes, zero between calls) | records | piece list
    size_t batch_bytes = 0;
    unsigned wide_calls = 0;         // wide-plan calls issued on this block (which of its two flags is this call's)
    // Pageable host descriptors of up to kDescStageBytes per call are copied by the
    // CPU into one of two pinned buffers which the kernels read in place (no copy
    // command: hipMemcpyAsync from pageable memory costs 100-300 us of fixed time
    // at these sizes).  A buffer is reused once the call that read it has finished.
    // Arrays of up to kDescSmallBytes (4096 items) go through a ring of 16 small
    // buffers in two halves with ONE event per half, recorded behind the call that
    // uses a half's last buffer and waited for before its first buffer is used again:
    // an event record costs the stream about 3 us, a third of a small batch call.
    uint8_t* h_desc = nullptr;       // 2 * kDescStageBytes + kDescSmallSlots * kDescSmallBytes, pinned
    uint8_t* d_desc = nullptr;       // the same memory as the device addresses it
    hipEvent_t desc_read[2] = {nullptr, nullptr};
    bool desc_used[2] = {false, false};
    unsigned desc_next = 0;
    hipEvent_t half_read[2] = {nullptr, nullptr};
    bool half_begun[2] = {false, false};   // a buffer of this half has been handed out since its last wait
    bool half_closed[2] = {false, false};  // ... and the event behind its last buffer has been recorded
    unsigned small_next = 0;
};
constexpr size_t kDescStageBytes = 1u << 20;
constexpr size_t kDescSmallBytes = 64u << 10;
constexpr unsigned kDescSmallSlots = 16;

// What a stream's scratch is filed under.  A hipStream_t names one stream -- except
// hipStreamPerThread, ONE handle value that names a different stream in every thread:
// calls made with it are keyed by (handle, calling thread), or two threads would run
// concurrently on one exchange block.  NULL and hipStreamLegacy are the same stream.
struct StreamKey {
    hipStream_t stream = nullptr;
    uint64_t thread = 0;  // 0 unless stream == hipStreamPerThread
    bool operator==(const StreamKey& o) const { return stream == o.stream && thread == o.thread; }
};
struct StreamKeyHash {
    size_t operator()(const StreamKey& k) const {
        return std::hash<void*>()(static_cast<void*>(k.stream)) ^ (size_t)(k.thread * 0x9E3779B97F4A7C15ull);
    }
};
uint64_t this_thread_id() {
    static std::atomic<uint64_t> next{1};
    thread_local const uint64_t id = next.fetch_add(1);
    return id;
}
StreamKey key_of(hipStream_t s) {
    if (s == hipStreamPerThread) return StreamKey{s, this_thread_id()};
    if (s == hipStreamLegacy) return StreamKey{nullptr, 0};
    return StreamKey{s, 0};
}
// Bumped whenever an entry leaves any per_stream map (sift_release_stream,
// sift_shutdown): invalidates the threads' one-entry lookup caches.
std::atomic<uint64_t> g_scratch_epoch{1};

struct DeviceCtx {
    int device = -1;
    int groups_max = 0;  // workgroups per launch = CUs (one 132 KiB-LDS workgroup per CU)
    DeviceTables* d_tabs = nullptr;
    uint32_t* static_scratch = nullptr;  // this device's copy of `scratch` (hipGetSymbolAddress)
    std::mutex scratch_mutex;            // guards the per_stream map only (entries have their own mutex)
    std::unordered_map<StreamKey, std::unique_ptr<StreamScratch>, StreamKeyHash> per_stream;
    // host-input staging (sift_crc32_compute_host)
    bool staging_ready = false;
    uint8_t* d_stage[2] = {nullptr, nullptr};
    uint32_t* d_state = nullptr;  // chained running state for staged pieces
    // Results of the synchronous entry points land in pinned host memory the
    // GPU writes directly (no device->host copy command): kResultSlots slots, 64 B
    // apart, one per calling thread (a thread is inside one synchronous call at a
    // time, so a slot is never shared by two calls in flight unless more than
    // kResultSlots threads are inside such calls at once).
    uint32_t* h_results = nullptr;
    hipStream_t copy_stream = nullptr, work_stream = nullptr;
    hipEvent_t copied[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
    // pinned bounce ring for pageable input
    uint8_t* h_bounce = nullptr;  // kBounceSlots * kBounceSlotBytes, pinned
    uint8_t* d_bounce = nullptr;  // the same memory as the device addresses it
    hipEvent_t bounce_read[kBounceSlots] = {};
    CopyCrew* crew = nullptr;
    std::mutex host_path_mutex;
    // measurement support (sift_profile_begin / _end): event pairs around the
    // tile kernel of each launch, recorded on the launch's own stream
    std::mutex prof_mutex;
    std::vector<hipEvent_t> prof_events;
    std::vector<uint8_t> prof_kind;  // per pair: 0 tile kernel, 1 batch (wave) kernel, 2 piece kernel
    int prof_used = 0;
    int prof_stride = 1, prof_seen = 0;
    std::atomic<bool> prof_on{false};
};

// Frees what a stream's entry holds.  The device is current and nothing that uses
// the blocks is still running.
void free_stream_scratch(StreamScratch& ss) {
    (void)hipFree(ss.compute_ws.exchange(nullptr));
    (void)hipFree(ss.batch);  // from hipMallocAsync: hipFree takes such blocks too
    ss.batch = nullptr;
    ss.batch_bytes = 0;
    if (ss.h_desc) (void)hipHostFree(ss.h_desc);
    ss.h_desc = ss.d_desc = nullptr;
    for (hipEvent_t& e : ss.desc_read)
        if (e) (void)hipEventDestroy(e), e = nullptr;
    for (hipEvent_t& e : ss.half_read)
        if (e) (void)hipEventDestroy(e), e = nullptr;
}

void destroy_ctx(DeviceCtx* c) {  // the device is current
    if (!c) return;
    (void)hipFree(c->d_tabs);
    (void)hipFree(c->d_state);
    if (c->h_results) (void)hipHostFree(c->h_results);
    delete c->crew;
    if (c->h_bounce) (void)hipHostFree(c->h_bounce);
    for (hipEvent_t e : c->bounce_read)
        if (e) (void)hipEventDestroy(e);
    for (auto& kv : c->per_stream) free_stream_scratch(*kv.second);
    c->per_stream.clear();
    for (hipEvent_t e : c->prof_events) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) {
        if (c->d_stage[i]) (void)hipFree(c->d_stage[i]);
        if (c->copied[i]) (void)hipEventDestroy(c->copied[i]);
        if (c->consumed[i]) (void)hipEventDestroy(c->consumed[i]);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->work_stream) (void)hipStreamDestroy(c->work_stream);
    delete c;
}

std::mutex g_ctx_mutex;
std::vector<DeviceCtx*> g_ctx;  // indexed by device; under g_ctx_mutex
// The same pointers where every call finds its context without a lock (a context
// lives from the device's first use to sift_shutdown).
constexpr int kMaxDevices = 64;
std::atomic<DeviceCtx*> g_ctx_fast[kMaxDevices];

void build_tables(DeviceTables* t) {
    const uint32_t y = gf2_xpow(x2n(), 8ull * kTileBytes);
    const uint32_t x32 = gf2_xpow(x2n(), 32);
    for (int k = 0; k < 4; ++k)
        for (uint32_t b = 0; b < 256; ++b) {
            t->Y[k * 256 + b] = gf2_mulmod(b << (8 * k), y);
            t->X[k * 256 + b] = gf2_mulmod(b << (8 * k), x32);
        }
    for (int k = 0; k < 4; ++k)
        for (uint32_t b = 0; b < 256; ++b) {
            t->Yt[4 * b + k] = t->Y[k * 256 + b];
            t->Xt[4 * b + k] = t->X[k * 256 + b];
        }
    const uint32_t y1 = gf2_xpow(x2n(), 8ull * 1024);
    for (int k = 0; k < 4; ++k)
        for (uint32_t b = 0; b < 256; ++b) t->Y1[k * 256 + b] = gf2_mulmod(b << (8 * k), y1);
    const uint32_t y256 = gf2_xpow(x2n(), 8ull * 256);
    for (int k = 0; k < 4; ++k)
        for (uint32_t b = 0; b < 256; ++b) t->Y256[k * 256 + b] = gf2_mulmod(b << (8 * k), y256);
    for (uint32_t b = 0; b < 256; ++b) t->T8[b] = crc_step_byte(0u, b);
    for (int j = 0; j < 4; ++j) {
        const uint32_t kj = gf2_xpow(x2n(), 128ull << j);
        for (int k = 0; k < 4; ++k)
            for (uint32_t b = 0; b < 256; ++b) t->K[j][k * 256 + b] = gf2_mulmod(b << (8 * k), kj);
    }
    uint32_t c1 = gf2_xpow(x2n(), 128ull * kThreads);
    for (int i = 0; i < 32; ++i) {
        t->C1[i] = c1;
        c1 = gf2_mulx(c1);
    }
    for (int sq = 0; sq < 32; ++sq) {
        uint32_t r = x2n().v[sq];
        for (int k = 0; k < 32; ++k) {
            t->X2[sq][k] = r;
            r = gf2_mulx(r);
        }
    }
    for (uint64_t h = 0; h < 16; ++h) t->XinvH[h] = gf2_xpow(x2n(), 0xFFFFFFFFull - 8ull * h);
    // the batch kernel's LDS image (layout: KiLane in crc32_kernels.hip)
    for (uint32_t b = 0; b < 256; ++b)
        for (uint32_t k = 0; k < 4; ++k) {
            for (uint32_t c = 0; c < 8; ++c) {
                t->batch_image[b * 64 + 4 * c + k] = t->Y256[k * 256 + b];             // block 0: G
                t->batch_image[b * 64 + 32 + 4 * c + k] = t->X[k * 256 + b];           //          X
                t->batch_image[16384 + b * 64 + 4 * c + k] = t->K[2][k * 256 + b];     // block 1: K2
            }
            for (uint32_t c = 0; c < 8; ++c) t->batch_image[32768 + b * 32 + 4 * c + k] = t->Y1[k * 256 + b];  // block 2: Y1
            for (uint32_t c = 0; c < 2; ++c) {
                t->batch_image[16384 + b * 64 + 32 + 4 * c + k] = t->K[1][k * 256 + b];  // K1
                t->batch_image[16384 + b * 64 + 56 + 4 * c + k] = t->K[3][k * 256 + b];  // K3
            }
            for (uint32_t c = 0; c < 4; ++c) t->batch_image[16384 + b * 64 + 40 + 4 * c + k] = t->K[0][k * 256 + b];  // K0
        }
    const uint32_t x128 = gf2_xpow(x2n(), 128);
    uint32_t p = kOne;
    for (int tdx = kThreads - 1; tdx >= 0; --tdx) {
        t->Kt[tdx] = p;
        p = gf2_mulmod(p, x128);
    }
}

// RAII device switch for calls that allocate or launch.
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && (prev == dev || hipSetDevice(dev) == hipSuccess))
            ok = true;
    }
    ~DeviceGuard() {
        if (ok && prev >= 0) (void)hipSetDevice(prev);
    }
};

int init_ctx(DeviceCtx* c, int device) {  // the device is current
    c->device = device;
    hipDeviceProp_t prop;
    SIFT_HIP(hipGetDeviceProperties(&prop, device));
    c->groups_max = std::max(1, std::min(prop.multiProcessorCount * kGroupsPerCu, kMaxGroups));
    std::vector<DeviceTables> h(1);
    build_tables(h.data());
    SIFT_HIP(hipMalloc(&c->d_tabs, sizeof(DeviceTables)));
    SIFT_HIP(hipMemcpy(c->d_tabs, h.data(), sizeof(DeviceTables), hipMemcpyHostToDevice));
    SIFT_HIP(hipMalloc(&c->d_state, 256));
    // This device's address of the static scratch array (this device is current).
    SIFT_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&c->static_scratch), HIP_SYMBOL(scratch)));
    SIFT_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_results), kResultSlots * 64, hipHostMallocDefault));
    // Once per device: the tables are in place before any stream (non-blocking ones
    // do not wait for the null stream's copies) launches a kernel that reads them.
    SIFT_HIP(hipDeviceSynchronize());
    return SIFT_OK;
}

int get_ctx(int device, DeviceCtx** out) {
    if (device >= 0 && device < kMaxDevices) {
        if (DeviceCtx* c = g_ctx_fast[device].load(std::memory_order_acquire)) {
            *out = c;
            return SIFT_OK;
        }
    }
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_last_error = std::string("hipGetDeviceCount: ") +
                       (e != hipSuccess ? hipGetErrorString(e) : "no devices");
        return SIFT_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) {
        g_last_error = "device index out of range";
        return SIFT_ERR_NO_DEVICE;
    }
    if ((int)g_ctx.size() < count) g_ctx.resize(count, nullptr);
    if (g_ctx[device]) {
        *out = g_ctx[device];
        return SIFT_OK;
    }
    // Build into a local and publish only on full success; the caller's current
    // device is restored on every path.
    DeviceGuard guard(device);
    if (!guard.ok) {
        g_last_error = "hipSetDevice failed";
        return SIFT_ERR_HIP;
    }
    DeviceCtx* c = new DeviceCtx;
    const int rc = init_ctx(c, device);
    if (rc != SIFT_OK) {
        destroy_ctx(c);
        return rc;
    }
    g_ctx[device] = c;
    if (device < kMaxDevices) g_ctx_fast[device].store(c, std::memory_order_release);
    *out = c;
    return SIFT_OK;
}

// x^(-8h) mod P for h = 0..15 (P is primitive: x^-k = x^(2^32-1-k)).
const uint32_t* xinv8() {
    static const struct Tab {
        uint32_t v[16];
        Tab() {
            for (uint64_t h = 0; h < 16; ++h) v[h] = gf2_xpow(x2n(), 0xFFFFFFFFull - 8ull * h);
        }
    } tab;
    return tab.v;
}

// Matrices of z^(2^s), z = x^(8 * tpg * kTileBytes), for the group-weighting
// step.  They depend on (tpg, groups) only, so calls of one size class reuse
// them; a small cache per calling thread (no lock on the call path) saves the
// ~10 GF(2) multiplies.
const CombineMats& combine_mats(int64_t tpg, int groups) {
    struct Entry {
        int64_t tpg = -1;
        int groups = -1;
        CombineMats m;
    };
    thread_local std::array<Entry, 8> cache;
    thread_local unsigned next = 0;
    for (auto& e : cache)
        if (e.tpg == tpg && e.groups == groups) return e.m;
    Entry& slot = cache[next++ % cache.size()];
    slot.tpg = tpg;
    slot.groups = groups;
    uint32_t z = gf2_xpow(x2n(), 8ull * (uint64_t)tpg * kTileBytes);
    for (int s = 0; s < kTreeLevels; ++s) {
        uint32_t r = z;
        for (int k = 0; k < 32; ++k) {
            slot.m.z[s][k] = r;
            r = gf2_mulx(r);
        }
        if ((1 << (s + 1)) < groups) z = gf2_mulmod(z, z);
    }
    return slot.m;
}

// Measurement: an event pair for the next launch of a kernel of `kind`, if
// sift_profile_begin asked for one (the stride counts tile-kernel launches only).
void prof_take(DeviceCtx* c, int kind, hipEvent_t* ev0, hipEvent_t* ev1) {
    *ev0 = *ev1 = nullptr;
    if (!c->prof_on.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lock(c->prof_mutex);
    if (!c->prof_on) return;
    if (kind == 0 && (c->prof_seen++ % c->prof_stride) != 0) return;
    if ((size_t)(2 * c->prof_used + 1) >= c->prof_events.size()) return;
    *ev0 = c->prof_events[2 * c->prof_used];
    *ev1 = c->prof_events[2 * c->prof_used + 1];
    c->prof_kind[c->prof_used] = (uint8_t)kind;
    ++c->prof_used;
}

// Geometry + launch of one compute(state, buf) over device memory.
int launch_compute(DeviceCtx* c, uint32_t state, const uint32_t* d_state_in, const void* d_buf,
                   uint64_t len, uint32_t xor_out, uint32_t* d_out, uint32_t* workspace,
                   hipStream_t stream, uint32_t tag = 0) {
    if (!d_out) {
        g_last_error = "d_out is NULL";
        return SIFT_ERR_INVALID_ARG;
    }
    if (len && !d_buf) {
        g_last_error = "d_buf is NULL with len > 0";
        return SIFT_ERR_INVALID_ARG;
    }
    const uint64_t addr = reinterpret_cast<uint64_t>(d_buf);
    const uint64_t v0 = addr >> 4, v1 = (addr + len) >> 4;
    const int64_t nvec = (int64_t)(v1 - v0);

    FinalArgs f;
    std::memset(&f, 0, sizeof(f));
    f.xor_out = xor_out;
    f.out = d_out;
    f.tag = tag;

    if (len < 32) {
        // Too short to hold the folded state safely inside whole vectors: one
        // lane walks the bytes from the starting register.
        f.state_imm = state;
        f.state_ptr = d_state_in;
        f.tail = static_cast<const uint8_t*>(d_buf);
        f.tail_len = (uint32_t)len;
        f.groups = 0;
        CombineMats none;
        std::memset(&none, 0, sizeof(none));
        launch_combine(workspace, f, none, stream);
        SIFT_HIP(hipGetLastError());
        return SIFT_OK;
    }

    const uint32_t head_skip = (uint32_t)(addr & 15u);
    f.tail = reinterpret_cast<const uint8_t*>(v1 << 4);
    f.tail_len = (uint32_t)((addr + len) & 15u);

    const int64_t ntile = (nvec + kTileVecs - 1) / kTileVecs;
    int64_t groups = std::min<int64_t>(ntile, c->groups_max);
    const int64_t tpg = (ntile + groups - 1) / groups;
    groups = (ntile + tpg - 1) / tpg;
    f.groups = (int32_t)groups;
    if ((uint64_t)tpg * kTileBytes >= (1ull << 31) - (1ull << 24)) {
        // a workgroup addresses its run through a signed 32-bit buffer offset
        g_last_error = "buffer too large for this device's workgroup count";
        return SIFT_ERR_INVALID_ARG;
    }


    TileArgs t;
    t.vbase = reinterpret_cast<const u32x4*>(v0 << 4);
    t.nvec = nvec;
    t.tiles_per_group = tpg;
    t.virtual_tiles = groups * tpg;
    t.pad_tiles = groups * tpg - ntile;
    t.tabs = c->d_tabs;
    t.partial = workspace;
    t.exchange = workspace + kExchangeWord;
    t.head_skip = head_skip;
    t.xinv_h = xinv8()[head_skip];
    t.state_ptr = d_state_in;
    t.state_fold = d_state_in ? 0u : (head_skip ? gf2_mulmod(state, t.xinv_h) : state);
    t.fuse_final = groups == 1 ? 1u : 0u;
    t.single_launch = (groups > 1 && g_single_launch) ? 1u : 0u;

    static const CombineMats no_mats = {};
    const CombineMats& m = groups > 1 ? combine_mats(tpg, (int)groups) : no_mats;

    // measurement: bracket this launch's tile kernel with an event pair
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    prof_take(c, 0, &ev0, &ev1);
    // Measurement (sift_profile_begin/_end): the event pair rides on the tile kernel's own
    // dispatch (hipExtLaunchKernelGGL: its start and end timestamps) -- what rocprofv3
    // reports for the launch -- instead of two marker packets around it, which put 2-3 us
    // of their own between the events and ~8 us of idle stream around every timed launch.
    // SIFT_PROF_EXT=0 brings the markers back.
    static const bool ext_events = [] {
        const char* e = getenv("SIFT_PROF_EXT");
        return !(e && *e == '0');
    }();
    if (ev0 && !ext_events) SIFT_HIP(hipEventRecord(ev0, stream));
    launch_tiles(t, f, m, (int)groups, tpg > 1, stream, ext_events ? ev0 : nullptr, ext_events ? ev1 : nullptr);
    SIFT_HIP(hipGetLastError());
    if (ev1 && !ext_events) SIFT_HIP(hipEventRecord(ev1, stream));
    if (groups > 1 && !t.single_launch) {
        launch_combine(workspace, f, m, stream);
        SIFT_HIP(hipGetLastError());
    }
    return SIFT_OK;
}

// Pinned result slots of the synchronous entry points: a thread leases one on its
// first synchronous call and gives it back when it exits, so no two threads ever
// poll one word.  More than kResultSlots threads alive inside the library at once
// get an error, not an alias.
struct SlotPool {
    std::mutex m;
    std::vector<int> free_slots;
    SlotPool() {
        for (int i = (int)kResultSlots - 1; i >= 0; --i) free_slots.push_back(i);
    }
};
SlotPool& slot_pool() {
    static SlotPool* pool = new SlotPool;  // leaked on purpose: threads may exit after static destruction
    return *pool;
}
struct SlotLease {
    int idx = -1;
    ~SlotLease() {
        if (idx < 0) return;
        SlotPool& p = slot_pool();
        std::lock_guard<std::mutex> lock(p.m);
        p.free_slots.push_back(idx);
    }
};
int result_slot(DeviceCtx* c, volatile uint32_t** out) {
    thread_local SlotLease lease;
    if (lease.idx < 0) {
        SlotPool& p = slot_pool();
        std::lock_guard<std::mutex> lock(p.m);
        if (p.free_slots.empty()) {
            g_last_error = "more than 1024 threads are using the synchronous entry points at once";
            return SIFT_ERR_TOO_MANY_THREADS;
        }
        lease.idx = p.free_slots.back();
        p.free_slots.pop_back();
    }
    *out = c->h_results + (size_t)lease.idx * 16;
    return SIFT_OK;
}

// A tag no other call in flight carries (never 0).
std::atomic<uint32_t> g_result_tags{0};
uint32_t next_tag() {
    uint32_t t = g_result_tags.fetch_add(1) + 1u;
    while (t == 0u) t = g_result_tags.fetch_add(1) + 1u;
    return t;
}

// Waits for the kernel's 8-byte store (tag << 32 | result) into the pinned slot.
// The store is the last thing the call's last kernel does, and the GPU's writes to
// coherent pinned memory are visible to the host as they land; polling for it
// returns several microseconds before hipStreamSynchronize would (completion
// signal, wake-up).  Short calls are what this is for: after kPollNs of polling
// the thread stops burning its core and waits for the stream instead, which also
// is where a failed launch or a faulting kernel surfaces with its error.
constexpr long long kPollNs = 60 * 1000;
int wait_result(hipStream_t stream, volatile uint32_t* slot, uint32_t tag, uint32_t* out) {
    volatile unsigned long long* word = reinterpret_cast<volatile unsigned long long*>(slot);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 1;; ++spin) {
        const unsigned long long v = *word;
        if ((uint32_t)(v >> 32) == tag) {
            *out = (uint32_t)v;
            return SIFT_OK;
        }
        if ((spin & 63u) == 0 &&
            std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() > kPollNs)
            break;
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    SIFT_HIP(hipStreamSynchronize(stream));
    const unsigned long long w = *word;
    if ((uint32_t)(w >> 32) != tag) {
        g_last_error = "the kernel finished without storing its result";
        return SIFT_ERR_HIP;
    }
    *out = (uint32_t)w;
    return SIFT_OK;
}

// True when `stream` is being captured into a graph.  The library's own scratch
// is tied to the stream a call is issued on; a captured call would bake that
// block into a graph that may later replay on any stream, beside eager calls
// on this one.  Entry points that would use library scratch refuse instead.
bool capturing(hipStream_t stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st != hipStreamCaptureStatusNone;
}

// The scratch entry of `stream` on this device (created empty on first use).  The
// calling thread remembers the last entry it looked up, so a loop of calls on one
// stream takes no lock here.
StreamScratch* stream_scratch(DeviceCtx* c, hipStream_t stream) {
    struct Last {
        DeviceCtx* c = nullptr;
        StreamKey key;
        uint64_t epoch = 0;
        StreamScratch* ss = nullptr;
    };
    thread_local Last last;
    const StreamKey key = key_of(stream);
    const uint64_t epoch = g_scratch_epoch.load(std::memory_order_acquire);
    if (last.ss && last.c == c && last.key == key && last.epoch == epoch) return last.ss;
    std::lock_guard<std::mutex> lock(c->scratch_mutex);
    std::unique_ptr<StreamScratch>& slot = c->per_stream[key];
    if (!slot) slot.reset(new StreamScratch);
    last.c = c;
    last.key = key;
    last.epoch = g_scratch_epoch.load(std::memory_order_acquire);
    last.ss = slot.get();
    return last.ss;
}

// The small scratch block of a stream (created zeroed on the stream's first call).
// ss->m is held.
int stream_block_held(StreamScratch* ss, hipStream_t stream, uint8_t** blk) {
    uint32_t* p = ss->compute_ws.load(std::memory_order_relaxed);
    if (!p) {
        SIFT_HIP(hipMalloc(&p, kStreamBlockBytes));
        // On the stream itself: its first kernel is ordered behind the clearing.  (A
        // plain hipMemset runs on the null stream and may still be running when it
        // returns; a non-blocking stream does not wait for it, and a clearing that
        // lands in the middle of the first call's exchange breaks every call after it.)
        hipError_t e = hipMemsetAsync(p, 0, kStreamBlockBytes, stream);
        if (e != hipSuccess) {
            (void)hipFree(p);
            return fail_hip(e, "hipMemsetAsync(workspace)");
        }
        ss->compute_ws.store(p, std::memory_order_release);
    }
    *blk = reinterpret_cast<uint8_t*>(p);
    return SIFT_OK;
}
int stream_block(StreamScratch* ss, hipStream_t stream, uint8_t** blk) {
    if (uint32_t* p = ss->compute_ws.load(std::memory_order_acquire)) {
        *blk = reinterpret_cast<uint8_t*>(p);
        return SIFT_OK;
    }
    std::lock_guard<std::mutex> lock(ss->m);
    return stream_block_held(ss, stream, blk);
}

// The compute workspace of `stream`.
int stream_workspace(DeviceCtx* c, hipStream_t stream, uint32_t** ws) {
    if (capturing(stream)) {
        g_last_error = "stream is being captured: pass a workspace of sift_crc32_workspace_bytes(), or use "
                       "sift_crc32_compute_device_static (the library's per-stream scratch cannot go into a graph)";
        return SIFT_ERR_INVALID_ARG;
    }
    uint8_t* blk = nullptr;
    const int rc = stream_block(stream_scratch(c, stream), stream, &blk);
    if (rc != SIFT_OK) return rc;
    *ws = reinterpret_cast<uint32_t*>(blk);
    return SIFT_OK;
}

// A caller's workspace: the exchange words are 8-byte atomics on 128-byte lines of
// their own, laid out from the start of the buffer.
int check_workspace(const void* workspace) {
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) {
        g_last_error = "workspace must be 256-byte aligned (as hipMalloc returns it)";
        return SIFT_ERR_INVALID_ARG;
    }
    return SIFT_OK;
}

int ensure_staging(DeviceCtx* c) {  // under host_path_mutex
    if (c->staging_ready) return SIFT_OK;
    for (int i = 0; i < 2; ++i) {
        if (!c->d_stage[i]) SIFT_HIP(hipMalloc(&c->d_stage[i], kStagePiece));
        if (!c->copied[i]) SIFT_HIP(hipEventCreateWithFlags(&c->copied[i], hipEventDisableTiming));
        if (!c->consumed[i]) SIFT_HIP(hipEventCreateWithFlags(&c->consumed[i], hipEventDisableTiming));
    }
    if (!c->copy_stream) SIFT_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!c->work_stream) SIFT_HIP(hipStreamCreateWithFlags(&c->work_stream, hipStreamNonBlocking));
    if (!c->h_bounce) {
        SIFT_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_bounce), kBounceSlots * kBounceSlotBytes,
                               hipHostMallocDefault));
        SIFT_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_bounce), c->h_bounce, 0));
    }
    for (int i = 0; i < kBounceSlots; ++i)
        if (!c->bounce_read[i]) SIFT_HIP(hipEventCreateWithFlags(&c->bounce_read[i], hipEventDisableTiming));
    if (!c->crew) c->crew = new CopyCrew(std::max(0, g_host_copy_threads.load()));
    c->staging_ready = true;  // set last: a partial failure above is retried, not skipped
    return SIFT_OK;
}

}  // namespace
}  // namespace sift

using namespace sift;

extern "C" {

const char* sift_strerror(int status) {
    switch (status) {
        case SIFT_OK: return "ok";
        case SIFT_ERR_READ: return "Cannot read from file";                       // container.rs:8
        case SIFT_ERR_WRITE: return "Cannot write to file";                       // container.rs:10
        case SIFT_ERR_READ_INVALID_CHUNK_SIZE: return "Read a chunk with an invalid chunk size";  // :12
        case SIFT_ERR_WRITE_INVALID_CHUNK_SIZE: return "Chunk size too large to be written";      // :14
        case SIFT_ERR_READ_CRC_MISMATCH: return "CRC mismatch when reading data";                 // :16
        case SIFT_ERR_NO_DEVICE: return "no usable HIP device";
        case SIFT_ERR_HIP: return "HIP runtime error";
        case SIFT_ERR_INVALID_ARG: return "invalid argument";
        case SIFT_ERR_TOO_MANY_THREADS: return "too many threads hold a result slot";
        default: return "unknown status";
    }
}

const char* sift_last_error(void) { return g_last_error.c_str(); }

int sift_init(int device) {
    DeviceCtx* c = nullptr;
    return get_ctx(device, &c);
}

void sift_shutdown(void) {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    for (auto& slot : g_ctx_fast) slot.store(nullptr, std::memory_order_release);
    g_scratch_epoch.fetch_add(1, std::memory_order_acq_rel);
    for (DeviceCtx* c : g_ctx) {
        if (!c) continue;
        DeviceGuard guard(c->device);
        (void)hipDeviceSynchronize();
        destroy_ctx(c);
    }
    g_ctx.clear();
}

int sift_reset_workspace(int device) {
    DeviceCtx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc != SIFT_OK) return rc;
    DeviceGuard guard(device);
    if (!guard.ok) {
        g_last_error = "hipSetDevice failed";
        return SIFT_ERR_HIP;
    }
    // No new stream entry and no launch sequence on an existing one while the
    // blocks are cleared: the map's lock, then every entry's.
    std::lock_guard<std::mutex> lock(c->scratch_mutex);
    std::vector<std::unique_lock<std::mutex>> held;
    for (auto& kv : c->per_stream) held.emplace_back(kv.second->m);
    SIFT_HIP(hipDeviceSynchronize());
    for (auto& kv : c->per_stream) {
        StreamScratch& ss = *kv.second;
        if (uint32_t* ws = ss.compute_ws.load()) SIFT_HIP(hipMemset(ws, 0, kStreamBlockBytes));
        if (ss.batch) SIFT_HIP(hipMemset(ss.batch, 0, kCutHeadBytes));
    }
    SIFT_HIP(hipMemset(c->static_scratch, 0, kStaticScratchWords * sizeof(uint32_t)));
    SIFT_HIP(hipDeviceSynchronize());  // the clearing may outlive hipMemset's return
    return SIFT_OK;
}

int sift_release_stream(int device, void* stream) {
    DeviceCtx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc != SIFT_OK) return rc;
    DeviceGuard guard(device);
    if (!guard.ok) {
        g_last_error = "hipSetDevice failed";
        return SIFT_ERR_HIP;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    std::unique_ptr<StreamScratch> ss;
    {
        std::lock_guard<std::mutex> lock(c->scratch_mutex);
        auto it = c->per_stream.find(key_of(st));
        if (it == c->per_stream.end()) return SIFT_OK;  // nothing kept for this stream
        ss = std::move(it->second);
        c->per_stream.erase(it);
        g_scratch_epoch.fetch_add(1, std::memory_order_acq_rel);
    }
    // the calls that used the blocks have to be done before the blocks go
    hipError_t e = hipStreamSynchronize(st);
    {
        std::lock_guard<std::mutex> lock(ss->m);
        free_stream_scratch(*ss);
    }
    if (e != hipSuccess) return fail_hip(e, "hipStreamSynchronize(released stream)");
    return SIFT_OK;
}

int sift_profile_begin(int device, int max_launches, int stride) {
    DeviceCtx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc != SIFT_OK) return rc;
    DeviceGuard guard(device);
    std::lock_guard<std::mutex> lock(c->prof_mutex);
    if (max_launches < 0) max_launches = 0;
    while ((int)c->prof_events.size() < 2 * max_launches) {
        hipEvent_t e;
        SIFT_HIP(hipEventCreate(&e));
        c->prof_events.push_back(e);
    }
    c->prof_kind.assign((size_t)max_launches, 0);
    c->prof_used = 0;
    c->prof_seen = 0;
    c->prof_stride = stride > 0 ? stride : 1;
    c->prof_on = true;
    return SIFT_OK;
}

int sift_profile_end(int device, double* tile_kernel_ms, int* launches) {
    return sift_profile_end_samples(device, tile_kernel_ms, launches, nullptr, 0);
}

int sift_profile_end_samples(int device, double* tile_kernel_ms, int* launches, float* samples_ms,
                             int max_samples) {
    double ms[3] = {0, 0, 0};
    int n[3] = {0, 0, 0};
    const int rc = sift_profile_end_kernels(device, ms, n, samples_ms, max_samples);
    if (tile_kernel_ms) *tile_kernel_ms = ms[0];
    if (launches) *launches = n[0];
    return rc;
}

int sift_profile_end_kernels(int device, double kernel_ms[3], int launches[3], float* tile_samples_ms,
                             int max_samples) {
    DeviceCtx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc != SIFT_OK) return rc;
    DeviceGuard guard(device);
    std::lock_guard<std::mutex> lock(c->prof_mutex);
    c->prof_on = false;
    double total[3] = {0, 0, 0};
    int count[3] = {0, 0, 0};
    for (int i = 0; i < c->prof_used; ++i) {
        SIFT_HIP(hipEventSynchronize(c->prof_events[2 * i + 1]));
        float ms = 0.f;
        SIFT_HIP(hipEventElapsedTime(&ms, c->prof_events[2 * i], c->prof_events[2 * i + 1]));
        const int kind = c->prof_kind[i] < 3 ? c->prof_kind[i] : 0;
        if (kind == 0 && tile_samples_ms && count[0] < max_samples) tile_samples_ms[count[0]] = ms;
        total[kind] += ms;
        ++count[kind];
    }
    for (int k = 0; k < 3; ++k) {
        if (kernel_ms) kernel_ms[k] = total[k];
        if (launches) launches[k] = count[k];
    }
    c->prof_used = 0;
    return SIFT_OK;
}

int sift_tune(int key, int value) {
    switch (key) {
        case SIFT_TUNE_SINGLE_LAUNCH:
            g_single_launch = value != 0;
            return SIFT_OK;
        case SIFT_TUNE_BATCH_CUT:
            g_batch_cut = value != 0;
            return SIFT_OK;
        case SIFT_TUNE_HOST_COPY_THREADS:
            if (value < 0 || value > 16) break;
            g_host_copy_threads = value;  // takes effect for devices not yet used for host input
            return SIFT_OK;
        case SIFT_TUNE_BATCH_PLAN:
            g_batch_plan = value != 0;
            return SIFT_OK;
        case SIFT_TUNE_BATCH_WAVE_MAX:
            if (value != 0 && value < 4096) break;
            g_batch_wave_max = value;
            return SIFT_OK;
        case SIFT_TUNE_BATCH_PIECE_LOG2:
            if (value < 14 || value > 24) break;  // whole 16 KiB tiles
            g_batch_piece_log2 = value;
            return SIFT_OK;
        default:
            g_last_error = "unknown tuning key";
            return SIFT_ERR_INVALID_ARG;
    }
    g_last_error = "tuning value out of range";
    return SIFT_ERR_INVALID_ARG;
}

size_t sift_crc32_workspace_bytes(void) { return kSlotWords * sizeof(uint32_t); }

int sift_crc32_compute_device(int device, uint32_t state, const uint32_t* d_state_in,
                              const void* d_buf, uint64_t len, uint32_t* d_out, void* workspace,
                              void* stream) {
    DeviceCtx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc != SIFT_OK) return rc;
    DeviceGuard guard(device);
    if (!guard.ok) {
        g_last_error = "hipSetDevice failed";
        return SIFT_ERR_HIP;
    }
    uint32_t* ws = static_cast<uint32_t*>(workspace);
    if (ws && (rc = check_workspace(ws)) != SIFT_OK) return rc;
    if (!ws && (rc = stream_workspace(c, static_cast<hipStream_t>(stream), &ws)) != SIFT_OK) return rc;
    return launch_compute(c, state, d_state_in, d_buf, len, 0u, d_out, ws,
                          static_cast<hipStream_t>(stream));
}

int sift_crc32_compute_device_static(int device, uint32_t state, const uint32_t* d_state_in,
                                     const void* d_buf, uint64_t len, uint32_t* d_out,
                                     void* stream) {
    // Allocates nothing: the context (tables, this device's scratch address) must
    // exist already, so sift_init(device) has to come first.
    DeviceCtx* c = nullptr;
    if (device >= 0 && device < kMaxDevices) c = g_ctx_fast[device].load(std::memory_order_acquire);
    if (!c || !c->static_scratch) {
        g_last_error = "sift_crc32_compute_device_static: call sift_init(device) first "
                       "(this entry point allocates nothing, so it cannot set the device up)";
        return SIFT_ERR_INVALID_ARG;
    }
    if (len > SIFT_STATIC_MAX_LEN) {
        g_last_error = "sift_crc32_compute_device_static: len exceeds SIFT_STATIC_MAX_LEN (256 GiB)";
        return SIFT_ERR_INVALID_ARG;
    }
    DeviceGuard guard(device);
    if (!guard.ok) {
        g_last_error = "hipSetDevice failed";
        return SIFT_ERR_HIP;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint32_t* ws = c->static_scratch;
    // The array keeps whatever the last call on this GPU left.  The kernels leave
    // the accumulator and the arrival counter at zero when they finish, but a call
    // that was aborted does not: clear the exchange words on the stream first.
    // By a kernel, not hipMemsetAsync: captured into a HIP graph, the memset node gave
    // wrong results from the second replay on (ROCm 7.2, gfx950; the kernel form and
    // the eager memset are both fine).
    launch_clear(ws + kExchangeWord, (int)kExchangeWords, s);
    SIFT_HIP(hipGetLastError());
    return launch_compute(c, state, d_state_in, d_buf, len, 0u, d_out, ws, s);
}

int sift_crc32_compute_device_sync(int device, uint32_t state, const void* d_buf, uint64_t len,
                                   uint32_t* out, void* stream) {
    if (!out) {
        g_last_error = "out is NULL";
        return SIFT_ERR_INVALID_ARG;
    }
    DeviceCtx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc != SIFT_OK) return rc;
    DeviceGuard guard(device);
    if (!guard.ok) {
        g_last_error = "hipSetDevice failed";
        return SIFT_ERR_HIP;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint32_t* ws = nullptr;
    if ((rc = stream_workspace(c, s, &ws)) != SIFT_OK) return rc;
    volatile uint32_t* res = nullptr;
    if ((rc = result_slot(c, &res)) != SIFT_OK) return rc;
    const uint32_t tag = next_tag();
    rc = launch_compute(c, state, nullptr, d_buf, len, 0u, const_cast<uint32_t*>(res), ws, s, tag);
    if (rc != SIFT_OK) return rc;
    return wait_result(s, res, tag, out);
}

int sift_crc32_compute_host(int device, uint32_t state, const void* buf, uint64_t len,
                            uint32_t* out) {
    if (!out || (len && !buf)) {
        g_last_error = "NULL pointer";
        return SIFT_ERR_INVALID_ARG;
    }
    DeviceCtx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc != SIFT_OK) return rc;
    DeviceGuard guard(device);
    if (!guard.ok) {
        g_last_error = "hipSetDevice failed";
        return SIFT_ERR_HIP;
    }
    std::lock_guard<std::mutex> lock(c->host_path_mutex);
    rc = ensure_staging(c);
    if (rc != SIFT_OK) return rc;

    const uint8_t* src = static_cast<const uint8_t*>(buf);
    uint32_t* ws = nullptr;
    if ((rc = stream_workspace(c, c->work_stream, &ws)) != SIFT_OK) return rc;
    volatile uint32_t* res = nullptr;
    if ((rc = result_slot(c, &res)) != SIFT_OK) return rc;
    const uint32_t tag = next_tag();  // carried by the call's last launch

    // Pinned (hipHostMalloc / hipHostRegister) input is device-addressable: the
    // tile kernel streams it over PCIe directly, each byte crossing once, with
    // no staging copy.  Pageable input is staged below.
    if (len >= 4096) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, buf) == hipSuccess && attr.type == hipMemoryTypeHost &&
            attr.devicePointer != nullptr) {
            rc = launch_compute(c, state, nullptr, attr.devicePointer, len, 0u,
                                const_cast<uint32_t*>(res), ws, c->work_stream, tag);
            if (rc != SIFT_OK) return rc;
            return wait_result(c->work_stream, res, tag, out);
        }
        (void)hipGetLastError();  // an unregistered pointer is not an error here
    }
    // Small pageable input: this thread copies it into the first pinned slot and the
    // kernel reads it there -- no copy command (a pageable hipMemcpyAsync costs tens of
    // microseconds before a byte moves).
    if (len <= kSmallHostMax) {
        if (len) std::memcpy(c->h_bounce, src, len);
        rc = launch_compute(c, state, nullptr, c->d_bounce, len, 0u, const_cast<uint32_t*>(res), ws,
                            c->work_stream, tag);
        if (rc != SIFT_OK) return rc;
        return wait_result(c->work_stream, res, tag, out);
    }
    // Running state lives in c->d_state between pieces; piece 0 starts from
    // the immediate.  Empty input still runs one (empty) launch so that the
    // result provably comes from the device path.
    uint64_t off = 0;
    int piece = 0;
    if (len >= kBounceMin && g_host_copy_threads.load() > 0) {
        // Pageable input, large: host threads fill pinned slots, the kernel reads
        // each slot in place; slot k % kBounceSlots is refilled once the kernel
        // that read it has finished.
        // Piece k's copy runs on the helper threads while this thread launches the
        // kernel over piece k-1.
        // Piece sizes ramp up 1, 2, 4, 8 MiB and taper down again towards the end: the
        // first copy and the last kernel are the two steps nothing overlaps.
        auto piece_bytes = [&](int k, uint64_t at) -> uint64_t {
            const uint64_t left = len - at, mib = 1ull << 20;
            if (left <= mib) return left;
            const uint64_t grow = mib << std::min(k, 3);
            const uint64_t half = std::max<uint64_t>(mib, (left / 2 + mib - 1) / mib * mib);
            return std::min<uint64_t>(std::min<uint64_t>(kBounceSlotBytes, grow), half);
        };
        auto slot_of = [&](int k) { return (size_t)(k % kBounceSlots) * kBounceSlotBytes; };
        auto start_copy = [&](int k, uint64_t at) -> hipError_t {
            if (k >= kBounceSlots) {  // the kernel that read this slot last must be done
                hipError_t e = hipEventSynchronize(c->bounce_read[k % kBounceSlots]);
                if (e != hipSuccess) return e;
            }
            c->crew->begin(c->h_bounce + slot_of(k), src + at, piece_bytes(k, at));
            return hipSuccess;
        };
        hipError_t e = start_copy(0, 0);
        while (e == hipSuccess) {
            const uint64_t n = piece_bytes(piece, off);
            const bool last = off + n >= len;
            c->crew->wait();  // piece `piece` is in its slot
            if (!last && (e = start_copy(piece + 1, off + n)) != hipSuccess) break;
            rc = launch_compute(c, state, piece ? c->d_state : nullptr, c->d_bounce + slot_of(piece), n, 0u,
                                last ? const_cast<uint32_t*>(res) : c->d_state, ws, c->work_stream, last ? tag : 0u);
            if (rc != SIFT_OK) break;
            if ((e = hipEventRecord(c->bounce_read[piece % kBounceSlots], c->work_stream)) != hipSuccess) break;
            off += n;
            ++piece;
            if (last) break;
        }
        if (rc != SIFT_OK || e != hipSuccess) {
            c->crew->wait();
            (void)hipStreamSynchronize(c->work_stream);
            return rc != SIFT_OK ? rc : fail_hip(e, "pinned bounce ring");
        }
        return wait_result(c->work_stream, res, tag, out);
    }
    // An error in the middle leaves copies from the caller's buffer in flight:
    // drain both streams before handing the buffer back.
    auto bail = [&](int code) {
        (void)hipStreamSynchronize(c->copy_stream);
        (void)hipStreamSynchronize(c->work_stream);
        return code;
    };
#define SIFT_STAGE(call)                                          \
    do {                                                          \
        hipError_t e__ = (call);                                  \
        if (e__ != hipSuccess) return bail(fail_hip(e__, #call)); \
    } while (0)
    do {
        const uint64_t n = std::min<uint64_t>(kStagePiece, len - off);
        const int b = piece & 1;
        if (piece >= 2) SIFT_STAGE(hipStreamWaitEvent(c->copy_stream, c->consumed[b], 0));
        if (n)
            SIFT_STAGE(hipMemcpyAsync(c->d_stage[b], src + off, n, hipMemcpyHostToDevice,
                                      c->copy_stream));
        SIFT_STAGE(hipEventRecord(c->copied[b], c->copy_stream));
        SIFT_STAGE(hipStreamWaitEvent(c->work_stream, c->copied[b], 0));
        const bool last = off + n >= len;
        rc = launch_compute(c, state, piece ? c->d_state : nullptr, c->d_stage[b], n, 0u,
                            last ? const_cast<uint32_t*>(res) : c->d_state, ws, c->work_stream, last ? tag : 0u);
        if (rc != SIFT_OK) return bail(rc);
        SIFT_STAGE(hipEventRecord(c->consumed[b], c->work_stream));
        off += n;
        ++piece;
    } while (off < len);
#undef SIFT_STAGE
    return wait_result(c->work_stream, res, tag, out);
}

void sift_digest_new(sift_digest* d, int device) {
    d->value = SIFT_CRC_INIT;  // crc.rs:42
    d->device = device;
}

int sift_digest_update(sift_digest* d, const void* buf, uint64_t len) {
    uint32_t v = 0;
    int rc = sift_crc32_compute_host(d->device, d->value, buf, len, &v);  // crc.rs:46
    if (rc == SIFT_OK) d->value = v;
    return rc;
}

int sift_digest_update_device(sift_digest* d, const void* d_buf, uint64_t len, void* stream) {
    uint32_t v = 0;
    int rc = sift_crc32_compute_device_sync(d->device, d->value, d_buf, len, &v, stream);
    if (rc == SIFT_OK) d->value = v;
    return rc;
}

uint32_t sift_digest_finalize(const sift_digest* d) { return d->value ^ 0xFFFFFFFFu; }  // crc.rs:51

uint32_t sift_crc32_concat(uint32_t state_a, uint32_t state_b0, uint64_t len_b) {
    return gf2_mulmod(state_a, gf2_xpow(x2n(), 8ull * len_b)) ^ state_b0;
}

namespace {

// Where a descriptor array lives.
enum class Where { Host, Pinned, Device };
Where where_is(const void* p, const void** dev_ptr) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) == hipSuccess) {
        if (attr.type == hipMemoryTypeDevice) {
            *dev_ptr = p;
            return Where::Device;
        }
        if (attr.type == hipMemoryTypeHost && attr.devicePointer) {
            *dev_ptr = attr.devicePointer;
            return Where::Pinned;
        }
    }
    (void)hipGetLastError();  // a plain host pointer is not an error here
    return Where::Host;
}

// The batch entry points.  Two launches on `stream`:
//   1. crc32_batch_kernel   one wave per item; an item whose CRC span is long is
//                           filed (record + pieces) instead of walked
//   2. crc32_pieces_kernel  one workgroup per run of filed pieces (tile algorithm);
//                           the run that completes a record gives the item its verdict
// The lengths that decide what is long are read on the device (from `lens`, or
// from each chunk's header in the verify modes), so it makes no difference
// whether the descriptors came from the host or were already in HBM.
int run_batch(int device, const void* d_base, const uint64_t* offsets, const uint64_t* lens,
              uint32_t n, uint32_t state, uint32_t xor_out, int verify, uint32_t* d_out,
              uint32_t* d_data_len, hipStream_t stream) {
    if (n == 0) return SIFT_OK;
    if (!offsets || !lens || !d_out || !d_base) {
        g_last_error = "NULL pointer";
        return SIFT_ERR_INVALID_ARG;
    }
    if (n >= 0x80000000u) {
        g_last_error = "too many buffers in one batch";
        return SIFT_ERR_INVALID_ARG;
    }
    DeviceCtx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc != SIFT_OK) return rc;
    DeviceGuard guard(device);
    if (!guard.ok) {
        g_last_error = "hipSetDevice failed";
        return SIFT_ERR_HIP;
    }
    // Descriptors: device arrays are used in place; host arrays are copied or read in
    // place as described where the copy is made, below.
    const void *dv_off = nullptr, *dv_len = nullptr;
    const Where w_off = where_is(offsets, &dv_off), w_len = where_is(lens, &dv_len);
    if ((w_off == Where::Device) != (w_len == Where::Device)) {
        g_last_error = "offsets and lens must both be host arrays or both be device arrays";
        return SIFT_ERR_INVALID_ARG;
    }

    BatchArgs b;
    std::memset(&b, 0, sizeof(b));
    b.base = static_cast<const uint8_t*>(d_base);
    b.n = n;
    b.state = state;
    b.state_m32 = gf2_mulmod(verify ? 0xFFFFFFFFu : state, gf2_xpow(x2n(), 0xFFFFFFFFull - 32ull));
    b.xor_out = xor_out;
    b.verify = (uint32_t)verify;
    b.chunk_limit = SIFT_CHUNK_SIZE_LIMIT;
    b.out = d_out;
    b.data_len = d_data_len;
    b.tabs = c->d_tabs;

    // Scratch of the cutting pass.  The host cannot see the lengths, so the
    // capacities come from what fits in the allocation `d_base` points into:
    // filed spans are longer than wave_max, so non-overlapping items give at most
    // extent / wave_max records, and a record of body B has at most B / piece + 1
    // pieces.  Items that overlap can exceed this; a span that finds the arrays
    // full is walked by its own wave instead (same bits, slower).
    bool cut = g_batch_cut.load() != 0;
    const int tuned_wave_max = g_batch_wave_max.load(), piece_log2 = g_batch_piece_log2.load();
    // The longest span a wave keeps: from the lengths themselves (crc32_device.h, wave_max_of),
    // here if the host can read them, else by the kernels.
    const bool host_lens = w_len != Where::Device;
    const uint32_t sample_stride = wave_sample_stride(n);
    const bool wave_adapt = tuned_wave_max <= 0 && !host_lens;
    uint32_t wave_max = tuned_wave_max > 0 ? (uint32_t)tuned_wave_max : kWaveMin;
    if (tuned_wave_max <= 0 && host_lens && n) {
        uint32_t cls[4] = {0, 0, 0, 0};
        for (uint32_t k = 0; k < std::min(n, kWaveSample); ++k) ++cls[wave_class(lens[wave_sample_item(k, n)])];
        wave_max = wave_max_of(cls[1], cls[2], cls[3], sample_stride);
    }
    // Lengths the host can read (host or pinned arrays), not too many of them: if no
    // item can hold a span longer than wave_max, nothing will be filed and the call is
    // the wave kernel alone -- no scratch, no second launch (6 us of an otherwise 8 us call).
    if (cut && host_lens && n > kPlanMax && n <= 4096u) {
        uint64_t longest = 0;
        for (uint32_t i = 0; i < n; ++i) longest = std::max(longest, lens[i]);
        if (longest <= wave_max) cut = false;  // a span is never longer than its item
    }
    // Few items (and not the document mode, whose second chunk needs the wave
    // kernel): no filing pass, the piece kernel plans for itself.
    const bool plan = cut && n <= kPlanMax && verify != 2 && g_batch_plan.load() != 0;
    // Up to kWideMax items (same exception): the batch kernel leaves the long spans alone and
    // the piece kernel finds them itself -- no filing, no piece list.
    const bool wide = cut && !plan && n <= kWideMax && verify != 2 && g_batch_plan.load() != 0;
    uint64_t extent = 1ull << 30;
    if (cut && !plan && !wide) {  // (the capacities below are the filed form's)
        hipDeviceptr_t abase = nullptr;
        size_t asize = 0;
        if (hipMemGetAddressRange(&abase, &asize, const_cast<void*>(d_base)) == hipSuccess && asize) {
            extent = reinterpret_cast<uint64_t>(abase) + asize - reinterpret_cast<uint64_t>(d_base);
        } else {
            (void)hipGetLastError();
        }
    }
    // (filed spans are longer than the threshold, at least kWaveMin when the kernels work it out;
    // at most a million records unless the threshold itself allows more)
    const uint64_t cap_recs =
        std::min<uint64_t>({(uint64_t)n, extent / (wave_adapt ? kWaveMin : wave_max) + 1,
                            std::max<uint64_t>(extent / (256u << 10) + 1, 1u << 20)});
    const uint64_t cap_pieces = std::min<uint64_t>((extent >> piece_log2) + cap_recs + 1, 0x7FFFFFFFull);
    hipStream_t st = stream;
    if (capturing(st)) {
        g_last_error = "the batch entry points cannot be captured into a graph (their scratch belongs to the stream)";
        return SIFT_ERR_INVALID_ARG;
    }
    // One batch call at a time issues its launch sequence on a stream's scratch
    // (shared by that stream's calls in stream order); calls on other streams of
    // the device are not held up.
    StreamScratch& ss = *stream_scratch(c, st);
    std::lock_guard<std::mutex> lock(ss.m);
    if (plan) {
        uint8_t* blk = nullptr;
        rc = stream_block_held(&ss, st, &blk);
        if (rc != SIFT_OK) return rc;
        b.cut.recs = reinterpret_cast<CutRecord*>(blk + kPlanRecsOffset);
        b.cut.wave_max = wave_max;
        b.cut.piece_min_log2 = (uint32_t)piece_log2;
    } else if (cut) {
        const size_t need = wide ? kCutHeadBytes : kCutHeadBytes + sizeof(CutRecord) * cap_recs + 4 * cap_pieces;
        if (ss.batch_bytes < need) {
            // grow-only; the new block is published before the old one is freed (in
            // stream order, behind the calls that used it), so a failure in between
            // leaves nothing dangling
            const size_t want = std::max<size_t>(need + need / 2, 512u << 10);
            uint8_t* fresh = nullptr;
            SIFT_HIP(hipMallocAsync(reinterpret_cast<void**>(&fresh), want, st));
            // the head is zero between calls: counters, flags, the wide plan's accumulators
            launch_clear(reinterpret_cast<uint32_t*>(fresh), (int)(kCutHeadBytes / 4), st);
            ss.wide_calls = 0;
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) {
                (void)hipFreeAsync(fresh, st);
                return fail_hip(e, "launch_clear(batch scratch)");
            }
            uint8_t* old = ss.batch;
            ss.batch = fresh;
            ss.batch_bytes = want;
            if (old) SIFT_HIP(hipFreeAsync(old, st));
        }
        uint8_t* blk = ss.batch;
        if (wide) {
            // flags take turns: this call's, and the next wide call's (cleared by this call's batch kernel)
            uint32_t* flags = reinterpret_cast<uint32_t*>(blk + kCutFlagsOffset);
            b.cut.long_flag = flags + (ss.wide_calls & 1u);
            b.cut.clear_flag = flags + ((ss.wide_calls + 1u) & 1u);
            ++ss.wide_calls;
            b.cut.recs = reinterpret_cast<CutRecord*>(blk + kCutWideOffset);  // pairs of words (crc32_pieces_kernel<2>)
        } else {
            b.cut.counts = reinterpret_cast<uint32_t*>(blk);
            b.cut.left = reinterpret_cast<uint32_t*>(blk + 64);
            b.cut.pieces = reinterpret_cast<unsigned long long*>(blk + 128);
            b.cut.recs = reinterpret_cast<CutRecord*>(blk + kCutHeadBytes);
            b.cut.piece_rec = reinterpret_cast<uint32_t*>(blk + kCutHeadBytes + sizeof(CutRecord) * cap_recs);
            b.cut.cap_recs = (uint32_t)cap_recs;
            b.cut.cap_pieces = (uint32_t)cap_pieces;
        }
        b.cut.wave_max = wave_max;
        b.cut.wave_adapt = wave_adapt ? 1u : 0u;
        b.cut.piece_min_log2 = (uint32_t)piece_log2;
    }
    // Few items with descriptors the host can read: they go into the kernel arguments.
    InlineDesc inl;
    const bool use_inline = plan && host_lens && w_off != Where::Device;
    if (use_inline) {
        for (uint32_t i = 0; i < n; ++i) {
            inl.v[2 * i] = offsets[i];
            inl.v[2 * i + 1] = lens[i];
        }
        for (uint32_t i = 2 * n; i < 2 * kPlanMax; ++i) inl.v[i] = 0;
    }
    // Host descriptor arrays of up to kDescStageBytes (64 K items when both are host
    // arrays) are COPIED before the call returns, pageable and pinned alike, so the
    // caller may refill or free them at once; larger pinned arrays are read in place
    // by the GPU (they must stay unchanged until `stream` has passed the call),
    // larger pageable ones go through a stream-ordered allocation.
    const size_t host_bytes = (w_off != Where::Device ? 8ull * n : 0) + (w_len != Where::Device ? 8ull * n : 0);
    const bool stage_small = !use_inline && host_bytes && host_bytes <= kDescStageBytes;
    const bool copy_off = !use_inline && (stage_small ? w_off != Where::Device : w_off == Where::Host);
    const bool copy_len = !use_inline && (stage_small ? w_len != Where::Device : w_len == Where::Host);
    const size_t copy_bytes = (copy_off ? 8ull * n : 0) + (copy_len ? 8ull * n : 0);
    // Whatever happens after this point, the bookkeeping of the descriptor buffers
    // must follow what was actually queued: `finish` runs on every exit.
    uint8_t* d_desc = nullptr;   // stream-ordered allocation (large descriptor arrays)
    int stage_buf = -1;          // or one of the stream's two pinned buffers (up to 1 MiB)
    int small_slot = -1;         // or one of its ring of small ones
    bool launched = false;       // a kernel that reads the descriptors may be queued
    struct Finish {
        StreamScratch& ss;
        hipStream_t st;
        uint8_t*& d_desc;
        int& stage_buf;
        int& small_slot;
        bool& launched;
        ~Finish() {
            if (d_desc) (void)hipFreeAsync(d_desc, st);  // stream order: behind the kernels that read it
            if (stage_buf >= 0 && launched) {
                if (hipEventRecord(ss.desc_read[stage_buf], st) == hipSuccess) {
                    ss.desc_used[stage_buf] = true;
                } else {
                    (void)hipStreamSynchronize(st);  // no event to wait for later: wait now
                    ss.desc_used[stage_buf] = false;
                }
            }
            if (small_slot >= 0 && small_slot % (int)(kDescSmallSlots / 2) == (int)(kDescSmallSlots / 2) - 1) {
                const int half = small_slot / (int)(kDescSmallSlots / 2);
                // not recorded (error, or nothing launched): half_closed stays false and the
                // next round of this half waits for the stream instead
                if (launched && hipEventRecord(ss.half_read[half], st) == hipSuccess) ss.half_closed[half] = true;
            }
        }
    } finish{ss, st, d_desc, stage_buf, small_slot, launched};
    if (copy_bytes && copy_bytes <= kDescStageBytes) {
        if (!ss.h_desc) {
            uint8_t* h = nullptr;
            SIFT_HIP(hipHostMalloc(reinterpret_cast<void**>(&h),
                                   2 * kDescStageBytes + kDescSmallSlots * kDescSmallBytes, hipHostMallocDefault));
            ss.h_desc = h;
            SIFT_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&ss.d_desc), h, 0));
            for (int i = 0; i < 2; ++i) {
                SIFT_HIP(hipEventCreateWithFlags(&ss.desc_read[i], hipEventDisableTiming));
                SIFT_HIP(hipEventCreateWithFlags(&ss.half_read[i], hipEventDisableTiming));
            }
        }
        size_t at;  // where the buffer starts in h_desc / d_desc
        if (copy_bytes <= kDescSmallBytes) {
            const int slot = (int)(ss.small_next % kDescSmallSlots);
            const int half = slot / (int)(kDescSmallSlots / 2);
            if (slot % (int)(kDescSmallSlots / 2) == 0) {  // entering a half: its last round must have been read
                if (ss.half_begun[half]) {
                    if (ss.half_closed[half]) SIFT_HIP(hipEventSynchronize(ss.half_read[half]));
                    else SIFT_HIP(hipStreamSynchronize(st));  // a call of that round failed before its record
                }
                ss.half_begun[half] = true;
                ss.half_closed[half] = false;
            }
            ++ss.small_next;
            small_slot = slot;
            at = 2 * kDescStageBytes + (size_t)small_slot * kDescSmallBytes;
        } else {
            const int buf = (int)(ss.desc_next & 1u);
            if (ss.desc_used[buf]) SIFT_HIP(hipEventSynchronize(ss.desc_read[buf]));
            ss.desc_used[buf] = false;
            ++ss.desc_next;
            stage_buf = buf;
            at = (size_t)stage_buf * kDescStageBytes;
        }
        uint8_t* hp = ss.h_desc + at;
        const uint8_t* dp = ss.d_desc + at;
        if (copy_off) {
            std::memcpy(hp, offsets, 8ull * n);
            dv_off = dp;
            hp += 8ull * n;
            dp += 8ull * n;
        }
        if (copy_len) {
            std::memcpy(hp, lens, 8ull * n);
            dv_len = dp;
        }
    } else if (copy_bytes) {
        SIFT_HIP(hipMallocAsync(reinterpret_cast<void**>(&d_desc), copy_bytes, st));
        uint8_t* d_next = d_desc;
        if (copy_off) {
            // pageable source: the runtime has consumed it when the call returns
            SIFT_HIP(hipMemcpyAsync(d_next, offsets, 8ull * n, hipMemcpyHostToDevice, st));
            dv_off = d_next;
            d_next += 8ull * n;
        }
        if (copy_len) {
            SIFT_HIP(hipMemcpyAsync(d_next, lens, 8ull * n, hipMemcpyHostToDevice, st));
            dv_len = d_next;
        }
    }
    b.offsets = static_cast<const uint64_t*>(dv_off);
    b.lens = static_cast<const uint64_t*>(dv_len);
    // Items a wave takes at a time (one per lane): a big batch pays descriptor and
    // header latency once per 64 items, a smaller one keeps about four turns per wave
    // so that the chip stays balanced (items differ in length), a small one spreads
    // one item per wave.
    const int waves_per_group = kThreads / 64;
    const uint64_t chip_waves = (uint64_t)c->groups_max * waves_per_group;
    uint32_t per_wave = 1;
    while (per_wave < 64u && (uint64_t)per_wave * 2 * 4 * chip_waves <= n) per_wave *= 2;
    b.items_per_wave = per_wave;
    const uint64_t wave_chunks = ((uint64_t)n + per_wave - 1) / per_wave;
    // turns are dealt across the workgroups first (crc32_batch_kernel): a small batch spreads over the chip
    // (descriptors read in place in pinned host memory: whole 64-byte lines a workgroup for a
    // small batch, whole 128-byte lines -- what an XCD's L2 fetches -- for a large one)
    b.deal_log2 = (w_off != Where::Device || w_len != Where::Device) && !use_inline ? (n <= 4096u ? 3u : 4u) : 0u;
    const int groups = (int)std::min<uint64_t>((wave_chunks + (1u << b.deal_log2) - 1) >> b.deal_log2,
                                               (uint64_t)c->groups_max);
    launched = true;  // from here on a kernel may be queued even if a later step fails
    hipEvent_t e0 = nullptr, e1 = nullptr;  // measurement: pairs that ride on the dispatches
    if (plan) {
        prof_take(c, 2, &e0, &e1);
        launch_pieces(b, c->groups_max, 1, use_inline ? &inl : nullptr, st, e0, e1);
        SIFT_HIP(hipGetLastError());
    } else {
        prof_take(c, 1, &e0, &e1);
        launch_batch(b, groups, st, e0, e1);
        SIFT_HIP(hipGetLastError());
        if (cut) {
            prof_take(c, 2, &e0, &e1);
            launch_pieces(b, c->groups_max, wide ? 2 : 0, nullptr, st, e0, e1);
            SIFT_HIP(hipGetLastError());
        }
    }
    return SIFT_OK;
}

}  // namespace

int sift_crc32_batch_device(int device, uint32_t state, uint32_t xor_out, const void* d_base,
                            const uint64_t* offsets, const uint64_t* lens, uint32_t n,
                            uint32_t* d_out, void* stream) {
    return run_batch(device, d_base, offsets, lens, n, state, xor_out, 0, d_out, nullptr,
                     static_cast<hipStream_t>(stream));
}

int sift_read_chunks_device(int device, const void* d_base, const uint64_t* offsets,
                            const uint64_t* avail, uint32_t n, uint32_t* d_status,
                            uint32_t* d_data_len, void* stream) {
    return run_batch(device, d_base, offsets, avail, n, 0, 0, 1, d_status, d_data_len,
                     static_cast<hipStream_t>(stream));
}

int sift_write_chunk_crcs_device(int device, void* d_base, const uint64_t* offsets,
                                 const uint64_t* avail, uint32_t n, uint32_t* d_status,
                                 void* stream) {
    return run_batch(device, d_base, offsets, avail, n, 0, 0, 3, d_status, nullptr,
                     static_cast<hipStream_t>(stream));
}

int sift_read_documents_device(int device, const void* d_base, const uint64_t* offsets,
                               const uint64_t* avail, uint32_t n, uint32_t* d_status,
                               uint32_t* d_amrg_len, void* stream) {
    return run_batch(device, d_base, offsets, avail, n, 0, 0, 2, d_status, d_amrg_len,
                     static_cast<hipStream_t>(stream));
}

int sift_chunk_crc_device(int device, const void* d_type_and_data, uint64_t data_len,
                          uint32_t* d_out, void* workspace, void* stream) {
    DeviceCtx* c = nullptr;
    int rc = get_ctx(device, &c);
    if (rc != SIFT_OK) return rc;
    DeviceGuard guard(device);
    if (!guard.ok) {
        g_last_error = "hipSetDevice failed";
        return SIFT_ERR_HIP;
    }
    uint32_t* ws = static_cast<uint32_t*>(workspace);
    if (ws && (rc = check_workspace(ws)) != SIFT_OK) return rc;
    if (!ws && (rc = stream_workspace(c, static_cast<hipStream_t>(stream), &ws)) != SIFT_OK) return rc;
    // container.rs:80-83: new() -> update(type) -> update(data) -> finalize().
    // Two updates over adjacent bytes equal one update over their concatenation.
    return launch_compute(c, SIFT_CRC_INIT, nullptr, d_type_and_data, data_len + 4u, 0xFFFFFFFFu,
                          d_out, ws, static_cast<hipStream_t>(stream));
}

}  // extern "C"
