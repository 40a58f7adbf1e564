// batch_bench.hip -- host-wall and device time per call of the batch entry points,
// without Python in the way.  Measurement tool, not part of the product.
//   batch_bench                 runs the whole table
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/sift_amd.h"

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            exit(2);                                                                  \
        }                                                                             \
    } while (0)
#define SK(x)                                                                         \
    do {                                                                              \
        int r_ = (x);                                                                 \
        if (r_ != SIFT_OK) {                                                          \
            fprintf(stderr, "%s: %s (%s)\n", #x, sift_strerror(r_), sift_last_error()); \
            exit(3);                                                                  \
        }                                                                             \
    } while (0)

__global__ void fill(uint32_t* p, size_t nwords, uint32_t seed) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < nwords; i += stride) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
        p[i] = x;
    }
}

// Reads the blob for `ms` milliseconds: the memory clock falls back while the GPU idles or
// runs host-paced small calls, and takes tens of milliseconds of load to come back (the 1
// GiB single-chunk row read 162 us behind busy rows and 190-205 us behind the sync_call
// section or in a run of its own, with the same library).
__global__ void heat(const uint32_t* p, size_t nwords, uint32_t* sink) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t x = 0;
    for (; i < nwords; i += stride) x ^= p[i];
    if (x == 0x9E3779B9u) *sink = x;
}

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    const size_t total = 1ull << 30;
    uint8_t* blob = nullptr;
    CK(hipMalloc(&blob, total + 4096));
    fill<<<1024, 256>>>(reinterpret_cast<uint32_t*>(blob), (total + 4096) / 4, 0x51F7);
    CK(hipDeviceSynchronize());
    SK(sift_init(0));
    // SIFT_BENCH_WAVE_MAX=<bytes>: the longest span a wave keeps (above it the item is cut into pieces)
    if (const char* e = getenv("SIFT_BENCH_WAVE_MAX")) SK(sift_tune(SIFT_TUNE_BATCH_WAVE_MAX, atoi(e)));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    uint32_t* d_out = nullptr;
    CK(hipMalloc(&d_out, 4ull << 20));
    uint64_t *d_off = nullptr, *d_len = nullptr;
    CK(hipMalloc(&d_off, 8ull << 20));
    CK(hipMalloc(&d_len, 8ull << 20));
    std::vector<uint64_t> off(1 << 20), len(1 << 20);
    auto preheat = [&](double ms) {
        const double t0 = now();
        while ((now() - t0) * 1e3 < ms) {
            for (int i = 0; i < 8; ++i) heat<<<2048, 1024, 0, s>>>(reinterpret_cast<const uint32_t*>(blob), total / 4, d_out);
            if (hipStreamSynchronize(s) != hipSuccess) return;
        }
    };
    preheat(80);

    // batch_bench compute [<bytes>]: sift_crc32_compute_device back to back over rotating slices of the
    // blob (caller's workspace), host issue time and wall time per call; under rocprofv3 --kernel-trace
    // the same run gives the kernel's duration and its start-to-start interval (tools/start_to_start.py)
    if (argc >= 2 && !strcmp(argv[1], "compute")) {
        const size_t only = argc >= 3 ? strtoull(argv[2], nullptr, 0) : 0;
        void* ws = nullptr;
        CK(hipMalloc(&ws, sift_crc32_workspace_bytes()));
        CK(hipMemset(ws, 0, sift_crc32_workspace_bytes()));
        CK(hipDeviceSynchronize());
        printf("{\"rows\": [\n");
        bool first_row = true;
        for (size_t size : {(size_t)1 << 20, (size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20, (size_t)256 << 20}) {
            if (only && size != only) continue;
            const size_t nsl = std::max<size_t>(2, std::min<size_t>(total / size, std::max<size_t>(8, (512u << 20) / size)));
            const size_t pitch = (total / nsl) / 4096 * 4096;
            const int calls = size <= (16u << 20) ? 2000 : 400;
            preheat(40);
            double best_issue = 1e9, best_wall = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                for (int i = 0; i < 20; ++i)
                    SK(sift_crc32_compute_device(0, 0xFFFFFFFFu, nullptr, blob + (i % nsl) * pitch, size, d_out, ws, s));
                CK(hipStreamSynchronize(s));
                const double t0 = now();
                for (int i = 0; i < calls; ++i)
                    SK(sift_crc32_compute_device(0, 0xFFFFFFFFu, nullptr, blob + (i % nsl) * pitch, size, d_out, ws, s));
                const double t_issue = now() - t0;
                CK(hipStreamSynchronize(s));
                const double t_wall = now() - t0;
                best_issue = std::min(best_issue, t_issue / calls);
                best_wall = std::min(best_wall, t_wall / calls);
            }
            printf("%s {\"kind\": \"compute_async\", \"bytes\": %zu, \"slices\": %zu, \"calls\": %d, "
                   "\"issue_us_per_call\": %.2f, \"wall_us_per_call\": %.2f, \"GBps_wall\": %.1f}",
                   first_row ? "" : ",\n", size, nsl, calls, best_issue * 1e6, best_wall * 1e6, (double)size / best_wall / 1e9);
            first_row = false;
            fflush(stdout);
        }
        printf("\n]}\n");
        return 0;
    }

    // batch_bench chunk <bytes>: only Chunk::read on one chunk of that many data bytes (for rocprofv3)
    const bool only_items = argc >= 3 && !strcmp(argv[1], "items");  // batch_bench items <bytes>: one crc_batch size (for rocprofv3 --pmc)
    const bool only_short = argc >= 2 && !strcmp(argv[1], "short");  // batch_bench short: items up to 64 KiB and the documents
    const bool only_chunk = (argc >= 3 && !strcmp(argv[1], "chunk")) || only_items;
    const size_t only_n = only_chunk ? strtoull(argv[2], nullptr, 0) : 0;
    printf("{\"rows\": [\n");
    bool first = true;
    // plain buffers back to back in the blob, descriptors in HBM and on the host
    // (batch_bench items <bytes> [count]: any size, and fewer buffers than fill the blob)
    std::vector<size_t> sizes = {64, 100, 128, 256, 512, 1000, 1024, 3000, 16384, 65536, 262144, 1 << 20, 4 << 20, 16 << 20, 64 << 20, 256 << 20, 1 << 30};
    if (only_items) sizes = {only_n};
    const size_t only_count = only_items && argc >= 4 ? strtoull(argv[3], nullptr, 0) : 0;
    for (size_t size : sizes) {
        if (!only_items && only_chunk) continue;
        if (only_short && size > 65536) continue;
        uint32_t n = (uint32_t)std::min<size_t>(total / size, 1 << 20);
        if (only_count && only_count < n) n = (uint32_t)only_count;
        for (uint32_t i = 0; i < n; ++i) {
            off[i] = (uint64_t)i * size;
            len[i] = size;
        }
        CK(hipMemcpy(d_off, off.data(), 8ull * n, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_len, len.data(), 8ull * n, hipMemcpyHostToDevice));
        preheat(20);
        for (int where = 0; where < (only_items || only_short ? 1 : 2); ++where) {
            const uint64_t* po = where ? off.data() : d_off;
            const uint64_t* pl = where ? len.data() : d_len;
            for (int i = 0; i < 3; ++i)
                SK(sift_crc32_batch_device(0, 0xFFFFFFFFu, 0xFFFFFFFFu, blob, po, pl, n, d_out, s));
            CK(hipStreamSynchronize(s));
            const int reps = 20;
            CK(hipEventRecord(e0, s));
            const double t0 = now();
            for (int i = 0; i < reps; ++i)
                SK(sift_crc32_batch_device(0, 0xFFFFFFFFu, 0xFFFFFFFFu, blob, po, pl, n, d_out, s));
            const double t_issue = now() - t0;
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            const double t_wall = now() - t0;
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s {\"kind\": \"crc_batch\", \"descriptors\": \"%s\", \"buffer_bytes\": %zu, \"buffers\": %u, "
                   "\"issue_us_per_call\": %.2f, \"wall_us_per_call\": %.2f, \"device_us_per_call\": %.2f, "
                   "\"GBps_wall\": %.1f, \"GBps_device\": %.1f}",
                   first ? "" : ",\n", where ? "host" : "device", size, n, t_issue / reps * 1e6, t_wall / reps * 1e6,
                   ms * 1e3 / reps, (double)n * size / (t_wall / reps) / 1e9, (double)n * size / (ms * 1e-3 / reps) / 1e9);
            first = false;
            fflush(stdout);
        }
    }
    // Mixed lengths in one batch: every turn of a wave holds spans of very different sizes
    for (size_t top : {(size_t)512, (size_t)4080}) {
        if (only_chunk) break;
        uint64_t x = 0x9E3779B97F4A7C15ull, pos = 0;
        uint32_t n = 0;
        while (n < (1u << 20)) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            const uint64_t sz = 32 + (x >> 33) % (top - 31);
            if (pos + sz > total) break;
            off[n] = pos;
            len[n] = sz;
            pos += sz;
            ++n;
        }
        CK(hipMemcpy(d_off, off.data(), 8ull * n, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_len, len.data(), 8ull * n, hipMemcpyHostToDevice));
        for (int i = 0; i < 3; ++i) SK(sift_crc32_batch_device(0, 0xFFFFFFFFu, 0xFFFFFFFFu, blob, d_off, d_len, n, d_out, s));
        CK(hipStreamSynchronize(s));
        const int reps = 20;
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) SK(sift_crc32_batch_device(0, 0xFFFFFFFFu, 0xFFFFFFFFu, blob, d_off, d_len, n, d_out, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s {\"kind\": \"crc_batch_mixed\", \"descriptors\": \"device\", \"buffer_bytes\": \"32..%zu\", \"buffers\": %u, "
               "\"device_us_per_call\": %.2f, \"GBps_device\": %.1f}",
               first ? "" : ",\n", top, n, ms * 1e3 / reps, (double)pos / (ms * 1e-3 / reps) / 1e9);
        first = false;
        fflush(stdout);
    }
    // Small batches: n buffers of 4 KiB (what a call costs when there is little to do)
    for (uint32_t n : {1u, 16u, 64u, 256u, 4096u}) {
        if (only_chunk || only_short) break;
        const size_t size = 4096;
        for (uint32_t i = 0; i < n; ++i) {
            off[i] = (uint64_t)i * size;
            len[i] = size;
        }
        CK(hipMemcpy(d_off, off.data(), 8ull * n, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_len, len.data(), 8ull * n, hipMemcpyHostToDevice));
        for (int where = 0; where < 2; ++where) {
            const uint64_t* po = where ? off.data() : d_off;
            const uint64_t* pl = where ? len.data() : d_len;
            for (int i = 0; i < 3; ++i)
                SK(sift_crc32_batch_device(0, 0xFFFFFFFFu, 0xFFFFFFFFu, blob, po, pl, n, d_out, s));
            CK(hipStreamSynchronize(s));
            const int reps = 200;
            CK(hipEventRecord(e0, s));
            const double t0 = now();
            for (int i = 0; i < reps; ++i)
                SK(sift_crc32_batch_device(0, 0xFFFFFFFFu, 0xFFFFFFFFu, blob, po, pl, n, d_out, s));
            const double t_issue = now() - t0;
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            const double t_wall = now() - t0;
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s {\"kind\": \"small_batch\", \"descriptors\": \"%s\", \"buffer_bytes\": %zu, \"buffers\": %u, "
                   "\"issue_us_per_call\": %.2f, \"wall_us_per_call\": %.2f, \"device_us_per_call\": %.2f}",
                   first ? "" : ",\n", where ? "host" : "device", size, n, t_issue / reps * 1e6, t_wall / reps * 1e6,
                   ms * 1e3 / reps);
            first = false;
            fflush(stdout);
        }
    }
    // Synchronous single calls through the C ABI (no Python in between): device-resident,
    // pageable host and pinned host input
    if (!only_chunk && !only_short) {
        std::vector<uint8_t> hbuf(1 << 20, 0x5A);
        uint8_t* pin = nullptr;
        CK(hipHostMalloc(reinterpret_cast<void**>(&pin), 1 << 20, hipHostMallocDefault));
        memset(pin, 0x5A, 1 << 20);
        for (size_t size : {(size_t)4, (size_t)4096, (size_t)65536, (size_t)1 << 20}) {
            double us[3] = {0, 0, 0};
            for (int kind = 0; kind < 3; ++kind) {
                uint32_t r = 0;
                const int reps = 300;
                for (int i = 0; i < 5; ++i) {
                    if (kind == 0) SK(sift_crc32_compute_device_sync(0, 0xFFFFFFFFu, blob, size, &r, s));
                    else SK(sift_crc32_compute_host(0, 0xFFFFFFFFu, kind == 1 ? hbuf.data() : pin, size, &r));
                }
                const double t0 = now();
                for (int i = 0; i < reps; ++i) {
                    if (kind == 0) SK(sift_crc32_compute_device_sync(0, 0xFFFFFFFFu, blob, size, &r, s));
                    else SK(sift_crc32_compute_host(0, 0xFFFFFFFFu, kind == 1 ? hbuf.data() : pin, size, &r));
                }
                us[kind] = (now() - t0) / reps * 1e6;
            }
            printf("%s {\"kind\": \"sync_call\", \"bytes\": %zu, \"device_us\": %.2f, \"pageable_host_us\": %.2f, "
                   "\"pinned_host_us\": %.2f}",
                   first ? "" : ",\n", size, us[0], us[1], us[2]);
            first = false;
            fflush(stdout);
        }
        CK(hipHostFree(pin));
    }
    // Chunk::read on ONE serialized chunk: length only in its header
    for (size_t n : {(size_t)1 << 20, (size_t)16 << 20, (size_t)64 << 20, (size_t)256 << 20, ((size_t)1 << 30) - 16}) {
        if ((only_chunk && n != only_n) || only_short) continue;
        preheat(40);
        uint8_t hdr[8] = {(uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n, 'A', 'M', 'R', 'G'};
        CK(hipMemcpy(blob, hdr, 8, hipMemcpyHostToDevice));
        uint32_t crc = 0;
        SK(sift_crc32_compute_device_sync(0, 0xFFFFFFFFu, blob + 4, n + 4, &crc, s));
        crc ^= 0xFFFFFFFFu;
        uint8_t be[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
        CK(hipMemcpy(blob + 8 + n, be, 4, hipMemcpyHostToDevice));
        uint64_t o = 0, av = n + 12;
        CK(hipMemcpy(d_off, &o, 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_len, &av, 8, hipMemcpyHostToDevice));
        for (int i = 0; i < 3; ++i) SK(sift_read_chunks_device(0, blob, d_off, d_len, 1, d_out, d_out + 1, s));
        CK(hipStreamSynchronize(s));
        uint32_t st[2];
        CK(hipMemcpy(st, d_out, 8, hipMemcpyDeviceToHost));
        if (st[0] != 0 || st[1] != (uint32_t)n) {
            fprintf(stderr, "chunk %zu: status %u len %u\n", n, st[0], st[1]);
            return 4;
        }
        const int reps = 20;
        CK(hipEventRecord(e0, s));
        const double t0 = now();
        for (int i = 0; i < reps; ++i) SK(sift_read_chunks_device(0, blob, d_off, d_len, 1, d_out, d_out + 1, s));
        const double t_issue = now() - t0;
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        const double t_wall = now() - t0;
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s {\"kind\": \"read_one_chunk\", \"descriptors\": \"device\", \"chunk_data_bytes\": %zu, "
               "\"issue_us_per_call\": %.2f, \"wall_us_per_call\": %.2f, \"device_us_per_call\": %.2f, "
               "\"GBps_wall\": %.1f, \"GBps_device\": %.1f}",
               first ? "" : ",\n", n, t_issue / reps * 1e6, t_wall / reps * 1e6, ms * 1e3 / reps, (double)(n + 4) / (t_wall / reps) / 1e9,
               (double)(n + 4) / (ms * 1e-3 / reps) / 1e9);
        first = false;
        fflush(stdout);
    }
    // read_document's framing on MANY small save files (header | AMRG chunk | SEND chunk),
    // the shape of the reference's real data: one valid file built with the library's
    // own host entry point, repeated back to back
    if (!only_chunk) {
        for (size_t payload : {(size_t)100, (size_t)200, (size_t)500, (size_t)2000, (size_t)8000}) {
            if (!only_short && payload < 500) continue;
            std::vector<uint8_t> pl(payload), file(payload + 33);
            for (size_t i = 0; i < payload; ++i) pl[i] = (uint8_t)(i * 131u + 7u);
            const int64_t flen = sift_write_document(0, pl.data(), payload, file.data(), file.size());
            if (flen != (int64_t)file.size()) return 5;
            const size_t pitch = (file.size() + 15) & ~size_t(15);
            const uint32_t n = (uint32_t)std::min<size_t>(total / pitch, 1 << 20);
            std::vector<uint8_t> host((size_t)n * pitch, 0);
            for (uint32_t i = 0; i < n; ++i) {
                memcpy(host.data() + (size_t)i * pitch, file.data(), file.size());
                off[i] = (uint64_t)i * pitch;
                len[i] = file.size();
            }
            CK(hipMemcpy(blob, host.data(), host.size(), hipMemcpyHostToDevice));
            CK(hipMemcpy(d_off, off.data(), 8ull * n, hipMemcpyHostToDevice));
            CK(hipMemcpy(d_len, len.data(), 8ull * n, hipMemcpyHostToDevice));
            preheat(40);
            for (int i = 0; i < 3; ++i) SK(sift_read_documents_device(0, blob, d_off, d_len, n, d_out, nullptr, s));
            CK(hipStreamSynchronize(s));
            std::vector<uint32_t> st(n);
            CK(hipMemcpy(st.data(), d_out, 4ull * n, hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < n; ++i)
                if (st[i] != 0) {
                    fprintf(stderr, "document %u: status %u\n", i, st[i]);
                    return 6;
                }
            const int reps = 20;
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) SK(sift_read_documents_device(0, blob, d_off, d_len, n, d_out, nullptr, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf(",\n {\"kind\": \"read_documents\", \"descriptors\": \"device\", \"file_bytes\": %zu, \"files\": %u, "
                   "\"device_us_per_call\": %.2f, \"Mfiles_per_s\": %.1f, \"GBps_device\": %.1f}",
                   file.size(), n, ms * 1e3 / reps, n / (ms * 1e-3 / reps) / 1e6,
                   (double)n * file.size() / (ms * 1e-3 / reps) / 1e9);
            fflush(stdout);
        }
    }
    // the same on files of MIXED sizes packed back to back (every alignment, every tail):
    // 256 distinct valid files with payloads of 40..top bytes, repeated
    for (size_t top : {(size_t)500, (size_t)4000}) {
        if (only_chunk) break;
        std::vector<std::vector<uint8_t>> kinds(256);
        uint64_t x = 0x2545F4914F6CDD1Dull;
        for (auto& f : kinds) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            const size_t payload = 40 + (x >> 33) % (top - 39);
            std::vector<uint8_t> pl(payload);
            for (size_t i = 0; i < payload; ++i) pl[i] = (uint8_t)(i * 131u + (x >> 20));
            f.resize(payload + 33);
            if (sift_write_document(0, pl.data(), payload, f.data(), f.size()) != (int64_t)f.size()) return 5;
        }
        std::vector<uint8_t> host;
        host.reserve(total);
        uint32_t n = 0;
        while (n < (1u << 20)) {
            const auto& f = kinds[(n * 2654435761u) >> 24];
            if (host.size() + f.size() > total) break;
            off[n] = host.size();
            len[n] = f.size();
            host.insert(host.end(), f.begin(), f.end());
            ++n;
        }
        CK(hipMemcpy(blob, host.data(), host.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(d_off, off.data(), 8ull * n, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_len, len.data(), 8ull * n, hipMemcpyHostToDevice));
        preheat(40);
        for (int i = 0; i < 3; ++i) SK(sift_read_documents_device(0, blob, d_off, d_len, n, d_out, nullptr, s));
        CK(hipStreamSynchronize(s));
        std::vector<uint32_t> st(n);
        CK(hipMemcpy(st.data(), d_out, 4ull * n, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n; ++i)
            if (st[i] != 0) {
                fprintf(stderr, "mixed document %u: status %u\n", i, st[i]);
                return 6;
            }
        const int reps = 20;
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) SK(sift_read_documents_device(0, blob, d_off, d_len, n, d_out, nullptr, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf(",\n {\"kind\": \"read_documents_mixed\", \"descriptors\": \"device\", \"file_bytes\": \"73..%zu\", \"files\": %u, "
               "\"device_us_per_call\": %.2f, \"Mfiles_per_s\": %.1f, \"GBps_device\": %.1f}",
               top + 33, n, ms * 1e3 / reps, n / (ms * 1e-3 / reps) / 1e6, (double)host.size() / (ms * 1e-3 / reps) / 1e9);
        fflush(stdout);
    }
    printf("\n]}\n");
    return 0;
}
