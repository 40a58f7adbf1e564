// trace_pieces.hip -- where a single-chunk Chunk::read call spends its time (measurement tool).
// Needs a -DSIFT_TRACE_PIECES build of the library (LD_LIBRARY_PATH): the piece kernel then
// leaves six 100 MHz timestamps per workgroup in the status array (words 64 + slot * 256 + group):
// 0 entry, 1 planned and staged, 2 first run streamed, 3 all runs reduced, 4 two commit steps
// drained (XOR and arrival count have returned), 5 leaving.
//   trace_pieces <data bytes>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/sift_amd.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define SK(x) do { int r_ = (x); if (r_ != SIFT_OK) { fprintf(stderr, "%s: %s (%s)\n", #x, sift_strerror(r_), sift_last_error()); exit(3); } } while (0)
int main(int argc, char** argv) {
    const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 0) : (16u << 20);
    uint8_t* blob = nullptr;
    CK(hipMalloc(&blob, n + 4096));
    CK(hipMemset(blob, 0x5A, n + 4096));
    SK(sift_init(0));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    uint32_t* d_out = nullptr;
    CK(hipMalloc(&d_out, 1 << 20));
    uint64_t *d_off, *d_len;
    CK(hipMalloc(&d_off, 64));
    CK(hipMalloc(&d_len, 64));
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
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<uint32_t> w(64 + 6 * 256);
    for (int it = 0; it < 8; ++it) {
        for (int i = 0; i < 10; ++i) SK(sift_read_chunks_device(0, blob, d_off, d_len, 1, d_out, d_out + 1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 10; ++i) SK(sift_read_chunks_device(0, blob, d_off, d_len, 1, d_out, d_out + 1, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(w.data(), d_out, w.size() * 4, hipMemcpyDeviceToHost));
        if (w[0] != 0) { fprintf(stderr, "status %u\n", w[0]); return 4; }
        uint32_t t0 = 0xFFFFFFFFu;
        for (int g = 0; g < 256; ++g) t0 = std::min(t0, w[64 + g]);
        printf("call %.1f us:", ms * 100.0);
        for (int slot = 0; slot < 6; ++slot) {
            uint32_t lo = 0xFFFFFFFFu, hi = 0;
            for (int g = 0; g < 256; ++g) { lo = std::min(lo, w[64 + slot * 256 + g] - t0); hi = std::max(hi, w[64 + slot * 256 + g] - t0); }
            printf("  [%d] %.2f..%.2f", slot, lo / 100.0, hi / 100.0);
        }
        printf("\n");
    }
    return 0;
}
