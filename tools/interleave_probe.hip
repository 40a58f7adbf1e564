// interleave_probe.hip -- measurement tool: does HBM read faster when the 256 workgroups walk
// ONE moving window together (workgroup g reads tiles g, g + G, g + 2G, ...) than when each
// streams its own contiguous run (the tile kernel's assignment)?  Pure reads (XOR only), the tile
// kernel's geometry: 1024 lanes x 16 B per tile, two tiles per batch, one batch in flight, nt loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>  // 0 blocked, 1 interleaved by tile, 2 interleaved by batch (2 tiles)
__global__ __launch_bounds__(1024) void rd(const u32x4* p, long tiles, unsigned* out) {
    const long G = gridDim.x, g = blockIdx.x, per = tiles / G;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(p), 0, 0x7FFFFFFF, 0x00020000);
    u32x4 acc = {0, 0, 0, 0};
    const int voff = threadIdx.x * 16;
    for (long t = 0; t < per; t += 2) {
        long t0, t1;
        if (MODE == 0) { t0 = g * per + t; t1 = t0 + 1; }
        else if (MODE == 1) { t0 = t * G + g; t1 = (t + 1) * G + g; }
        else { t0 = (t / 2 * G + g) * 2; t1 = t0 + 1; }
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(srd, voff, (int)(t0 * 16384), 2);
        const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(srd, voff, (int)(t1 * 16384), 2);
        acc ^= a ^ b;
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[0] = 1;
}
int main() {
    const size_t n = 1ull << 30, R = 4;
    unsigned char* buf;
    CK(hipMalloc(&buf, n * R));
    CK(hipMemset(buf, 0x5A, n * R));
    unsigned* out;
    CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const long tiles = n / 16384;
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            auto launch = [&](int i) {
                const u32x4* p = reinterpret_cast<const u32x4*>(buf + (size_t)(i % R) * n);
                if (mode == 0) rd<0><<<256, 1024>>>(p, tiles, out);
                else if (mode == 1) rd<1><<<256, 1024>>>(p, tiles, out);
                else rd<2><<<256, 1024>>>(p, tiles, out);
            };
            for (int i = 0; i < 50; ++i) launch(i);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < 200; ++i) launch(i);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("mode %d (%s): %.1f us per GiB = %.2f TB/s\n", mode,
                   mode == 0 ? "blocked" : mode == 1 ? "interleaved by tile" : "interleaved by batch", ms * 5.0,
                   n / (ms * 5e-6) / 1e12);
        }
    return 0;
}
