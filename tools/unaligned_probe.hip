// How fast are 16-byte global loads whose address is not 16-byte (or 4-byte) aligned?
//   tools/bin/unaligned_probe        one line per byte offset
// Every lane reads 16 bytes at base + 16 * i + offset, waves stride through 1 GiB.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            return 2;                                                         \
        }                                                                     \
    } while (0)

__global__ __launch_bounds__(1024) void probe(const uint8_t* base, uint64_t nvec, uint32_t offset, uint32_t* out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const u32x4 v = *reinterpret_cast<const u32x4_u*>(base + 16ull * i + offset);
        acc ^= v;
    }
    const uint32_t r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (r == 0x12345678u) out[0] = r;  // never, keeps the loads
}

int main() {
    const uint64_t bytes = 1ull << 30;
    uint8_t* d = nullptr;
    uint32_t* out = nullptr;
    CK(hipMalloc(&d, bytes + 256));
    CK(hipMalloc(&out, 4));
    CK(hipMemset(d, 0x5A, bytes + 256));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (uint32_t offset : {0u, 16u, 4u, 8u, 1u, 2u, 3u, 7u, 13u}) {
        for (int i = 0; i < 3; ++i) probe<<<2048, 1024>>>(d, bytes / 16, offset, out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 10;
        for (int i = 0; i < reps; ++i) probe<<<2048, 1024>>>(d, bytes / 16, offset, out);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("offset %2u: %.1f us per GiB, %.0f GB/s\n", offset, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e9);
    }
    return 0;
}
