// hbm_read_probe.hip -- measurement tool (not part of the product).
// Times pure-read kernels over the bench workload (1 GiB) next to the
// library's tile kernel, interleaved in one process, to place the practical
// HBM read ceiling on the same device the CRC kernel runs on.
//   build: make -C tools   (compiles the library sources in, so -DSIFT_* variants can be compared)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/sift_amd.h"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));            \
            std::exit(1);                                                          \
        }                                                                          \
    } while (0)

__global__ void fill(uint32_t* p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t x = (uint32_t)i * 2654435761u;
        x ^= x >> 15;
        x *= 2246822519u;
        x ^= x >> 13;
        p[i] = x;
    }
}

// Same geometry as crc32_tiles_kernel: one 1024-lane workgroup per CU walks a
// contiguous run of 16 KiB tiles, U loads in flight per lane; XOR only.
template <int U, bool NT>
__global__ __launch_bounds__(1024) void read_runs(const u32x4* base, size_t tiles_per_group,
                                                  uint32_t* out) {
    const u32x4* p = base + (size_t)blockIdx.x * tiles_per_group * 1024 + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    size_t t = 0;
    for (; t + U <= tiles_per_group; t += U) {
        u32x4 w[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            w[u] = NT ? __builtin_nontemporal_load(p + (t + u) * 1024) : p[(t + u) * 1024];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= w[u];
    }
    for (; t < tiles_per_group; ++t) acc ^= p[t * 1024];
    uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x12345678u) out[blockIdx.x] = r;  // keeps the loads alive
}

// read_runs, but the groups with (blockIdx & mask) != 0 walk their run from its
// last tile down: does the direction of a stream matter to the memory system?
template <int U>
__global__ __launch_bounds__(1024) void read_runs_updown(const u32x4* base, size_t tiles_per_group,
                                                         uint32_t mask, uint32_t* out) {
    const bool down = (blockIdx.x & mask) != 0;
    const u32x4* p = base + (size_t)blockIdx.x * tiles_per_group * 1024 + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t t = 0; t + U <= tiles_per_group; t += U) {
        u32x4 w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t tt = down ? tiles_per_group - 1 - (t + u) : t + u;
            w[u] = __builtin_nontemporal_load(p + tt * 1024);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= w[u];
    }
    uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x12345678u) out[blockIdx.x] = r;
}

// read_runs<2> plus synthetic work per loaded vector: V rounds of 4 VALU ops per
// word and L conflict-free LDS lookups per word (same table layout as the CRC
// kernel).  Separates "the loop has arithmetic" from "the loop reads LDS".
template <int V, int L>
__global__ __launch_bounds__(1024) void read_runs_work(const u32x4* base, size_t tiles_per_group,
                                                       uint32_t* out) {
    __shared__ uint32_t tab[32768];
    for (int i = threadIdx.x; i < 32768; i += 1024) tab[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t lane32 = threadIdx.x & 31u;
    const u32x4* p = base + (size_t)blockIdx.x * tiles_per_group * 1024 + threadIdx.x;
    u32x4 acc = {1, 2, 3, 4};
    u32x4 cur[2], nxt[2];
    cur[0] = __builtin_nontemporal_load(p);
    cur[1] = __builtin_nontemporal_load(p + 1024);
    for (size_t t = 2; t + 2 <= tiles_per_group; t += 2) {
        nxt[0] = __builtin_nontemporal_load(p + t * 1024);
        nxt[1] = __builtin_nontemporal_load(p + (t + 1) * 1024);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint32_t a = acc[c];
#pragma unroll
                for (int v = 0; v < V; ++v) a = ((a >> 3) ^ (a << 5)) + (a & 0x55aa55aau);
                uint32_t x = 0;
#pragma unroll
                for (int l = 0; l < L; ++l) x ^= tab[l * 8192 + (((a >> (8 * l)) & 0xffu) << 5) + lane32];
                acc[c] = (a ^ x) ^ cur[u][c];
            }
        }
        cur[0] = nxt[0];
        cur[1] = nxt[1];
    }
    acc ^= cur[0] ^ cur[1];
    uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x12345678u) out[blockIdx.x] = r;
}

// Round-robin tiles: workgroup g reads tiles g, g+G, g+2G, ... (the whole chip
// sweeps the buffer front to back together).
template <int U>
__global__ __launch_bounds__(1024) void read_roundrobin(const u32x4* base, size_t tiles, uint32_t* out) {
    const size_t G = gridDim.x;
    const u32x4* p = base + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    size_t t = blockIdx.x;
    for (; t + (U - 1) * G < tiles; t += U * G) {
        u32x4 w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = __builtin_nontemporal_load(p + (t + u * G) * 1024);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= w[u];
    }
    for (; t < tiles; t += G) acc ^= p[t * 1024];
    uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x12345678u) out[blockIdx.x] = r;
}

// Contiguous runs, but each wave owns its own run (4096 streams instead of 256).
template <int U>
__global__ __launch_bounds__(1024) void read_wave_runs(const u32x4* base, size_t vec_per_wave, uint32_t* out) {
    const size_t wave = blockIdx.x * 16 + (threadIdx.x >> 6);
    const u32x4* p = base + wave * vec_per_wave + (threadIdx.x & 63);
    u32x4 acc = {0, 0, 0, 0};
    size_t t = 0;
    const size_t n = vec_per_wave / 64;
    for (; t + U <= n; t += U) {
        u32x4 w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = __builtin_nontemporal_load(p + (t + u) * 64);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= w[u];
    }
    uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x12345678u) out[blockIdx.x] = r;
}

// Conventional streaming read: 256-lane workgroups, grid-stride.
template <bool NT>
__global__ __launch_bounds__(256) void read_gridstride(const u32x4* base, size_t nvec, uint32_t* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        u32x4 a = NT ? __builtin_nontemporal_load(base + i) : base[i];
        u32x4 b = NT ? __builtin_nontemporal_load(base + i + stride) : base[i + stride];
        u32x4 c = NT ? __builtin_nontemporal_load(base + i + 2 * stride) : base[i + 2 * stride];
        u32x4 d = NT ? __builtin_nontemporal_load(base + i + 3 * stride) : base[i + 3 * stride];
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < nvec; i += stride) acc ^= base[i];
    uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x12345678u) out[blockIdx.x] = r;
}

__global__ __launch_bounds__(1024) void empty_kernel(uint32_t* out) {
    if (threadIdx.x == 5000) out[0] = 1;
}

// One load per lane + LDS reduce + one atomic per group: the irreducible part
// of a one-tile-per-group call.
__global__ __launch_bounds__(1024) void one_tile_atomic(const u32x4* base, unsigned long long* acc) {
    __shared__ uint32_t red[16];
    u32x4 w = __builtin_nontemporal_load(base + (size_t)blockIdx.x * 1024 + threadIdx.x);
    uint32_t r = w.x ^ w.y ^ w.z ^ w.w;
    for (int off = 32; off >= 1; off >>= 1) r ^= __shfl_xor(r, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = r;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t x = 0;
        for (int i = 0; i < 16; ++i) x ^= red[i];
        __hip_atomic_fetch_xor(acc, (unsigned long long)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 0) : (size_t)1 << 30;
    const int rounds = argc > 2 ? std::atoi(argv[2]) : 30;
    if (n < ((size_t)64 << 20) || n % ((size_t)256 * 1024 * 16) != 0) {
        std::fprintf(stderr, "bytes must be a multiple of 4 MiB and at least 64 MiB\n");
        return 2;  // every kernel below assumes 256 whole runs of whole tiles inside the buffer
    }
    CK(hipSetDevice(0));
    uint8_t* buf;
    uint32_t* out;
    CK(hipMalloc(&buf, n));
    CK(hipMalloc(&out, 1 << 16));
    fill<<<4096, 256>>>((uint32_t*)buf, n / 4);
    CK(hipDeviceSynchronize());
    if (sift_init(0) != SIFT_OK) {
        std::fprintf(stderr, "sift_init: %s\n", sift_last_error());
        return 1;
    }
    const size_t nvec = n / 16, tiles = nvec / 1024, tpg = tiles / 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char* names[] = {"read_runs<4,nt>  (crc geometry)", "read_runs<8,nt>", "read_runs<4,plain>",
                           "gridstride 256x2048 nt", "gridstride 256x4096 plain",
                           "sift tile kernel (events)", "sift compute, whole call",
                           "read_runs<2,nt>", "roundrobin<2> 256 groups", "roundrobin<4> 256 groups",
                           "wave_runs<4> 4096 streams", "read_runs<4,nt> 512 groups"};
    const int nv = 12;
    std::vector<std::vector<float>> ms(nv);
    for (int r = 0; r < rounds + 3; ++r) {
        for (int v = 0; v < nv; ++v) {
            float t = 0;
            if (v == 5) {
                sift_profile_begin(0, 1, 1);
                sift_crc32_compute_device(0, 0xFFFFFFFFu, nullptr, buf, n, out, nullptr, nullptr);
                double kms = 0;
                int k = 0;
                sift_profile_end(0, &kms, &k);
                t = (float)kms;
            } else {
                CK(hipEventRecord(e0));
                switch (v) {
                    case 0: read_runs<4, true><<<256, 1024>>>((const u32x4*)buf, tpg, out); break;
                    case 1: read_runs<8, true><<<256, 1024>>>((const u32x4*)buf, tpg, out); break;
                    case 2: read_runs<4, false><<<256, 1024>>>((const u32x4*)buf, tpg, out); break;
                    case 3: read_gridstride<true><<<2048, 256>>>((const u32x4*)buf, nvec, out); break;
                    case 4: read_gridstride<false><<<4096, 256>>>((const u32x4*)buf, nvec, out); break;
                    case 6:
                        sift_crc32_compute_device(0, 0xFFFFFFFFu, nullptr, buf, n, out, nullptr, nullptr);
                        break;
                    case 7: read_runs<2, true><<<256, 1024>>>((const u32x4*)buf, tpg, out); break;
                    case 8: read_roundrobin<2><<<256, 1024>>>((const u32x4*)buf, tiles, out); break;
                    case 9: read_roundrobin<4><<<256, 1024>>>((const u32x4*)buf, tiles, out); break;
                    case 10: read_wave_runs<4><<<256, 1024>>>((const u32x4*)buf, nvec / 4096, out); break;
                    case 11: read_runs<4, true><<<512, 1024>>>((const u32x4*)buf, tpg / 2, out); break;
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&t, e0, e1));
            }
            if (r >= 3) ms[v].push_back(t);
        }
    }
    std::printf("bytes %zu, rounds %d (interleaved)\n", n, rounds);
    for (int v = 0; v < nv; ++v) {
        std::sort(ms[v].begin(), ms[v].end());
        float med = ms[v][ms[v].size() / 2], mn = ms[v][0];
        std::printf("%-34s median %8.2f us (%7.1f GB/s)   min %8.2f us (%7.1f GB/s)\n", names[v],
                    med * 1e3, n / (med * 1e-3) / 1e9, mn * 1e3, n / (mn * 1e-3) / 1e9);
    }
    {
        std::vector<float> t0, t1;
        for (int r = 0; r < 20; ++r) {
            float t;
            CK(hipEventRecord(e0));
            empty_kernel<<<256, 1024>>>(out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&t, e0, e1));
            t0.push_back(t);
            CK(hipEventRecord(e0));
            one_tile_atomic<<<256, 1024>>>((const u32x4*)buf, (unsigned long long*)(out + 64));
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&t, e0, e1));
            t1.push_back(t);
        }
        std::sort(t0.begin(), t0.end());
        std::sort(t1.begin(), t1.end());
        std::printf("floor: empty 256x1024 kernel median %6.2f us; one load + reduce + atomic per group %6.2f us\n",
                    t0[10] * 1e3, t1[10] * 1e3);
    }
    // Does a power-of-two distance between the 256 streams matter?  Same kernel,
    // runs of 255 / 256 / 253 / 251 tiles (GB/s on the bytes actually read).
    {
        const size_t tp[4] = {255, 256, 253, 251};  // never more than tpg: the buffer holds 256 runs of tpg tiles
        for (int v = 0; v < 4; ++v) {
            if (tp[v] > tpg) continue;  // would read past the buffer
            std::vector<float> tt;
            for (int r = 0; r < 12; ++r) {
                float t;
                CK(hipEventRecord(e0));
                read_runs<2, true><<<256, 1024>>>((const u32x4*)buf, tp[v], out);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&t, e0, e1));
                if (r >= 2) tt.push_back(t);
            }
            std::sort(tt.begin(), tt.end());
            const double bytes = 256.0 * tp[v] * 16384.0;
            std::printf("stream spacing: runs of %zu tiles  median %7.2f us  %7.1f GB/s\n", tp[v],
                        tt[5] * 1e3, bytes / (tt[5] * 1e-3) / 1e9);
        }
    }
    // Direction of the streams: all up, odd groups down, all down.
    {
        const uint32_t masks[3] = {0u, 1u, 0xFFFFFFFFu};
        const char* dn[3] = {"all groups walk up", "odd groups walk down", "all groups walk down"};
        for (int rep = 0; rep < 2; ++rep)
            for (int v = 0; v < 3; ++v) {
                std::vector<float> tt;
                for (int r = 0; r < 12; ++r) {
                    float t;
                    CK(hipEventRecord(e0));
                    if (masks[v] == 0xFFFFFFFFu)
                        read_runs_updown<2><<<256, 1024>>>((const u32x4*)buf + 0, tpg, 0x1FFu, out);  // every group: (g & 0x1ff) != 0 except g = 0
                    else
                        read_runs_updown<2><<<256, 1024>>>((const u32x4*)buf, tpg, masks[v], out);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&t, e0, e1));
                    if (r >= 2) tt.push_back(t);
                }
                std::sort(tt.begin(), tt.end());
                std::printf("direction: %-24s median %7.2f us  min %7.2f us\n", dn[v], tt[5] * 1e3, tt[0] * 1e3);
            }
    }
    // Where does the CRC loop's ~3 % streaming loss come from?  Pure read with
    // synthetic VALU-only, LDS-only and combined work of the CRC loop's size.
    {
        const char* wn[5] = {"read + nothing", "read + 12 VALU/word (V=3)", "read + 4 LDS lookups/word",
                             "read + 2 LDS lookups/word", "read + 3 VALU rounds + 4 lookups"};
        for (int v = 0; v < 5; ++v) {
            std::vector<float> tt;
            for (int r = 0; r < 12; ++r) {
                float t;
                CK(hipEventRecord(e0));
                switch (v) {
                    case 0: read_runs_work<0, 0><<<256, 1024>>>((const u32x4*)buf, tpg, out); break;
                    case 1: read_runs_work<3, 0><<<256, 1024>>>((const u32x4*)buf, tpg, out); break;
                    case 2: read_runs_work<0, 4><<<256, 1024>>>((const u32x4*)buf, tpg, out); break;
                    case 3: read_runs_work<0, 2><<<256, 1024>>>((const u32x4*)buf, tpg, out); break;
                    case 4: read_runs_work<3, 4><<<256, 1024>>>((const u32x4*)buf, tpg, out); break;
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&t, e0, e1));
                if (r >= 2) tt.push_back(t);
            }
            std::sort(tt.begin(), tt.end());
            std::printf("synthetic work: %-34s median %7.2f us\n", wn[v], tt[5] * 1e3);
        }
    }
    // Fixed cost of a call: same grid, 2 tiles per group (table fill + tail +
    // exchange, almost no streaming), and 1 tile per group (no y table).
    {
        const size_t small[2] = {(size_t)256 * 2 * 16384, (size_t)256 * 16384};
        for (int v = 0; v < 2; ++v) {
            std::vector<float> t;
            for (int r = 0; r < 20; ++r) {
                sift_profile_begin(0, 1, 1);
                sift_crc32_compute_device(0, 0xFFFFFFFFu, nullptr, buf, small[v], out, nullptr, nullptr);
                double kms = 0;
                int k = 0;
                sift_profile_end(0, &kms, &k);
                t.push_back((float)kms);
            }
            std::sort(t.begin(), t.end());
            std::printf("fixed cost: %zu KiB (%s) kernel median %6.2f us  min %6.2f us\n", small[v] >> 10,
                        v == 0 ? "2 tiles/group, y table" : "1 tile/group", t[10] * 1e3, t[0] * 1e3);
        }
    }
    // Back-to-back launches on one stream, no host sync in between: what a
    // sustained run (bench.py) sees per launch, gaps included.
    const int reps = 50;
    for (int v = 0; v < 2; ++v) {
        std::vector<float> per;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) {
                if (v == 0)
                    read_runs<4, true><<<256, 1024>>>((const u32x4*)buf, tpg, out);
                else
                    sift_crc32_compute_device(0, 0xFFFFFFFFu, nullptr, buf, n, out, nullptr, nullptr);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float t = 0;
            CK(hipEventElapsedTime(&t, e0, e1));
            per.push_back(t / reps);
        }
        std::sort(per.begin(), per.end());
        std::printf("back-to-back x%d %-22s median %8.2f us/launch (%7.1f GB/s)  min %8.2f\n", reps,
                    v == 0 ? "read_runs<4,nt>" : "sift compute", per[2] * 1e3,
                    n / (per[2] * 1e-3) / 1e9, per[0] * 1e3);
    }
    return 0;
}
