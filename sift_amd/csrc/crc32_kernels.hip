// crc32_kernels.hip -- gfx950 kernels for CRC-32/ISO-HDLC over device memory.
//
// Replaces the byte loop of the reference, sift-persist/src/crc.rs:28-34
//     c = CRC_TABLE[((c ^ byte) & 0xff)] ^ (c >> 8)
// which is a serial recurrence, by an algebraically identical evaluation that
// is parallel over every lane of the chip.  Bit-exact by construction: the CRC
// register is a polynomial remainder, remainders are linear over GF(2), and
// the kernels only regroup the sum (see gf2.h for the representation).
//
//   reg(s, M) = s * x^(8|M|)  +  sum_j  w_j * x^(32 * (words after j) + 32)
//
// Tile kernel: each workgroup owns a contiguous run of 16 KiB tiles and folds
// it to one 32-bit residue.  HBM-bound by design: one coalesced, non-temporal
// dwordx4 load per lane per tile, 16 LDS lookups and ~50 VALU ops per load.
// The tables sit in LDS replicated once per bank so that a lookup with 32
// different indices in a half-wave is conflict-free.
// Combine kernel: log-depth tree over the <= 1024 residues, then the incoming
// state and the < 16 trailing bytes.
#include <hip/hip_runtime.h>

#include "crc32_device.h"
#include "gf2.h"

namespace sift {

namespace {

constexpr int kYWords = 4 * 256 * kRepl;  // 32768 words = 128 KiB
constexpr int kXWords = 4 * 256;          // 4 KiB, not replicated (used 16x per lane per launch)
constexpr int kWaves = kThreads / 64;

// acc * y: four lookups in the bank-replicated table.  Entry (k, b) for bank
// r lives at word ((k*256 + b) * 32 + r); lane l uses r = l & 31, so the 32
// lanes of a ds_read_b32 lane group hit 32 different banks whatever b is.
__device__ __forceinline__ uint32_t mul_y(uint32_t a, const uint32_t* ldsY, uint32_t lane32) {
    uint32_t i0 = ((a & 0xffu) << 5) + lane32;
    uint32_t i1 = (((a >> 8) & 0xffu) << 5) + lane32;
    uint32_t i2 = (((a >> 16) & 0xffu) << 5) + lane32;
    uint32_t i3 = ((a >> 24) << 5) + lane32;
    return ldsY[i0] ^ ldsY[8192 + i1] ^ ldsY[16384 + i2] ^ ldsY[24576 + i3];
}

// v * x^32: the reference's table step applied to four bytes at once.
__device__ __forceinline__ uint32_t mul_x32(uint32_t v, const uint32_t* ldsX) {
    return ldsX[v & 0xffu] ^ ldsX[256 + ((v >> 8) & 0xffu)] ^ ldsX[512 + ((v >> 16) & 0xffu)] ^
           ldsX[768 + (v >> 24)];
}

__device__ __forceinline__ u32x4 load_vec(const u32x4* p) { return __builtin_nontemporal_load(p); }

// Zero the first `skip` bytes (1..15) of a vector: they precede the message.
__device__ __forceinline__ u32x4 mask_head(u32x4 w, uint32_t skip) {
    uint32_t full = skip >> 2, part = skip & 3u;
    uint32_t m = part ? (0xffffffffu << (8u * part)) : 0xffffffffu;
#pragma unroll
    for (uint32_t c = 0; c < 4; ++c) {
        uint32_t keep = (c < full) ? 0u : ((c == full) ? m : 0xffffffffu);
        w[c] &= keep;
    }
    return w;
}

__device__ __forceinline__ uint32_t wave_xor(uint32_t r) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) r ^= __shfl_xor(r, off, 64);
    return r;
}

// Incoming state and trailing bytes; one lane.
__device__ __forceinline__ void finish(uint32_t body, const FinalArgs& f) {
    uint32_t s = f.state_ptr ? *f.state_ptr : f.state_imm;
    uint32_t r = gf2_mulmod(s, f.xpow_body) ^ body;
    for (uint32_t i = 0; i < f.tail_len; ++i) r = crc_step_byte(r, f.tail[i]);
    *f.out = r ^ f.xor_out;
}

}  // namespace

template <bool USE_Y>
__global__ __launch_bounds__(kThreads) void crc32_tiles_kernel(TileArgs a, FinalArgs f) {
    // One array (see the guide's second-__shared__-object trap); the one-tile
    // variant needs no y table and so leaves the CU's LDS to other workgroups.
    constexpr int kYUsed = USE_Y ? kYWords : 0;
    __shared__ uint32_t lds[kYUsed + kXWords + kWaves];
    uint32_t* ldsY = lds;
    uint32_t* ldsX = lds + kYUsed;
    uint32_t* red = ldsX + kXWords;

    const uint32_t tid = threadIdx.x;
    const uint32_t lane32 = tid & 31u;
    const int64_t g = blockIdx.x;

    // This workgroup's virtual tiles [tv0, tv1); the ones before pad_tiles are
    // all-zero padding in front of the message and contribute nothing.
    int64_t tv0 = g * a.tiles_per_group;
    const int64_t tv1 = tv0 + a.tiles_per_group;
    if (tv0 < a.pad_tiles) tv0 = a.pad_tiles;
    int64_t ntiles = tv1 - tv0;

    u32x4 acc = {0u, 0u, 0u, 0u};
    const u32x4* p = nullptr;
    if (ntiles > 0) {
        // Tiles are aligned to the END of the body, so only the first real tile
        // can start before vector 0.
        const int64_t vi = a.nvec - (a.virtual_tiles - tv0) * (int64_t)kTileVecs + (int64_t)tid;
        p = a.vbase + vi;
        if (vi >= 0) {
            acc = load_vec(p);
            if (vi == 0 && a.head_skip) acc = mask_head(acc, a.head_skip);
        }
        p += kTileVecs;
        --ntiles;
    }

    // Tables: global (L2-resident, 8 KiB) -> LDS, issued behind the first load.
    if (USE_Y) {
#pragma unroll 4
        for (int i = 0; i < kYWords / kThreads; ++i)
            ldsY[i * kThreads + tid] = a.tabs->Y[i * (kThreads / kRepl) + (tid >> 5)];
    }
    ldsX[tid] = a.tabs->X[tid];
    const uint32_t kt = a.tabs->Kt[tid];
    __syncthreads();

    if (USE_Y) {
        constexpr int U = 4;
        if (ntiles >= U) {
            u32x4 cur[U];
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = load_vec(p + u * kTileVecs);
            p += U * kTileVecs;
            ntiles -= U;
            while (ntiles >= U) {
                u32x4 nxt[U];
#pragma unroll
                for (int u = 0; u < U; ++u) nxt[u] = load_vec(p + u * kTileVecs);
                p += U * kTileVecs;
                ntiles -= U;
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = mul_y(acc[c], ldsY, lane32) ^ cur[u][c];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) cur[u] = nxt[u];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = mul_y(acc[c], ldsY, lane32) ^ cur[u][c];
            }
        }
        while (ntiles > 0) {
            u32x4 w = load_vec(p);
            p += kTileVecs;
            --ntiles;
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = mul_y(acc[c], ldsY, lane32) ^ w[c];
        }
    }

    // Lane residue over its four words, then its place in the tile, then XOR
    // over the workgroup.
    uint32_t r = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) r = mul_x32(r ^ acc[c], ldsX);
    r = gf2_mulmod(r, kt);
    r = wave_xor(r);
    if ((tid & 63u) == 0) red[tid >> 6] = r;
    __syncthreads();
    if (tid == 0) {
        uint32_t x = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) x ^= red[w];
        if (a.fuse_final)
            finish(x, f);
        else
            a.partial[g] = x;
    }
}

// Tree over the workgroup residues.  Residue of group g carries weight
// z^(G-1-g); index h = G-1-g so the weights are z^h and level s multiplies the
// upper half of each pair by z^(2^s).
__global__ __launch_bounds__(kMaxGroups) void crc32_combine_kernel(const uint32_t* partial,
                                                                   FinalArgs f) {
    __shared__ uint32_t sh[kMaxGroups];
    const int h = threadIdx.x;
    const int G = f.groups;
    uint32_t v = (h < G) ? partial[G - 1 - h] : 0u;
    sh[h] = v;
    for (int s = 0; (1 << s) < G; ++s) {
        __syncthreads();
        const bool active = ((h & ((2 << s) - 1)) == 0) && (h + (1 << s) < G);
        if (active) {
            v ^= gf2_mulmod(sh[h + (1 << s)], f.zpow[s]);
            sh[h] = v;
        }
    }
    if (h == 0) finish(v, f);
}

void launch_tiles(const TileArgs& t, const FinalArgs& f, int groups, bool use_y, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (use_y)
        hipLaunchKernelGGL(crc32_tiles_kernel<true>, dim3(groups), dim3(kThreads), 0, s, t, f);
    else
        hipLaunchKernelGGL(crc32_tiles_kernel<false>, dim3(groups), dim3(kThreads), 0, s, t, f);
}

void launch_combine(const uint32_t* partial, const FinalArgs& f, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    int block = 64;
    while (block < f.groups) block <<= 1;
    hipLaunchKernelGGL(crc32_combine_kernel, dim3(1), dim3(block), 0, s, partial, f);
}

}  // namespace sift
