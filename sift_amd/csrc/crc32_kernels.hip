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
// crc32_tiles_kernel   one buffer, whole chip.  Each workgroup owns a
//     contiguous run of 16 KiB tiles and folds it to one 32-bit residue.
//     HBM-bound by design: one coalesced, non-temporal dwordx4 load per lane
//     per tile, 16 LDS lookups and ~25 VALU ops per load.  The tables sit in
//     LDS in rows by byte value (KiLane): 8 copies are conflict-free and a
//     lookup's address is one v_perm_b32.  The workgroups exchange residues
//     through one returning 64-bit atomic per level; the last to arrive
//     finishes (single launch).
// crc32_batch_kernel   many buffers (also Chunk::read / Chunk::write's crc step /
//     read_document's framing x n): a wave takes 64 items a turn, one lane per
//     item for parsing and verdict; the CRC spans by one lane, a group of 2 / 4 / 8
//     lanes or the whole wave, by length.  Longer spans go to
// crc32_pieces_kernel  the tile kernel's algorithm over pieces of long items:
//     planned by the kernel itself for up to 4096 items, filed by the batch
//     kernel's waves beyond.
// crc32_combine_kernel inputs under 32 bytes (one lane), and the second launch
//     of the two-launch A/B form (tree over the <= 1024 residues).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <type_traits>

#include "crc32_device.h"
#include "gf2.h"

#ifndef SIFT_LOAD_AUX
#define SIFT_LOAD_AUX 2  // nt
#endif
#ifndef SIFT_WAVE_BATCH
#define SIFT_WAVE_BATCH 2  // tiles of 1 KiB a wave asks for at a time on a span of its own (wave_crc; 3 and 4: no faster)
#endif

namespace sift {

namespace {

constexpr int kYWords = 256 * 64;          // the y table by byte value in rows of 256 B: 64 KiB (half of each row in use)
constexpr int kWaves = kThreads / 64;

// a ^ b ^ c in one instruction (v_bitop3_b32, truth table 0x96: gfx950 has no v_xor3).
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

// a ^ (b & m) in one instruction (truth table 0x78): a step of a multiply by a constant
// whose rows are in registers, m = one bit of the other factor spread over the word.
__device__ __forceinline__ uint32_t xor_and(uint32_t a, uint32_t b, uint32_t m) {
    return __builtin_amdgcn_bitop3_b32(a, b, m, 0x78);
}

// ---------------------------------------------------------------------------
// Row-interleaved LDS tables of the batch kernel.
//
// A multiply-by-constant table T[k][b] (k = byte position 0..3, b = byte value) used to
// sit in LDS as [k][b][copy]: a lookup was a byte extract, a shift-add and the read, and a
// lane group's 32 lookups were conflict-free only with 32 copies (128 KiB a table).  Here a
// table is laid out BY BYTE VALUE: row b holds, for each of C copies, the four entries
// T[0..3][b] next to each other (word 4c + k), and lane l makes its four lookups of one
// multiply in the order k_j = ((l & 3) + j) & 3, copy c = (l >> 2) & (C - 1).  For every j
// the 32 lanes of a ds_read_b32 lane group then read 4 C different banks whatever the data:
// conflict-free with C = 8 (32 KiB a table instead of 128), two-way with C = 4 (lanes l and
// l + 16 share).  And with rows of 256 bytes the whole address is ONE v_perm_b32: byte 1 =
// the operand's byte k_j, byte 0 = the lane's offset in the row, byte 2 = the 64 KiB block:
//     block 0  [0, 64 K)      rows of 256 B:  G = x^(8*256) C = 8 | X = x^32 C = 8
//     block 1  [64 K, 128 K)  rows of 256 B:  K2 = x^(128*4) C = 8 | K1 = x^(128*2) C = 2 | K0 = x^128 C = 4 |
//                                             K3 = x^(128*8) C = 2
//     block 2  [128 K, 160 K) rows of 128 B:  Y1 = x^(8*1024) C = 8 (the wave path; the address is
//                                             the permute of doubled offsets shifted right by one)
// (K2 is the 4 x 4 shape's chain step, 48 lookups a pass at 256 bytes; K0, K1 and K3 are used
// 4 times a pass each since the two-chain shapes fold before they join.  With Y1 at C = 4
// 16 KiB items were 9 % slower.)
// The reference's own CRC_TABLE (crc.rs:2-20) is X's k = 3 slice: (b << 24) * x^32 = b * x^8.
// The image is built on the host (DeviceTables::batch_image) and copied in 16 bytes a lane.
// The tables start at LDS address 0 (the kernel's one __shared__ array; checked in its
// prologue), and a lookup reads the absolute address permute + IMM, so that IMM rides in
// the ds_read's offset field: with the array's base passed down as a pointer the callee
// had to add it to every address (one more instruction a lookup).
typedef __attribute__((address_space(3))) const uint8_t* LdsBytes;
typedef __attribute__((address_space(3))) const uint32_t* LdsWord;
template <int IMM>
__device__ __forceinline__ uint32_t ki_read(uint32_t at) {
    return *reinterpret_cast<LdsWord>(static_cast<uintptr_t>(at + (uint32_t)IMM));
}
struct KiLane {
    uint32_t sel[4];  // v_perm_b32 selectors, step j: [S1 byte (j & 1), operand byte k_j, S1 byte 2, 0]
    uint32_t p8, q8;  // C = 8 rows of block 0: row offsets of steps 0, 1 (p8) and 2, 3 (q8), block byte 0
};
__device__ __forceinline__ KiLane ki_lane(uint32_t lane) {
    KiLane L;
    const uint32_t s = lane & 3u, c8 = (lane >> 2) & 7u;
    uint32_t o[4];
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) {
        const uint32_t k = (s + j) & 3u;
        L.sel[j] = 0x0c020000u | ((4u + k) << 8) | (j & 1u);
        o[j] = 16u * c8 + 4u * k;
    }
    L.p8 = o[0] | (o[1] << 8);
    L.q8 = o[2] | (o[3] << 8);
    return L;
}
enum : int { kTabG = 0, kTabX = 128 };                                   // block 0, C = 8
constexpr int kBatchLdsWords = 40960;                                    // 160 KiB: all of the CU's LDS
// a * (the table's constant): four lookups, one v_perm_b32 each.  p, q: the lane's row
// offsets (and block) for the table's geometry; IMM: where the table starts in a row.
template <int IMM>
__device__ __forceinline__ uint32_t ki_mul(uint32_t a, const KiLane& L, uint32_t p, uint32_t q) {
    const uint32_t a0 = __builtin_amdgcn_perm(a, p, L.sel[0]), a1 = __builtin_amdgcn_perm(a, p, L.sel[1]);
    const uint32_t a2 = __builtin_amdgcn_perm(a, q, L.sel[2]), a3 = __builtin_amdgcn_perm(a, q, L.sel[3]);
    return xor3(ki_read<IMM>(a0), ki_read<IMM>(a1), ki_read<IMM>(a2)) ^ ki_read<IMM>(a3);
}
// a * (the table's constant) + w: the fifth term rides in the second three-way XOR
template <int IMM>
__device__ __forceinline__ uint32_t ki_mul_add(uint32_t a, uint32_t w, const KiLane& L, uint32_t p, uint32_t q) {
    const uint32_t a0 = __builtin_amdgcn_perm(a, p, L.sel[0]), a1 = __builtin_amdgcn_perm(a, p, L.sel[1]);
    const uint32_t a2 = __builtin_amdgcn_perm(a, q, L.sel[2]), a3 = __builtin_amdgcn_perm(a, q, L.sel[3]);
    return xor3(xor3(ki_read<IMM>(a0), ki_read<IMM>(a1), ki_read<IMM>(a2)), ki_read<IMM>(a3), w);
}
template <int IMM>
__device__ __forceinline__ uint32_t ki_mul8(uint32_t a, const KiLane& L) {  // G, X
    return ki_mul<IMM>(a, L, L.p8, L.q8);
}
// block 2 (rows of 128 bytes, C = 8): the permute of doubled offsets with block byte 4, halved
__device__ __forceinline__ uint32_t ki_py(uint32_t p8) { return (p8 << 1) | 0x40000u; }
template <int IMM>
__device__ __forceinline__ uint32_t ki_mul2(uint32_t a, const KiLane& L, uint32_t py, uint32_t qy) {
    const uint32_t a0 = __builtin_amdgcn_perm(a, py, L.sel[0]) >> 1, a1 = __builtin_amdgcn_perm(a, py, L.sel[1]) >> 1;
    const uint32_t a2 = __builtin_amdgcn_perm(a, qy, L.sel[2]) >> 1, a3 = __builtin_amdgcn_perm(a, qy, L.sel[3]) >> 1;
    return xor3(ki_read<IMM>(a0), ki_read<IMM>(a1), ki_read<IMM>(a2)) ^ ki_read<IMM>(a3);
}
template <int IMM>
__device__ __forceinline__ uint32_t ki_mul2_add(uint32_t a, uint32_t w, const KiLane& L, uint32_t py, uint32_t qy) {
    const uint32_t a0 = __builtin_amdgcn_perm(a, py, L.sel[0]) >> 1, a1 = __builtin_amdgcn_perm(a, py, L.sel[1]) >> 1;
    const uint32_t a2 = __builtin_amdgcn_perm(a, qy, L.sel[2]) >> 1, a3 = __builtin_amdgcn_perm(a, qy, L.sel[3]) >> 1;
    return xor3(xor3(ki_read<IMM>(a0), ki_read<IMM>(a1), ki_read<IMM>(a2)), ki_read<IMM>(a3), w);
}
__device__ __forceinline__ uint32_t ki_muly_add(uint32_t a, uint32_t w, const KiLane& L, uint32_t py, uint32_t qy) {
    return ki_mul2_add<0>(a, w, L, py, qy);
}
// a * x^(128 * 2^J) + w: the lane-join / chain-join tables, wherever each one lives
template <int J>
__device__ __forceinline__ uint32_t ki_mulK_add(uint32_t a, uint32_t w, const KiLane& L) {
    static_assert(J >= 0 && J <= 3, "K0..K3");
    if constexpr (J == 2) return ki_mul_add<0>(a, w, L, L.p8 | 0x10000u, L.q8 | 0x10000u);
    else if constexpr (J == 0) return ki_mul_add<160>(a, w, L, (L.p8 & 0x3f3fu) | 0x10000u, (L.q8 & 0x3f3fu) | 0x10000u);
    else return ki_mul_add<(J == 1 ? 128 : 224)>(a, w, L, (L.p8 & 0x1f1fu) | 0x10000u, (L.q8 & 0x1f1fu) | 0x10000u);
}
// One byte of the reference's recurrence WITH its table (crc.rs:31), the table being X's
// k = 3 slice; t8 = the lane's offset of such an entry in a row (ki_t8).
__device__ __forceinline__ uint32_t ki_t8(uint32_t lane) { return 16u * ((lane >> 2) & 7u) + 12u; }
__device__ __forceinline__ uint32_t ki_step_byte(uint32_t c, uint32_t byte, uint32_t t8) {
    const uint32_t at = __builtin_amdgcn_perm(c ^ byte, t8, 0x0c0c0400u);  // [t8, low byte of c ^ byte, 0, 0]
    return ki_read<kTabX>(at) ^ (c >> 8);
}

// acc * y + w, the tile and piece kernels' Horner step of one word: the y table sits at
// the start of LDS in the rows-by-byte-value layout above (8 copies, the first half of
// each 256-byte row), so a lookup is one v_perm_b32 and a conflict-free ds_read_b32.
// (Until round 3: [k][b][32 copies], 128 KiB, a byte extract and a shift-add per lookup.)
__device__ __forceinline__ uint32_t mul_y_add(uint32_t a, uint32_t w, const KiLane& L) {
    return ki_mul_add<0>(a, w, L, L.p8, L.q8);
}

// v * x^32: the reference's table step applied to four bytes at once.
template <typename P>
__device__ __forceinline__ uint32_t mul_x32(uint32_t v, P ldsX) {
    return xor3(ldsX[v & 0xffu], ldsX[256 + ((v >> 8) & 0xffu)], ldsX[512 + ((v >> 16) & 0xffu)]) ^
           ldsX[768 + (v >> 24)];
}

// The same two multiplies from tables with 2^RLOG2 copies of every entry in a row
// (the batch kernel's): lane-chosen copy `lr` < 2^RLOG2.
template <int RLOG2, typename P>
__device__ __forceinline__ uint32_t mul_tab(uint32_t a, P tab, uint32_t lr) {
    constexpr int T = 256 << RLOG2;  // words per byte-position table
    const uint32_t i0 = ((a & 0xffu) << RLOG2) + lr;
    const uint32_t i1 = (((a >> 8) & 0xffu) << RLOG2) + lr;
    const uint32_t i2 = (((a >> 16) & 0xffu) << RLOG2) + lr;
    const uint32_t i3 = ((a >> 24) << RLOG2) + lr;
    return tab[i0] ^ tab[T + i1] ^ tab[2 * T + i2] ^ tab[3 * T + i3];
}
template <int RLOG2, typename P>
__device__ __forceinline__ void horner_tab(u32x4& acc, const u32x4 w, P tab, uint32_t lr) {
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = mul_tab<RLOG2>(acc[c], tab, lr) ^ w[c];
}

// One byte of the reference's recurrence WITH its table (crc.rs:31):
// c = CRC_TABLE[(c ^ byte) & 0xff] ^ (c >> 8).
template <typename P>
__device__ __forceinline__ uint32_t table_step_byte(uint32_t c, uint32_t byte, P ldsT8) {
    return ldsT8[(c ^ byte) & 0xffu] ^ (c >> 8);
}

// One Horner step for a lane's four words: acc <- acc * y + w.
#ifndef SIFT_ABLATE_WORDS
#define SIFT_ABLATE_WORDS 4  // timing-only builds thin the step to fewer real multiplies
#endif
__device__ __forceinline__ void horner_step(u32x4& acc, const u32x4 w, const KiLane& L) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
        acc[c] = c < SIFT_ABLATE_WORDS ? mul_y_add(acc[c], w[c], L) : acc[c] ^ w[c];
}

// The y table into LDS: row b (256 bytes) gets eight copies of the four entries
// Y[0..3][b] (word 4c + k).  The host keeps the table transposed (DeviceTables::Yt), so a
// lane fetches the four entries of a row with ONE 16-byte load and stores them as one
// 16-byte word per copy: 2048 (row, copy) pairs, two a lane at 1024 lanes.
constexpr int kStagePairs = 256 * 8;
__device__ __forceinline__ u32x4 stage_fetch(const uint32_t* yt, uint32_t pair) {
    return reinterpret_cast<const u32x4*>(yt)[pair >> 3];
}
__device__ __forceinline__ void stage_store(uint32_t* lds, const u32x4 e, uint32_t pair, uint32_t half = 0) {
    reinterpret_cast<u32x4*>(lds)[(pair >> 3) * 16u + half * 8u + (pair & 7u)] = e;  // half 1: the rows' second 128 bytes
}

// A value every lane of the wave agrees on, moved to scalar registers (the
// compiler sometimes computes uniform 64-bit selects on the vector unit, and a
// buffer descriptor built from vector registers costs a loop per load).
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ u32x4 load_vec(const u32x4* p) { return __builtin_nontemporal_load(p); }

// One lane's vector of a whole tile: buffer_load_dwordx4, non-temporal (the
// stream is read once; aux bit 1 = nt).
#ifndef SIFT_ABL
#define SIFT_ABL 0
#endif
__device__ __forceinline__ u32x4 load_tile(__amdgpu_buffer_rsrc_t srd, int voff, int soff) {
#if SIFT_ABL == 2  // timing-only: no tile comes from memory
    return u32x4{(uint32_t)voff, (uint32_t)soff, 1u, 2u};
#endif
    return __builtin_amdgcn_raw_buffer_load_b128(srd, voff, soff, /*aux*/ SIFT_LOAD_AUX);
}

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

// XOR over lanes with data-parallel-primitive modifiers on the XOR itself (a few
// cycles a step) instead of ds_bpermute shuffles (an LDS round trip a step: the
// reductions at the end of a call sit on its critical path, some 0.3 us a reduction).
// After xor_rows16 every lane holds the XOR of its row of 16 lanes.
__device__ __forceinline__ uint32_t xor_rows16(uint32_t v) {
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);  // row_half_mirror
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true);  // row_mirror
    return v;
}
// XOR over each half-wave, in every lane (uniform per half: returned from lane 31 / 63
// by the caller's choice).  Rows 1 and 3 take in the row before them.
__device__ __forceinline__ uint32_t half_xor(uint32_t v) {
    v = xor_rows16(v);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);  // row_bcast15 into rows 1, 3
    return v;  // lanes 16..31 hold the XOR of lanes 0..31, lanes 48..63 that of 32..63
}
// XOR over the wave, in every lane.
__device__ __forceinline__ uint32_t wave_xor(uint32_t r) {
    r = half_xor(r);
    r ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x143, 0xc, 0xf, true);  // row_bcast31 into rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)r, 63);
}

__device__ __forceinline__ void store_result(uint32_t r, const FinalArgs& f);
// Trailing bytes and the store; one lane.  `body` already carries the state.
// `tailv`: the 16-byte block the trailing bytes lie in, fetched in the prologue (asked for
// here, byte by byte, they were up to 15 dependent trips to memory at the very end of
// the call).
__device__ __forceinline__ void finish(uint32_t body, const FinalArgs& f, u32x4 tailv) {
    uint32_t r = body;
    for (uint32_t i = 0; i < f.tail_len; ++i) {
        r = crc_step_byte(r, tailv[0] & 0xffu);
#pragma unroll
        for (int c = 0; c < 3; ++c) tailv[c] = (tailv[c] >> 8) | (tailv[c + 1] << 24);
        tailv[3] >>= 8;
    }
    store_result(r, f);
}
__device__ __forceinline__ void finish(uint32_t body, const FinalArgs& f) {
    uint32_t r = body;
    for (uint32_t i = 0; i < f.tail_len; ++i) r = crc_step_byte(r, f.tail[i]);
    store_result(r, f);
}
__device__ __forceinline__ void store_result(uint32_t r, const FinalArgs& f) {
    if (f.tag)
        *reinterpret_cast<volatile unsigned long long*>(f.out) =
            ((unsigned long long)f.tag << 32) | (unsigned long long)(r ^ f.xor_out);
    else
        *f.out = r ^ f.xor_out;
}

}  // namespace

template <bool USE_Y>
__global__ __launch_bounds__(kThreads) void crc32_tiles_kernel(TileArgs a, FinalArgs f,
                                                               CombineMats m) {
    // One array (see the guide's second-__shared__-object trap): 256 rows of 256 bytes,
    // the y table in their first halves (not staged by the one-tile variant), the x^32
    // table in their second halves (8 copies each: conflict-free lookups; with one copy of
    // x^32 the 16 lookups a lane makes at the end cost the call half a microsecond).
    __shared__ __align__(16) uint32_t lds[kYWords + kWaves];
    uint32_t* ldsY = lds;
    uint32_t* red = lds + kYWords;

    const uint32_t tid = threadIdx.x;
    const KiLane L = ki_lane(tid);      // where this lane's lookups go in the y table's rows
    if (static_cast<uint32_t>(reinterpret_cast<uintptr_t>((LdsBytes)lds)) != 0u) __builtin_trap();  // ki_read
    const uint32_t row32 = tid & 31u;   // its row in a 32-lane product
    const int64_t g = blockIdx.x;
#if defined(SIFT_TRACE_FINISH)  // timing-only build (tools/trace_groups.py): 100 MHz clock at four points per group
#define SIFT_TRACE(slot) \
    if (tid == 0 && a.single_launch) a.partial[(slot) * 256 + g] = (uint32_t)wall_clock64()
#else
#define SIFT_TRACE(slot)
#endif
    SIFT_TRACE(0);

    // This workgroup's virtual tiles [tv0, tv1); the ones before pad_tiles are
    // all-zero padding in front of the message and contribute nothing.
    int64_t tv0 = g * a.tiles_per_group;
    const int64_t tv1 = tv0 + a.tiles_per_group;
    if (tv0 < a.pad_tiles) tv0 = a.pad_tiles;
    int64_t ntiles = tv1 - tv0;

    // Straight-line prologue, so that the compiler can count the loads in flight:
    // the table words this lane will put into LDS are asked for FIRST (three
    // 16-byte / 4-byte loads, L2 hits), the message right behind them.  Loads
    // return in order, so the tables are staged and the lane's matrix rows built
    // while the first message bytes are still on their way from HBM; asked for
    // after the message, they used to arrive behind it.
    constexpr int kStageRounds = USE_Y ? (kStagePairs + kThreads - 1) / kThreads : 0;  // 2 at 1024 lanes
    constexpr int kXRounds = (kStagePairs + kThreads - 1) / kThreads;
    u32x4 ty[kStageRounds > 0 ? kStageRounds : 1];
    u32x4 tx[kXRounds];
#if !defined(SIFT_ABLATE_FILL)
#pragma unroll
    for (int k = 0; k < kStageRounds; ++k) {
        const uint32_t q = tid + (uint32_t)(k * kThreads);
        ty[k] = stage_fetch(a.tabs->Yt, q < (uint32_t)kStagePairs ? q : 0u);
    }
#endif
#pragma unroll
    for (int k = 0; k < kXRounds; ++k) {
        const uint32_t i = tid + (uint32_t)(k * kThreads);
        tx[k] = stage_fetch(a.tabs->Xt, i < (uint32_t)kStagePairs ? i : 0u);
    }
    uint32_t kb[32];
    kb[0] = a.tabs->Kt[tid];
    // the block the trailing bytes lie in (16-byte aligned, inside the page of the last
    // byte), for whichever group turns out to finish the call
    const u32x4 tailv = *reinterpret_cast<const u32x4*>(f.tail_len ? f.tail : reinterpret_cast<const uint8_t*>(a.vbase));
    __builtin_amdgcn_sched_barrier(0);

    // A lane holds kVpl vectors of a tile (vector t + j*kThreads for j < kVpl),
    // each with its own four running words: kVpl independent Horner chains.
    // Tiles are aligned to the END of the body, so only the first real tile can
    // start before vector 0: lanes in front of it (and every lane of a group
    // without tiles) ask at an offset the descriptor does not cover and get zeros
    // without a memory access.  The head of the message is fixed up after staging.
    constexpr int kBeyond = 0x7FFFFFF0;
    const bool has = ntiles > 0;
    const int64_t v0 = (int64_t)uniform64((uint64_t)(has ? a.nvec - (a.virtual_tiles - tv0) * (int64_t)kTileVecs : 0));
    u32x4 acc[kVpl];
    {
        const __amdgpu_buffer_rsrc_t srd0 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<u32x4*>(a.vbase + v0), /*stride*/ 0, kTileBytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < kVpl; ++j) {
            const int64_t vi = v0 + (int64_t)(j * kThreads) + (int64_t)tid;
            acc[j] = load_tile(srd0, has && vi >= 0 ? (int)((j * kThreads + tid) * 16u) : kBeyond, 0);
        }
    }
    // `p` is the next tile's first vector: the same for every lane, so it and its
    // increments live in scalar registers and each load is base + lane offset.
    const u32x4* p = reinterpret_cast<const u32x4*>(
        uniform64(reinterpret_cast<uint64_t>(a.vbase + v0 + (has ? kTileVecs : 0))));
    ntiles = (int64_t)uniform64((uint64_t)(has ? ntiles - 1 : ntiles));

    // The rest of the run goes through a buffer descriptor: 128-bit SRD in
    // scalar registers, per-lane offset tid*16 in one VGPR, tile offset in an
    // SGPR -- no 64-bit vector address arithmetic in the loop.  Every tile from
    // here on is whole, so the range check never fires.
    const uint32_t run_bytes = (uint32_t)(ntiles > 0 ? ntiles : 0) * (uint32_t)kTileBytes;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u32x4*>(p), /*stride*/ 0, (int)run_bytes, 0x00020000);
    const int voff = (int)(tid * 16u);
    int soff = 0;
    constexpr int U = kUnroll;
    // First batch of the main loop, also before the tables are staged.
    u32x4 cur[U][kVpl];
    const bool batched = USE_Y && ntiles >= U;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < kVpl; ++j)
            cur[u][j] = load_tile(srd, batched ? voff : kBeyond, soff + u * kTileBytes + j * kThreads * 16);
    if (batched) {
        soff += U * kTileBytes;
        ntiles -= U;
    }
    __builtin_amdgcn_sched_barrier(0);

    // Tables -> LDS.
#if !defined(SIFT_ABLATE_FILL)
#pragma unroll
    for (int k = 0; k < kStageRounds; ++k) {
        const uint32_t q = tid + (uint32_t)(k * kThreads);
        if (q < (uint32_t)kStagePairs) stage_store(ldsY, ty[k], q);
    }
#endif
#pragma unroll
    for (int k = 0; k < kXRounds; ++k) {
        const uint32_t i = tid + (uint32_t)(k * kThreads);
        if (i < (uint32_t)kStagePairs) stage_store(lds, tx[k], i, 1u);
    }
    // This lane's place in the tile, Kt[tid], as the 32 rows kt * x^i of its
    // multiplication matrix: computed here, while the first loads are in flight
    // and the VALU is idle, and kept in registers through the main loop, so that
    // the multiply at the end is 2 instructions per bit instead of 7.  (The rows loaded from
    // a 1024 x 32 table instead: 2.5 us slower a call, eight more loads a lane in the memory pipe.)
#pragma unroll
    for (int i = 1; i < 32; ++i) kb[i] = gf2_mulx(kb[i - 1]);
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(kb[i]));  // pin: do not sink past the loop
    __syncthreads();
    SIFT_TRACE(1);

    // The head of the message: the lane that holds vector 0 zeroes the bytes in
    // front of it and folds the starting register in.
    if (has) {
#pragma unroll
        for (int j = 0; j < kVpl; ++j) {
            const int64_t vi = v0 + (int64_t)(j * kThreads) + (int64_t)tid;
            if (vi == 0) {
                if (a.head_skip) acc[j] = mask_head(acc[j], a.head_skip);
                acc[j][0] ^= a.state_ptr ? gf2_mulmod(*a.state_ptr, a.xinv_h) : a.state_fold;
            }
        }
    }
    // Single-launch finish (a.single_launch): wave 0 will weight this group's
    // residue by z^(G-1-g) at the end; its matrix rows have the whole stream to arrive.
    uint32_t rows[kTreeLevels];
    if (a.single_launch && tid < 64u) {
#pragma unroll
        for (int s = 0; s < kTreeLevels; ++s) rows[s] = m.z[s][row32];
    }

    if (USE_Y) {
        if (batched) {
            // Two register sets take turns (cur is folded while nxt loads, then
            // the other way round), so no batch is ever copied between registers.
            u32x4 nxt[U][kVpl];
            while (ntiles >= 2 * U) {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int j = 0; j < kVpl; ++j)
                        nxt[u][j] = load_tile(srd, voff, soff + u * kTileBytes + j * kThreads * 16);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int j = 0; j < kVpl; ++j) horner_step(acc[j], cur[u][j], L);
                // The batch before must have landed before the next one is asked
                // for (one batch in flight per wave: two cost 3 % of HBM rate), and
                // a batch's loads go out back to back.
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int j = 0; j < kVpl; ++j)
                        cur[u][j] = load_tile(srd, voff, soff + (U + u) * kTileBytes + j * kThreads * 16);
                __builtin_amdgcn_sched_barrier(0);
                soff += 2 * U * kTileBytes;
                ntiles -= 2 * U;
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int j = 0; j < kVpl; ++j) horner_step(acc[j], nxt[u][j], L);
            }
            if (ntiles >= U) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int j = 0; j < kVpl; ++j)
                        nxt[u][j] = load_tile(srd, voff, soff + u * kTileBytes + j * kThreads * 16);
                soff += U * kTileBytes;
                ntiles -= U;
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int j = 0; j < kVpl; ++j) horner_step(acc[j], cur[u][j], L);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int j = 0; j < kVpl; ++j) cur[u][j] = nxt[u][j];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < kVpl; ++j) horner_step(acc[j], cur[u][j], L);
        }
        while (ntiles > 0) {
#pragma unroll
            for (int j = 0; j < kVpl; ++j) {
                const u32x4 w = load_tile(srd, voff, soff + j * kThreads * 16);
                horner_step(acc[j], w, L);
            }
            soff += kTileBytes;
            --ntiles;
        }
    }

    SIFT_TRACE(2);
    // Lane residue: each of its vectors folded over its four words with the
    // x^32 table, earlier vectors carried forward by x^(128*kThreads) (uniform
    // constant, rows C1); then the lane's place in the tile; then XOR over the
    // workgroup.
    uint32_t r = 0;
#pragma unroll
    for (int j = 0; j < kVpl; ++j) {
        if (j > 0) {
            uint32_t prod = 0;
#pragma unroll
            for (int i = 0; i < 32; ++i) prod ^= a.tabs->C1[i] & (0u - ((r >> (31 - i)) & 1u));
            r = prod;
        }
        uint32_t rj = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) rj = ki_mul8<kTabX>(rj ^ acc[j][c], L);
        r ^= rj;
    }
#if defined(SIFT_ABLATE_MULMOD)  // timing-only build (tools/): wrong results by design
    r ^= kb[0] ^ kb[31];
#else
    {
        uint32_t prod = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) prod = xor_and(prod, kb[i], (uint32_t)((int32_t)(r << i) >> 31));
        r = prod;
    }
#endif
    r = wave_xor(r);
    if ((tid & 63u) == 0) red[tid >> 6] = r;
    __syncthreads();
    if (tid >= 64u) return;

    // Wave 0 finishes the group.
    uint32_t x = wave_xor(tid < (uint32_t)kWaves ? red[tid] : 0u);
    if (a.fuse_final) {  // one group did everything
        if (tid == 0) finish(x, f, tailv);
        return;
    }
    if (!a.single_launch) {  // two-launch form: raw residue, combine kernel weights it
        if (tid == 0) a.partial[g] = x;
        return;
    }

    SIFT_TRACE(3);
    // Weight: x * z^h, h = G-1-g, as a product of the z^(2^s) with bit s of h
    // set; each factor applied by the half-wave together (lane k holds row k).
    const uint32_t G = (uint32_t)f.groups;
    const uint32_t h = G - 1u - (uint32_t)g;
#if defined(SIFT_ABLATE_WEIGHT)
    x ^= rows[0] ^ h;
#else
#pragma unroll
    for (int s = 0; s < kTreeLevels; ++s) {
        if ((h >> s) & 1u) {  // uniform
            const uint32_t t = ((x >> (31u - row32)) & 1u) ? rows[s] : 0u;
            x = (uint32_t)__builtin_amdgcn_readlane((int)half_xor(t), 31);
        }
    }
#endif

#if defined(SIFT_ABLATE_EXCHANGE)
    if (tid == 0) a.partial[g] = x;
    return;
#endif
    // Exchange: ONE returning 8-byte agent-scope atomic per level.  Groups meet in
    // shards of 32 (g >> 5); a shard's word holds the XOR of the residues that
    // have arrived in its low half and one arrival bit per member in its high
    // half, and a group XORs in (its bit | its residue).  Read-modify-write atomics
    // on one word are serialised at the memory side (microarch guide, "Valid
    // forms": atomics both sides; neither L1 nor a per-XCD L2 holds a copy), so the
    // OLD value a group gets back tells it exactly which members came before it
    // and the XOR of their residues: the group that sees every other bit set is
    // the shard's last and already holds the shard's total (old ^ own) -- no
    // second atomic to count arrivals, no third to fetch the sum.  The shards'
    // last groups meet the same way on the top word (one bit per shard), and the
    // last of those finishes.  Each last arriver XORs the word's final value in
    // again, which puts it back to zero for the next call (not waited for: the
    // kernel's end orders it before the next launch).
    if (tid == 0) {
        unsigned long long* words = reinterpret_cast<unsigned long long*>(a.exchange);  // 128-B lines
        const uint32_t nshards = (G + 31u) >> 5;
        const uint32_t shard = (uint32_t)g >> 5, bit = (uint32_t)g & 31u;
        const uint32_t members = G - (shard << 5) < 32u ? G - (shard << 5) : 32u;
        const uint32_t full1 = members == 32u ? 0xFFFFFFFFu : (1u << members) - 1u;
        unsigned long long* w1 = words + 16u * (1u + shard);
        unsigned long long old = __hip_atomic_fetch_xor(w1, ((unsigned long long)(1u << bit) << 32) | x,
                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (((uint32_t)(old >> 32) | (1u << bit)) != full1) return;
        const uint32_t xs = (uint32_t)old ^ x;  // the shard's total
        __hip_atomic_fetch_xor(w1, ((unsigned long long)full1 << 32) | xs, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        uint32_t total = xs;
        if (nshards > 1u) {
            const uint32_t full2 = nshards == 32u ? 0xFFFFFFFFu : (1u << nshards) - 1u;
            old = __hip_atomic_fetch_xor(words, ((unsigned long long)(1u << shard) << 32) | xs,
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (((uint32_t)(old >> 32) | (1u << shard)) != full2) return;
            total = (uint32_t)old ^ xs;
            __hip_atomic_fetch_xor(words, ((unsigned long long)full2 << 32) | total, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        finish(total, f, tailv);
    }
}

// Tree over the workgroup residues.  Residue of group g carries weight
// z^(G-1-g); with h = G-1-g the weights are z^h, and level s folds element
// h + 2^s into element h (h a multiple of 2^(s+1)) after multiplying it by
// z^(2^s).  Each such product is done by 32 lanes together: lane k holds row k
// of the constant's matrix, zmat[s][k] = z^(2^s) * x^k (built on the host), and
// contributes it when bit 31-k of the operand is set; a 5-step XOR butterfly
// inside the half-wave sums the rows.  ~15 instructions per level instead of a
// 32-step serial multiply per lane.
__global__ __launch_bounds__(kMaxGroups) void crc32_combine_kernel(const uint32_t* partial,
                                                                   FinalArgs f, CombineMats m) {
    __shared__ uint32_t sh[kMaxGroups];
    const int tid = threadIdx.x;
    const int G = f.groups;
    if (tid < G) sh[tid] = partial[G - 1 - tid];
    const int k = tid & 31;
    const int hw = tid >> 5;
    constexpr int kHalfWaves = kMaxGroups / 32;
    // All ten rows up front: one memory latency instead of one per level.
    uint32_t rows[kTreeLevels];
#pragma unroll
    for (int s = 0; s < kTreeLevels; ++s) rows[s] = m.z[s][k];
#pragma unroll
    for (int s = 0; s < kTreeLevels; ++s) {
        if ((1 << s) >= G) break;
        __syncthreads();
        const uint32_t row = rows[s];
        const int stride = 2 << s;
        const int nprod = (G - (1 << s) + stride - 1) / stride;
        for (int j0 = 0; j0 < nprod; j0 += kHalfWaves) {
            const int j = j0 + hw;
            const bool valid = j < nprod;
            const uint32_t a = valid ? sh[j * stride + (1 << s)] : 0u;
            uint32_t t = ((a >> (31 - k)) & 1u) ? row : 0u;
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) t ^= __shfl_xor(t, off, 64);
            if (valid && k == 0) sh[j * stride] ^= t;
        }
    }
    __syncthreads();
    if (tid == 0) {
        // G == 0: serial path, the whole (< 32 byte) input is the "tail"
        const uint32_t start = f.state_ptr ? *f.state_ptr : f.state_imm;
        finish(G > 0 ? sh[0] : start, f);
    }
}


// ---------------------------------------------------------------------------
// Batch kernel: many independent buffers, ONE WAVE PER BUFFER.  This is the
// data-parallel shape of the reference's load path when many files / chunks
// are checked at once (Chunk::read x n, container.rs:47-66).  Same algebra as
// the tile kernel at wave scale: a tile is 64 lanes x 16 B = 1 KiB, the Horner
// step multiplies by y1 = x^(8*1024).  The starting register is folded into
// the first four message bytes (valid for len >= 4: byte k of the state has
// the same weight x^(8(len-k)) as message byte k), so no x^(8 len) is needed.
// Buffers shorter than 32 bytes are walked by one lane.
namespace {

// XOR the bytes of the starting register that belong to vector `vi` into it:
// message byte k (k = 0..3) sits h + k bytes into vector 0.
__device__ __forceinline__ u32x4 fold_state(u32x4 w, int64_t vi, uint32_t h, uint32_t state) {
    if (vi > 1) return w;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t pos = h + k;
        if ((int64_t)(pos >> 4) == vi) {
            const uint32_t word = (pos & 15u) >> 2;
            const uint32_t bits = ((state >> (8u * k)) & 0xffu) << (8u * (pos & 3u));
            w[0] ^= word == 0 ? bits : 0u;
            w[1] ^= word == 1 ? bits : 0u;
            w[2] ^= word == 2 ? bits : 0u;
            w[3] ^= word == 3 ? bits : 0u;
        }
    }
    return w;
}

// The same for a vector known to be vector 0 or 1, with shifts instead of a byte loop
// (the group paths, where this runs once per pass): the state is a 32-bit word laid at
// byte h of vectors 0|1, i.e. at byte sh = h - 16 * vi of this one, straddling words
// sh >> 2 and (sh >> 2) + 1.
__device__ __forceinline__ u32x4 fold_state_word(u32x4 w, int32_t vi, uint32_t h, uint32_t state) {
    const int32_t sh = (int32_t)h - 16 * vi;  // -16 .. 15; under -3 nothing lands here
    const uint32_t bs = 8u * ((uint32_t)sh & 3u);
    const uint32_t lo = state << bs, hi = bs ? state >> (32u - bs) : 0u;
    const int32_t wi = sh >> 2;  // -4 .. 3
#pragma unroll
    for (int c = 0; c < 4; ++c) w[c] ^= (wi == c ? lo : 0u) ^ (wi + 1 == c ? hi : 0u);
    return w;
}

// The 16 bytes of v moved up by d (0..15) byte positions, zeros behind them.
__device__ __forceinline__ u32x4 shl_bytes(u32x4 v, uint32_t d) {
    const uint32_t down = 32u - 8u * (d & 3u), q = d >> 2;  // a 64-bit shift by 32 is allowed: d & 3 == 0
    u32x4 t;
    t[0] = (uint32_t)(((uint64_t)v[0] << 32) >> down);
#pragma unroll
    for (int c = 1; c < 4; ++c) t[c] = (uint32_t)((((uint64_t)v[c] << 32) | v[c - 1]) >> down);
    u32x4 o;
    o[0] = q == 0 ? t[0] : 0u;
    o[1] = q == 0 ? t[1] : q == 1 ? t[0] : 0u;
    o[2] = q == 0 ? t[2] : q == 1 ? t[1] : q == 2 ? t[0] : 0u;
    o[3] = q == 0 ? t[3] : q == 1 ? t[2] : q == 2 ? t[1] : t[0];
    return o;
}

// The first two tiles and the trailing bytes of a span the whole wave will take, requested
// ahead of time: every load unconditional (what the span does not have is read at vector 0
// and dropped later), so that they stay in flight across whatever comes next.
struct WaveHead {
    u32x4 t0, t1;
    uint32_t tb;
};
__device__ __forceinline__ void wave_request(const uint8_t* ptr, uint64_t len, uint32_t lane, WaveHead& hd) {
    typedef __attribute__((address_space(1))) const u32x4* GlobalVec;
    typedef __attribute__((address_space(1))) const uint8_t* GlobalByte;
    const uint64_t addr = reinterpret_cast<uint64_t>(ptr);
    const uint64_t v0 = addr >> 4, v1 = (addr + len) >> 4;
    const int64_t nvec = (int64_t)(v1 - v0);
    const GlobalVec vbase = (GlobalVec)(v0 << 4);
    const int64_t ntile = (nvec + 63) >> 6;
    const int64_t vi = nvec - ntile * 64 + (int64_t)lane;
    const uint32_t tail_len = (uint32_t)((addr + len) & 15u);
    hd.t0 = vbase[vi >= 0 ? vi : 0];
    hd.t1 = vbase[ntile > 1 ? vi + 64 : 0];
    hd.tb = ((GlobalByte)((tail_len ? v1 : v0) << 4))[lane < tail_len ? lane : 0u];
}

// CRC of a span of more than kGroupMax bytes by the whole wave.  `hd` holds the span's
// first tiles (wave_request); the NEXT span the wave will take (np, nl; this one again if
// there is none) is requested into `nxt` before this span's lanes are joined: the join
// is a chain of dependent lookups and shuffles with nothing in flight, and the next
// span's first tile used to be asked for only behind it -- two idle stretches per span.
__device__ __forceinline__ uint32_t wave_crc(const uint8_t* ptr, uint64_t len, uint32_t state,
                                             const KiLane& L, uint32_t lane, const WaveHead& hd, const uint8_t* np,
                                             uint64_t nl, WaveHead& nxt) {
    const uint32_t py = ki_py(L.p8), qy = ki_py(L.q8);  // the wave table's rows (block 2)
    auto horner_y = [&](u32x4& a, const u32x4 w) {
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] = ki_muly_add(a[c], w[c], L, py, qy);
    };
    const uint64_t addr = reinterpret_cast<uint64_t>(ptr);
    const uint64_t v0 = addr >> 4, v1 = (addr + len) >> 4;
    const int64_t nvec = (int64_t)(v1 - v0);  // >= 1, body >= 17 bytes
    const uint32_t h = (uint32_t)(addr & 15u);
    // The address came through lane shuffles as an integer: say that it is global memory,
    // or the loads below are flat loads, which count as LDS traffic too -- every wait for
    // a table lookup then waits for the data in flight as well.
    typedef __attribute__((address_space(1))) const u32x4* GlobalVec;
    const GlobalVec vbase = (GlobalVec)(v0 << 4);
    const int64_t ntile = (nvec + 63) >> 6;
    int64_t vi = nvec - ntile * 64 + (int64_t)lane;  // tiles aligned to the end of the body

    // Tile 0 may start before vector 0 (front padding, head mask).  The four
    // state bytes land in vectors 0 and 1, which are in tile 0 or, when tile 0
    // holds vector 0 alone, in tiles 0 and 1: both are peeled.
    u32x4 acc;
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = vi >= 0 ? hd.t0[c] : 0u;
    // the < 16 bytes behind the last whole vector: lane k fetched byte k beside the data
    const uint32_t tail_len = (uint32_t)((addr + len) & 15u);
    const uint32_t tb = lane < tail_len ? hd.tb : 0u;
    if (vi >= 0) {
        if (vi == 0 && h) acc = mask_head(acc, h);
        acc = fold_state(acc, vi, h, state);
    }
    vi += 64;
    int64_t left = ntile - 1;
    GlobalVec p = vbase + vi;
    if (left > 0) {
        const u32x4 w = fold_state(hd.t1, vi, h, state);
        horner_y(acc, w);
        p += 64;
        --left;
    }
    constexpr int B = SIFT_WAVE_BATCH;  // tiles in flight while the B before them are folded
    if (left >= B) {
        u32x4 cc[B], nn[B];
#pragma unroll
        for (int u = 0; u < B; ++u) cc[u] = __builtin_nontemporal_load(p + 64 * u);
        p += 64 * B;
        for (left -= B; left >= B; left -= B) {
#pragma unroll
            for (int u = 0; u < B; ++u) nn[u] = __builtin_nontemporal_load(p + 64 * u);
            p += 64 * B;
#pragma unroll
            for (int u = 0; u < B; ++u) horner_y(acc, cc[u]);
#pragma unroll
            for (int u = 0; u < B; ++u) cc[u] = nn[u];
        }
        // the tiles behind the last whole batch are asked for before that batch is folded
#pragma unroll
        for (int u = 0; u < B - 1; ++u)
            if (left > u) nn[u] = __builtin_nontemporal_load(p + 64 * u);
#pragma unroll
        for (int u = 0; u < B; ++u) horner_y(acc, cc[u]);
#pragma unroll
        for (int u = 0; u < B - 1; ++u)
            if (left > u) horner_y(acc, nn[u]);
        left = 0;
    }
    for (; left > 0; --left) {
        const u32x4 w = __builtin_nontemporal_load(p);
        p += 64;
        horner_y(acc, w);
    }
    wave_request(np, nl, lane, nxt);
    uint32_t r = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) r = ki_mul8<kTabX>(r ^ acc[c], L);
    // the 64 lanes joined pairwise, as the lanes of a group are: at step j the lane whose
    // bit j is clear holds the earlier 2^j vectors, multiplies by x^(128 * 2^j) and takes
    // in its partner's.  Steps 0-3 look up the lane-join tables, step 4 the group path's
    // x^(8*256) table, step 5 the same twice.  (Until round 2 every lane multiplied by its
    // place, x^(128 * (63 - lane)), in 32 shift-and-xor steps: some 220 instructions a
    // span against 90 here.  The time per span did not change -- this path waits for
    // memory -- but the kernel needs 9-15 registers fewer and no longer spills.)
    r = ki_mulK_add<0>(r, __shfl_xor(r, 1, 64), L);
    r = ki_mulK_add<1>(r, __shfl_xor(r, 2, 64), L);
    r = ki_mulK_add<2>(r, __shfl_xor(r, 4, 64), L);
    r = ki_mulK_add<3>(r, __shfl_xor(r, 8, 64), L);
    r = ki_mul_add<kTabG>(r, __shfl_xor(r, 16, 64), L, L.p8, L.q8);
    r = ki_mul_add<kTabG>(ki_mul8<kTabG>(r, L), __shfl_xor(r, 32, 64), L, L.p8, L.q8);
    r = __shfl(r, 0, 64);
    // trailing bytes: every lane walks them alike (no divergence, no broadcast)
    const uint32_t t8 = ki_t8(lane);
    for (uint32_t i = 0; i < tail_len; ++i) r = ki_step_byte(r, __shfl(tb, (int)i, 64), t8);
    return r;
}

// CRC of a span of 32 .. kGroupMax bytes by ONE GROUP OF GW LANES (8, 4 or 2), 64/GW spans
// per wave at a time: the wave kernel's algebra at group scale.  The span is cut into
// tiles of GW vectors, aligned to its end so that only the first can be partial; lane l
// of the group takes vector l of every tile.  A lane keeps NCH accumulators ("chains"):
// the tiles, counted from the end, go to the chains round-robin.  An 8-lane group has
// two, so a chain steps over 16 vectors and every Horner step multiplies by x^(8*256)
// from the one kGroupRepl-copy table (mul_tab on ldsG), as a 16-lane group with one chain
// would; the 4- and 2-lane shapes only take spans that fit their chains (4 x 4, 4 x 2
// and 2 x 2 vectors: 256, 128 and 64 bytes) and have no Horner loop at all.  At the end
// a lane's chains are joined with x^(128*GW) (the lane-join table of that constant), then
// the lanes.  The work per byte is the same for every shape; what a narrow group saves
// is the work per PASS (the x^32 fold, the lane join, the head, the call), which 8, 16
// or 32 spans share, and the lanes left idle by a span shorter than a tile.  (Round 2
// began with 16-lane groups, four spans a pass: 64-byte items 4.4x slower, 1 KiB items
// 15 %, mixed 32..4080-byte items 5 %; nothing was faster.)  Every lane of the wave calls
// this; `on` says whether the lane's group has a span.  The register over the whole span
// comes back in lane 0 of the group.  Not inlined: inside the batch kernel's item loop,
// beside the wave path, it pushed the kernel past 128 VGPRs; as a call it costs a few
// instructions per pass -- but the kernel around it can only keep what the callee
// leaves of the 128 registers, which is why the rounds below are written to stay near 50.
// The first round's vectors of a group's span (and, in the aligned framing, the vector its
// trailing bytes lie in): what group_crc_body loads first, as a request of its own, so that
// two passes' loads can be in flight before the first pass waits (group_crc_pair).  Same
// geometry and the same unconditional loads as in group_crc_body.
template <int NCH>
struct GroupLoads {
    u32x4 ch[NCH];
    u32x4 tw;
    u32x4 h0;  // vector 0 of the group's span, in every lane of the group
};
template <int GW, int NCH, bool ENDF>
__device__ __forceinline__ void group_request(const uint8_t* ptr, uint32_t len, bool on, uint32_t lane,
                                              GroupLoads<NCH>& g) {
    constexpr int LG = GW == 8 ? 3 : GW == 4 ? 2 : 1;
    constexpr int LCH = NCH == 4 ? 2 : 1;
    constexpr bool ONE_ROUND = GW < 8;
    typedef u32x4 __attribute__((aligned(1))) u32x4_u;
    typedef __attribute__((address_space(1))) const u32x4_u* GlobalVecU;
    typedef __attribute__((address_space(1))) const u32x4* GlobalVec;
    const uint32_t lg = lane & (uint32_t)(GW - 1);
    const uint64_t addr = reinterpret_cast<uint64_t>(ptr);
    const int32_t nvec = ENDF ? (int32_t)((len + 15u) >> 4) : (int32_t)(((addr + len) >> 4) - (addr >> 4));
    const uint32_t lead = ENDF ? (uint32_t)nvec * 16u - len : (uint32_t)(addr & 15u);
    const uint32_t tail_len = ENDF || !on ? 0u : (uint32_t)((addr + len) & 15u);
    const int32_t ntile = on ? (nvec + GW - 1) >> LG : 0;
    auto vec = [&](int32_t vi) {
        if constexpr (ENDF) return (GlobalVecU)(addr + (uint64_t)(uint32_t)(vi > 0 ? 16 * vi - (int32_t)lead : 0));
        else return (GlobalVec)(((addr >> 4) + (uint64_t)(uint32_t)(vi > 0 ? vi : 0)) << 4);
    };
    const int32_t nround = ONE_ROUND ? (ntile > 0 ? 1 : 0) : (ntile + NCH - 1) >> LCH;
    const int32_t tile0 = NCH * (nround - 1);
    const int32_t vi0 = nvec - GW * (tile0 + 1) + (int32_t)lg;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int32_t vi = vi0 - GW * c;
        g.ch[c] = *vec(on && tile0 + c < ntile && vi >= 0 ? vi : 0);
    }
    if constexpr (!ENDF) g.tw = *vec(tail_len ? nvec : 0);
    if constexpr (ONE_ROUND) g.h0 = *vec(0);
}

template <int GW, int NCH, bool ENDF>
__device__ __forceinline__ uint32_t group_crc_body(const uint8_t* ptr, uint32_t len, uint64_t state2, bool on,
                                                   const KiLane& L, uint32_t lane,
                                                   const GroupLoads<NCH>* pre = nullptr) {
    static_assert((GW == 8 && NCH == 2) || (GW == 4 && (NCH == 4 || NCH == 2)) || (GW == 2 && NCH == 2), "group shape");
    constexpr int LG = GW == 8 ? 3 : GW == 4 ? 2 : 1;   // log2(GW): steps of the lane join
    constexpr int LCH = NCH == 4 ? 2 : 1;               // log2(chains per lane)
    constexpr bool ONE_ROUND = GW < 8;                  // the span is at most GW * NCH vectors
    // A call passes a generic pointer; say which memory it points into, or every
    // table lookup becomes a flat load that waits on both counters.
    typedef u32x4 __attribute__((aligned(1))) u32x4_u;
    typedef __attribute__((address_space(1))) const u32x4_u* GlobalVecU;
    typedef __attribute__((address_space(1))) const u32x4* GlobalVec;
    const uint32_t lg = lane & (uint32_t)(GW - 1);
    // Two framings.  ENDF: the span is framed from its END: vector v covers the 16 bytes
    // that end 16 * (nvec - 1 - v) bytes before the span's end, whatever their address (the
    // hardware takes 16-byte loads at any byte address; off a 4-byte boundary they stream
    // at 5.1 instead of 6.5 TB/s).  Nothing trails the last vector -- no byte-at-a-time
    // tail, 8 instructions a byte for every lane of the wave, as many bytes as the longest
    // tail of the pass -- and vector 0 is the short one: it begins `lead` bytes before the
    // span does.  Those bytes are not read (they may lie before the start of an
    // allocation): vector 0 is read AT the span's first byte and moved up instead.
    // Otherwise: vectors are the aligned 16-byte blocks of memory; vector 0 has `lead`
    // bytes of something else in front, masked off, and up to 15 bytes trail the last one.
    // The first wins while spans are short (a mix of 32..512-byte spans at all alignments:
    // 15 % faster), the second when the loads matter more than the tail (a mix of
    // 32..4080: 5 % slower framed from the end, 2 KB documents 10 %).
    const uint64_t addr = reinterpret_cast<uint64_t>(ptr);
    const int32_t nvec = ENDF ? (int32_t)((len + 15u) >> 4) : (int32_t)(((addr + len) >> 4) - (addr >> 4));  // 2..255
    const uint32_t lead = ENDF ? (uint32_t)nvec * 16u - len : (uint32_t)(addr & 15u);                        // 0..15
    const uint32_t tail_len = ENDF || !on ? 0u : (uint32_t)((addr + len) & 15u);
    const int32_t ntile = on ? (nvec + GW - 1) >> LG : 0;
    auto vec = [&](int32_t vi) {  // vector vi > 0; anything else reads vector 0 (ENDF: the span's first 16 bytes)
        if constexpr (ENDF) return (GlobalVecU)(addr + (uint64_t)(uint32_t)(vi > 0 ? 16 * vi - (int32_t)lead : 0));
        else return (GlobalVec)(((addr >> 4) + (uint64_t)(uint32_t)(vi > 0 ? vi : 0)) << 4);
    };
    // the < 16 bytes behind the last whole vector: every lane fetches the vector they lie
    // in (one address per group), issued behind the first data vectors
    u32x4 tw = {0u, 0u, 0u, 0u};
    u32x4 acc = {0u, 0u, 0u, 0u};
    uint32_t folded = 0;     // two-chain shapes: the lane's register, already folded and joined
    bool is_folded = false;  // wave-uniform
    {
        // A round is one tile per chain; rounds are counted from the first (t = 0, the
        // only one that can be short of tiles).  In round t chain c holds tile
        // NCH * (nround - 1 - t) + c, counted from the END of the body, which is vectors
        // nvec - GW * (tile + 1) .. + GW - 1.  Every load is unconditional (a vector the
        // lane does not have is read at a clamped index and zeroed by a select): a load
        // under a divergent branch makes the compiler wait for it on the spot, one memory
        // latency per chain.  Within a round all loads are issued first and the chains
        // are then stepped one after the other (scheduling barriers): interleaved by the
        // compiler they took 70-100 registers, and a callee that big makes the kernel
        // around it spill.  4-lane groups only take spans of one round (kGroup4Max <= 256
        // bytes: at most 16 vectors at any alignment).
        const int32_t nround = ONE_ROUND ? (ntile > 0 ? 1 : 0) : (ntile + NCH - 1) >> LCH;
        const int32_t tile0 = NCH * (nround - 1);
        const int32_t vi0 = nvec - GW * (tile0 + 1) + (int32_t)lg;  // round 0, chain c: vector vi0 - GW * c
        u32x4 ch[NCH];
        bool head1 = false;  // 8-lane shape: this lane holds vector 1 in round 1 (state bytes reach into it)
        if constexpr (ONE_ROUND || ENDF) {
            // Vector 0 needs moving up (or its head masking) and the starting register.  The
            // register goes into vector 0 alone: as the word state * x^-32 in the four zero bytes
            // just BEFORE the first message byte when there are four (lead >= 4; a word there
            // weighs x^(8 len + 32)), else as the state itself over the first four message bytes
            // (bytes lead .. lead + 3 < 8).  (Round 2 always used the second form, which reaches
            // into vector 1 for lead > 12: a second fix-up, in another chain or another round.)
            // Every lane fetches vector 0 of its group's span itself (h0: one address a group)
            // and works the fix-up out; the lane that holds vector 0 -- in round 0, chain
            // vi0 >> LG, if vi0 is a multiple of GW -- takes the result.  (Round 2 picked the
            // vector out of the lane's chains and put it back: 8 selects a chain.)
            const bool own0 = on && vi0 >= 0 && (vi0 & (GW - 1)) == 0;
            const int32_t hc = vi0 >> LG;
            {
                bool ex[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int32_t vi = vi0 - GW * c;
                    ex[c] = on && tile0 + c < ntile && vi >= 0;
                    ch[c] = pre ? pre->ch[c] : *vec(ex[c] ? vi : 0);
                }
                if constexpr (!ENDF) tw = pre ? pre->tw : *vec(tail_len ? nvec : 0);
                // (the 8-lane shape picks vector 0 out of its two chains instead: there the
                // extra load costs more than the 8 selects)
                u32x4 h0 = {0u, 0u, 0u, 0u};
                if constexpr (ONE_ROUND) h0 = pre ? pre->h0 : *vec(0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!ONE_ROUND) {
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const bool hd = own0 && hc == c;
#pragma unroll
                        for (int k = 0; k < 4; ++k) h0[k] = hd ? ch[c][k] : h0[k];
                    }
                }
                const u32x4 ones = {~0u, ~0u, ~0u, ~0u};
                u32x4 up = ENDF ? shl_bytes(h0, lead) : h0 & mask_head(ones, lead);
                {
                    const uint32_t sw = lead >= 4u ? (uint32_t)(state2 >> 32) : (uint32_t)state2;
                    const uint32_t pb = lead >= 4u ? lead - 4u : lead;  // 0 .. 11: the word lies inside vector 0
                    const uint32_t bs = 8u * (pb & 3u), wi = pb >> 2;
                    const uint32_t lo = sw << bs, hi = bs ? sw >> (32u - bs) : 0u;
#pragma unroll
                    for (uint32_t k = 0; k < 4u; ++k) up[k] ^= (wi == k ? lo : 0u) ^ (wi + 1u == k ? hi : 0u);
                }
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const bool hd = own0 && hc == c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) ch[c][k] = hd ? up[k] : ex[c] ? ch[c][k] : 0u;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // The 8-lane shape on memory's own 16-byte blocks (spans over 1 KiB) keeps round 2's
            // fix-up (the state over the first four message bytes, which may reach into vector 1;
            // the vector picked out of the lane's chains): with the form above mixed
            // 32..4080-byte items were 6 % slower (framed from the end, 1000-byte items are 8 %
            // faster with it).
            // Vector 0 needs moving up (or its head masking) and vectors 0 and 1 the state bytes
            // (at bytes lead .. lead + 3 of the two).  A lane holds at most one of them: in round 0, chain vi0 >> LG (its
            // vector there is vi0 mod GW), or -- when vector 0 is alone in its tile -- vector 1
            // in round 1, last chain.
            const bool head0 = on && vi0 >= 0 && (vi0 & (GW - 1)) <= 1;
            head1 = on && vi0 == 1 - GW;
            const int32_t hvi = head1 ? 1 : (vi0 & (GW - 1)), hc = vi0 >> LG;
            const u32x4 delta = fold_state_word(u32x4{0u, 0u, 0u, 0u}, hvi, lead, (uint32_t)state2);
            {
                bool ex[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int32_t vi = vi0 - GW * c;
                    ex[c] = on && tile0 + c < ntile && vi >= 0;
                    ch[c] = pre ? pre->ch[c] : *vec(ex[c] ? vi : 0);
                }
                if constexpr (!ENDF) tw = pre ? pre->tw : *vec(tail_len ? nvec : 0);
                __builtin_amdgcn_sched_barrier(0);
                u32x4 hw = {0u, 0u, 0u, 0u};  // the lane's vector 0 or 1, if it has one
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const bool hd = head0 && hc == c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        ch[c][k] = ex[c] ? ch[c][k] : 0u;
                        hw[k] = hd ? ch[c][k] : hw[k];
                    }
                }
                const u32x4 ones = {~0u, ~0u, ~0u, ~0u};
                const u32x4 up = ENDF ? shl_bytes(hw, lead) : hw & mask_head(ones, lead);
#pragma unroll
                for (int k = 0; k < 4; ++k) hw[k] = (hvi == 0 ? up[k] : hw[k]) ^ delta[k];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const bool hd = head0 && hc == c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) ch[c][k] = hd ? hw[k] : ch[c][k];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // Rounds 1 .. : round t + 1's vectors are requested before round t is stepped (a
        // round read and then stepped is a trip to memory per 256 bytes of span with
        // nothing else in flight for the group).
        // (Only while some group of the pass has a round to ask for: with two-round spans
        // alone -- 0.5 KB documents -- the idle requests cost 12 %.)
        if (!ONE_ROUND && __any(nround > 1)) {
            u32x4 w[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) w[c] = __builtin_nontemporal_load(vec(nround > 1 ? vi0 + 16 - GW * c : 0));
            for (int32_t t = 1; __any(t < nround); ++t) {
                const bool ha = t < nround;
                const int32_t vn = vi0 + 16 * (t + 1);  // next round, chain c: vector vn - GW * c
                u32x4 wn[NCH];
                if (__any(t + 1 < nround)) {
#pragma unroll
                    for (int c = 0; c < NCH; ++c) wn[c] = __builtin_nontemporal_load(vec(t + 1 < nround ? vn - GW * c : 0));
                } else {
#pragma unroll
                    for (int c = 0; c < NCH; ++c) wn[c] = u32x4{0u, 0u, 0u, 0u};
                }
                __builtin_amdgcn_sched_barrier(0);
                if (t == 1 && __any(head1)) {  // worked out again here: four registers fewer across the loop
                    const u32x4 d1 = fold_state_word(u32x4{0u, 0u, 0u, 0u}, 1, lead, (uint32_t)state2);
#pragma unroll
                    for (int k = 0; k < 4; ++k) w[NCH - 1][k] ^= head1 ? d1[k] : 0u;
                }
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (ha) {  // two words at a time: half the lookups in flight, half the registers
#pragma unroll
                        for (int k = 0; k < 2; ++k) ch[c][k] = ki_mul_add<kTabG>(ch[c][k], w[c][k], L, L.p8, L.q8);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k = 2; k < 4; ++k) ch[c][k] = ki_mul_add<kTabG>(ch[c][k], w[c][k], L, L.p8, L.q8);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int c = 0; c < NCH; ++c) w[c] = wn[c];
            }
        }
        // the lane's chains: chain c is GW vectors (x^(128*GW), lane-join table LG) ahead
        // of chain c - 1.  Steps no group of the wave needs are skipped (all zero so far).
        // With two chains each is folded over its four words FIRST (x^32 table, 8 copies)
        // and the two 32-bit registers joined with 4 lookups: joining the 128-bit
        // accumulators took 16 in a table that only has room for 2 or 4 copies (LDS bank
        // conflicts 41 % of the LDS cycles at 512-byte items).
        if constexpr (NCH == 2) {
            if (__any(ntile > 1)) {
                uint32_t f1 = 0, f0 = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f1 = ki_mul8<kTabX>(f1 ^ ch[1][k], L);
                    f0 = ki_mul8<kTabX>(f0 ^ ch[0][k], L);
                }
                folded = ki_mulK_add<LG>(f1, f0, L);
                is_folded = true;
            } else {
                acc = ch[0];  // chain 1 is empty in every group of the wave
            }
        } else {
            acc = ch[NCH - 1];
#pragma unroll
            for (int c = NCH - 2; c >= 0; --c) {
                if (__any(ntile > c + 1)) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = ki_mulK_add<LG>(acc[k], ch[c][k], L);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] ^= ch[c][k];
                }
            }
        }
    }
    // the lane's four words, then the GW lanes joined pairwise: at step j the
    // lane whose bit j is clear holds the earlier 2^j vectors, multiplies by
    // x^(128 * 2^j) (four lookups in K[j]) and takes in its partner's.  Lane 0 of the
    // group ends up with the span's register.  (A 32-step register multiply by the
    // lane's place, x^(128 * (GW - 1 - lane)), costs 7 instructions a bit; this path is
    // VALU-bound -- a wave64 instruction takes four cycles -- so lookups win.  Four
    // register multiplies instead of the x^32 table were tried too: 1.9x slower.)
    uint32_t r = folded;
    if (!is_folded) {  // wave-uniform
        r = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) r = ki_mul8<kTabX>(r ^ acc[c], L);
    }
    r = ki_mulK_add<0>(r, __shfl_xor(r, 1, 64), L);
    if constexpr (LG > 1) r = ki_mulK_add<1>(r, __shfl_xor(r, 2, 64), L);
    if constexpr (LG > 2) r = ki_mulK_add<2>(r, __shfl_xor(r, 4, 64), L);
    // trailing bytes, with the reference's own table; only lane 0 of a group holds a
    // meaningful register, the others walk along
    if constexpr (!ENDF) {
        const uint32_t t8 = ki_t8(lane);
#pragma unroll
        for (uint32_t i = 0; i < 15u; ++i) {
            if (!__any(i < tail_len)) break;
            const uint32_t step = ki_step_byte(r, (tw[i >> 2] >> (8u * (i & 3u))) & 0xffu, t8);
            r = i < tail_len ? step : r;
        }
    }
    return r;
}

template <int GW, int NCH, bool ENDF>
__device__ __attribute__((noinline)) uint32_t group_crc(const uint8_t* ptr, uint32_t len, uint64_t state2, bool on,
                                              KiLane L, uint32_t lane) {
    return group_crc_body<GW, NCH, ENDF>(ptr, len, state2, on, L, lane);
}

// Two passes in one: both spans' vectors are requested before the first is worked on, so a
// turn makes half as many dependent trips to memory (and half as many calls).  Results in
// the low (a) and high (b) word.
template <int GW, int NCH, bool ENDF>
__device__ __forceinline__ uint64_t group_crc_pair_body(const uint8_t* pa, uint32_t la, bool ona, const uint8_t* pb,
                                                        uint32_t lb, bool onb, uint64_t state2, const KiLane& L,
                                                        uint32_t lane) {
    GroupLoads<NCH> ga, gb;
    group_request<GW, NCH, ENDF>(pa, la, ona, lane, ga);
    group_request<GW, NCH, ENDF>(pb, lb, onb, lane, gb);
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t ra = group_crc_body<GW, NCH, ENDF>(pa, la, state2, ona, L, lane, &ga);
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t rb = group_crc_body<GW, NCH, ENDF>(pb, lb, state2, onb, L, lane, &gb);
    return ((uint64_t)rb << 32) | ra;
}
template <int GW, int NCH, bool ENDF>
__device__ __attribute__((noinline)) uint64_t group_crc_pair(const uint8_t* pa, uint32_t la, bool ona,
                                                             const uint8_t* pb, uint32_t lb, bool onb, uint64_t state2,
                                                             KiLane L, uint32_t lane) {
    return group_crc_pair_body<GW, NCH, ENDF>(pa, la, ona, pb, lb, onb, state2, L, lane);
}

// CRC of a span under 32 bytes by ONE LANE (64 spans per wave at a time): the
// bytes are fetched first, all at once, then walked with the reference's table step.
__device__ __forceinline__ uint32_t lane_crc(const uint8_t* ptr_, uint32_t len, uint32_t state, bool on,
                                             uint32_t lane) {
    const uint32_t t8 = ki_t8(lane);
    uint32_t r = state;
    on = on && len != 0u;
    typedef __attribute__((address_space(1))) const uint8_t* GlobalByte;  // see wave_crc
    const GlobalByte ptr = (GlobalByte) reinterpret_cast<uint64_t>(ptr_);
    uint32_t w[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    if (on) {
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) w[k >> 2] |= (uint32_t)ptr[k < len ? k : len - 1u] << (8u * (k & 3u));
    }
    if (__any(on && len > 8u)) {  // most such spans are a bare chunk type (4 bytes)
        if (on) {
#pragma unroll
            for (uint32_t k = 8; k < 31u; ++k)
                w[k >> 2] |= (uint32_t)ptr[k < len ? k : len - 1u] << (8u * (k & 3u));
        }
    }
    if (on) {
#pragma unroll
        for (uint32_t k = 0; k < 31u; ++k)
            if (k < len) r = ki_step_byte(r, (w[k >> 2] >> (8u * (k & 3u))) & 0xffu, t8);
    }
    return r;
}

__device__ __forceinline__ uint32_t be32(const uint8_t* p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

// N little-endian words from ANY byte address p (p .. p + 4N - 1 must be readable): N + 1
// aligned dword loads and N funnel shifts.  The chunk headers and stored CRCs of the
// container format sit at arbitrary addresses; read a byte at a time (be32) a document's
// 37 header and verdict bytes were 37 load instructions whose 64 lanes each touch a
// different cache line -- four times as many requests to the CU's one memory pipe as the
// document's payload makes.  The dword behind the last byte wanted may lie outside the
// buffer, so it is only asked for when the window really reaches into it.
template <int N>
__device__ __forceinline__ void load_words(const uint8_t* p, uint32_t (&out)[N]) {
    typedef __attribute__((address_space(1))) const uint32_t* GlobalWord;
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    const uint32_t o = (uint32_t)a & 3u;
    const GlobalWord q = (GlobalWord)(a & ~3ull);
    uint32_t v[N + 1];
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = q[j];
    v[N] = q[o ? N : N - 1];
#pragma unroll
    for (int j = 0; j < N; ++j) out[j] = __builtin_amdgcn_alignbyte(v[j + 1], v[j], o);
}

// x <- x * c, c given as the 32 rows c * x^k of its matrix: lane k of each
// half-wave holds row k and contributes it when bit 31-k of x is set; a 5-step
// butterfly sums the rows.  Every lane of the wave returns the product.
__device__ __forceinline__ uint32_t mul_rows(uint32_t x, uint32_t row, uint32_t row32) {
    const uint32_t t = ((x >> (31u - row32)) & 1u) ? row : 0u;
    return (uint32_t)__builtin_amdgcn_readlane((int)half_xor(t), 31);
}

// Level and piece count of a span whose whole vectors cover `body` bytes.
__device__ __forceinline__ uint32_t span_level(uint64_t body, uint32_t piece_min_log2) {
    uint32_t level = 0;
    while (level < kMaxLevel - (piece_min_log2 - (uint32_t)kPieceMinLog2) &&
           (body >> (piece_min_log2 + level)) >= (uint64_t)kMaxPiecesPerItem)
        ++level;
    return level;
}

// A wave that meets a long CRC span files it for the piece kernel instead of
// walking it.  Wave-uniform.  Returns false (nothing changed) when cutting is off,
// the span is short, or the record / piece arrays are full: the caller then
// walks the span itself, which is slow for a long span but gives the same bits.
__device__ __forceinline__ bool file_span(const BatchArgs& b, uint32_t item, const uint8_t* ptr,
                                          uint64_t len, uint32_t cont, uint32_t dlen,
                                          uint32_t lane) {
    const CutArgs& c = b.cut;
    if (!c.counts || len <= (uint64_t)c.wave_max) return false;
    const uint64_t A = reinterpret_cast<uint64_t>(ptr);
    const uint64_t body = ((A + len) & ~15ull) - (A & ~15ull);  // whole vectors, > 0 here
    const uint32_t level = span_level(body, c.piece_min_log2);
    const uint32_t plog = c.piece_min_log2 + level;
    const uint64_t np64 = (body + (1ull << plog) - 1) >> plog;
    if (np64 > 0x40000000ull) return false;
    const uint32_t np = (uint32_t)np64;
    // Reserve a record slot and np piece slots: two independent atomic adds (plain
    // adds, not compare-and-swap: a thousand waves filing at once would retry
    // each other for milliseconds).  A reservation that does not fit is voided:
    // whatever part of it lies inside the arrays is marked so that the piece
    // kernel and the finisher skip it, and the caller walks the span itself.
    uint32_t slot = 0;
    unsigned long long pb = 0;
    if (lane == 0) slot = __hip_atomic_fetch_add(c.counts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 1)
        pb = __hip_atomic_fetch_add(c.pieces, (unsigned long long)np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    slot = __shfl(slot, 0, 64);
    pb = ((unsigned long long)__shfl((uint32_t)(pb >> 32), 1, 64) << 32) | __shfl((uint32_t)pb, 1, 64);
    if (slot >= c.cap_recs || pb + np > (unsigned long long)c.cap_pieces) {
        if (lane == 0 && slot < c.cap_recs) c.recs[slot].np = 0u;
        const unsigned long long stop = pb + np < c.cap_pieces ? pb + np : c.cap_pieces;
        for (unsigned long long t = pb + lane; t < stop; t += 64u) c.piece_rec[t] = kVoidSlot;
        return false;
    }
    const uint32_t pbase = (uint32_t)pb;
    for (uint32_t t = lane; t < np; t += 64u) c.piece_rec[pbase + t] = slot;
    if (lane == 0) {
        CutRecord r;
        r.span_off = (uint64_t)(ptr - b.base);
        r.span_len = len;
        r.item = item;
        r.piece_base = pbase;
        r.np = np;
        r.level = level;
        r.cont = cont;
        r.dlen = dlen;
        r.acc = 0ull;
        r.arrived = 0ull;
        r.pad[0] = r.pad[1] = 0u;
        c.recs[slot] = r;
    }
    return true;
}

// Chunk::read (container.rs:47-66) by one wave on the serialized chunk at ptr,
// `avail` bytes readable.  Returns SIFT_OK (0), Error::Read (1),
// ReadInvalidChunkSize (3) or ReadCrcMismatch (5), raised in the reference's
// order; *dlen is the chunk's data length once the length field has been read.
// The checks that come before the CRC (:48-60) are split off so that the caller
// lane that parses a chunk can hand its span on (wave, group, lane, or the piece kernel).
__device__ __forceinline__ uint32_t read_chunk_head(const uint8_t* ptr, uint64_t avail,
                                                    uint32_t limit, uint32_t* dlen) {
    *dlen = 0;
    if (avail < 4) return 1;                      // read_u32 fails, :48
    const uint32_t n = be32(ptr);
    if (n > limit) return 3;                      // :49-52
    *dlen = n;
    if (avail < 12ull + (uint64_t)n) return 1;    // read_exact on type / data / crc fails, :54-60
    return 0;
}

}  // namespace

// ---------------------------------------------------------------------------
// Batch kernel.  A wave takes W consecutive items at a time (W = 1 .. 64, chosen
// by the host so that every wave of the chip has work), ONE ITEM PER LANE for
// everything that is not the CRC itself:
//   A. lane j reads item j's descriptor and, in the chunk modes, its header
//      (length, type, stored crc) and runs the checks that precede the CRC in the
//      reference (container.rs:48-60, :109-113): descriptor and header latency is
//      paid once per W items, not once per item;
//   B. the CRC spans, by length:  under 32 bytes    one lane each, 64 at a time
//                                 32 .. 4080 bytes  one group of lanes each, by the turn's
//                                                   longest: 2 lanes up to 64 bytes (32 at
//                                                   a time), 4 up to 256 (16), else 8 (8)
//                                 up to wave_max    the whole wave, one at a time
//                                 longer            filed for the piece kernel
//   C. lane j turns its item's registers into the verdict and stores it.
// MODE is BatchArgs::verify.  In the document mode an item has two spans (AMRG
// chunk, SEND chunk); the SEND span is done first so that a long AMRG chunk can
// be filed with everything that follows its CRC check already decided.
namespace {

constexpr uint32_t kTinyMax = 31, kGroupMax = 4080;  // at most 255 vectors = 16 rounds of an 8-lane group
#ifndef SIFT_GROUP4_MAX
#define SIFT_GROUP4_MAX 256
#endif
#ifndef SIFT_END_FRAMED_MAX
#define SIFT_END_FRAMED_MAX 1024
#endif
#ifndef SIFT_DOC_PAIR
#define SIFT_DOC_PAIR 0  // 1: the document mode pairs its passes of up to 128 bytes too (it then spills 48 bytes a lane)
#endif
constexpr uint32_t kGroup4Max = SIFT_GROUP4_MAX;  // widest span of a turn that runs on 4-lane groups
static_assert(kGroup4Max >= 128 && kGroup4Max <= 256, "4-lane groups take one round of 2 or 4 chains");
constexpr uint32_t kEndFramedMax = SIFT_END_FRAMED_MAX;  // ... that is framed from the spans' ends
static_assert(kEndFramedMax >= kGroup4Max, "4-lane groups are framed from the end");

// wave_max from the lengths (crc32_device.h, wave_max_of): lane l of EVERY wave looks at
// sample members l, l + 64, l + 128, l + 192 (wave_sample asks for them early), and the wave
// counts their classes by ballot.
__device__ __forceinline__ void wave_sample(const BatchArgs& b, uint64_t (&len4)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t m = (threadIdx.x & 63u) + 64u * (uint32_t)k;
        len4[k] = m < b.n && m < kWaveSample ? b.lens[wave_sample_item(m, b.n)] : 0ull;
    }
}
__device__ __forceinline__ uint32_t wave_threshold(const BatchArgs& b, const uint64_t (&len4)[4]) {
    uint32_t cnt[3] = {0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t cl = wave_class(len4[k]);  // (a member the batch does not have has length 0: no class)
#pragma unroll
        for (uint32_t q = 0; q < 3u; ++q) cnt[q] += (uint32_t)__builtin_popcountll(__ballot(cl == q + 1u));
    }
    return wave_max_of(cnt[0], cnt[1], cnt[2], wave_sample_stride(b.n));
}

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) {
    return ((uint64_t)__shfl((uint32_t)(v >> 32), src, 64) << 32) | __shfl((uint32_t)v, src, 64);
}

// The spans `slot` of this wave's items: registers over the whole span into `res`
// (lanes with want == true), except for lanes whose span was filed (`filed`).
// `may_file`: long spans may go to the piece kernel (primary span only).
template <int MODE>
__device__ __forceinline__ void crc_spans(const BatchArgs& b, bool want, uint64_t sp, uint64_t sl,
                                          uint32_t item, uint32_t cont, uint32_t dlen, bool may_file,
                                          uint32_t& res, bool& filed, const KiLane& L,
                                          uint32_t lane) {
    const uint32_t state = MODE == 0 ? b.state : 0xFFFFFFFFu;
    const uint64_t state2 = ((uint64_t)b.state_m32 << 32) | state;  // with state * x^-32 (group_crc_body's head)
    // under 32 bytes: a lane each
    const bool tiny = want && sl <= (uint64_t)kTinyMax;
    if (__any(tiny)) {
        const uint32_t r = lane_crc(reinterpret_cast<const uint8_t*>(sp), (uint32_t)sl, state, tiny, lane);
        if (tiny) res = r;
    }
    // 32 .. 4080 bytes: a group of lanes each.  The spans of the turn are taken in ORDER OF
    // LENGTH (classes of 64, 128, 256 bytes, then one per 256 bytes; lane order inside a
    // class) and every pass picks its own shape from the spans it holds: 2 lanes while the
    // next 32 are all of 64 bytes or less, 4 lanes while the next 16 fit 256, else 8 lanes for
    // the next 8.  A pass lasts as long as its longest span, so spans of like length belong
    // together: taken in lane order with ONE shape for the turn (round 2), a turn of mixed
    // 32..4080-byte spans kept its groups busy little more than half the time, and one long
    // span made a turn of short ones run 8 to a pass instead of 32.
    const bool mid = want && sl > (uint64_t)kTinyMax && sl <= (uint64_t)kGroupMax;
    const unsigned long long m = __ballot(mid);
    if (m) {
        static_assert(kGroup4Max == 256 && kEndFramedMax == 1024, "the length classes below");
        constexpr uint32_t kClass64 = 0, kClass128 = 1, kClass256 = 2, kClassEndFramed = 5, kClasses = 18;
        const uint32_t len32 = (uint32_t)sl;
        const uint32_t cls = !mid ? 31u : len32 <= 64u ? kClass64 : len32 <= 128u ? kClass128 : 2u + ((len32 - 1u) >> 8);
        const uint32_t cnt = (uint32_t)__builtin_popcountll(m);
        // rank of every span in (class, lane) order; lanes without one come behind
        const uint32_t below = (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        uint32_t rank = mid ? below : cnt + (lane - below);
        const uint32_t cls0 = (uint32_t)__builtin_amdgcn_readlane((int)cls, __builtin_ctzll(m));
        if (__ballot(cls == cls0) != m) {  // more than one class in the turn (else lane order is class order)
            uint32_t base = 0;
            for (uint32_t v = 0; v < kClasses; ++v) {
                const unsigned long long mv = __ballot(cls == v);
                if (!mv) continue;
                const uint32_t within =
                    __builtin_amdgcn_mbcnt_hi((uint32_t)(mv >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mv, 0u));
                rank = cls == v ? base + within : rank;
                base += (uint32_t)__builtin_popcountll(mv);
            }
        }
        // lane r learns which lane holds the span of rank r, and that span's class
        const uint32_t holder = (uint32_t)__builtin_amdgcn_ds_permute((int)(rank << 2), (int)lane);
        const uint32_t cls_at = __shfl(cls, (int)holder, 64);
        auto class_at = [&](uint32_t r) { return (uint32_t)__builtin_amdgcn_readlane((int)cls_at, (int)(r < cnt ? r : cnt - 1u)); };
        for (uint32_t first = 0; first < cnt;) {
            // this pass: its shape from the longest class among the spans it would hold
            const uint32_t c32 = class_at(first + 31u), c16 = class_at(first + 15u), c8 = class_at(first + 7u);
            const uint32_t lgw = c32 == kClass64 ? 1u : c16 <= kClass256 ? 2u : 3u;  // log2(lanes per group)
            const uint32_t per_pass = 64u >> lgw;
            const uint32_t top = lgw == 1u ? c32 : lgw == 2u ? c16 : c8;             // longest class of the pass
            // two passes at a time while the next one has the same shape
            // (plain buffers: every end-framed shape; chunk modes: up to 128 bytes, larger pairs
            // make them spill; the document mode: none, it has no registers for that)
            const uint32_t top2 = class_at(first + 2u * per_pass - 1u);  // >= top: the spans are in class order
            const bool same2 = lgw == 1u ? top2 == kClass64 : lgw == 2u ? top2 <= kClass256 : top2 <= kClassEndFramed;
            const bool pair = first + per_pass < cnt && same2 &&
                              ((top2 <= kClass128 && (MODE != 2 || SIFT_DOC_PAIR)) || (MODE == 0 && top2 <= kClassEndFramed));
            if (pair) {
                const uint32_t ia = first + (lane >> lgw), ib = ia + per_pass;
                const bool ona = ia < cnt, onb = ib < cnt;
                const int ma = (int)__shfl(holder, (int)(ona ? ia : 0u), 64), mb = (int)__shfl(holder, (int)(onb ? ib : 0u), 64);
                const uint8_t* pa = reinterpret_cast<const uint8_t*>(shfl64(sp, ma));
                const uint8_t* pb = reinterpret_cast<const uint8_t*>(shfl64(sp, mb));
                const uint32_t la = __shfl(len32, ma, 64), lb = __shfl(len32, mb, 64);
                // (the smallest shape is inlined: 3 % off a turn of 64-byte chunks; the next one
                // inlined as well makes the document mode spill)
                const uint64_t rr = lgw == 1u ? group_crc_pair_body<2, 2, true>(pa, la, ona, pb, lb, onb, state2, L, lane)
                                    : lgw == 3u ? group_crc_pair<8, 2, true>(pa, la, ona, pb, lb, onb, state2, L, lane)
                                    : top2 <= kClass128 ? group_crc_pair<4, 2, true>(pa, la, ona, pb, lb, onb, state2, L, lane)
                                                        : group_crc_pair<4, 4, true>(pa, la, ona, pb, lb, onb, state2, L, lane);
                const uint32_t k = rank - first;  // which of the 2 * per_pass spans, for the lanes it concerns
                const int src = (int)((k & (per_pass - 1u)) << lgw);
                const uint32_t va = __shfl((uint32_t)rr, src, 64), vb = __shfl((uint32_t)(rr >> 32), src, 64);
                if (mid && rank >= first && k < 2u * per_pass) res = k < per_pass ? va : vb;
                first += 2u * per_pass;
            } else {
                const uint32_t idx = first + (lane >> lgw);
                const bool on = idx < cnt;
                const int mine = (int)__shfl(holder, (int)(on ? idx : 0u), 64);
                const uint8_t* p = reinterpret_cast<const uint8_t*>(shfl64(sp, mine));
                const uint32_t l = __shfl(len32, mine, 64);
                uint32_t r;
                if (lgw == 1u) r = group_crc<2, 2, true>(p, l, state2, on, L, lane);
                else if (lgw == 2u && top <= kClass128) r = group_crc<4, 2, true>(p, l, state2, on, L, lane);
                else if (lgw == 2u) r = group_crc<4, 4, true>(p, l, state2, on, L, lane);
                else if (top <= kClassEndFramed) r = group_crc<8, 2, true>(p, l, state2, on, L, lane);
                else r = group_crc<8, 2, false>(p, l, state2, on, L, lane);
                // back to the lanes that hold the spans: rank - first is the group that did it
                const uint32_t v = __shfl(r, (int)(((rank - first) & (per_pass - 1u)) << lgw), 64);
                if (mid && rank >= first && rank < first + per_pass) res = v;
                first += per_pass;
            }
        }
    }
    // longer: the whole wave, one at a time -- or filed for the piece kernel
    unsigned long long ml = __ballot(want && sl > (uint64_t)kGroupMax);
    WaveHead hd;
    int requested = -1;  // the lane whose span `hd` was requested for
    while (ml) {
        const int s = (int)__builtin_ctzll(ml);
        ml &= ml - 1;
        const uint8_t* p = reinterpret_cast<const uint8_t*>(uniform64(shfl64(sp, s)));
        const uint64_t l = uniform64(shfl64(sl, s));
        if (may_file && b.cut.long_flag && l > (uint64_t)b.cut.wave_max) {
            // wide plan: the piece kernel finds this span itself (plan_wide makes the same test)
            if (lane == 0) *b.cut.long_flag = 1u;
            if ((int)lane == s) filed = true;
            continue;
        }
        if (may_file && file_span(b, __shfl(item, s, 64), p, l, __shfl(cont, s, 64), __shfl(dlen, s, 64), lane)) {
            if ((int)lane == s) filed = true;
            continue;
        }
        if (requested != s) wave_request(p, l, lane, hd);
        // the span behind this one (it may yet be filed: its two tiles are then dropped);
        // behind the last one, this span's last 64 bytes (just read: no new traffic)
        const int ns = ml ? (int)__builtin_ctzll(ml) : -1;
        const uint8_t* np = ns >= 0 ? reinterpret_cast<const uint8_t*>(uniform64(shfl64(sp, ns))) : p + (l - 64u);
        const uint64_t nl = ns >= 0 ? uniform64(shfl64(sl, ns)) : 64u;
        WaveHead nxt;
        const uint32_t r = wave_crc(p, l, state, L, lane, hd, np, nl, nxt);
        hd = nxt;
        requested = ns;
        if ((int)lane == s) res = r;
    }
}

}  // namespace

template <int MODE>
__global__ __launch_bounds__(kThreads) void crc32_batch_kernel(BatchArgs b) {
    // all 160 KiB of the CU's LDS: the row-interleaved tables (layout at KiLane), copied
    // from the image the host built, 16 bytes a lane at a time
    __shared__ __align__(16) uint32_t lds[kBatchLdsWords];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    uint64_t len4[4] = {0, 0, 0, 0};
    {
        const u32x4* img = reinterpret_cast<const u32x4*>(b.tabs->batch_image);
        static_assert(kBatchLdsWords % (4 * kThreads) == 0, "whole rounds of 16-byte copies");
        u32x4 t[kBatchLdsWords / (4 * kThreads)];
#pragma unroll
        for (int r = 0; r < kBatchLdsWords / (4 * kThreads); ++r) t[r] = img[tid + (uint32_t)(r * kThreads)];
        if (b.cut.wave_adapt) wave_sample(b, len4);  // asked for behind the tables, looked at behind the barrier
#pragma unroll
        for (int r = 0; r < kBatchLdsWords / (4 * kThreads); ++r)
            reinterpret_cast<u32x4*>(lds)[tid + (uint32_t)(r * kThreads)] = t[r];
    }
    // the lookups address LDS absolutely (ki_read): the tables must start at address 0
    if (static_cast<uint32_t>(reinterpret_cast<uintptr_t>((LdsBytes)lds)) != 0u) __builtin_trap();
    const KiLane L = ki_lane(lane);
    if (b.cut.clear_flag && blockIdx.x == 0 && tid == 0) *b.cut.clear_flag = 0u;  // the next wide-plan call's flag
    __syncthreads();
    if (b.cut.wave_adapt) b.cut.wave_max = wave_threshold(b, len4);

    // Turns of W items: turn q goes to workgroup q mod G, then to its wave (q / G) mod 16 --
    // ACROSS the workgroups first.  (Consecutive turns on the waves of one workgroup put a
    // batch of 1024 long items, one a wave, on 64 of the 256 CUs: 2.7 TB/s, bound by those
    // CUs' vector issue.)
    // Descriptors the GPU reads in place in pinned host memory: 2^deal_log2 = 8 consecutive
    // turns stay on one workgroup (a 64-byte line of each array crosses PCIe once, not once
    // per XCD that holds one of its turns).
    const uint32_t W = b.items_per_wave;
    const uint32_t wv = tid >> 6, dl = b.deal_log2;
    const uint64_t wave = ((((uint64_t)(wv >> dl) * gridDim.x + blockIdx.x) << dl) + (wv & ((1u << dl) - 1u)));
    const uint64_t step = (uint64_t)gridDim.x * kWaves * W;
    for (uint64_t first = wave * W; first < (uint64_t)b.n; first += step) {
        const bool have = lane < W && first + lane < (uint64_t)b.n;
        const uint32_t i = (uint32_t)(first + lane);

        // ---- A: descriptor, header, the checks that come before the CRC ----
        // Kept per lane across the CRCs: the item's address, its data length(s) and
        // what the checks said; stored crcs and chunk types are read again in C.
        uint32_t status = 0, dlen = 0;      // status != 0: verdict already known
        uint32_t st2 = 0, d2 = 0;           // MODE 2: the SEND chunk's checks and data length
        uint32_t type2 = 0;                 // MODE 2: the SEND chunk's type bytes (used right after A)
        uint64_t len0 = 0;                  // MODE 0: the buffer's length
        bool span0 = false, span1 = false;  // CRC spans still to do
        const uint8_t* ptr = b.base;
        // Wide reads (load_words) of what the verdict will compare, made here, beside the
        // headers: the stored CRC(s), and in the document mode the AMRG chunk's type.  Only
        // when every item of the turn is long enough for the windows to lie inside it
        // (wave-uniform; any valid item is); otherwise the byte-wise code below and in C.
        bool pre = false;                   // stored0 (and type0) are in registers
        bool pre2 = false;                  // MODE 2: so are the SEND chunk's header and, if it is empty, its CRC
        uint32_t stored0 = 0, type0 = 0, stored2 = 0;
        uint64_t avail_w = 0;
        if (MODE != 0 && have) {
            avail_w = b.lens[i];
            ptr = b.base + b.offsets[i];
        }
        // lanes without an item read where lane 0's item is (the wave's first lane always has one)
        const uint8_t* ptr0 = reinterpret_cast<const uint8_t*>(uniform64(reinterpret_cast<uint64_t>(ptr)));
        if (MODE == 1 && __all(!have || avail_w >= 12u)) {  // Chunk::read, container.rs:47-66
            pre = true;
            uint32_t w[1];
            load_words<1>(have ? ptr : ptr0, w);
            const uint32_t n = __builtin_bswap32(w[0]);
            if (have) {
                if (n > b.chunk_limit) status = 3;  // :49-52
                else {
                    dlen = n;
                    if (avail_w < 12ull + (uint64_t)n) status = 1;  // read_exact fails, :54-60
                }
            }
            span0 = have && status == 0;
            load_words<1>(span0 ? ptr + 8ull + dlen : ptr0, w);
            stored0 = __builtin_bswap32(w[0]);
        } else if (MODE == 2 && __all(!have || avail_w >= 32u)) {  // read_document's framing, document.rs:62-77
            pre = true;
            uint32_t h[5];
            load_words<5>(have ? ptr : ptr0, h);  // magic (9 bytes) | length | type
            if (have) {
                if (h[0] != 0x46495389u || h[1] != 0x1A0A0D54u || (h[2] & 0xffu) != 0x0Au) status = 1;  // container.rs:113
                if (status == 0) {  // read_chunk_head on the AMRG chunk
                    const uint32_t n = __builtin_bswap32(__builtin_amdgcn_alignbyte(h[3], h[2], 1));
                    if (n > b.chunk_limit) status = 3;
                    else {
                        dlen = n;
                        if (avail_w - 9u < 12ull + (uint64_t)n) status = 1;
                    }
                }
                type0 = __builtin_bswap32(__builtin_amdgcn_alignbyte(h[4], h[3], 1));
            }
            span0 = have && status == 0;
            const uint64_t pos2 = 21ull + dlen;  // the SEND chunk starts right behind
            if (__all(!span0 || avail_w >= pos2 + 12u)) {
                // one window: the AMRG chunk's stored CRC | the SEND chunk's length | type | its CRC if it is empty
                pre2 = true;
                uint32_t w[4];
                load_words<4>(span0 ? ptr + pos2 - 4u : ptr0, w);
                stored0 = __builtin_bswap32(w[0]);
                type2 = __builtin_bswap32(w[2]);
                stored2 = __builtin_bswap32(w[3]);
                if (span0) {  // read_chunk_head on the SEND chunk
                    const uint32_t n2 = __builtin_bswap32(w[1]);
                    if (n2 > b.chunk_limit) st2 = 3;
                    else {
                        d2 = n2;
                        if (avail_w - pos2 < 12ull + (uint64_t)n2) st2 = 1;
                    }
                    span1 = st2 == 0;
                }
            } else if (span0) {
                uint32_t w[1];
                load_words<1>(ptr + pos2 - 4u, w);  // inside the item: avail >= 21 + dlen
                stored0 = __builtin_bswap32(w[0]);
                if (avail_w - pos2 >= 8u) type2 = be32(ptr + pos2 + 4);
                st2 = read_chunk_head(ptr + pos2, avail_w - pos2, b.chunk_limit, &d2);
                span1 = st2 == 0;
            }
        } else if (have) {
            const uint64_t off = b.offsets[i], avail = b.lens[i];
            ptr = b.base + off;
            if (MODE == 0) {  // crc.rs:28-34 on buffer i
                span0 = true;
                len0 = avail;
            } else if (MODE == 1) {  // Chunk::read, container.rs:47-66
                status = read_chunk_head(ptr, avail, b.chunk_limit, &dlen);
                span0 = status == 0;
            } else if (MODE == 3) {  // Chunk::write's crc step, container.rs:75; avail = bytes of the slot
                if (avail < 12u) {
                    status = 2;  // Error::Write: no room
                } else {
                    dlen = be32(ptr);
                    if (avail < 12ull + (uint64_t)dlen) status = 2;
                }
                span0 = status == 0;
            } else {  // read_document's framing, document.rs:62-77
                if (avail < 9u) {
                    status = 1;  // read_exact on the header fails (container.rs:109)
                } else {
                    const uint8_t magic[9] = {0x89, 0x53, 0x49, 0x46, 0x54, 0x0D, 0x0A, 0x1A, 0x0A};
#pragma unroll
                    for (int k = 0; k < 9; ++k)
                        if (ptr[k] != magic[k]) status = 1;  // InvalidData (container.rs:113)
                }
                if (status == 0) status = read_chunk_head(ptr + 9, avail - 9, b.chunk_limit, &dlen);
                if (status == 0) {
                    span0 = true;
                    // the SEND chunk starts right behind; whether it is reached at all is
                    // decided in C, in the reference's order
                    const uint64_t pos2 = 21ull + dlen;
                    if (avail - pos2 >= 8u) type2 = be32(ptr + pos2 + 4);  // beside the length field, not behind it
                    st2 = read_chunk_head(ptr + pos2, avail - pos2, b.chunk_limit, &d2);
                    span1 = st2 == 0;
                }
            }
        }
        // where the spans are: type | data of the chunk at `at` (container.rs:79-84)
        constexpr uint32_t kChunkAt = MODE == 2 ? 9u : 0u;  // first chunk, from the item's start
        const uint8_t* send = ptr + 21ull + dlen;           // MODE 2: the second chunk

        // ---- B: the CRCs ----  (one instance of the span code: the document mode
        // runs it twice, SEND span first)
        uint32_t r0 = 0, r1 = 0;
        uint32_t after0 = 0;  // MODE 2: the file's status if its AMRG chunk's CRC matches
        bool filed = false;
        bool send_done = false;
        if (MODE == 2 && !__any(span1 && d2 != 0u)) {
            // Every SEND chunk of the turn is empty (document.rs:59 writes them so): its CRC
            // span is the four type bytes that came with its header -- no second trip to
            // memory, no pass of the span code.
            uint32_t r = 0xFFFFFFFFu;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                r = ki_step_byte(r, (type2 >> (24 - 8 * k)) & 0xffu, ki_t8(lane));
            r1 = span1 ? r : 0u;
            send_done = true;
        }
        for (int slot = MODE == 2 && !send_done ? 1 : 0; slot >= 0; --slot) {
            uint32_t r = 0, cont = 0;
            bool f = false;
            const uint8_t* sp = MODE == 0 ? ptr : (slot ? send + 4 : ptr + kChunkAt + 4);
            const uint64_t sl = MODE == 0 ? len0 : 4ull + (slot ? d2 : dlen);
            const bool may_file = slot == 0 && (b.cut.counts != nullptr || b.cut.long_flag != nullptr);
            if (MODE == 2 && slot == 0) {
                // Everything the reference does AFTER the AMRG chunk's CRC check (document.rs:
                // 65-74) is decided HERE, before that chunk's passes: the SEND chunk's register
                // is known, and one word (`after0`) crosses the passes instead of the SEND
                // chunk's checks, length, type, stored CRC and register.  A long AMRG chunk
                // takes the same word to the piece kernel.
                if (span0) {
                    const uint32_t t0 = pre ? type0 : be32(ptr + 13);
                    cont = t0 != 0x414D5247u ? 1u : st2;  // expect_type(AMRG), then read_chunk on the SEND chunk
                    if (cont == 0) {
                        const uint32_t s2 = pre2 && d2 == 0u ? stored2 : be32(send + 8ull + d2);
                        const uint32_t t2 = pre2 ? type2 : be32(send + 4);
                        cont = (r1 ^ 0xFFFFFFFFu) != s2 ? 5u : t2 != 0x53454E44u ? 1u : 0u;  // its crc, expect_type(SEND)
                    }
                }
                after0 = cont;
            }
            crc_spans<MODE>(b, slot ? span1 : span0, reinterpret_cast<uint64_t>(sp), sl, i, cont, dlen, may_file,
                            r, f, L, lane);
            if (slot) {
                r1 = r;
            } else {
                r0 = r;
                filed = f;
            }
        }

        // ---- C: the verdict ----
        if (!have || filed) continue;
        if (MODE == 0) {
            b.out[i] = r0 ^ b.xor_out;
            continue;
        }
        if (status == 0) {
            const uint8_t* chunk = ptr + kChunkAt;
            const uint32_t crc = r0 ^ 0xFFFFFFFFu;  // container.rs:83
            if (MODE == 3) {
                uint8_t* slot = const_cast<uint8_t*>(chunk) + 8ull + dlen;
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) slot[k] = (uint8_t)(crc >> (24u - 8u * k));
            } else if (MODE == 1) {
                if (crc != (pre ? stored0 : be32(chunk + 8ull + dlen))) status = 5;  // container.rs:60-63
            } else {
                const uint32_t s0 = pre ? stored0 : be32(chunk + 8ull + dlen);
                status = crc != s0 ? 5u : after0;  // container.rs:60-63, then what was decided before the passes
            }
        }
        b.out[i] = status;
        if (b.data_len) b.data_len[i] = status == 0 ? dlen : 0u;
    }
}

// ---------------------------------------------------------------------------
// Piece kernel: the pieces of the long items (planned by the kernel itself, or filed by
// the batch kernel: see the PLAN forms below), a workgroup per run of pieces, with
// the tile kernel's algorithm (16 KiB tiles, stride-one-tile Horner, y table in
// rows by byte value).  Piece j of a record covers
//     [Eb - (np-j)*P, Eb - (np-1-j)*P),   Eb = end of the span rounded down to 16,
// piece 0 starting at the span's first byte instead; its residue is weighted by
// x^(8*P*(np-1-j)), so that the XOR of a record's residues is the register at Eb.
// The kernel is persistent over the piece list: tables are staged once per
// workgroup, and the workgroups stride through the list in windows of g pieces.
// Pieces are small (64 KiB) so that one long item alone still spreads over the
// chip; when the list is long the kernel, which knows its length, regroups:
// neighbouring pieces of one record are adjacent in memory, so a run of them
// inside a window is streamed as ONE longer piece.
//
// A run's weighted residue is XORed into its record's accumulator and the run's
// piece count added to the record's arrival count, the same two-atomic exchange
// as the tile kernel's; the run whose add brings the count to np has seen every
// other run's XOR performed, swaps the accumulator out, walks the < 16 tail bytes
// and gives the item its verdict (what the batch kernel does after a wave_crc).
namespace {

// One lane.  `r` is the register over the whole span: what the batch kernel does
// after a wave_crc, per mode.
__device__ __forceinline__ void verdict_record(const BatchArgs& b, const CutRecord* rp, uint32_t r,
                                               bool have_stored = false, uint32_t stored = 0) {
    const uint8_t* end = b.base + rp->span_off + rp->span_len;
    const uint32_t item = rp->item;
    if (!b.verify) {
        b.out[item] = r ^ b.xor_out;
        return;
    }
    const uint32_t crc = r ^ 0xFFFFFFFFu;  // container.rs:83
    uint32_t status;
    if (b.verify == 3) {
        uint8_t* slot = const_cast<uint8_t*>(end);  // Chunk::write: the crc goes behind the data, :75
        for (uint32_t k = 0; k < 4u; ++k) slot[k] = (uint8_t)(crc >> (24u - 8u * k));
        status = 0;
    } else {
        status = crc == (have_stored ? stored : be32(end)) ? rp->cont : 5u;  // container.rs:60-63, then what follows
    }
    b.out[item] = status;
    if (b.data_len) b.data_len[item] = status == 0 ? rp->dlen : 0u;
}

// One lane.  `r` is the register at the end of the span's body: the < 16 tail
// bytes, then the verdict.
__device__ __forceinline__ void finish_record(const BatchArgs& b, const CutRecord* rp, uint32_t r) {
    const uint8_t* end = b.base + rp->span_off + rp->span_len;
    const uint32_t tail_len = (uint32_t)(reinterpret_cast<uint64_t>(end) & 15u);
    for (uint32_t i = 0; i < tail_len; ++i) r = crc_step_byte(r, end[(int)i - (int)tail_len]);
    verdict_record(b, rp, r);
}
// The same with the 16-byte block the tail bytes lie in and the stored CRC already in
// registers (the self-planned form asks for them with the header: fetched here they
// were up to 19 dependent byte loads behind the record's last atomic).
__device__ __forceinline__ void finish_record(const BatchArgs& b, const CutRecord* rp, uint32_t r, u32x4 tailv,
                                              uint32_t stored) {
    const uint8_t* end = b.base + rp->span_off + rp->span_len;
    const uint32_t tail_len = (uint32_t)(reinterpret_cast<uint64_t>(end) & 15u);
    for (uint32_t i = 0; i < tail_len; ++i) {
        r = crc_step_byte(r, tailv[0] & 0xffu);
#pragma unroll
        for (int c = 0; c < 3; ++c) tailv[c] = (tailv[c] >> 8) | (tailv[c + 1] << 24);
        tailv[3] >>= 8;
    }
    verdict_record(b, rp, r, true, stored);
}

// Few items (n <= kPlanMax, not the document mode): no filing pass.  Every
// workgroup of the piece kernel works the plan out for itself, identically: lane
// i of wave 0 reads item i's descriptor (and chunk header), runs the checks that
// come before the CRC, and turns its span into np pieces; a wave scan gives the
// piece bases.  The plan lands in LDS as CutRecords.  Items that end without a
// span to stream (an error before the CRC, or a span under 32 bytes, walked by
// the lane) get their verdict from workgroup 0 alone.  Returns the piece total.
__device__ __forceinline__ uint32_t plan_items(const BatchArgs& b, const uint64_t* inl, CutRecord* prec,
                                               uint32_t* pbase_lds, uint32_t* total_lds, uint64_t d_off,
                                               uint64_t d_avail, u32x4& pre_tail, uint32_t& pre_stored) {
    const uint32_t tid = threadIdx.x;
    pre_tail = u32x4{0u, 0u, 0u, 0u};
    pre_stored = 0u;
    if (tid < 64u) {
        const uint32_t i = tid;
        uint32_t np = 0, level = 0, dlen = 0;
        uint64_t span_off = 0, span_len = 0;
        if (i < b.n) {
            const uint64_t off = d_off, avail = d_avail;  // asked for at the top of the kernel
            const uint8_t* ptr = b.base + off;
            uint32_t status = 0;
            bool pending = true;
            if (!b.verify) {
                span_off = off;
                span_len = avail;
            } else {
                if (b.verify == 1) {
                    status = read_chunk_head(ptr, avail, b.chunk_limit, &dlen);  // container.rs:48-60
                } else {  // 3: Chunk::write's crc step, avail = bytes of the slot
                    if (avail < 12u) {
                        status = 2;
                    } else {
                        dlen = be32(ptr);
                        if (avail < 12ull + (uint64_t)dlen) status = 2;
                    }
                }
                span_off = off + 4u;
                span_len = 4ull + dlen;
                if (status) {
                    pending = false;
                    if (blockIdx.x == 0) {
                        b.out[i] = status;
                        if (b.data_len) b.data_len[i] = 0u;
                    }
                }
            }
            CutRecord r;
            r.span_off = span_off;
            r.span_len = span_len;
            r.acc = r.arrived = 0ull;
            r.item = i;
            r.cont = 0u;
            r.dlen = dlen;
            r.pad[0] = r.pad[1] = 0u;
            if (pending && span_len < 32u) {
                // too short to carry the state inside whole vectors: one lane walks it
                if (blockIdx.x == 0) {
                    uint32_t reg = b.verify ? 0xFFFFFFFFu : b.state;
                    const uint8_t* sp = b.base + span_off;
                    for (uint32_t k = 0; k < (uint32_t)span_len; ++k) reg = crc_step_byte(reg, sp[k]);
                    r.piece_base = r.np = r.level = 0u;
                    verdict_record(b, &r, reg);
                }
                pending = false;
            }
            if (pending) {
                const uint64_t A = reinterpret_cast<uint64_t>(b.base) + span_off;
                // what the record's finisher will need, asked for now (lane i of wave 0 keeps
                // item i's): the block its trailing bytes lie in and the stored CRC behind it
                const uint64_t E = A + span_len;
                pre_tail = *reinterpret_cast<const u32x4*>((E & 15u) ? (E & ~15ull) : (A & ~15ull));
                if (b.verify == 1) pre_stored = be32(reinterpret_cast<const uint8_t*>(E));
                const uint64_t body = ((A + span_len) & ~15ull) - (A & ~15ull);
                level = span_level(body, b.cut.piece_min_log2);
                const uint32_t plog = b.cut.piece_min_log2 + level;
                np = (uint32_t)((body + (1ull << plog) - 1) >> plog);
            }
            r.np = np;
            r.level = level;
            // exclusive scan of np over the lanes below n (a lane only reads lower lanes,
            // all of which are in this branch)
            uint32_t incl = np;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = __shfl_up(incl, d, 64);
                if ((int)tid >= d) incl += up;
            }
            r.piece_base = incl - np;
            prec[i] = r;
            pbase_lds[i] = r.piece_base;
            if (i == b.n - 1u) *total_lds = incl;
        } else {
            pbase_lds[i] = 0xFFFFFFFFu;
        }
    }
    __syncthreads();
    return *total_lds;
}

// The wide plan (kPlanMax < n <= kWideMax, not the document mode).  The batch kernel has
// taken every item whose span fits one wave and left the longer ones alone; every workgroup
// of the piece kernel now finds those for itself, identically: lane t takes items 4t .. 4t+3
// (descriptors asked for at the top of the kernel), repeats the checks that come before the
// CRC, and counts the pieces of the spans over wave_max -- the batch kernel's own test
// (crc_spans).  wide_end[i] = pieces of items 0 .. i (a scan over the workgroup).  Nothing
// is filed and no word of device memory is written: a run finds its item by two 64-way
// looks into wide_end, and the item's span in
// spans[2i], spans[2i+1] (offset, length) -- a run of 64 KiB lasts 2.4 us at the workgroup's
// share of the memory system, and a descriptor read again from memory was a third trip
// in front of it.  Returns the piece total.
__device__ __forceinline__ uint32_t plan_wide(BatchArgs& b, const uint64_t (&d_off)[4],
                                              const uint64_t (&d_avail)[4], const uint64_t (&sample4)[4],
                                              uint32_t* wide_end, uint64_t* spans, uint32_t* parts) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if (b.cut.wave_adapt) b.cut.wave_max = wave_threshold(b, sample4);  // the threshold the batch kernel worked out
    uint32_t np[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = 4u * tid + (uint32_t)k;
        uint32_t dlen = 0;
        np[k] = 0;
        uint64_t span_off_k = 0, span_len_k = 0;
        if (i < b.n) {
            const uint64_t off = d_off[k], avail = d_avail[k];
            const uint8_t* ptr = b.base + off;
            uint32_t status = 0;
            uint64_t span_off = off, span_len = avail;
            if (b.verify) {
                if (b.verify == 1) {
                    status = read_chunk_head(ptr, avail, b.chunk_limit, &dlen);  // container.rs:48-60
                } else if (avail < 12u) {  // 3: Chunk::write's crc step, avail = bytes of the slot
                    status = 2;
                } else {
                    dlen = be32(ptr);
                    if (avail < 12ull + (uint64_t)dlen) status = 2;
                }
                span_off = off + 4u;
                span_len = 4ull + dlen;
            }
            if (status == 0 && span_len > (uint64_t)b.cut.wave_max) {
                const uint64_t A = reinterpret_cast<uint64_t>(b.base) + span_off;
                const uint64_t body = ((A + span_len) & ~15ull) - (A & ~15ull);
                const uint32_t plog = b.cut.piece_min_log2 + span_level(body, b.cut.piece_min_log2);
                np[k] = (uint32_t)((body + (1ull << plog) - 1) >> plog);
            }
            span_off_k = span_off;
            span_len_k = span_len;
        }
        spans[2u * i] = span_off_k;
        spans[2u * i + 1u] = span_len_k;
    }
    const uint32_t mine = np[0] + np[1] + np[2] + np[3];
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += up;
    }
    if (lane == 63u) parts[tid >> 6] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        const uint32_t pw = parts[w];
        before += (uint32_t)w < (tid >> 6) ? pw : 0u;
        total += pw;
    }
    uint32_t run = before + incl - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        run += np[k];
        wide_end[4u * tid + (uint32_t)k] = run;
    }
    __syncthreads();
    return total;
}

// Pieces per window: two windows a workgroup once the list is long enough, at most 64 pieces.
// Rounded UP: rounded down, 4000 pieces made 572 windows of 7 -- 60 workgroups with a third
// window while 196 had left, 21 runs against 15.6 on average.
__device__ __forceinline__ uint32_t window_pieces(uint32_t total, uint32_t groups) {
    const uint32_t g = (total + groups * 2u - 1u) / (groups * 2u);
    return g < 1u ? 1u : (g > 64u ? 64u : g);
}

// A workgroup of the piece kernel leaves: the last one out puts the call's
// counters back to zero for the next call on this scratch block.
__device__ __forceinline__ void leave_pieces(const CutArgs& c, uint32_t groups) {
    const uint32_t t = __hip_atomic_fetch_add(c.left, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == groups - 1u) {
        __hip_atomic_exchange(c.counts, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_exchange(c.pieces, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_exchange(c.left, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace

// PLAN = 0: the piece list was filed by crc32_batch_kernel (c.recs, c.piece_rec).
// PLAN = 1: few items; every workgroup plans them itself (plan_items), the
//           records are in LDS and only their accumulators (c.recs[i].acc /
//           .arrived, zero between calls) in device memory.
//           INL: their descriptors are in the second argument instead of b.offsets / b.lens.
// PLAN = 2: the wide plan (plan_wide): only the spans the batch kernel left alone; piece
//           ends and data lengths in LDS, accumulators in c.recs taken as pairs of words.
struct NoInline {
    uint32_t unused;
};
template <int PLAN, bool INL>
__global__ __launch_bounds__(kThreads) void crc32_pieces_kernel(BatchArgs b,
                                                                std::conditional_t<INL, InlineDesc, NoInline> d) {
    static_assert(PLAN == 1 || !INL, "inline descriptors: the narrow plan only");
    constexpr int kZWords = 32 * 32;
    constexpr int kPlanWords = PLAN == 1   ? (int)(kPlanMax * sizeof(CutRecord) / 4 + kPlanMax + 16)
                               : PLAN == 2 ? (int)(5 * kWideMax + kWaves)
                                           : 0;
    __shared__ __align__(16) uint32_t lds[kYWords + kZWords + 2 * kWaves + kPlanWords];  // rows: y | x^32, as the tile kernel's
    uint32_t* ldsY = lds;
    uint32_t* ldsZ = lds + kYWords;
    uint32_t* red = ldsZ + kZWords;
    CutRecord* prec = reinterpret_cast<CutRecord*>(red + 2 * kWaves);  // PLAN 1
    uint32_t* plan_pbase = reinterpret_cast<uint32_t*>(prec + kPlanMax);
    uint64_t* wide_span = reinterpret_cast<uint64_t*>(red + 2 * kWaves);  // PLAN 2: (offset, length) of item i's span
    uint32_t* wide_end = reinterpret_cast<uint32_t*>(wide_span + 2 * kWideMax);
    uint32_t* wide_parts = wide_end + kWideMax;
    static_assert((kYWords + kZWords + 2 * kWaves) % 2 == 0, "wide_span: 8-byte words");

    const CutArgs& c = b.cut;
    unsigned long long* wide_acc = reinterpret_cast<unsigned long long*>(c.recs);  // PLAN 2: (acc, arrived) of item i at 2i
    const uint32_t tid = threadIdx.x;
#if defined(SIFT_TRACE_PIECES)  // timing-only build (tools/trace_pieces.hip): 100 MHz clock at six points per workgroup, into out[]
#define SIFT_PTRACE(slot) \
    if (tid == 0) b.out[64 + (slot) * 256 + blockIdx.x] = (uint32_t)wall_clock64()
#else
#define SIFT_PTRACE(slot)
#endif
    SIFT_PTRACE(0);
    // What the plan hangs on is asked for first: the items' descriptors (lane i of wave 0:
    // item i; wide plan: four a lane), or the length of the filed piece list.  Then the table
    // words this lane will put into LDS.  Tables are staged and the lane's matrix rows
    // computed while those first answers -- and, in the self-planned forms, the chunk headers
    // behind them -- are on their way (round 2 staged after the plan: two trips to memory
    // with idle lanes, then 0.8 us of vector work with an idle memory system).
    uint64_t d_off = 0, d_avail = 0;
    uint64_t w_off[4] = {0, 0, 0, 0}, w_avail[4] = {0, 0, 0, 0}, w_sample[4] = {0, 0, 0, 0};
    unsigned long long reserved = 0;
    if (PLAN == 1) {
        const uint64_t* inl = nullptr;
        if constexpr (INL) inl = d.v;
        if (tid < 64u && tid < b.n) {
            d_off = inl ? inl[2u * tid] : b.offsets[tid];
            d_avail = inl ? inl[2u * tid + 1u] : b.lens[tid];
        }
    } else if (PLAN == 2) {
        reserved = *c.long_flag;  // raised by the batch kernel if it left any span alone
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = 4u * tid + (uint32_t)k;
            w_off[k] = b.offsets[i < b.n ? i : 0u];
            w_avail[k] = b.lens[i < b.n ? i : 0u];
        }
        if (b.cut.wave_adapt) wave_sample(b, w_sample);
    } else {
        reserved = *c.pieces;
    }
    static_assert(kZWords == 1024, "table of 1024 words");
    constexpr int kRounds = (1024 + kThreads - 1) / kThreads;  // 1 at 1024 lanes
    constexpr int kYRounds = (kStagePairs + kThreads - 1) / kThreads;  // 2 at 1024 lanes
    u32x4 ty[kYRounds], tx[kYRounds];
    uint32_t tz[kRounds];
#pragma unroll
    for (int k = 0; k < kYRounds; ++k) {
        const uint32_t q = tid + (uint32_t)(k * kThreads);
        ty[k] = stage_fetch(b.tabs->Yt, q < (uint32_t)kStagePairs ? q : 0u);
        tx[k] = stage_fetch(b.tabs->Xt, q < (uint32_t)kStagePairs ? q : 0u);
    }
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {
        const uint32_t q = tid + (uint32_t)(k * kThreads) < 1024u ? tid + (uint32_t)(k * kThreads) : 0u;
        tz[k] = (&b.tabs->X2[0][0])[q];
    }
    uint32_t kb[32];
    kb[0] = b.tabs->Kt[tid];
    __builtin_amdgcn_sched_barrier(0);
    uint32_t total = 0, g = 1;
    if (PLAN == 0) {
        // the filed list: most batches file nothing, and a workgroup without a window leaves
        // before it stages anything
        // nothing reserved: no record was filed either (a record has at least one piece), the
        // call's counters are all zero as they stand, and there is nothing to count out of
        // (256 workgroups counting themselves out on one word was half of this launch's 4 us)
        if (reserved == 0ull) return;
        total = reserved < c.cap_pieces ? (uint32_t)reserved : c.cap_pieces;
        g = window_pieces(total, gridDim.x);
        if ((uint64_t)blockIdx.x * g >= total) {
            if (tid == 0) leave_pieces(c, gridDim.x);
            return;
        }
    } else if (PLAN == 2) {
        if (reserved == 0ull) return;  // every span fitted its wave
    }

    const KiLane L = ki_lane(tid);
    if (static_cast<uint32_t>(reinterpret_cast<uintptr_t>((LdsBytes)lds)) != 0u) __builtin_trap();  // ki_read
    const uint32_t row32 = tid & 31u;
#pragma unroll
    for (int k = 0; k < kYRounds; ++k) {
        const uint32_t q = tid + (uint32_t)(k * kThreads);
        if (q < (uint32_t)kStagePairs) {
            stage_store(ldsY, ty[k], q);
            stage_store(ldsY, tx[k], q, 1u);
        }
    }
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {
        const uint32_t q = tid + (uint32_t)(k * kThreads);
        if (q < 1024u) ldsZ[q] = tz[k];
    }
#pragma unroll
    for (int i = 1; i < 32; ++i) kb[i] = gf2_mulx(kb[i - 1]);
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(kb[i]));
    __builtin_amdgcn_sched_barrier(0);

    u32x4 pre_tail = {0u, 0u, 0u, 0u};  // PLAN 1, wave 0, lane i: item i's tail block and stored CRC
    uint32_t pre_stored = 0;            // PLAN 2, wave 0: those of the run's item, if the run is all of it
    if (PLAN == 1) {
        const uint64_t* inl = nullptr;
        if constexpr (INL) inl = d.v;
        // (ends with the workgroup's barrier, which also covers the tables staged above)
        total = plan_items(b, inl, prec, plan_pbase, plan_pbase + kPlanMax, d_off, d_avail, pre_tail, pre_stored);
    } else if (PLAN == 2) {
        total = plan_wide(b, w_off, w_avail, w_sample, wide_end, wide_span, wide_parts);  // (ends with a barrier too)
    } else {
        __syncthreads();
    }
    if (PLAN != 0) {
        g = window_pieces(total, gridDim.x);
        if ((uint64_t)blockIdx.x * g >= total) return;
    }
    SIFT_PTRACE(1);

    // What the commit pipeline below needs of a record, per form.
    auto acc_word = [&](uint32_t slot) -> unsigned long long* {
        return PLAN == 2 ? wide_acc + 2u * slot : &c.recs[slot].acc;
    };
    auto arrived_word = [&](uint32_t slot) -> unsigned long long* {
        return PLAN == 2 ? wide_acc + 2u * slot + 1u : &c.recs[slot].arrived;
    };
    auto pieces_of = [&](uint32_t slot) -> uint32_t {
        if (PLAN == 2) return wide_end[slot] - (slot ? wide_end[slot - 1u] : 0u);
        return PLAN == 1 ? prec[slot].np : c.recs[slot].np;
    };
    // PLAN 2: item `slot` as a record (one lane)
    auto wide_record = [&](uint32_t slot) -> CutRecord {
        CutRecord t;
        t.span_off = wide_span[2u * slot];
        t.span_len = wide_span[2u * slot + 1u];
        t.dlen = b.verify ? (uint32_t)t.span_len - 4u : 0u;
        t.item = slot;
        t.cont = 0u;
        return t;
    };

    // Wave 0 hands residues to their records through a three-stage pipeline, so
    // that it never waits for an atomic it has just issued (the other 15 waves
    // would wait for it at the next run's barrier):
    //   cur      residues of consecutive runs of ONE record, summed in a register
    //   stage 1  record whose XOR into `acc` has been issued
    //   stage 2  record whose `arrived` add has been issued (after stage 1's XOR
    //            returned: 8-byte agent-scope atomics on both sides, the XOR
    //            performed at the memory side before the add issues, as in
    //            crc32_tiles_kernel); when the add returns np - count, this was
    //            the record's last commit and lane 0 finishes the item.
    // The pipeline moves one step each time the workgroup changes record, a whole
    // run after the previous step, by which time the awaited atomic has long returned.
    uint32_t cur_slot = kVoidSlot, cur_x = 0, cur_cnt = 0;
    uint32_t s1_slot = kVoidSlot, s1_cnt = 0, s2_slot = kVoidSlot, s2_cnt = 0;
    unsigned long long s1_old = 0, s2_before = 0;
#define SIFT_COMMIT_ADVANCE()                                                                         \
    do {                                                                                              \
        if (s2_slot != kVoidSlot && tid == 0) {                                                       \
            asm volatile("s_waitcnt vmcnt(0)" ::"v"(s2_before) : "memory");                           \
            if (s2_before + s2_cnt == (unsigned long long)pieces_of(s2_slot)) {                       \
                const unsigned long long v_ = __hip_atomic_exchange(                                  \
                    acc_word(s2_slot), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);             \
                if (PLAN != 0) /* no filing pass re-initialises the record: leave it zero */          \
                    __hip_atomic_exchange(arrived_word(s2_slot), 0ull, __ATOMIC_RELAXED,              \
                                          __HIP_MEMORY_SCOPE_AGENT);                                  \
                if (PLAN == 1) {                                                                      \
                    u32x4 t_;                                                                         \
                    for (int q_ = 0; q_ < 4; ++q_)                                                    \
                        t_[q_] = (uint32_t)__builtin_amdgcn_readlane((int)pre_tail[q_], (int)s2_slot); \
                    finish_record(b, prec + s2_slot, (uint32_t)v_, t_,                                \
                                  (uint32_t)__builtin_amdgcn_readlane((int)pre_stored, (int)s2_slot)); \
                } else if (PLAN == 2) {                                                               \
                    const CutRecord t_ = wide_record(s2_slot);                                        \
                    finish_record(b, &t_, (uint32_t)v_);                                              \
                } else {                                                                              \
                    finish_record(b, c.recs + s2_slot, (uint32_t)v_);                                 \
                }                                                                                     \
            }                                                                                         \
        }                                                                                             \
        s2_slot = s1_slot;                                                                            \
        s2_cnt = s1_cnt;                                                                              \
        if (s1_slot != kVoidSlot && tid == 0) {                                                       \
            asm volatile("s_waitcnt vmcnt(0)" ::"v"(s1_old) : "memory");                              \
            s2_before = __hip_atomic_fetch_add(arrived_word(s1_slot), (unsigned long long)s1_cnt,     \
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);           \
        }                                                                                             \
        s1_slot = cur_slot;                                                                           \
        s1_cnt = cur_cnt;                                                                             \
        if (cur_slot != kVoidSlot && tid == 0)                                                        \
            s1_old = __hip_atomic_fetch_xor(acc_word(cur_slot), (unsigned long long)cur_x,            \
                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);              \
    } while (0)

    const uint32_t start_state = b.verify ? 0xFFFFFFFFu : b.state;
    // The runs are taken two at a time: run k+1 is located and its first tiles are asked for
    // BEFORE run k is reduced (fold, place, XOR over the workgroup, barrier, verdict: 1.3-1.6 us
    // with nothing in flight -- as much as a run of 64 KiB takes to stream -- followed by a
    // trip to memory before the next run's first fold).
    struct Run {
        uint32_t slot, np, plog, run, jl;
        bool whole;
        int64_t ntiles;  // tiles behind the first one and the first batch
        const u32x4* pt;
        int soff;
        bool batch;      // the first batch (cur) was asked for
        bool early;      // ... and the last tiles behind it too (odd): a short run is asked for whole
        u32x4 acc, cur[kUnroll], odd[kUnroll > 1 ? kUnroll - 1 : 1];
        u32x4 tail;      // PLAN 2, whole, wave 0: the block the trailing bytes lie in, the stored CRC
        uint32_t stored;
    };
    constexpr int U = kUnroll;
    const int voff = (int)(tid * 16u);
    uint64_t w = blockIdx.x;  // window index: this workgroup takes w, w + grid, ...
    uint32_t p = (uint32_t)w * g, wend = p + g < total ? p + g : total;
    // the next piece to take and its record's slot; false when the workgroup's windows are done
    auto locate = [&](uint32_t& slot) __attribute__((always_inline)) -> bool {
        for (;;) {
            if (p >= wend) {
                w += gridDim.x;
                if (w * g >= total) return false;
                p = (uint32_t)(w * g);
                wend = p + g < total ? p + g : total;
            }
            if (PLAN == 1) {
                // the last item whose piece base is <= p (items without pieces share the next one's base)
                const uint32_t pb = plan_pbase[tid & 63u];
                slot = 63u - (uint32_t)__builtin_clzll(__ballot(pb <= p));
            } else if (PLAN == 2) {
                // the first item whose pieces end behind p: which block of 64 items, then which of them
                // (items past n and items without pieces end where their predecessor does)
                const uint32_t blk = (uint32_t)__builtin_ctzll(__ballot(wide_end[(tid & 63u) * 64u + 63u] > p));
                slot = blk * 64u + (uint32_t)__builtin_ctzll(__ballot(wide_end[blk * 64u + (tid & 63u)] > p));
            } else {
                slot = c.piece_rec[p];
            }
            if (slot != kVoidSlot) return true;
            ++p;  // reservation that did not fit: its span was walked in place
        }
    };
    // the run at p: its extent, and its first tile and first batch asked for
    auto begin = [&](uint32_t slot) __attribute__((always_inline)) -> Run {
        Run R;
        R.slot = slot;
        uint64_t span_off, span_len;
        uint32_t piece_base;
        if (PLAN == 2) {
            span_off = wide_span[2u * slot];
            span_len = wide_span[2u * slot + 1u];
            piece_base = slot ? wide_end[slot - 1u] : 0u;
            R.np = wide_end[slot] - piece_base;
            const uint64_t A0 = reinterpret_cast<uint64_t>(b.base) + span_off;
            R.plog = c.piece_min_log2 + span_level(((A0 + span_len) & ~15ull) - (A0 & ~15ull), c.piece_min_log2);
        } else {
            const CutRecord* rp = PLAN == 1 ? prec + slot : c.recs + slot;
            span_off = rp->span_off;
            span_len = rp->span_len;
            R.np = rp->np;
            piece_base = rp->piece_base;
            R.plog = c.piece_min_log2 + rp->level;
        }
        const uint32_t j = p - piece_base;
        // the run: pieces j .. j+run-1 of this record, inside the window, under 1 GiB
        R.run = wend - p;
        if (R.run > R.np - j) R.run = R.np - j;
        if (R.run > (1u << (30u - R.plog))) R.run = 1u << (30u - R.plog);
        R.jl = j + R.run - 1u;  // last piece of the run: the run is weighted as jl is
        const uint64_t A = reinterpret_cast<uint64_t>(b.base) + span_off;
        const uint64_t Eb = (A + span_len) & ~15ull;
        const uint64_t end_j = Eb - ((uint64_t)(R.np - 1u - R.jl) << R.plog);
        const uint64_t start_j = j == 0 ? (A & ~15ull) : end_j - ((uint64_t)R.run << R.plog);
        const int64_t nvec = (int64_t)((end_j - start_j) >> 4);
        int64_t ntiles = (nvec + kTileVecs - 1) / kTileVecs;
        const u32x4* vbase = reinterpret_cast<const u32x4*>(start_j);
        // PLAN 2: the run is the whole span -- no other workgroup has a part of it, so its
        // register needs no accumulator; wave 0 asks now for what the verdict will need
        // (the block the trailing bytes lie in, the stored CRC) and gives it after the run.
        R.whole = PLAN == 2 && R.run == R.np;
        R.tail = u32x4{0u, 0u, 0u, 0u};
        R.stored = 0u;
        if (R.whole && tid < 64u) {
            const uint64_t E = A + span_len;
            R.tail = *reinterpret_cast<const u32x4*>((E & 15u) ? (E & ~15ull) : (A & ~15ull));
            if (b.verify == 1) R.stored = be32(reinterpret_cast<const uint8_t*>(E));
        }
        // First tile: the only one that can be partial (tiles are aligned to the
        // END of the piece); in piece 0 vector 0 carries the head mask and the
        // starting register.
        R.acc = u32x4{0u, 0u, 0u, 0u};
        const int64_t v0 = nvec - ntiles * (int64_t)kTileVecs;
        const int64_t vi = v0 + (int64_t)tid;
        if (vi >= 0) {
            R.acc = load_vec(vbase + vi);
            if (j == 0 && vi == 0) {
                const uint32_t h = (uint32_t)(A & 15u);
                if (h) R.acc = mask_head(R.acc, h);
                R.acc[0] ^= h ? gf2_mulmod(start_state, b.tabs->XinvH[h]) : start_state;
            }
        }
        --ntiles;
        R.pt = vbase + v0 + kTileVecs;
        R.soff = 0;
        R.batch = ntiles >= U;
        R.early = false;
#pragma unroll
        for (int u = 0; u < U; ++u) R.cur[u] = u32x4{0u, 0u, 0u, 0u};
        if (R.batch) {
            const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<u32x4*>(R.pt), /*stride*/ 0, (int)((uint32_t)ntiles * (uint32_t)kTileBytes), 0x00020000);
#pragma unroll
            for (int u = 0; u < U; ++u) R.cur[u] = load_tile(srd, voff, u * kTileBytes);
            R.soff = U * kTileBytes;
            ntiles -= U;
            // a run that ends here (up to 2U tiles, 64 KiB) is asked for whole: its last tiles
            // asked for when the run is streamed came a whole trip behind the reduction in between
            R.early = ntiles < U;
#pragma unroll
            for (int u = 0; u < U - 1; ++u) {
                R.odd[u] = u32x4{0u, 0u, 0u, 0u};
                if (R.early && ntiles > u) R.odd[u] = load_tile(srd, voff, R.soff + u * kTileBytes);
            }
        }
        R.ntiles = ntiles;
        return R;
    };
    // the rest of the run's tiles: its register over the run's vectors, per lane
    auto stream = [&](Run& R) __attribute__((always_inline)) {
        u32x4 acc = R.acc;
        int64_t ntiles = R.ntiles;
        int soff = R.soff;
        const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<u32x4*>(R.pt), /*stride*/ 0, (int)((uint32_t)soff + (uint32_t)ntiles * (uint32_t)kTileBytes),
            0x00020000);
        if (R.batch) {
            u32x4 cur[U];
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = R.cur[u];
            // as in the tile kernel: two register sets in turn, one batch in flight
            u32x4 nxt[U];
            while (ntiles >= 2 * U) {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
                for (int u = 0; u < U; ++u) nxt[u] = load_tile(srd, voff, soff + u * kTileBytes);
#pragma unroll
                for (int u = 0; u < U; ++u) horner_step(acc, cur[u], L);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
                for (int u = 0; u < U; ++u) cur[u] = load_tile(srd, voff, soff + (U + u) * kTileBytes);
                __builtin_amdgcn_sched_barrier(0);
                soff += 2 * U * kTileBytes;
                ntiles -= 2 * U;
#pragma unroll
                for (int u = 0; u < U; ++u) horner_step(acc, nxt[u], L);
            }
            if (ntiles >= U) {
#pragma unroll
                for (int u = 0; u < U; ++u) nxt[u] = load_tile(srd, voff, soff + u * kTileBytes);
                soff += U * kTileBytes;
                ntiles -= U;
#pragma unroll
                for (int u = 0; u < U; ++u) horner_step(acc, cur[u], L);
#pragma unroll
                for (int u = 0; u < U; ++u) cur[u] = nxt[u];
            }
            // the tiles behind the last whole batch are asked for before that batch is folded
            // (asked for behind it, a run of 64 KiB -- tiles 1 | 2 3 | 4 -- made a third trip)
            u32x4 odd[U > 1 ? U - 1 : 1];
#pragma unroll
            for (int u = 0; u < U - 1; ++u) {
                odd[u] = R.odd[u];
                if (!R.early && ntiles > u) odd[u] = load_tile(srd, voff, soff + u * kTileBytes);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) horner_step(acc, cur[u], L);
#pragma unroll
            for (int u = 0; u < U - 1; ++u)
                if (ntiles > u) horner_step(acc, odd[u], L);
            ntiles = 0;
        }
        while (ntiles > 0) {
            const u32x4 w = load_tile(srd, voff, soff);
            horner_step(acc, w, L);
            soff += kTileBytes;
            --ntiles;
        }
        R.acc = acc;
    };
    uint32_t parity = 0;
    // the run's residue: lane residue -> place in the tile -> XOR over the workgroup -> its record
    auto reduce = [&](const Run& R) __attribute__((always_inline)) {
#if SIFT_ABL == 1  // timing-only: no reduction, no verdict
        if (R.acc[0] == 0x12345u && R.acc[1] == 7u) b.out[0] = 1u;
        return;
#endif
        uint32_t r = 0;
#pragma unroll
        for (int cw = 0; cw < 4; ++cw) r = ki_mul8<kTabX>(r ^ R.acc[cw], L);
        {
            uint32_t prod = 0;
#pragma unroll
            for (int i = 0; i < 32; ++i) prod = xor_and(prod, kb[i], (uint32_t)((int32_t)(r << i) >> 31));
            r = prod;
        }
        r = wave_xor(r);
        uint32_t* redp = red + parity * kWaves;  // two sets: wave 0 may still be reading the other
        parity ^= 1u;
        if ((tid & 63u) == 0) redp[tid >> 6] = r;
        __syncthreads();
        if (tid < 64u) {
            uint32_t x = wave_xor(tid < (uint32_t)kWaves ? redp[tid] : 0u);
            // weight x^(8 * 2^plog * k), k = pieces behind this one: the product of
            // x^(2^(3+plog+bit)) over the set bits of k (x^(2^s) has period 32 in s)
            uint32_t k = R.np - 1u - R.jl;
            for (uint32_t bit = 0; k; ++bit, k >>= 1)
                if (k & 1u) x = mul_rows(x, ldsZ[((3u + R.plog + bit) & 31u) * 32u + row32], row32);
            if (R.whole) {
                if (tid == 0) {
                    const CutRecord t = wide_record(R.slot);
                    finish_record(b, &t, x, R.tail, R.stored);
                }
            } else if (R.slot == cur_slot) {  // into the record, through wave 0's commit pipeline
                cur_x ^= x;
                cur_cnt += R.run;
            } else {
                SIFT_COMMIT_ADVANCE();
                cur_slot = R.slot;
                cur_x = x;
                cur_cnt = R.run;
            }
        }
    };
    {
        Run R, N;
        bool have = false;
        uint32_t slot = kVoidSlot;
        bool more = locate(slot);
        while (more || have) {
            if (more) N = begin(slot);
            if (have) {
                SIFT_PTRACE(2);
                reduce(R);
            }
            have = more;
            if (more) {
                stream(N);
                R = N;
                p += R.run;
                more = locate(slot);
            }
        }
    }
    SIFT_PTRACE(3);
    if (PLAN == 1 && b.n == 1u) {
        // ONE item: every workgroup holds its whole share of the item's register in cur_x, and
        // the workgroups that took part are 0 .. P - 1.  They meet as the tile kernel's groups
        // do: shards of 32 on a 64-bit word each (arrival bits | XOR of the residues), one
        // returning atomic per level, the last arriver of the last level finishes the item.
        // (Through the record's accumulator and arrival count -- XOR, then add, then two
        // swaps, each waited for, all 256 workgroups on the same two words -- a single
        // 16 MiB chunk spent 6.5 of its 15 us here.)  The words are the first 8 bytes of
        // every other record slot of the stream's block: zero between calls, put back to
        // zero by each level's last arriver.
        if (tid == 0) {
            unsigned long long* words = reinterpret_cast<unsigned long long*>(c.recs);  // 128 bytes apart
            const uint32_t P = (total + g - 1u) / g < gridDim.x ? (total + g - 1u) / g : gridDim.x;
            const uint32_t nshards = (P + 31u) >> 5;
            const uint32_t shard = blockIdx.x >> 5, bit = blockIdx.x & 31u;
            const uint32_t members = P - (shard << 5) < 32u ? P - (shard << 5) : 32u;
            const uint32_t full1 = members == 32u ? 0xFFFFFFFFu : (1u << members) - 1u;
            const uint32_t x = cur_slot == kVoidSlot ? 0u : cur_x;
            unsigned long long* w1 = words + 16u * (1u + shard);
            unsigned long long old = __hip_atomic_fetch_xor(w1, ((unsigned long long)(1u << bit) << 32) | x,
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (((uint32_t)(old >> 32) | (1u << bit)) == full1) {
                const uint32_t xs = (uint32_t)old ^ x;
                __hip_atomic_fetch_xor(w1, ((unsigned long long)full1 << 32) | xs, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                uint32_t sum = xs;
                bool last = true;
                if (nshards > 1u) {
                    const uint32_t full2 = nshards == 32u ? 0xFFFFFFFFu : (1u << nshards) - 1u;
                    old = __hip_atomic_fetch_xor(words, ((unsigned long long)(1u << shard) << 32) | xs,
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    last = ((uint32_t)(old >> 32) | (1u << shard)) == full2;
                    if (last) {
                        sum = (uint32_t)old ^ xs;
                        __hip_atomic_fetch_xor(words, ((unsigned long long)full2 << 32) | sum, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (last) finish_record(b, prec, sum, pre_tail, pre_stored);  // lane 0 holds item 0's
            }
        }
        SIFT_PTRACE(5);
        return;
    }
    if (tid < 64u) {  // drain the pipeline: three stages
        SIFT_COMMIT_ADVANCE();
        cur_slot = kVoidSlot;
        SIFT_COMMIT_ADVANCE();
        SIFT_PTRACE(4);
        SIFT_COMMIT_ADVANCE();
    }
    SIFT_PTRACE(5);
#undef SIFT_COMMIT_ADVANCE
    if (PLAN == 0 && tid == 0) leave_pieces(c, gridDim.x);
}

// Zeroes `n` words of a workspace.
__global__ __launch_bounds__(256) void crc32_clear_kernel(uint32_t* words, int n) {
    for (int i = threadIdx.x; i < n; i += 256) words[i] = 0u;
}

void launch_clear(uint32_t* words, int n, void* stream) {
    hipLaunchKernelGGL(crc32_clear_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), words, n);
}

// With an event pair the launch goes through hipExtLaunchKernelGGL: the events take the
// dispatch's own start and end timestamps (measurement only, sift_profile_begin).
#define SIFT_LAUNCH(kernel, grid, s, e0, e1, ...)                                                      \
    do {                                                                                               \
        if (e0 && e1)                                                                                  \
            hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), 0, s, static_cast<hipEvent_t>(e0), \
                                  static_cast<hipEvent_t>(e1), 0, __VA_ARGS__);                        \
        else                                                                                           \
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), 0, s, __VA_ARGS__);                 \
    } while (0)

void launch_batch(const BatchArgs& b, int groups, void* stream, void* ev_start, void* ev_stop) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (b.verify) {
        case 0: SIFT_LAUNCH(crc32_batch_kernel<0>, groups, s, ev_start, ev_stop, b); break;
        case 1: SIFT_LAUNCH(crc32_batch_kernel<1>, groups, s, ev_start, ev_stop, b); break;
        case 2: SIFT_LAUNCH(crc32_batch_kernel<2>, groups, s, ev_start, ev_stop, b); break;
        default: SIFT_LAUNCH(crc32_batch_kernel<3>, groups, s, ev_start, ev_stop, b); break;
    }
}

void launch_pieces(const BatchArgs& b, int groups, int plan, const InlineDesc* inl, void* stream, void* ev_start,
                   void* ev_stop) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (plan == 1 && inl)
        SIFT_LAUNCH((crc32_pieces_kernel<1, true>), groups, s, ev_start, ev_stop, b, *inl);
    else if (plan == 1)
        SIFT_LAUNCH((crc32_pieces_kernel<1, false>), groups, s, ev_start, ev_stop, b, NoInline{0u});
    else if (plan == 2)
        SIFT_LAUNCH((crc32_pieces_kernel<2, false>), groups, s, ev_start, ev_stop, b, NoInline{0u});
    else
        SIFT_LAUNCH((crc32_pieces_kernel<0, false>), groups, s, ev_start, ev_stop, b, NoInline{0u});
}
#undef SIFT_LAUNCH

void launch_tiles(const TileArgs& t, const FinalArgs& f, const CombineMats& m, int groups,
                  bool use_y, void* stream, void* ev_start, void* ev_stop) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (ev_start && ev_stop) {
        // measurement: the two events take the start and end timestamps of this very
        // dispatch (no marker packets before and behind it on the stream)
        hipEvent_t e0 = static_cast<hipEvent_t>(ev_start), e1 = static_cast<hipEvent_t>(ev_stop);
        if (use_y)
            hipExtLaunchKernelGGL(crc32_tiles_kernel<true>, dim3(groups), dim3(kThreads), 0, s, e0, e1, 0, t, f, m);
        else
            hipExtLaunchKernelGGL(crc32_tiles_kernel<false>, dim3(groups), dim3(kThreads), 0, s, e0, e1, 0, t, f, m);
        return;
    }
    if (use_y)
        hipLaunchKernelGGL(crc32_tiles_kernel<true>, dim3(groups), dim3(kThreads), 0, s, t, f, m);
    else
        hipLaunchKernelGGL(crc32_tiles_kernel<false>, dim3(groups), dim3(kThreads), 0, s, t, f, m);
}

void launch_combine(const uint32_t* partial, const FinalArgs& f, const CombineMats& m,
                    void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    // Tiny inputs (no residues) need one lane only; otherwise all 32 half-waves.
    const int block = f.groups > 0 ? kMaxGroups : 64;
    hipLaunchKernelGGL(crc32_combine_kernel, dim3(1), dim3(block), 0, s, partial, f, m);
}

}  // namespace sift
