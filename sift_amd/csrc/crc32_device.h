// crc32_device.h -- structures shared by the kernels and the host API.
//
// Geometry.  One workgroup of kThreads lanes owns one CU (its LDS tables fill
// most of the 160 KiB).  A "vector" is 16 bytes (one dwordx4 load per lane); a
// "tile" is one vector per lane = kTileBytes contiguous bytes.  A workgroup
// walks a contiguous run of tiles; lane t keeps four running words (one per
// dword of its vector) and advances them with a Horner step whose stride is
// one tile:
//       acc <- acc * y + w,    y = x^(8 * kTileBytes) mod P
// (the `* y` is four LDS table lookups, exactly as many as the reference's
// byte table would need for the same four bytes, crc.rs:31).
#pragma once

#include <stdint.h>

namespace sift {

// Tunable at build time (-DSIFT_THREADS=..., tools/hbm_read_probe.hip builds
// variants to compare); the shipped values are the measured best.
#ifndef SIFT_THREADS
#define SIFT_THREADS 1024
#endif
#ifndef SIFT_REPL
#define SIFT_REPL 32
#endif
#ifndef SIFT_UNROLL
#define SIFT_UNROLL 2
#endif
#ifndef SIFT_VPL
#define SIFT_VPL 1
#endif
#ifndef SIFT_GROUPS_PER_CU
#define SIFT_GROUPS_PER_CU 1
#endif

constexpr int kThreads = SIFT_THREADS;      // lanes per workgroup
constexpr int kVpl = SIFT_VPL;              // vectors per lane per tile (independent Horner chains)
constexpr int kTileVecs = kThreads * kVpl;  // 16-byte vectors per tile; lane t holds vectors t + j*kThreads
constexpr int kTileBytes = kTileVecs * 16;  // 16 KiB at 1024 lanes x 1 vector
constexpr int kRepl = SIFT_REPL;            // LDS copies of each table row (32 = one per bank)
constexpr int kReplLog2 = kRepl == 32 ? 5 : kRepl == 16 ? 4 : kRepl == 8 ? 3 : -1;
constexpr int kUnroll = SIFT_UNROLL;        // tiles in flight per lane, twice (current + next batch)
constexpr int kGroupsPerCu = SIFT_GROUPS_PER_CU;  // resident workgroups per CU the LDS budget allows
static_assert(kReplLog2 > 0, "SIFT_REPL must be 8, 16 or 32");
static_assert(kThreads % 64 == 0 && kThreads >= 64 && kThreads <= 1024, "SIFT_THREADS");
// LDS copies of the batch kernel's tables (SQ counters, profiles/r02_batch_pmc.json: with 2
// copies of the group path's Horner table and 1 of the x^32 table, two thirds of the LDS
// cycles at 1-3 KB items were bank conflicts).  160 KiB hold: wave path's table kWaveRepl
// copies, group path's kGroupRepl, x^32 kXRepl, the lane-join tables and T8 one each.
#ifndef SIFT_WAVE_REPL_LOG2
#define SIFT_WAVE_REPL_LOG2 4
#endif
#ifndef SIFT_GROUP_REPL_LOG2
#define SIFT_GROUP_REPL_LOG2 4
#endif
#ifndef SIFT_X_REPL_LOG2
#define SIFT_X_REPL_LOG2 1
#endif
constexpr int kWaveReplLog2 = SIFT_WAVE_REPL_LOG2, kWaveRepl = 1 << kWaveReplLog2;
constexpr int kGroupReplLog2 = SIFT_GROUP_REPL_LOG2, kGroupRepl = 1 << kGroupReplLog2;
constexpr int kXReplLog2 = SIFT_X_REPL_LOG2, kXRepl = 1 << kXReplLog2;
constexpr int kMaxGroups = 1024;            // workgroups one call may launch
constexpr int kTreeLevels = 10;             // log2(kMaxGroups)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Constant tables, built on the host once per device (crc32_api.hip).
struct DeviceTables {
    uint32_t Y[4 * 256];     // Y[k*256+b] = (b << 8k) * x^(8*kTileBytes) mod P
    __attribute__((aligned(16))) uint32_t Yt[256 * 4];  // the same by byte value: Yt[4b + k] = Y[k*256+b] (how the kernels' LDS rows want it)
    uint32_t X[4 * 256];     // X[k*256+b] = (b << 8k) * x^32 mod P
    __attribute__((aligned(16))) uint32_t Xt[256 * 4];  // Xt[4b + k] = X[k*256+b]
    uint32_t Kt[kThreads];   // Kt[t] = x^(128 * (kThreads-1-t)) mod P
    uint32_t Y1[4 * 256];    // Y1[k*256+b] = (b << 8k) * x^(8*1024) mod P  (one-wave tiles, batch kernel)
    uint32_t Y256[4 * 256];  // Y256[k*256+b] = (b << 8k) * x^(8*256) mod P  (a chain of a lane group steps over 16 vectors, batch kernel)
    uint32_t T8[256];        // T8[b] = b * x^8 mod P: the reference's CRC_TABLE itself (crc.rs:2-20), for
                             // bytes that are walked one at a time (tails, spans under 32 bytes)
    uint32_t K[4][4 * 256];  // K[j][k*256+b] = (b << 8k) * x^(128 * 2^j): joins neighbouring lanes' residues
    uint32_t C1[32];         // C1[i] = x^(128*kThreads) * x^i: from one of a lane's vectors to its next
    uint32_t X2[32][32];     // X2[s][k] = x^(2^s) * x^k: row k of the matrix of x^(2^s) (piece weights)
    uint32_t XinvH[16];      // XinvH[h] = x^(-8h): folds a starting register into a vector with h pad bytes
    // The batch kernel's LDS, word for word (160 KiB; layout: KiLane in crc32_kernels.hip):
    // Y256 and X by byte value in rows of 256 B, K[0..3] likewise, Y1 in rows of 128 B.
    __attribute__((aligned(16))) uint32_t batch_image[40960];
};

// Arguments of the tile kernel.
struct TileArgs {
    const u32x4* vbase;        // 16-byte aligned; vector 0 holds the first message byte
    int64_t nvec;              // whole vectors in the body (vector 0 may be front-masked)
    int64_t tiles_per_group;   // virtual tiles each workgroup owns
    int64_t virtual_tiles;     // groups * tiles_per_group
    int64_t pad_tiles;         // all-zero virtual tiles in front of the first real one
    const DeviceTables* tabs;
    uint32_t* partial;         // [groups] residues, one per workgroup
    uint32_t* exchange;        // single-launch finish: 1 + kMaxGroups/32 u64 words, each on its own 128-B
                               // line (top word, then one per shard of 32 groups); all zero between calls
    // Starting register, folded into the message: with h = head_skip leading
    // zero bytes the padded message is n + h bytes long and
    //   state * x^(8n) = (state * x^(-8h)) * x^(8(n+h)),
    // which is the weight of the padded message's first word.  So the lane that
    // holds vector 0 XORs state * x^(-8h) into word 0.  Immediate state: the host
    // does the multiply (state_fold).  Device state (state_ptr): that lane does
    // it with xinv_h = x^(-8h).
    uint32_t state_fold;
    uint32_t xinv_h;
    const uint32_t* state_ptr;
    uint32_t head_skip;        // bytes of vector 0 that precede the message (zeroed)
    uint32_t fuse_final;       // groups == 1: finish in this launch, no exchange
    uint32_t single_launch;    // groups > 1: last group to arrive combines and finishes
};

// Arguments of the finishing step (combine kernel, or fused into the tile
// kernel when one workgroup did all the work).
struct FinalArgs {
    // Serial path only (inputs under 32 bytes, groups == 0): the starting register,
    // as an immediate or, if state_ptr is non-null, read on the device.  On the
    // vector path the state is folded into vector 0 by the tile kernel (TileArgs).
    uint32_t state_imm;
    const uint32_t* state_ptr;
    const uint8_t* tail;         // bytes after the last whole vector (or the whole tiny input)
    uint32_t tail_len;
    uint32_t xor_out;            // 0, or 0xFFFFFFFF to finalize (Chunk::compute_crc)
    uint32_t* out;
    int32_t groups;
    // Synchronous entry points: `out` is an 8-byte slot in pinned host memory and the
    // result goes out as ONE 8-byte store (tag << 32 | result), which the calling thread
    // polls for instead of waiting for the stream (0: plain 4-byte store).
    uint32_t tag;
};

// Row k of level s: z^(2^s) * x^k, z = x^(8 * tiles_per_group * kTileBytes).
struct CombineMats {
    uint32_t z[kTreeLevels][32];
};

// Device-side cutting of long items (batch kernels).  A long CRC span is not
// walked by the wave that finds it: that wave files a record and the span is cut
// into np pieces of (kPieceMinBytes << level) bytes, each reduced by a whole
// workgroup (crc32_pieces_kernel, the tile kernel's algorithm), and a last small
// kernel XORs the weighted piece residues and finishes the item.  Cuts are at
// absolute 16-byte boundaries counted back from the end of the span's body, so
// only piece 0 has a ragged head and only the finisher sees the < 16 tail bytes.
constexpr int kPieceMinLog2 = 16;            // 64 KiB: four 16 KiB tiles
constexpr uint32_t kMaxPiecesPerItem = 1024; // level is raised until np fits (up to kMaxLevel)
constexpr uint32_t kMaxLevel = 14;
constexpr uint32_t kPlanMax = 64;             // up to this many items the piece kernel plans for itself (one lane each)
constexpr uint32_t kWideMax = 4096;           // up to this many it plans the LONG ones (four items a lane, all lanes), the
                                              // batch kernel keeps the others: no filing (one returning atomic pair per long
                                              // item on two words: 23 ns an item, 25 us of a 190 us call on 1024 x 1 MiB)
// The longest span a wave keeps (wave_max), from the lengths themselves.  A wave alone streams
// about 4.5 GB/s, whole workgroups 27 GB/s each but at 2 us a run: waves pay when there are
// thousands of spans of a length, pieces when there are few (profiles/r03_cut_grid.json:
// crossover near 1100 items at 64 KiB, 1500-2000 at 128 KiB, 2500-3500 at 256 KiB).  The item
// COUNT says nothing about that: 3500 items of 64 bytes and one of 250 KiB took 62 us with
// the 250 KiB on one wave, 18 us with it cut.  So lengths are counted in three classes
// (32-64, 64-128, 128-256 KiB) over a sample of kWaveSample items -- four runs of 64 at the
// quarter points of the batch, which every wave reads and counts for itself (ballots; the
// same 2 KiB for all of them) -- and the threshold is the top of the highest class with
// enough members.  Same function on the host (lengths it can read) and in both kernels
// (lengths in HBM).  (A sample of 4096 evenly spaced items counted by the workgroup together
// cost a million-item call 5 us, 1024 still 2.5, the workgroup's barriers alone 2 -- and a
// batch of a hundred items 3.7.)
constexpr uint32_t kWaveSample = 256;
constexpr uint32_t kWaveMin = 32u << 10;
// what one sample member stands for, and member k's item (k < n, k < kWaveSample)
__host__ __device__ inline uint32_t wave_sample_stride(uint32_t n) {
    return n <= kWaveSample ? 1u : n / kWaveSample;
}
__host__ __device__ inline uint32_t wave_sample_item(uint32_t k, uint32_t n) {
    return n <= kWaveSample ? k : (n / 4u) * (k >> 6) + (k & 63u);
}
__host__ __device__ inline uint32_t wave_class(uint64_t len) {  // 0: none, 1..3
    return len <= (32u << 10) || len > (256u << 10) ? 0u : len <= (64u << 10) ? 1u : len <= (128u << 10) ? 2u : 3u;
}
__host__ __device__ inline uint32_t wave_max_of(uint32_t c1, uint32_t c2, uint32_t c3, uint32_t stride) {
    uint32_t t = kWaveMin;  // counts are of the sample: times the stride for the batch
    if ((uint64_t)c1 * stride >= 1100u) t = 64u << 10;
    if ((uint64_t)c2 * stride >= 1500u) t = 128u << 10;
    if ((uint64_t)c3 * stride >= 2500u) t = 256u << 10;
    return t;
}
constexpr uint32_t kVoidSlot = 0xFFFFFFFFu;  // piece_rec entry of a reservation that did not fit           // pieces of at most 1 GiB (32-bit buffer offsets)

struct CutRecord {
    uint64_t span_off;           // first byte of the CRC span, from BatchArgs::base
    uint64_t span_len;
    unsigned long long acc;      // XOR of the weighted residues of the runs that have arrived
    unsigned long long arrived;  // pieces those runs covered; the run that brings it to np finishes the item
    uint32_t item;               // index into out / data_len
    uint32_t piece_base;         // its pieces are piece_base .. piece_base + np
    uint32_t np;                 // 0: voided reservation
    uint32_t level;
    uint32_t cont;               // status the item gets if the span's CRC matches (verify modes)
    uint32_t dlen;               // verify modes: data length of the chunk
    uint32_t pad[2];
};
static_assert(sizeof(CutRecord) == 64, "CutRecord layout");

// Scratch of one batch call.  The three counters sit at the head of the scratch
// block, are zero between calls, and are put back to zero by the piece kernel's
// last workgroup to leave.
struct CutArgs {
    uint32_t* counts;            // records filed (may run past cap_recs); null = never cut
    uint32_t* left;              // piece-kernel workgroups that have left
    unsigned long long* pieces;  // piece slots reserved (may run past cap_pieces)
    CutRecord* recs;
    uint32_t* piece_rec;         // piece -> record slot, or kVoidSlot
    uint32_t cap_recs, cap_pieces;
    uint32_t wave_max;           // spans of up to this many bytes stay on the wave that found them
    // wave_adapt: wave_max is worked out by the kernels from the items' lengths (wave_threshold:
    // every wave of both kernels, identically), because the host cannot read them
    uint32_t wave_adapt;
    uint32_t piece_min_log2;     // smallest piece, log2 bytes (kPieceMinLog2 unless tuned)
    // Wide plan (64 < n <= kWideMax): a wave that meets a long span leaves it alone and raises
    // *long_flag; the piece kernel returns at once when it is not raised.  Two flag words take
    // turns from call to call: the batch kernel of a call clears the one the NEXT call will use.
    // `recs` is then kWideMax pairs of 64-bit words (accumulator, pieces arrived), zero between calls.
    uint32_t* long_flag;
    uint32_t* clear_flag;
};
constexpr size_t kCutFlagsOffset = 192;                     // counts @0, left @64, pieces @128, the two flags @192
constexpr size_t kCutWideOffset = 256;                      // wide plan's accumulators
constexpr size_t kCutHeadBytes = 256 + 16 * kWideMax;       // head of the batch scratch block: zero between calls

// Batch kernel (one wave per buffer).  Buffer i is `lens[i]` bytes at
// `base + offsets[i]`.  In the verify / write modes the item is a serialized
// chunk or file (container.rs:47-77): `lens[i]` is the number of bytes
// available from its offset, its own length comes from its header.
struct BatchArgs {
    const uint8_t* base;
    const uint64_t* offsets;  // device arrays of n entries
    const uint64_t* lens;
    uint32_t n;
    uint32_t state;       // starting register for every buffer (crc mode)
    uint32_t state_m32;   // (the mode's starting register) * x^-32: 0xFFFFFFFF in the chunk modes, `state` in crc mode
    uint32_t xor_out;     // crc mode: XORed into each result
    uint32_t verify;      // 0: out[i] = compute(state, buffer i) ^ xor_out
                          // 1: out[i] = status of Chunk::read on item i (SIFT_OK / SIFT_ERR_*)
                          // 2: out[i] = status of read_document's framing on file i
                          // 3: store the CRC field of serialized chunk i (Chunk::write), out[i] = status
    uint32_t chunk_limit; // verify: container.rs:49 size cap
    uint32_t items_per_wave;  // batch kernel: consecutive items a wave takes at a time, one per lane (1..64)
    uint32_t deal_log2;       // ... and 2^deal_log2 consecutive turns stay on one workgroup (0..4)
    uint32_t* out;
    uint32_t* data_len;   // verify, optional: data length of each chunk that parsed
    const DeviceTables* tabs;
    CutArgs cut;
};

// Up to kPlanMax items whose descriptors the host can read travel in the kernel
// arguments themselves (offset, length pairs): no staging copy and no PCIe read.
// A second argument of the planned piece kernel only, so that no other launch
// carries the kilobyte.
struct InlineDesc {
    uint64_t v[2 * kPlanMax];
};

void launch_batch(const BatchArgs& b, int groups, void* stream, void* ev_start = nullptr, void* ev_stop = nullptr);
// plan: 0 the filed list, 1 up to kPlanMax items planned by one lane each, 2 the wide plan
void launch_pieces(const BatchArgs& b, int groups, int plan, const InlineDesc* inl, void* stream,
                   void* ev_start = nullptr, void* ev_stop = nullptr);
void launch_clear(uint32_t* words, int n, void* stream);
void launch_tiles(const TileArgs& t, const FinalArgs& f, const CombineMats& m, int groups,
                  bool use_y, void* stream, void* ev_start = nullptr, void* ev_stop = nullptr);
void launch_combine(const uint32_t* partial, const FinalArgs& f, const CombineMats& m,
                    void* stream);

}  // namespace sift
