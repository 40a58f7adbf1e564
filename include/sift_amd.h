/*
 * sift_amd.h -- C ABI of the MI355X (gfx950) CRC-32 / container hot path.
 *
 * What this replaces.  matta/sift has no FFI, plugin or operator boundary
 * (SURVEY.md section 8(b)): `mod crc` is private to the sift-persist crate.  The
 * entry points below are therefore the ones a maintainer WOULD bind if the
 * crate's private functions were swapped for an accelerator: one entry per
 * reference function, same argument meaning, same state-in/state-out
 * convention, same error cases.  Each declaration cites the reference
 * function it stands in for.  INTEGRATION.md shows the Rust `extern "C"`
 * block and the two-line patch to crc.rs / container.rs.
 *
 * Conventions
 *  - plain C: pointers, sizes, integer status codes; no C++ or torch types.
 *  - every input byte is read and reduced on the GPU.  The host computes only
 *    geometry-dependent constants (a dozen GF(2) multiplies).  There is no CPU fallback:
 *    without a HIP device every compute call returns SIFT_ERR_NO_DEVICE.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *  - "state" is the running CRC register exactly as the reference keeps it in
 *    Digest.value (crc.rs:36-53): initial 0xFFFFFFFF, final XOR by the caller
 *    (or use the sift_digest_* helpers, which do both).
 */
#ifndef SIFT_AMD_H
#define SIFT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
enum {
    SIFT_OK = 0,
    /* container.rs:4-18, enum Error */
    SIFT_ERR_READ = 1,                     /* Error::Read: short input / bad magic / wrong type */
    SIFT_ERR_WRITE = 2,                    /* Error::Write: output buffer too small */
    SIFT_ERR_READ_INVALID_CHUNK_SIZE = 3,  /* Error::ReadInvalidChunkSize */
    SIFT_ERR_WRITE_INVALID_CHUNK_SIZE = 4, /* Error::WriteInvalidChunkSize */
    SIFT_ERR_READ_CRC_MISMATCH = 5,        /* Error::ReadCrcMismatch */
    /* errors the reference cannot have (it never leaves the CPU) */
    SIFT_ERR_NO_DEVICE = 16,               /* no HIP device / device index out of range */
    SIFT_ERR_HIP = 17,                     /* a HIP runtime call failed */
    SIFT_ERR_INVALID_ARG = 18,             /* NULL pointer with non-zero length, etc. */
    SIFT_ERR_TOO_MANY_THREADS = 19         /* more than 1024 threads alive that have used the synchronous
                                              entry points (each keeps a pinned result slot until it exits) */
};

/* Human-readable text for a status code (static storage). */
const char *sift_strerror(int status);
/* Detail of the last failure on the calling thread (HIP error text), or "". */
const char *sift_last_error(void);

#define SIFT_HEADER_LEN 9u                               /* container.rs:104 */
#define SIFT_CHUNK_SIZE_LIMIT (1024u * 1024u * 1024u)    /* container.rs:49  */
#define SIFT_CRC_INIT 0xFFFFFFFFu                        /* crc.rs:42        */

/* ------------------------------------------------------------ device setup */
/* Builds the per-device tables and scratch.  Optional: every compute call
 * initialises its device on first use.  Returns SIFT_OK or SIFT_ERR_*. */
int sift_init(int device);
/* Frees everything the library allocated on every device. */
void sift_shutdown(void);
/* Waits for the device, then re-zeroes every piece of scratch the library keeps
 * on it (per-stream workspaces, batch counters, the static array).  Between
 * calls that scratch is zero by construction; only a kernel that was aborted
 * half-way can leave it otherwise, and this puts it right.  No reference
 * counterpart. */
int sift_reset_workspace(int device);
/* Gives back everything the library keeps for `stream` on `device` (its 12 KiB
 * workspace block, its batch scratch, its pinned descriptor buffers and events),
 * after waiting for the stream.  Call it before hipStreamDestroy on a stream that
 * was passed to this library with workspace = NULL or to a batch entry point;
 * without it those blocks stay until sift_shutdown, and a later stream that
 * happens to get the same handle value inherits them (harmless: they are zero
 * between calls).  For hipStreamPerThread it releases the calling thread's entry.
 * A stream the library holds nothing for is not an error.  No reference
 * counterpart. */
int sift_release_stream(int device, void *stream);
/* Bytes of device workspace one in-flight sift_crc32_compute_device call
 * needs when the caller brings its own (see `workspace` below).  A caller's
 * workspace must be 256-byte aligned (as hipMalloc returns it: the exchange words
 * are 8-byte atomics on 128-byte lines of their own; a misaligned pointer is
 * refused with SIFT_ERR_INVALID_ARG) and zero-filled once before its first use;
 * the library leaves it reusable (its arrival counter back at zero) after every
 * call. */
size_t sift_crc32_workspace_bytes(void);

/* ------------------------------------------------------------ measurement
 * No reference counterpart.  Between _begin and _end every `stride`-th
 * sift_crc32_compute_* / sift_chunk_crc_device launch on `device` times its
 * tile kernel (the dominant kernel) with a HIP event pair on the launch's own
 * stream, up to `max_launches` timed launches.  The pair is attached to the
 * kernel's dispatch (hipExtLaunchKernelGGL: the events take the dispatch's start
 * and end timestamps, the quantity rocprofv3 reports); with SIFT_PROF_EXT=0 in
 * the environment the events are recorded before and behind the launch instead
 * (round 1's method: 2-3 us of marker packets inside the bracket and a few
 * microseconds of idle stream around it, hence the stride).  _end waits
 * for them and returns the summed kernel time in milliseconds and the number
 * of launches measured.
 * bench.py divides algorithmic bytes by this for `roofline.achieved`. */
int sift_profile_begin(int device, int max_launches, int stride);
int sift_profile_end(int device, double *tile_kernel_ms, int *launches);
/* Same, and also copies the first `max_samples` per-launch kernel times (ms)
 * into `samples_ms` (may be NULL). */
int sift_profile_end_samples(int device, double *tile_kernel_ms, int *launches, float *samples_ms,
                             int max_samples);
/* Same for every kernel of the library: between _begin and _end the batch entry
 * points time their kernels as well (every launch, whatever `stride`; each timed
 * launch takes one of the `max_launches` pairs).  kernel_ms[k] / launches[k]:
 * k = 0 the tile kernel, 1 the batch (wave) kernel, 2 the piece kernel.
 * `tile_samples_ms` as samples_ms above. */
int sift_profile_end_kernels(int device, double kernel_ms[3], int launches[3], float *tile_samples_ms,
                             int max_samples);

/* Measurement knobs, process-wide.  No reference counterpart. */
enum {
    SIFT_TUNE_SINGLE_LAUNCH = 1, /* 1 (default): the last workgroup to arrive combines the
                                    residues inside the tile kernel; 0: a second, combine
                                    kernel does (A/B measurement only, same results) */
    SIFT_TUNE_BATCH_CUT = 2,     /* 1 (default): the batch entry points cut long items on the
                                    device; 0: one wave per item whatever its length */
    SIFT_TUNE_BATCH_WAVE_MAX = 3,/* bytes of CRC span one wave keeps before the item is cut;
                                    0 (default) = from the items' lengths: 32 KiB, or 64 / 128 /
                                    256 KiB when a sample of the batch holds thousands of items
                                    that long (a wave alone streams 4.5 GB/s: waves pay once there
                                    are thousands; a lone long item is always cut) */
    SIFT_TUNE_BATCH_PIECE_LOG2 = 4, /* log2 of the smallest piece, 14..24 (default 16 = 64 KiB) */
    SIFT_TUNE_HOST_COPY_THREADS = 6, /* helper threads that fill the pinned bounce ring for pageable
                                    input of more than 512 KiB, 1..16 (default 4; the calling thread
                                    copies nothing, it launches the kernel over the slot filled before);
                                    0: always the staged hipMemcpyAsync path.  Set before the first
                                    host-input call on a device. */
    SIFT_TUNE_BATCH_PLAN = 5     /* 1 (default): up to 64 items (not whole documents) go straight to the
                                    piece kernel, which plans them itself, and up to 4096 it finds the
                                    long ones itself; 0: the wave that meets a long item files it */
};
int sift_tune(int key, int value);

/* ------------------------------------------------------- crc.rs:28-34 ----
 * fn compute(crc: u32, buf: &[u8]) -> u32
 *
 * Device-resident input, asynchronous on `stream`.  Reads `len` bytes at
 * device address `d_buf` (any alignment), starting from running state
 * `state`, and stores the new running state to device address `d_out`
 * (4-byte aligned).  If `d_state_in` is non-NULL the starting state is read
 * from that device word instead of `state` (lets Digest::update calls chain
 * on a stream with no host round trip).  `workspace` is either a device buffer
 * of sift_crc32_workspace_bytes() that no other in-flight call uses, or NULL:
 * the library then uses the workspace it keeps for `stream` (one per stream,
 * allocated and zeroed on that stream's first call, so that first call
 * allocates; calls on one stream run in order and share it safely, calls on
 * different streams never share one, however many are in flight;
 * hipStreamPerThread, one handle value that names a different stream in every
 * thread, is keyed by the calling thread as well; NULL and hipStreamLegacy are one
 * stream; sift_release_stream gives a stream's block back).  Under HIP
 * graph capture a NULL workspace is refused (SIFT_ERR_INVALID_ARG): a graph may
 * replay on any stream, so bring a workspace, or use
 * sift_crc32_compute_device_static.
 */
int sift_crc32_compute_device(int device, uint32_t state, const uint32_t *d_state_in,
                              const void *d_buf, uint64_t len, uint32_t *d_out,
                              void *workspace, void *stream);

/* The same computation for callers that cannot allocate: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  NO COUNTERPART
 * IN THE REFERENCE (it never leaves the CPU).  Like sift_crc32_compute_device with
 * workspace = NULL, except that the workspace is an 8.3 KiB array in the
 * library's own static device memory (one copy per GPU, held until the process
 * ends) instead of the block allocated for the stream, and that the call itself makes
 * no hipMalloc / hipFree or any other allocating call.  Requirements:
 *  - sift_init(device) has returned SIFT_OK earlier (set-up allocates the
 *    constant tables; this call does not, and returns SIFT_ERR_INVALID_ARG if the
 *    device was never initialised);
 *  - len <= SIFT_STATIC_MAX_LEN, else SIFT_ERR_INVALID_ARG (the scratch holds one
 *    residue per workgroup whatever the length, so the limit is an addressing
 *    one, not a capacity one);
 *  - every call on one GPU shares that GPU's array: two calls that overlap in
 *    time on one GPU (different streams or threads) race and both results are
 *    undefined.  Calls queued on ONE stream are fine.  The accumulator and the
 *    arrival counter are cleared on the stream at the start of every call. */
#define SIFT_STATIC_MAX_LEN (1ull << 38) /* 256 GiB */
int sift_crc32_compute_device_static(int device, uint32_t state, const uint32_t *d_state_in,
                                     const void *d_buf, uint64_t len, uint32_t *d_out,
                                     void *stream);

/* Same as sift_crc32_compute_device, then waits and returns the new state through `*out` (host).
 * The wait is a poll of a pinned 8-byte word (tag | result) that the call's last kernel stores
 * and that belongs to the calling thread, not a hipStreamSynchronize: the call returns when the
 * result is there (a few microseconds earlier); after 60 us of polling it waits for the stream
 * instead, which is also where a failed kernel's error surfaces.
 * Work queued on `stream` BEFORE the call has finished by then; the call's own kernel has done
 * all its reads of `d_buf` and is at most retiring. */
int sift_crc32_compute_device_sync(int device, uint32_t state, const void *d_buf, uint64_t len,
                                   uint32_t *out, void *stream);

/* Host-resident input, reduced on the GPU; returns the new state through
 * `*out`.  Pinned memory (hipHostMalloc / hipHostRegister) is read by the
 * kernel in place over PCIe.  Pageable memory of up to 512 KiB is copied by the
 * calling thread into a pinned slot and read there; more than that is copied by
 * a few host threads into a ring of pinned 8 MiB slots which the kernel reads in
 * place, slot k+1 being filled while slot k is read (with
 * SIFT_TUNE_HOST_COPY_THREADS = 0: staged with hipMemcpyAsync instead).  Calls
 * on one device are serialised.  This is the signature the reference's
 * compute(crc, buf) maps to one-to-one. */
int sift_crc32_compute_host(int device, uint32_t state, const void *buf, uint64_t len,
                            uint32_t *out);

/* ------------------------------------------------------- crc.rs:36-53 ----
 * struct Digest { value: u32 }  /  new / update / finalize                  */
typedef struct sift_digest {
    uint32_t value;
    int32_t device;
} sift_digest;

void sift_digest_new(sift_digest *d, int device);                          /* crc.rs:41-43 */
int sift_digest_update(sift_digest *d, const void *buf, uint64_t len);    /* crc.rs:45-47, host bytes */
int sift_digest_update_device(sift_digest *d, const void *d_buf, uint64_t len, void *stream);
uint32_t sift_digest_finalize(const sift_digest *d);                       /* crc.rs:50-52 */

/* ---------------------------------------------------- multi-GPU stitching
 * a buffer split into contiguous ranges A|B.  Given
 * state_a = compute(s, A) and state_b0 = compute(0, B), returns
 * compute(s, A|B) = state_a * x^(8*len_b) + state_b0.  Pure host GF(2)
 * arithmetic on two words; no reference counterpart (the reference is
 * single-process). */
uint32_t sift_crc32_concat(uint32_t state_a, uint32_t state_b0, uint64_t len_b);

/* ------------------------------------------------ container.rs:79-84 -----
 * Chunk::compute_crc: Digest over chunk_type then data, finalized.
 * `d_type_and_data` points at the 4 type bytes followed by `data_len` data
 * bytes, contiguous in device memory (as they are inside a serialized
 * chunk).  Result (finalized CRC) is stored to `d_out`. */
int sift_chunk_crc_device(int device, const void *d_type_and_data, uint64_t data_len,
                          uint32_t *d_out, void *workspace, void *stream);

/* ------------------------------------------------ many buffers at once -----
 * The data-parallel form of the two functions above, for callers that hold
 * many independent buffers / serialized chunks in HBM (bulk load of many
 * files).  `offsets` / `lens` are arrays of n entries (byte offset from `d_base`,
 * byte length): both device arrays (used in place), or host arrays.  Host arrays of
 * up to 1 MiB together (64 K items) are copied before the call returns, pinned or
 * not: the caller may refill or free them at once.  Larger pageable arrays are
 * copied as well (stream-ordered); larger PINNED arrays (hipHostMalloc /
 * hipHostRegister) are read in place by the GPU and must stay unchanged until
 * `stream` has passed the call.
 * Short items are taken 64 at a time by a wave (one lane per item for parsing,
 * checks and the verdict; the CRC itself by one lane, a group of 4 or 8 lanes or the
 * whole wave, by length).  An item whose CRC span is long
 * (over 32 KiB; over 64 / 128 / 256 KiB when the batch holds thousands that long) is cut ON THE DEVICE --
 * in the chunk modes the length is only in the chunk's header -- into pieces of
 * 64 KiB and up that whole workgroups reduce, so one 1 GiB chunk, 64 chunks of
 * 16 MiB and a million 64-byte chunks all run near the memory rate, whether the
 * descriptors came from the host or not.  Results go to device memory,
 * asynchronously on `stream`.  Scratch (a few hundred KiB, more for `d_base`
 * allocations beyond 1 GiB) is kept per stream and allocated when first needed;
 * the calls cannot be captured into a HIP graph.  Items that overlap each other
 * in memory are allowed; if there are so many long overlapping ones that the
 * scratch sized for non-overlapping items fills up, the rest are walked by one
 * wave each (same results, slower).
 *
 * sift_crc32_batch_device:  d_out[i] = compute(state, buffer i) ^ xor_out
 *                           (crc.rs:28-34 applied n times; state = 0xFFFFFFFF,
 *                           xor_out = 0xFFFFFFFF gives Digest new/update/finalize).
 * sift_read_chunks_device:  Chunk::read (container.rs:47-66) on n serialized
 *                           chunks; avail[i] = bytes readable from offsets[i].
 *                           d_status[i] = SIFT_OK, SIFT_ERR_READ,
 *                           SIFT_ERR_READ_INVALID_CHUNK_SIZE or
 *                           SIFT_ERR_READ_CRC_MISMATCH, raised in the reference's
 *                           order; d_data_len[i] (optional, may be NULL) = data
 *                           length of chunk i if it parsed, else 0.
 * sift_read_documents_device: the framing of read_document (document.rs:62-77)
 *                           on n whole save files: read_header, read_chunk +
 *                           expect_type(AMRG), read_chunk + expect_type(SEND).
 *                           The automerge decode between them is NOT done
 *                           (third-party codec, out of scope).  d_status[i] as
 *                           above (a wrong chunk type or bad magic is
 *                           SIFT_ERR_READ); d_amrg_len[i] (optional) = length of
 *                           the AMRG payload, which starts 17 bytes into file i. */
int sift_crc32_batch_device(int device, uint32_t state, uint32_t xor_out, const void *d_base,
                            const uint64_t *offsets, const uint64_t *lens, uint32_t n,
                            uint32_t *d_out, void *stream);
int sift_read_chunks_device(int device, const void *d_base, const uint64_t *offsets,
                            const uint64_t *avail, uint32_t n, uint32_t *d_status,
                            uint32_t *d_data_len, void *stream);
/* Chunk::write's CRC step (container.rs:75) on n serialized chunks whose
 * length | type | data are already laid out in device memory with a 4-byte slot
 * behind the data: computes each CRC over type | data and stores it big-endian
 * into the slot.  avail[i] = bytes of chunk i's slot (>= 12 + its length);
 * d_status[i] = SIFT_OK or SIFT_ERR_WRITE (slot too small). */
int sift_write_chunk_crcs_device(int device, void *d_base, const uint64_t *offsets,
                                 const uint64_t *avail, uint32_t n, uint32_t *d_status,
                                 void *stream);
int sift_read_documents_device(int device, const void *d_base, const uint64_t *offsets,
                               const uint64_t *avail, uint32_t n, uint32_t *d_status,
                               uint32_t *d_amrg_len, void *stream);

/* ------------------------------------------- container.rs:104-126 --------
 * write_header / read_header on byte slices.  Return bytes written/consumed
 * (9) or -SIFT_ERR_*. */
int64_t sift_write_header(uint8_t *out, size_t cap);
int64_t sift_read_header(const uint8_t *in, size_t avail);

/* ------------------------------------------- container.rs:68-77 ----------
 * Chunk::write: u32 BE length | type | data | u32 BE crc, CRC on the GPU.
 * Returns bytes written (len + 12) or -SIFT_ERR_*. */
int64_t sift_write_chunk(int device, const uint8_t chunk_type[4], const uint8_t *data,
                         uint64_t len, uint8_t *out, size_t cap);

/* ------------------------------------------- container.rs:47-66 ----------
 * Chunk::read: parses one chunk at `in`, verifies its CRC on the GPU.
 * On success returns bytes consumed and fills type / offset of data within
 * `in` / data length; on failure returns -SIFT_ERR_* in the order the
 * reference raises them (size limit before type/data reads, CRC last). */
int64_t sift_read_chunk(int device, const uint8_t *in, size_t avail, uint8_t chunk_type[4],
                        uint64_t *data_off, uint32_t *data_len);

/* ------------------------------------------- container.rs:86-96 ----------
 * Chunk::expect_type.  SIFT_OK or SIFT_ERR_READ. */
int sift_expect_type(const uint8_t chunk_type[4], const uint8_t expected[4]);

/* ------------------------------------------- document.rs:53-60 -----------
 * write_document's framing: header, an AMRG chunk holding `payload` (the bytes
 * encode_document returned: an automerge document, opaque here), an empty SEND
 * chunk; both CRCs on the GPU.  Returns bytes written (len + 33) or -SIFT_ERR_*
 * (the error of the first write that fails, as the reference's `?` chain). */
int64_t sift_write_document(int device, const uint8_t *payload, uint64_t len, uint8_t *out,
                            size_t cap);

/* ------------------------------------------- document.rs:62-77 -----------
 * read_document's framing, in the reference's order: read_header, read_chunk
 * (CRC verified on the GPU) + expect_type(AMRG), read_chunk + expect_type(SEND).
 * Returns bytes consumed or -SIFT_ERR_* (SIFT_ERR_READ for bad magic, a short
 * input or a wrong chunk type; trailing bytes are ignored, as the reference
 * ignores them).  `*stage` (optional) says how far it got; from
 * SIFT_DOC_SEND_CHUNK on, *payload_off / *payload_len locate the AMRG payload
 * inside `in`.  The reference decodes the payload (document.rs:69, automerge,
 * not part of this library) BEFORE it reads the SEND chunk: a caller that wants
 * the reference's error precedence decodes first whenever stage >= SEND_CHUNK
 * and lets a decode error win over the SEND chunk's. */
enum {
    SIFT_DOC_HEADER = 0,
    SIFT_DOC_AMRG_CHUNK = 1,
    SIFT_DOC_AMRG_TYPE = 2,
    SIFT_DOC_SEND_CHUNK = 3,
    SIFT_DOC_SEND_TYPE = 4,
    SIFT_DOC_DONE = 5
};
int64_t sift_read_document(int device, const uint8_t *in, size_t avail, uint64_t *payload_off,
                           uint32_t *payload_len, int *stage);

#ifdef __cplusplus
}
#endif
#endif /* SIFT_AMD_H */
