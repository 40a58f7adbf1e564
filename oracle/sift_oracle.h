/*
 * sift_oracle.h -- CPU ORACLE. TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the CRC-32 and container-framing algorithms of the
 * reference (matta/sift, sift-persist).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link or call this.  The product
 * (sift_amd/, include/) never does; the product path is HIP only and fails
 * loudly when its library is missing.
 *
 * Pinning: checked against every known-answer vector the reference's own
 * tests hold (sift-persist/src/crc.rs:71-96, :121-122, :127-128) and against
 * Python's stdlib zlib.crc32 (an independent implementation of the same
 * CRC-32/ISO-HDLC) on seeded random buffers -- see tests/test_oracle.py and
 * tests/golden/.  The reference itself is Rust and cannot be built here (no
 * cargo/rustc in the image), so there is no oracle/_ref.
 */
#ifndef SIFT_ORACLE_H
#define SIFT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* crc.rs:2-20 -- the 256-entry table, reflected polynomial 0xEDB88320. */
const uint32_t *oracle_crc_table(void);

/* crc.rs:28-34 -- compute(crc, buf): running state in, running state out. */
uint32_t oracle_crc_compute(uint32_t crc, const uint8_t *buf, size_t len);

/* crc.rs:36-53 -- Digest::{new, update, finalize}. */
typedef struct { uint32_t value; } oracle_digest;
void oracle_digest_new(oracle_digest *d);
void oracle_digest_update(oracle_digest *d, const uint8_t *buf, size_t len);
uint32_t oracle_digest_finalize(const oracle_digest *d);

/* container.rs:4-18 -- the Error enum, as integer codes. */
enum {
    ORACLE_OK = 0,
    ORACLE_ERR_READ = 1,                    /* Error::Read (short read / bad data) */
    ORACLE_ERR_WRITE = 2,                   /* Error::Write (output buffer too small) */
    ORACLE_ERR_READ_INVALID_CHUNK_SIZE = 3, /* Error::ReadInvalidChunkSize */
    ORACLE_ERR_WRITE_INVALID_CHUNK_SIZE = 4,/* Error::WriteInvalidChunkSize */
    ORACLE_ERR_READ_CRC_MISMATCH = 5        /* Error::ReadCrcMismatch */
};

#define ORACLE_HEADER_LEN 9
#define ORACLE_CHUNK_SIZE_LIMIT (1024u * 1024u * 1024u) /* container.rs:49 */

/* container.rs:79-84 -- Chunk::compute_crc: digest over type then data. */
uint32_t oracle_chunk_crc(const uint8_t chunk_type[4], const uint8_t *data, size_t len);

/* container.rs:104,124-126 -- write_header.  Returns bytes written or <0. */
int64_t oracle_write_header(uint8_t *out, size_t cap);
/* container.rs:107-121 -- read_header.  Returns bytes consumed or -ORACLE_ERR_*. */
int64_t oracle_read_header(const uint8_t *in, size_t avail);

/* container.rs:68-77 -- Chunk::write.  Returns bytes written or -ORACLE_ERR_*. */
int64_t oracle_write_chunk(const uint8_t chunk_type[4], const uint8_t *data, uint64_t len,
                           uint8_t *out, size_t cap);

/* container.rs:47-66 -- Chunk::read.  On success returns bytes consumed and
 * fills type / data offset / data length; on failure returns -ORACLE_ERR_*. */
int64_t oracle_read_chunk(const uint8_t *in, size_t avail, uint8_t chunk_type[4],
                          uint64_t *data_off, uint32_t *data_len);

/* document.rs:50-60 -- write_document's framing around an opaque payload (the
 * bytes encode_document returned): header, AMRG chunk, empty SEND chunk.
 * Returns bytes written or -ORACLE_ERR_*. */
int64_t oracle_write_document(const uint8_t *payload, uint64_t len, uint8_t *out, size_t cap);
/* document.rs:62-77 -- read_document's framing: read_header, read_chunk +
 * expect_type(AMRG), [decode_document: the caller's], read_chunk +
 * expect_type(SEND).  Returns bytes consumed or -ORACLE_ERR_*; *stage says how
 * far it got (0 header, 1 AMRG chunk, 2 AMRG type, 3 SEND chunk, 4 SEND type,
 * 5 all good); payload offset / length are valid from stage 3 on. */
int64_t oracle_read_document(const uint8_t *in, size_t avail, uint64_t *payload_off,
                             uint32_t *payload_len, int *stage);

/* Bounded timing helper for bench.py's cpu_baseline leg: runs
 * oracle_crc_compute over buf `passes` times, returns the final state. */
uint32_t oracle_crc_repeat(const uint8_t *buf, size_t len, int passes);

#ifdef __cplusplus
}
#endif
#endif
