// container.cpp -- the PNG-style container framing of the reference
// (sift-persist/src/container.rs) on byte slices, with every CRC computed by
// the GPU path (sift_crc32_compute_host -> crc32_kernels.hip).
//
// The reference reads from `impl Read` and writes to `impl Write`; across a C
// ABI those become (pointer, length) slices, and io errors become "slice too
// short" (Error::Read) / "output too small" (Error::Write).
#include <cstring>

#include "../../include/sift_amd.h"

namespace {

// container.rs:20-32 -- read_u32 / write_u32, big endian.
inline uint32_t be32_get(const uint8_t* p) {
    return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]);
}

inline void be32_put(uint8_t* p, uint32_t v) {
    p[0] = uint8_t(v >> 24);
    p[1] = uint8_t(v >> 16);
    p[2] = uint8_t(v >> 8);
    p[3] = uint8_t(v);
}

// container.rs:104
const uint8_t kHeader[SIFT_HEADER_LEN] = {0x89, 0x53, 0x49, 0x46, 0x54, 0x0D, 0x0A, 0x1A, 0x0A};

// container.rs:79-84 -- Chunk::compute_crc over `type | data`, which are
// adjacent both in a serialized chunk and in the slices handled here.
int chunk_crc(int device, const uint8_t* type_and_data, uint64_t data_len, uint32_t* crc) {
    sift_digest d;
    sift_digest_new(&d, device);
    int rc = sift_digest_update(&d, type_and_data, data_len + 4u);
    if (rc != SIFT_OK) return rc;
    *crc = sift_digest_finalize(&d);
    return SIFT_OK;
}

}  // namespace

extern "C" {

int64_t sift_write_header(uint8_t* out, size_t cap) {
    if (!out || cap < SIFT_HEADER_LEN) return -SIFT_ERR_WRITE;
    std::memcpy(out, kHeader, SIFT_HEADER_LEN);  // container.rs:124-126
    return SIFT_HEADER_LEN;
}

int64_t sift_read_header(const uint8_t* in, size_t avail) {
    if (!in || avail < SIFT_HEADER_LEN) return -SIFT_ERR_READ;                  // read_exact, :109
    if (std::memcmp(in, kHeader, SIFT_HEADER_LEN) != 0) return -SIFT_ERR_READ;  // InvalidData, :113
    return SIFT_HEADER_LEN;
}

int64_t sift_write_chunk(int device, const uint8_t chunk_type[4], const uint8_t* data,
                         uint64_t len, uint8_t* out, size_t cap) {
    if (len > 0xFFFFFFFFull) return -SIFT_ERR_WRITE_INVALID_CHUNK_SIZE;  // container.rs:69-71
    if (!chunk_type || (len && !data)) return -SIFT_ERR_INVALID_ARG;
    const uint64_t total = 12 + len;
    if (!out || cap < total) return -SIFT_ERR_WRITE;
    be32_put(out, uint32_t(len));          // :72
    std::memcpy(out + 4, chunk_type, 4);   // :73
    if (len) std::memcpy(out + 8, data, size_t(len));  // :74
    uint32_t crc = 0;
    int rc = chunk_crc(device, out + 4, len, &crc);    // :75
    if (rc != SIFT_OK) return -rc;
    be32_put(out + 8 + len, crc);
    return int64_t(total);
}

int64_t sift_read_chunk(int device, const uint8_t* in, size_t avail, uint8_t chunk_type[4],
                        uint64_t* data_off, uint32_t* data_len) {
    if (!in || !chunk_type || !data_off || !data_len) return -SIFT_ERR_INVALID_ARG;
    if (avail < 4) return -SIFT_ERR_READ;                                        // :48
    const uint32_t len = be32_get(in);
    if (len > SIFT_CHUNK_SIZE_LIMIT) return -SIFT_ERR_READ_INVALID_CHUNK_SIZE;   // :49-52
    if (avail < 8) return -SIFT_ERR_READ;                                        // :54
    std::memcpy(chunk_type, in + 4, 4);
    if (avail < 8 + uint64_t(len)) return -SIFT_ERR_READ;                        // :56
    if (avail < 12 + uint64_t(len)) return -SIFT_ERR_READ;                       // :60
    const uint32_t stored = be32_get(in + 8 + len);
    uint32_t crc = 0;
    int rc = chunk_crc(device, in + 4, len, &crc);
    if (rc != SIFT_OK) return -rc;
    if (stored != crc) return -SIFT_ERR_READ_CRC_MISMATCH;                       // :61-63
    *data_off = 8;
    *data_len = len;
    return int64_t(12 + uint64_t(len));
}

namespace {
const uint8_t kAutomergeChunk[4] = {'A', 'M', 'R', 'G'};  // document.rs:50
const uint8_t kEndChunk[4] = {'S', 'E', 'N', 'D'};        // document.rs:51
}  // namespace

int64_t sift_write_document(int device, const uint8_t* payload, uint64_t len, uint8_t* out, size_t cap) {
    int64_t n = sift_write_header(out, cap);  // document.rs:54
    if (n < 0) return n;
    size_t pos = size_t(n);
    n = sift_write_chunk(device, kAutomergeChunk, payload, len, out + pos, cap - pos);  // :55-56
    if (n < 0) return n;
    pos += size_t(n);
    n = sift_write_chunk(device, kEndChunk, nullptr, 0, out + pos, cap - pos);  // :57-58
    if (n < 0) return n;
    return int64_t(pos + size_t(n));
}

int64_t sift_read_document(int device, const uint8_t* in, size_t avail, uint64_t* payload_off,
                           uint32_t* payload_len, int* stage) {
    if (!payload_off || !payload_len) return -SIFT_ERR_INVALID_ARG;
    int dummy = 0;
    int* st = stage ? stage : &dummy;
    uint8_t type[4];
    uint64_t off = 0;
    uint32_t len = 0;
    *st = SIFT_DOC_HEADER;
    int64_t n = sift_read_header(in, avail);  // document.rs:63
    if (n < 0) return n;
    size_t pos = size_t(n);
    *st = SIFT_DOC_AMRG_CHUNK;
    n = sift_read_chunk(device, in + pos, avail - pos, type, &off, &len);  // :64
    if (n < 0) return n;
    *st = SIFT_DOC_AMRG_TYPE;
    if (sift_expect_type(type, kAutomergeChunk) != SIFT_OK) return -SIFT_ERR_READ;  // :65-67
    *payload_off = pos + off;
    *payload_len = len;
    pos += size_t(n);
    *st = SIFT_DOC_SEND_CHUNK;  // :69, decode_document, is the caller's
    n = sift_read_chunk(device, in + pos, avail - pos, type, &off, &len);  // :71
    if (n < 0) return n;
    *st = SIFT_DOC_SEND_TYPE;
    if (sift_expect_type(type, kEndChunk) != SIFT_OK) return -SIFT_ERR_READ;  // :72-74
    *st = SIFT_DOC_DONE;
    return int64_t(pos + size_t(n));
}

int sift_expect_type(const uint8_t chunk_type[4], const uint8_t expected[4]) {
    return std::memcmp(chunk_type, expected, 4) == 0 ? SIFT_OK : SIFT_ERR_READ;  // :86-95
}

}  // extern "C"
