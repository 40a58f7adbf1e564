/*
 * sift_oracle.c -- CPU ORACLE. TEST INFRASTRUCTURE ONLY (see sift_oracle.h).
 *
 * Each function restates one piece of the reference and cites it.  The CRC
 * loop is deliberately the reference's own byte-at-a-time table walk, not a
 * faster variant: it is also what bench.py times as the CPU baseline.
 */
#include "sift_oracle.h"

#include <string.h>

/* ---- crc.rs:2-20: CRC_TABLE ------------------------------------------- */
static uint32_t g_table[256];
static int g_table_ready;

static void build_table(void) {
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int i = 0; i < 8; i++)
            c = (c & 1u) ? (0xedb88320u ^ (c >> 1)) : (c >> 1);
        g_table[n] = c;
    }
    g_table_ready = 1;
}

const uint32_t *oracle_crc_table(void) {
    if (!g_table_ready) build_table();
    return g_table;
}

/* ---- crc.rs:28-34: compute -------------------------------------------- */
uint32_t oracle_crc_compute(uint32_t crc, const uint8_t *buf, size_t len) {
    const uint32_t *t = oracle_crc_table();
    uint32_t c = crc;
    for (size_t i = 0; i < len; i++)
        c = t[(c ^ (uint32_t)buf[i]) & 0xffu] ^ (c >> 8);
    return c;
}

/* ---- crc.rs:36-53: Digest --------------------------------------------- */
void oracle_digest_new(oracle_digest *d) { d->value = 0xffffffffu; }

void oracle_digest_update(oracle_digest *d, const uint8_t *buf, size_t len) {
    d->value = oracle_crc_compute(d->value, buf, len);
}

uint32_t oracle_digest_finalize(const oracle_digest *d) { return d->value ^ 0xffffffffu; }

/* ---- container.rs:79-84: Chunk::compute_crc --------------------------- */
uint32_t oracle_chunk_crc(const uint8_t chunk_type[4], const uint8_t *data, size_t len) {
    oracle_digest d;
    oracle_digest_new(&d);
    oracle_digest_update(&d, chunk_type, 4);
    oracle_digest_update(&d, data, len);
    return oracle_digest_finalize(&d);
}

/* ---- container.rs:20-32: read_u32 / write_u32 (big endian) ------------ */
static uint32_t be32_get(const uint8_t *p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

static void be32_put(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
}

/* ---- container.rs:104: HEADER ----------------------------------------- */
static const uint8_t k_header[ORACLE_HEADER_LEN] = {0x89, 0x53, 0x49, 0x46, 0x54,
                                                    0x0D, 0x0A, 0x1A, 0x0A};

/* ---- container.rs:124-126: write_header ------------------------------- */
int64_t oracle_write_header(uint8_t *out, size_t cap) {
    if (cap < ORACLE_HEADER_LEN) return -ORACLE_ERR_WRITE;
    memcpy(out, k_header, ORACLE_HEADER_LEN);
    return ORACLE_HEADER_LEN;
}

/* ---- container.rs:107-121: read_header -------------------------------- */
int64_t oracle_read_header(const uint8_t *in, size_t avail) {
    if (avail < ORACLE_HEADER_LEN) return -ORACLE_ERR_READ; /* read_exact fails */
    if (memcmp(in, k_header, ORACLE_HEADER_LEN) != 0) return -ORACLE_ERR_READ; /* InvalidData */
    return ORACLE_HEADER_LEN;
}

/* ---- container.rs:68-77: Chunk::write --------------------------------- */
int64_t oracle_write_chunk(const uint8_t chunk_type[4], const uint8_t *data, uint64_t len,
                           uint8_t *out, size_t cap) {
    if (len > 0xffffffffull) return -ORACLE_ERR_WRITE_INVALID_CHUNK_SIZE; /* u32::try_from */
    uint64_t total = 4 + 4 + len + 4;
    if (cap < total) return -ORACLE_ERR_WRITE;
    be32_put(out, (uint32_t)len);
    memcpy(out + 4, chunk_type, 4);
    if (len) memcpy(out + 8, data, (size_t)len);
    be32_put(out + 8 + len, oracle_chunk_crc(chunk_type, data, (size_t)len));
    return (int64_t)total;
}

/* ---- container.rs:47-66: Chunk::read ---------------------------------- */
int64_t oracle_read_chunk(const uint8_t *in, size_t avail, uint8_t chunk_type[4],
                          uint64_t *data_off, uint32_t *data_len) {
    if (avail < 4) return -ORACLE_ERR_READ;
    uint32_t len = be32_get(in);
    if (len > ORACLE_CHUNK_SIZE_LIMIT) return -ORACLE_ERR_READ_INVALID_CHUNK_SIZE;
    if (avail < 8) return -ORACLE_ERR_READ;
    memcpy(chunk_type, in + 4, 4);
    if (avail < 8 + (uint64_t)len) return -ORACLE_ERR_READ;
    if (avail < 8 + (uint64_t)len + 4) return -ORACLE_ERR_READ;
    uint32_t crc = be32_get(in + 8 + len);
    if (crc != oracle_chunk_crc(chunk_type, in + 8, len)) return -ORACLE_ERR_READ_CRC_MISMATCH;
    *data_off = 8;
    *data_len = len;
    return (int64_t)(8 + (uint64_t)len + 4);
}

/* ---- document.rs:50-51: chunk types ------------------------------------ */
static const uint8_t k_amrg[4] = {'A', 'M', 'R', 'G'};
static const uint8_t k_send[4] = {'S', 'E', 'N', 'D'};

/* ---- document.rs:53-60: write_document (payload = encode_document's bytes) */
int64_t oracle_write_document(const uint8_t *payload, uint64_t len, uint8_t *out, size_t cap) {
    int64_t n = oracle_write_header(out, cap);                       /* :54 */
    if (n < 0) return n;
    size_t pos = (size_t)n;
    n = oracle_write_chunk(k_amrg, payload, len, out + pos, cap - pos); /* :55-56 */
    if (n < 0) return n;
    pos += (size_t)n;
    n = oracle_write_chunk(k_send, NULL, 0, out + pos, cap - pos);   /* :57-58 */
    if (n < 0) return n;
    return (int64_t)(pos + (size_t)n);
}

/* ---- document.rs:62-77: read_document ---------------------------------- */
int64_t oracle_read_document(const uint8_t *in, size_t avail, uint64_t *payload_off,
                             uint32_t *payload_len, int *stage) {
    uint8_t type[4];
    uint64_t off = 0;
    uint32_t len = 0;
    *stage = 0;
    int64_t n = oracle_read_header(in, avail);                       /* :63 */
    if (n < 0) return n;
    size_t pos = (size_t)n;
    *stage = 1;
    n = oracle_read_chunk(in + pos, avail - pos, type, &off, &len);  /* :64 */
    if (n < 0) return n;
    *stage = 2;
    if (memcmp(type, k_amrg, 4) != 0) return -ORACLE_ERR_READ;       /* :65-67 */
    *payload_off = pos + off;
    *payload_len = len;
    pos += (size_t)n;
    *stage = 3;                                                      /* :69 decode_document is the caller's */
    n = oracle_read_chunk(in + pos, avail - pos, type, &off, &len);  /* :71 */
    if (n < 0) return n;
    *stage = 4;
    if (memcmp(type, k_send, 4) != 0) return -ORACLE_ERR_READ;       /* :72-74 */
    *stage = 5;
    return (int64_t)(pos + (size_t)n);
}

/* ---- bench helper ------------------------------------------------------ */
uint32_t oracle_crc_repeat(const uint8_t *buf, size_t len, int passes) {
    uint32_t c = 0xffffffffu;
    for (int p = 0; p < passes; p++) c = oracle_crc_compute(c, buf, len);
    return c;
}
