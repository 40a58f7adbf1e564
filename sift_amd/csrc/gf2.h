// gf2.h -- arithmetic in GF(2)[x] / P(x) for CRC-32/ISO-HDLC, shared by host
// and device code.
//
// Representation ("reflected", the one the reference's register uses,
// sift-persist/src/crc.rs:2-34): a uint32_t v stands for the polynomial
//     sum_k bit_k(v) * x^(31-k)
// so bit 31 is x^0 and 0x80000000 is the constant 1.  The reference's table
// step  c = TABLE[(c ^ byte) & 0xff] ^ (c >> 8)  (crc.rs:31) is then
//     c <- (c + byte) * x^8  mod P
// and a little-endian 32-bit load w of four message bytes enters as
//     c <- (c + w) * x^32 mod P.
// Everything the GPU path does follows from linearity of that map.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define SIFT_HD __host__ __device__
#else
#define SIFT_HD
#endif

namespace sift {

// Reflected CRC-32 polynomial, crc.rs:10.
constexpr uint32_t kPoly = 0xEDB88320u;
// The polynomial "1".
constexpr uint32_t kOne = 0x80000000u;

// b * x mod P.
SIFT_HD inline uint32_t gf2_mulx(uint32_t b) {
    return (b >> 1) ^ (kPoly & (0u - (b & 1u)));
}

// a * b mod P.  32 shift/xor steps; branch-free so every lane of a wave runs
// the same instruction stream.
SIFT_HD inline uint32_t gf2_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 31; i >= 0; --i) {
        p ^= b & (0u - ((a >> i) & 1u));
        b = gf2_mulx(b);
    }
    return p;
}

// x^(2^k) mod P for k = 0..31.  P is primitive, so x^(2^32) = x and the table
// is periodic in k with period 32.
struct X2nTable {
    uint32_t v[32];
};

inline X2nTable make_x2n_table() {
    X2nTable t;
    uint32_t p = kOne >> 1;  // x^1
    for (int k = 0; k < 32; ++k) {
        t.v[k] = p;
        p = gf2_mulmod(p, p);
    }
    return t;
}

// x^n mod P for any 64-bit n (n counts bits of message, so 8 * bytes).
inline uint32_t gf2_xpow(const X2nTable& t, uint64_t n) {
    uint32_t p = kOne;
    for (int k = 0; n; n >>= 1, ++k)
        if (n & 1u) p = gf2_mulmod(t.v[k & 31], p);
    return p;
}

// One byte of the reference's recurrence without a table (8 x mulx):
// c <- (c + byte) * x^8.  Used for the < 16 trailing bytes and for tiny
// inputs, on the device, by a single lane.
SIFT_HD inline uint32_t crc_step_byte(uint32_t c, uint32_t byte) {
    c ^= byte;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 8; ++i) c = gf2_mulx(c);
    return c;
}

}  // namespace sift
