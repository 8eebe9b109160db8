// sfsn_pack.cpp -- host-side weight packing for the int8 x 3-digit products (see include/sfsn.h).
// Pure integer / bit work on host memory; needs no device.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "sfsn.h"

#ifndef SFSN_SRC_HASH
#define SFSN_SRC_HASH "unknown"
#endif
// first 16 hex digits of sha256 over the library's sources (Makefile: SRCS), checked by the Python binding at load time
extern "C" const char* sfsn_source_hash(void) { return SFSN_SRC_HASH; }

static inline int tiles16(int n) { return (n + 15) / 16; }
static inline int steps64(int k) { return (k + 63) / 64; }

extern "C" size_t sfsn_w3_packed_bytes(int n_out, int k_in) {
    if (n_out <= 0 || k_in <= 0) return 0;
    return (size_t)3 * tiles16(n_out) * steps64(k_in) * 1024;
}

extern "C" int sfsn_w3_padded_rows(int n_out) { return n_out <= 0 ? 0 : tiles16(n_out) * 16; }

// Largest |q| three balanced base-256 digits in [-128,127] can hold with every digit still in range.
static const double QMAX = 127.0 * 65536.0 + 127.0 * 256.0 + 127.0;  // 8355711

// bits = 24: the exact mode (three digits).  bits = 16: the 16-bit-weight fast mode -- every weight is rounded to 16 significant
// bits of its row's fixed-point grid (|W - W~| <= 128 dq[n]), stored in the two upper digits; the least-significant digit plane
// is all zero, so kernels that know it may skip a third of their matrix instructions (the others just add zeros).
extern "C" int sfsn_w3_pack_bits(const float* w, int n_out, int k_in, int bits, int8_t* packed, float* dq) {
    if (!w || !packed || !dq || n_out <= 0 || k_in <= 0 || (bits != 24 && bits != 16)) return SFSN_EINVAL;
    const int NT = tiles16(n_out), KS = steps64(k_in);
    memset(packed, 0, sfsn_w3_packed_bytes(n_out, k_in));
    for (int n = 0; n < NT * 16; ++n) dq[n] = 0.0f;
    for (int n = 0; n < n_out; ++n) {
        const float* row = w + (size_t)n * k_in;
        double amax = 0.0;
        for (int k = 0; k < k_in; ++k) {
            if (!isfinite(row[k])) return SFSN_EINVAL;
            const double a = fabs((double)row[k]);
            if (a > amax) amax = a;
        }
        // power-of-two row scale 2^e with |W| / 2^e * 2^23 <= QMAX
        int e = 0;
        if (amax > 0.0) {
            e = (int)ceil(log2(amax * (8388608.0 / QMAX)));
            while (ldexp(amax, 23 - e) > QMAX) ++e;                 // guard log2 rounding
            while (e > -126 && ldexp(amax, 23 - (e - 1)) <= QMAX) --e;
        }
        if (e - 23 < -126 || e > 100) return SFSN_EUNSUPPORTED;      // dq must stay a normal fp32 number
        dq[n] = (float)ldexp(1.0, e - 23);
        const int nt = n >> 4, nn = n & 15;
        for (int k = 0; k < k_in; ++k) {
            long q = lrint(ldexp((double)row[k], 23 - e));            // exact scaling, round to nearest even
            if (bits == 16) {
                long q16 = lrint(ldexp((double)row[k], 15 - e));      // round to the 16-bit grid (nearest even)
                if (q16 > 32639) q16 = 32639;                         // (the row maximum may round up past the digit range)
                if (q16 < -32639) q16 = -32639;
                q = q16 * 256;
            }
            const int d0 = (int)(((q + 128) & 255) - 128);
            const long q1 = (q - d0) >> 8;  // exact: q - d0 is a multiple of 256
            const int d1 = (int)(((q1 + 128) & 255) - 128);
            const long d2 = (q1 - d1) >> 8;
            if (d2 < -128 || d2 > 127) return SFSN_EINVAL;            // cannot happen given QMAX
            const int ks = k >> 6, kk = k & 63, lane = ((kk >> 4) << 4) | nn, byte = kk & 15;
            const int digits[3] = {d0, d1, (int)d2};
            for (int d = 0; d < 3; ++d) {
                const size_t off = ((((size_t)d * NT + nt) * KS + ks) * 64 + lane) * 16 + byte;
                packed[off] = (int8_t)digits[d];
            }
        }
    }
    return SFSN_OK;
}

extern "C" int sfsn_w3_pack(const float* w, int n_out, int k_in, int8_t* packed, float* dq) {
    return sfsn_w3_pack_bits(w, n_out, k_in, 24, packed, dq);
}

extern "C" int sfsn_w3_unpack(const int8_t* packed, const float* dq, int n_out, int k_in, float* w) {
    if (!w || !packed || !dq || n_out <= 0 || k_in <= 0) return SFSN_EINVAL;
    const int NT = tiles16(n_out), KS = steps64(k_in);
    for (int n = 0; n < n_out; ++n)
        for (int k = 0; k < k_in; ++k) {
            const int nt = n >> 4, nn = n & 15, ks = k >> 6, kk = k & 63, lane = ((kk >> 4) << 4) | nn, byte = kk & 15;
            long q = 0;
            for (int d = 2; d >= 0; --d) {
                const size_t off = ((((size_t)d * NT + nt) * KS + ks) * 64 + lane) * 16 + byte;
                q = q * 256 + packed[off];
            }
            w[(size_t)n * k_in + k] = (float)((double)q * (double)dq[n]);
        }
    return SFSN_OK;
}
