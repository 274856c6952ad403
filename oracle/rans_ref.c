/*
 * CPU oracle for the MRS1 container ("manypriors rANS streams").  TEST INFRASTRUCTURE ONLY.
 *
 * The container is this project's own format (the reference's coder is torchac's
 * range coder, whose bytes are unpinned -- see range_coder.c), so this file is the
 * format's scalar specification: the GPU encoder must produce exactly these bytes
 * and the GPU decoder must accept them.
 *
 * Layout (little endian):
 *   u32 magic 'MRS1' | u32 n_streams | u32 n_symbols | u32 syms_per_stream
 *   u16 len_words[n_streams]        payload length of each stream, in 16-bit words
 *   payload                         streams back to back, each 2-byte aligned:
 *                                   u16 state_lo, u16 state_hi, then renormalisation
 *                                   words in the order the decoder consumes them
 * Stream j codes symbols [j*sps, min((j+1)*sps, n_symbols)) of the location-major /
 * channel-minor symbol array (manynets_compressor.py:264-269), each with the CDF row
 * table[idx[i / M]][i % M][:] (manynets_compressor.py:252-262), the last row entry
 * read as 2^16.
 *
 * rANS: 32-bit state in [2^16, 2^32), 16-bit probabilities, 16-bit renormalisation.
 *   encode (symbols in reverse):  if (x >= freq << 16) { emit(x & 0xffff); x >>= 16; }
 *                                 x = ((x / freq) << 16) + x % freq + start
 *   decode:  slot = x & 0xffff; s = row search; x = freq * (x >> 16) + slot - start;
 *            if (x < 2^16) x = (x << 16) | next_word
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MRS1_MAGIC 0x3153524Du
#define RANS_L (1u << 16)

static inline uint32_t cdf_at(const uint16_t *row, int s, int Lp) {
    return (s == Lp - 1) ? 0x10000u : (uint32_t)row[s];
}

static inline const uint16_t *row_of(const uint16_t *table, int M, int Lp, const int32_t *idx, size_t i) {
    return table + ((size_t)idx[i / (size_t)M] * M + (i % (size_t)M)) * (size_t)Lp;
}

size_t oracle_mrs1_header_bytes(uint32_t n_streams) { return 16 + 2 * (size_t)n_streams; }

/* returns total bytes, or <0 on error (-1 capacity, -2 symbol range, -3 zero frequency, -4 stream too long) */
long oracle_mrs1_encode(const uint16_t *table, int M, int Lp, const int32_t *idx, const int16_t *sym,
                        uint32_t n_symbols, uint32_t n_streams, uint32_t sps, uint8_t *out, size_t cap) {
    if ((uint64_t)n_streams * sps < n_symbols || sps + 2 > 65535u) return -4;
    size_t hdr = oracle_mrs1_header_bytes(n_streams);
    if (cap < hdr) return -1;
    uint32_t h4[4] = {MRS1_MAGIC, n_streams, n_symbols, sps};
    memcpy(out, h4, 16);
    uint16_t *tmp = (uint16_t *)malloc(((size_t)sps + 2) * sizeof(uint16_t));
    size_t off = hdr;
    for (uint32_t j = 0; j < n_streams; ++j) {
        size_t b = (size_t)j * sps, e = b + sps;
        if (b > n_symbols) b = n_symbols;
        if (e > n_symbols) e = n_symbols;
        size_t pos = (size_t)sps + 2;
        uint32_t x = RANS_L;
        for (size_t i = e; i-- > b;) {
            const uint16_t *row = row_of(table, M, Lp, idx, i);
            int s = sym[i];
            if (s < 0 || s > Lp - 2) { free(tmp); return -2; }
            uint32_t start = cdf_at(row, s, Lp), end = cdf_at(row, s + 1, Lp);
            if (end <= start) { free(tmp); return -3; }
            uint32_t freq = end - start;
            if ((uint64_t)x >= ((uint64_t)freq << 16)) { tmp[--pos] = (uint16_t)(x & 0xffffu); x >>= 16; }
            x = ((x / freq) << 16) + (x % freq) + start;
        }
        tmp[--pos] = (uint16_t)(x >> 16);
        tmp[--pos] = (uint16_t)(x & 0xffffu);
        size_t nw = (size_t)sps + 2 - pos;
        uint16_t nw16 = (uint16_t)nw;
        memcpy(out + 16 + 2 * (size_t)j, &nw16, 2);
        if (off + 2 * nw > cap) { free(tmp); return -1; }
        memcpy(out + off, tmp + pos, 2 * nw);
        off += 2 * nw;
    }
    free(tmp);
    return (long)off;
}

/* returns 0, or <0 on malformed input */
int oracle_mrs1_decode(const uint16_t *table, int M, int Lp, const int32_t *idx, const uint8_t *in,
                       size_t nbytes, int16_t *sym_out, uint32_t n_symbols_expected) {
    if (nbytes < 16) return -1;
    uint32_t h4[4];
    memcpy(h4, in, 16);
    if (h4[0] != MRS1_MAGIC) return -2;
    uint32_t n_streams = h4[1], n_symbols = h4[2], sps = h4[3];
    if (n_symbols != n_symbols_expected) return -3;
    size_t off = oracle_mrs1_header_bytes(n_streams);
    if (nbytes < off) return -1;
    for (uint32_t j = 0; j < n_streams; ++j) {
        uint16_t nw;
        memcpy(&nw, in + 16 + 2 * (size_t)j, 2);
        if (off + 2 * (size_t)nw > nbytes || nw < 2) return -4;
        const uint8_t *p = in + off, *pend = p + 2 * (size_t)nw;
        uint16_t w0, w1;
        memcpy(&w0, p, 2); memcpy(&w1, p + 2, 2); p += 4;
        uint32_t x = (uint32_t)w0 | ((uint32_t)w1 << 16);
        size_t b = (size_t)j * sps, e = b + sps;
        if (b > n_symbols) b = n_symbols;
        if (e > n_symbols) e = n_symbols;
        for (size_t i = b; i < e; ++i) {
            const uint16_t *row = row_of(table, M, Lp, idx, i);
            uint32_t slot = x & 0xffffu;
            int lo = 0, hi = Lp - 1;
            while (hi - lo > 1) {
                int mid = (lo + hi) >> 1;
                if (cdf_at(row, mid, Lp) <= slot) lo = mid; else hi = mid;
            }
            uint32_t start = cdf_at(row, lo, Lp), freq = cdf_at(row, lo + 1, Lp) - start;
            sym_out[i] = (int16_t)lo;
            x = freq * (x >> 16) + slot - start;
            if (x < RANS_L) {
                if (p + 2 > pend) return -5;
                uint16_t w; memcpy(&w, p, 2); p += 2;
                x = (x << 16) | w;
            }
        }
        if (x != RANS_L || p != pend) return -6;
        off += 2 * (size_t)nw;
    }
    return 0;
}

/* Ideal code length in bits: sum_i -log2((c_hi - c_lo) / 2^16), computed by the caller in
 * floating point from the (start, freq) pairs this returns. */
int oracle_symbol_freqs(const uint16_t *table, int M, int Lp, const int32_t *idx, const int16_t *sym,
                        uint32_t n_symbols, uint32_t *freq_out) {
    for (size_t i = 0; i < n_symbols; ++i) {
        const uint16_t *row = row_of(table, M, Lp, idx, i);
        int s = sym[i];
        if (s < 0 || s > Lp - 2) return -2;
        freq_out[i] = cdf_at(row, s + 1, Lp) - cdf_at(row, s, Lp);
    }
    return 0;
}
