/*
 * CPU oracle: adaptive-CDF binary range coder.  TEST INFRASTRUCTURE ONLY.
 *
 * Restates the published algorithm of `torchac` (module src/torchac of
 * github.com/trougnouf/L3C-PyTorch, which the reference calls at
 * src/compression/models/manynets_compressor.py:274,328,724,727 and
 * src/compression/models/abstract_model.py:344,347,423,431; no version is pinned by
 * the reference and the module is absent from its tree): Mark Nelson's 32-bit
 * low/high arithmetic coder with pending-bit carry resolution, one 16-bit CDF row
 * per symbol, the last CDF entry of a row read as 2^16.
 *
 * PARITY UNPINNED: no golden bytes exist for this coder (the reference only
 * self-checks encode -> decode equality, manynets_compressor.py:727-746).  It is
 * used as (i) a lossless round-trip check and (ii) the compressed-size comparator
 * for the GPU rANS coder.
 *
 * Two row sources are supported: materialised rows cdf[n_sym][Lp] (the tensor the
 * reference hands to torchac), or table[K][M][Lp] + idx[n_loc] (row of symbol
 * i = table[idx[i / M]][i % M]) which avoids the 514 B/symbol materialisation.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint8_t *buf;
    size_t cap, n;
    uint32_t acc;
    int nacc;
    int overflow;
} bitw_t;

static void bw_put(bitw_t *w, int bit) {
    w->acc = (w->acc << 1) | (uint32_t)(bit & 1);
    if (++w->nacc == 8) {
        if (w->n < w->cap) w->buf[w->n] = (uint8_t)w->acc; else w->overflow = 1;
        w->n++;
        w->acc = 0;
        w->nacc = 0;
    }
}

static void bw_put_with_pending(bitw_t *w, int bit, uint64_t *pending) {
    bw_put(w, bit);
    while (*pending) { bw_put(w, !bit); (*pending)--; }
}

static void bw_flush(bitw_t *w) {
    while (w->nacc) bw_put(w, 0);
}

typedef struct {
    const uint16_t *rows;    /* materialised rows or NULL */
    const uint16_t *table;   /* [K][M][Lp] */
    const int32_t *idx;      /* [n_loc] */
    int M, Lp;
} rowsrc_t;

static inline const uint16_t *row_of(const rowsrc_t *r, size_t i) {
    if (r->rows) return r->rows + i * (size_t)r->Lp;
    size_t loc = i / (size_t)r->M, c = i % (size_t)r->M;
    return r->table + ((size_t)r->idx[loc] * r->M + c) * (size_t)r->Lp;
}

static inline uint32_t cdf_at(const uint16_t *row, int s, int Lp) {
    return (s == Lp - 1) ? 0x10000u : (uint32_t)row[s];
}

static long rc_encode(const rowsrc_t *src, const int16_t *sym, size_t n_sym, uint8_t *out, size_t cap) {
    bitw_t w = {out, cap, 0, 0, 0, 0};
    uint32_t low = 0, high = 0xFFFFFFFFu;
    uint64_t pending = 0;
    const int Lp = src->Lp;
    for (size_t i = 0; i < n_sym; ++i) {
        const uint16_t *row = row_of(src, i);
        int s = sym[i];
        if (s < 0 || s > Lp - 2) return -2;
        uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        uint32_t c_lo = cdf_at(row, s, Lp), c_hi = cdf_at(row, s + 1, Lp);
        if (c_hi <= c_lo) return -3;
        high = (uint32_t)((uint64_t)low - 1 + ((span * c_hi) >> 16));
        low = (uint32_t)((uint64_t)low + ((span * c_lo) >> 16));
        for (;;) {
            if (high < 0x80000000u) {
                bw_put_with_pending(&w, 0, &pending);
            } else if (low >= 0x80000000u) {
                bw_put_with_pending(&w, 1, &pending);
            } else if (low >= 0x40000000u && high < 0xC0000000u) {
                pending++;
                low = (low << 1) & 0x7FFFFFFFu;
                high = (high << 1) | 0x80000001u;
                continue;
            } else {
                break;
            }
            low <<= 1;
            high = (high << 1) | 1u;
        }
    }
    pending++;
    bw_put_with_pending(&w, low < 0x40000000u ? 0 : 1, &pending);
    bw_flush(&w);
    if (w.overflow) return -1;
    return (long)w.n;
}

typedef struct {
    const uint8_t *buf;
    size_t n, pos;
    int bit;
} bitr_t;

static inline uint32_t br_get(bitr_t *r) {
    if (r->pos >= r->n) return 0;
    uint32_t b = (r->buf[r->pos] >> (7 - r->bit)) & 1u;
    if (++r->bit == 8) { r->bit = 0; r->pos++; }
    return b;
}

static int rc_decode(const rowsrc_t *src, const uint8_t *in, size_t nbytes, int16_t *sym, size_t n_sym) {
    bitr_t r = {in, nbytes, 0, 0};
    uint32_t low = 0, high = 0xFFFFFFFFu, value = 0;
    const int Lp = src->Lp;
    for (int i = 0; i < 32; ++i) value = (value << 1) | br_get(&r);
    for (size_t i = 0; i < n_sym; ++i) {
        const uint16_t *row = row_of(src, i);
        uint64_t span = (uint64_t)high - (uint64_t)low + 1;
        uint64_t count = ((((uint64_t)value - (uint64_t)low + 1) << 16) - 1) / span;
        int lo = 0, hi = Lp - 1;                 /* invariant: cdf[lo] <= count < cdf[hi] */
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (cdf_at(row, mid, Lp) <= count) lo = mid; else hi = mid;
        }
        int s = lo;
        sym[i] = (int16_t)s;
        uint32_t c_lo = cdf_at(row, s, Lp), c_hi = cdf_at(row, s + 1, Lp);
        high = (uint32_t)((uint64_t)low - 1 + ((span * c_hi) >> 16));
        low = (uint32_t)((uint64_t)low + ((span * c_lo) >> 16));
        for (;;) {
            if (high < 0x80000000u) {
                /* nothing to subtract */
            } else if (low >= 0x80000000u) {
                /* top bit set in all three; the shift discards it */
            } else if (low >= 0x40000000u && high < 0xC0000000u) {
                value -= 0x40000000u;
                low -= 0x40000000u;
                high -= 0x40000000u;
            } else {
                break;
            }
            low <<= 1;
            high = (high << 1) | 1u;
            value = (value << 1) | br_get(&r);
        }
    }
    return 0;
}

/* ---- exported entry points (ctypes) ------------------------------------------------ */

long oracle_range_encode_rows(const uint16_t *rows, int Lp, const int16_t *sym, size_t n_sym,
                              uint8_t *out, size_t cap) {
    rowsrc_t s = {rows, NULL, NULL, 1, Lp};
    return rc_encode(&s, sym, n_sym, out, cap);
}

int oracle_range_decode_rows(const uint16_t *rows, int Lp, const uint8_t *in, size_t nbytes,
                             int16_t *sym, size_t n_sym) {
    rowsrc_t s = {rows, NULL, NULL, 1, Lp};
    return rc_decode(&s, in, nbytes, sym, n_sym);
}

long oracle_range_encode_table(const uint16_t *table, int M, int Lp, const int32_t *idx,
                               const int16_t *sym, size_t n_sym, uint8_t *out, size_t cap) {
    rowsrc_t s = {NULL, table, idx, M, Lp};
    return rc_encode(&s, sym, n_sym, out, cap);
}

int oracle_range_decode_table(const uint16_t *table, int M, int Lp, const int32_t *idx,
                              const uint8_t *in, size_t nbytes, int16_t *sym, size_t n_sym) {
    rowsrc_t s = {NULL, table, idx, M, Lp};
    return rc_decode(&s, in, nbytes, sym, n_sym);
}
