// MRS1 entropy coder: tile-interleaved rANS streams over the per-location CDF rows
// table[idx[loc]][c][:].  Replaces torchac.encode_cdf / decode_cdf
// (src/compression/models/manynets_compressor.py:274,328,724,727).  The byte format is specified by
// the scalar implementation in oracle/rans_ref.c; the encoder here must produce exactly those bytes.
//
// Encoder = three launches.
//   1. symbol_params: one thread per symbol, fully parallel: gathers (start, freq) of the symbol from
//      the int16 table (2 B symbol + two 2 B table entries read, 4 B written) and stores them
//      transposed, params[r][stream], so that step 2 reads coalesced.
//   2. encode_streams: one LANE per stream (rANS is sequential within a stream).  Walks the stream's
//      symbols backwards, emits 16-bit words backwards into the stream's scratch slot.
//   3. compact: per image, exclusive scan of the stream lengths, header, and a copy of every stream's
//      words to its final offset.
// Decoder = one launch.  One 16-LANE GROUP per stream: lane t holds entries 16t..16t+15 of the
// symbol's 256-entry CDF row (32 B per lane, 512 B coalesced per group), counts entries <= slot and
// the group sums the counts -> symbol, with the next symbol's row already in flight.  The row
// address depends only on (idx, channel), never on decoded data, which is what makes the prefetch
// possible.
#include "mpc_internal.h"

namespace {

constexpr uint32_t MRS1_MAGIC = 0x3153524Du;
constexpr uint32_t RANS_L = 1u << 16;

struct Geometry {
    int n_images;
    int64_t n_loc;          // locations per image
    int M, L, K;
    int n_streams;          // per image
    int locs_per_stream;
    uint32_t sps;           // symbols per stream = locs_per_stream * M
    uint32_t n_symbols;     // per image = n_loc * M
    uint32_t cap_words;     // sps + 2
};

__global__ __launch_bounds__(256) void symbol_params_kernel(Geometry g, const uint16_t *__restrict__ table,
                                                             const int32_t *__restrict__ idx,
                                                             const int16_t *__restrict__ sym,
                                                             uint32_t *__restrict__ params,
                                                             int32_t *__restrict__ status) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)g.n_images * g.n_symbols;
    if (i >= total) return;
    int img = (int)(i / g.n_symbols);
    uint32_t local = (uint32_t)(i % g.n_symbols);
    int64_t loc = (int64_t)img * g.n_loc + local / g.M;
    int c = (int)(local % g.M);
    int k = idx[loc];
    int s = sym[i];
    bool bad = (k < 0) | (k >= g.K) | (s < 0) | (s > g.L - 2);
    k = bad ? 0 : k;
    s = bad ? 0 : s;
    const uint16_t *row = table + ((int64_t)k * g.M + c) * g.L;
    uint32_t start = row[s];
    uint32_t end = (s == g.L - 2) ? 0x10000u : (uint32_t)row[s + 1];
    if (end <= start) bad = true;
    uint32_t freq = bad ? 1u : end - start;
    if (bad) { start = 0; atomicMin(&status[img], -10); }
    uint32_t stream = local / g.sps, r = local % g.sps;
    int64_t G = (int64_t)g.n_images * g.n_streams;
    params[(int64_t)r * G + (int64_t)img * g.n_streams + stream] = start | (freq << 16);
}

__global__ __launch_bounds__(64) void encode_streams_kernel(Geometry g, const uint32_t *__restrict__ params,
                                                             uint16_t *__restrict__ scratch,
                                                             uint32_t *__restrict__ lens) {
    int64_t G = (int64_t)g.n_images * g.n_streams;
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= G) return;
    uint32_t stream = (uint32_t)(gid % g.n_streams);
    uint64_t b = (uint64_t)stream * g.sps, e = b + g.sps;
    if (b > g.n_symbols) b = g.n_symbols;
    if (e > g.n_symbols) e = g.n_symbols;
    uint32_t n = (uint32_t)(e - b);
    uint16_t *buf = scratch + gid * g.cap_words;
    uint32_t pos = g.cap_words;
    uint32_t x = RANS_L;
    const uint32_t *p = params + gid;
    auto step = [&](uint32_t sf) {
        uint32_t start = sf & 0xffffu, freq = sf >> 16;
        if (x >= (freq << 16)) { buf[--pos] = (uint16_t)(x & 0xffffu); x >>= 16; }
        uint32_t q = x / freq;
        x = (q << 16) + (x - q * freq) + start;
    };
    // symbols are consumed backwards; the (start,freq) words of the next 8 are loaded while the current 8
    // are coded (the addresses do not depend on the coder state)
    constexpr int U = 8;
    uint32_t r = n;
    uint32_t cur[U], nxt[U];
    if (r >= U) {
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = p[(int64_t)(r - 1 - u) * G];
        while (r >= 2 * U) {
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[u] = p[(int64_t)(r - U - 1 - u) * G];
#pragma unroll
            for (int u = 0; u < U; ++u) step(cur[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = nxt[u];
            r -= U;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) step(cur[u]);
        r -= U;
    }
    while (r-- > 0) step(p[(int64_t)r * G]);
    buf[--pos] = (uint16_t)(x >> 16);
    buf[--pos] = (uint16_t)(x & 0xffffu);
    lens[gid] = g.cap_words - pos;
}

// One block per image: scan stream lengths, write header and per-stream byte offsets.
__global__ __launch_bounds__(256) void header_kernel(Geometry g, const uint32_t *__restrict__ lens,
                                                      uint32_t *__restrict__ offs, uint8_t *__restrict__ out,
                                                      size_t out_stride, uint64_t *__restrict__ nbytes,
                                                      int32_t *__restrict__ status) {
    __shared__ uint32_t part[256];
    __shared__ uint32_t base_sh;
    const int img = blockIdx.x, t = threadIdx.x;
    const uint32_t *l = lens + (int64_t)img * g.n_streams;
    uint32_t *o = offs + (int64_t)img * g.n_streams;
    uint8_t *dst = out + (size_t)img * out_stride;
    const uint32_t hdr = 16 + 2 * (uint32_t)g.n_streams;
    if (t == 0) base_sh = hdr;
    __syncthreads();
    for (int j0 = 0; j0 < g.n_streams; j0 += 256) {
        int j = j0 + t;
        uint32_t v = (j < g.n_streams) ? 2 * l[j] : 0;
        part[t] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {          // Hillis-Steele inclusive scan
            uint32_t a = (t >= off) ? part[t - off] : 0;
            __syncthreads();
            part[t] += a;
            __syncthreads();
        }
        uint32_t base = base_sh;
        if (j < g.n_streams) o[j] = base + part[t] - v;
        __syncthreads();
        if (t == 255) base_sh = base + part[255];
        __syncthreads();
    }
    uint32_t total = base_sh;
    bool fits = (size_t)total <= out_stride;
    if (t == 0) {
        nbytes[img] = total;
        if (!fits) atomicMin(&status[img], -11);
    }
    if (!fits) return;
    if (t < 4) {
        uint32_t h4[4] = {MRS1_MAGIC, (uint32_t)g.n_streams, g.n_symbols, g.sps};
        ((uint32_t *)dst)[t] = h4[t];
    }
    for (int j = t; j < g.n_streams; j += 256) ((uint16_t *)(dst + 16))[j] = (uint16_t)l[j];
}

// One block per stream: copy its words to the final position.
__global__ __launch_bounds__(256) void copy_streams_kernel(Geometry g, const uint16_t *__restrict__ scratch,
                                                            const uint32_t *__restrict__ lens,
                                                            const uint32_t *__restrict__ offs,
                                                            uint8_t *__restrict__ out, size_t out_stride,
                                                            const uint64_t *__restrict__ nbytes) {
    int64_t gid = blockIdx.x;
    int img = (int)(gid / g.n_streams);
    if (nbytes[img] > out_stride) return;
    uint32_t nw = lens[gid];
    const uint16_t *src = scratch + gid * g.cap_words + (g.cap_words - nw);
    uint16_t *dst = (uint16_t *)(out + (size_t)img * out_stride + offs[gid]);
    for (uint32_t i = threadIdx.x; i < nw; i += 256) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------------
// Decoder
// ---------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(2))) Row16 {
    uint16_t v[16];
};

constexpr int GROUP = 16;                  // lanes per stream
constexpr int GROUPS_PER_BLOCK = 256 / GROUP;

__device__ __forceinline__ uint32_t group_sum(uint32_t v) {
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}

// Counts entries of the row that are <= slot.  Generic in S = L-1: the row is scanned in segments of
// 256 entries, lane t looking at entries seg*256 + 16t .. +15; entries at index >= S count as 2^16.
__device__ __forceinline__ uint32_t count_le(const uint16_t *row, int S, int t, uint32_t slot) {
    uint32_t cnt = 0;
    for (int seg = 0; seg < S; seg += 256) {
        int e0 = seg + 16 * t;
        if (e0 + 16 <= S) {
            Row16 r = *(const Row16 *)(row + e0);
#pragma unroll
            for (int j = 0; j < 16; ++j) cnt += (r.v[j] <= slot) ? 1u : 0u;
        } else {
            for (int j = 0; j < 16; ++j)
                if (e0 + j < S) cnt += (row[e0 + j] <= slot) ? 1u : 0u;
        }
    }
    return cnt;
}

__global__ __launch_bounds__(256) void decode_streams_kernel(Geometry g, const uint16_t *__restrict__ table,
                                                              const int32_t *__restrict__ idx,
                                                              const uint8_t *__restrict__ in, size_t in_stride,
                                                              const uint64_t *__restrict__ nbytes,
                                                              int16_t *__restrict__ sym_out,
                                                              int32_t *__restrict__ status) {
    const int t = threadIdx.x & (GROUP - 1);
    const int64_t gid = (int64_t)blockIdx.x * GROUPS_PER_BLOCK + (threadIdx.x / GROUP);
    const int64_t G = (int64_t)g.n_images * g.n_streams;
    if (gid >= G) return;                                   // whole groups leave together
    const int img = (int)(gid / g.n_streams);
    const uint32_t stream = (uint32_t)(gid % g.n_streams);
    const uint8_t *src = in + (size_t)img * in_stride;
    const uint64_t nb = nbytes[img];
    const uint32_t hdr = 16 + 2 * (uint32_t)g.n_streams;
    int err = 0;
    if (nb < hdr) err = -20;
    uint32_t my_len = 0, off = hdr;
    if (!err) {
        const uint32_t *h4 = (const uint32_t *)src;
        if (h4[0] != MRS1_MAGIC || h4[1] != (uint32_t)g.n_streams || h4[2] != g.n_symbols || h4[3] != g.sps) err = -21;
    }
    if (!err) {
        const uint16_t *lens = (const uint16_t *)(src + 16);
        uint32_t acc = 0;
        for (uint32_t j = t; j < stream; j += GROUP) acc += lens[j];
        off += 2 * group_sum(acc);
        my_len = lens[stream];
        if (my_len < 2 || (uint64_t)off + 2ull * my_len > nb) err = -22;
    }
    if (err) {
        if (t == 0) atomicMin(&status[img], err);
        return;
    }
    const uint16_t *words = (const uint16_t *)(src + off);
    const uint16_t *wend = words + my_len;
    uint32_t x = (uint32_t)words[0] | ((uint32_t)words[1] << 16);
    const uint16_t *wp = words + 2;
    uint32_t wnext = (wp < wend) ? *wp : 0;                 // next renormalisation word, loaded ahead

    uint64_t b = (uint64_t)stream * g.sps, e = b + g.sps;
    if (b > g.n_symbols) b = g.n_symbols;
    if (e > g.n_symbols) e = g.n_symbols;
    const int S = g.L - 1;
    const int32_t *img_idx = idx + (int64_t)img * g.n_loc;
    int16_t *dst = sym_out + (int64_t)img * g.n_symbols;

    for (uint64_t i = b; i < e;) {
        int64_t loc = (int64_t)(i / g.M);
        int k = img_idx[loc];
        if (k < 0 || k >= g.K) { err = -23; break; }
        const uint16_t *rows = table + (int64_t)k * g.M * g.L;
        int c0 = (int)(i % g.M);
        uint64_t loc_end = (uint64_t)(loc + 1) * g.M;
        if (loc_end > e) loc_end = e;
        for (int c = c0; i < loc_end; ++c, ++i) {
            const uint16_t *row = rows + (int64_t)c * g.L;
            uint32_t slot = x & 0xffffu;
            uint32_t cnt = group_sum(count_le(row, S, t, slot));
            if (cnt == 0) { err = -24; break; }
            int s = (int)cnt - 1;
            uint32_t start = row[s];
            uint32_t end = (s == S - 1) ? 0x10000u : (uint32_t)row[s + 1];
            x = (end - start) * (x >> 16) + slot - start;
            if (x < RANS_L) {
                if (wp >= wend) { err = -25; break; }
                x = (x << 16) | wnext;
                ++wp;
                wnext = (wp < wend) ? *wp : 0;
            }
            if (t == 0) dst[i] = (int16_t)s;
        }
        if (err) break;
    }
    if (!err && (x != RANS_L || wp != wend)) err = -26;
    if (err && t == 0) atomicMin(&status[img], err);
}

// ---------------------------------------------------------------------------------------------
// Fast decoder for 256-symbol rows (L = 257, the standard table): the row of every symbol is staged
// global -> registers -> LDS two symbols ahead of its use, and the symbol is found with a two-level
// search in LDS (16 segment heads, then the 16 entries of the segment), one 16-lane ballot per level.
// Per symbol the coder state waits on two LDS round trips instead of global-memory round trips.
// ---------------------------------------------------------------------------------------------
constexpr int RING = 4;                    // LDS row slots per group (rows i .. i+1 are live, 4 for slack)

struct RowRegs {
    uint4 a, b;                            // 16 u16 entries: 16t .. 16t+15 of the row
};

__device__ __forceinline__ RowRegs load_row(const uint16_t *row, int t) {
    Row16 r = *(const Row16 *)(row + 16 * t);
    return __builtin_bit_cast(RowRegs, r);
}

__global__ __launch_bounds__(256) void decode_streams_lds_kernel(Geometry g, const uint16_t *__restrict__ table,
                                                                  const int32_t *__restrict__ idx,
                                                                  const uint8_t *__restrict__ in, size_t in_stride,
                                                                  const uint64_t *__restrict__ nbytes,
                                                                  int16_t *__restrict__ sym_out,
                                                                  int32_t *__restrict__ status) {
    __shared__ __attribute__((aligned(16))) uint16_t rows_sh[GROUPS_PER_BLOCK][RING][256];
    __shared__ uint16_t heads_sh[GROUPS_PER_BLOCK][RING][16];
    const int t = threadIdx.x & (GROUP - 1);
    const int gi = threadIdx.x / GROUP;                    // group within the block
    const int gshift = (threadIdx.x & 63) & ~(GROUP - 1);  // bit offset of this group in a wave ballot
    const int64_t gid = (int64_t)blockIdx.x * GROUPS_PER_BLOCK + gi;
    const int64_t G = (int64_t)g.n_images * g.n_streams;
    if (gid >= G) return;
    const int img = (int)(gid / g.n_streams);
    const uint32_t stream = (uint32_t)(gid % g.n_streams);
    const uint8_t *src = in + (size_t)img * in_stride;
    const uint64_t nb = nbytes[img];
    const uint32_t hdr = 16 + 2 * (uint32_t)g.n_streams;
    int err = 0;
    if (nb < hdr) err = -20;
    uint32_t my_len = 0, off = hdr;
    if (!err) {
        const uint32_t *h4 = (const uint32_t *)src;
        if (h4[0] != MRS1_MAGIC || h4[1] != (uint32_t)g.n_streams || h4[2] != g.n_symbols || h4[3] != g.sps) err = -21;
    }
    if (!err) {
        const uint16_t *lens = (const uint16_t *)(src + 16);
        uint32_t acc = 0;
        for (uint32_t j = t; j < stream; j += GROUP) acc += lens[j];
        off += 2 * group_sum(acc);
        my_len = lens[stream];
        if (my_len < 2 || (uint64_t)off + 2ull * my_len > nb) err = -22;
    }
    if (err) {
        if (t == 0) atomicMin(&status[img], err);
        return;
    }
    const uint16_t *words = (const uint16_t *)(src + off);
    uint32_t x = (uint32_t)words[0] | ((uint32_t)words[1] << 16);
    // Renormalisation words are kept in a 32-word register window spread over the group's 16 lanes (lane t
    // holds words wbase+t and wbase+16+t) and read with a lane shuffle, so the coder state never waits on a
    // global load; the window is refilled 16 words ahead of its use.
    uint32_t wpos = 2, wbase = 2;                          // next word to consume; first word of the window
    auto fetch = [&](uint32_t k) -> uint32_t { return k < my_len ? (uint32_t)words[k] : 0u; };
    uint32_t win0 = fetch(wbase + t), win1 = fetch(wbase + 16 + t);

    uint32_t b = (uint32_t)min((uint64_t)stream * g.sps, (uint64_t)g.n_symbols);
    uint32_t e = (uint32_t)min((uint64_t)b + g.sps, (uint64_t)g.n_symbols);
    const int32_t *img_idx = idx + (int64_t)img * g.n_loc;
    int16_t *dst = sym_out + (int64_t)img * g.n_symbols;
    const uint32_t M = (uint32_t)g.M;
    const int64_t prior_stride = (int64_t)g.M * g.L;

    // Prefetch cursor: walks the rows of symbols b, b+1, ... (three ahead of the decode position).  The prior
    // index of a location is read one whole location (M symbols) before its rows are needed, so no row address
    // ever waits on an index load.
    const uint32_t loc_last = (e - 1) / M;                 // valid when b < e
    uint32_t pf_loc = b / M, pf_c = b % M;
    auto clamp_k = [&](int k) { return (k < 0 || k >= g.K) ? 0 : k; };
    int pf_k = (b < e) ? clamp_k(img_idx[pf_loc]) : 0;
    int pf_knext = (b < e && pf_loc < loc_last) ? clamp_k(img_idx[pf_loc + 1]) : pf_k;
    const uint16_t *pf_row = table + (int64_t)pf_k * prior_stride + (int64_t)pf_c * g.L;
    uint32_t pf_i = b;
    // returns the row of the cursor's symbol and advances the cursor (stays on the last symbol at the end)
    auto next_row = [&]() -> const uint16_t * {
        const uint16_t *r = pf_row;
        if (pf_i + 1 < e) {
            ++pf_i;
            if (++pf_c == M) {
                pf_c = 0;
                ++pf_loc;
                pf_k = pf_knext;
                pf_knext = (pf_loc < loc_last) ? clamp_k(img_idx[pf_loc + 1]) : pf_k;
                pf_row = table + (int64_t)pf_k * prior_stride;
            } else {
                pf_row += g.L;
            }
        }
        return r;
    };
    auto stage = [&](int slot, const RowRegs &r) {
        uint4 *d = (uint4 *)&rows_sh[gi][slot][16 * t];
        d[0] = r.a;
        d[1] = r.b;
        heads_sh[gi][slot][t] = (uint16_t)(r.a.x & 0xffffu);
    };
    // decode one symbol from LDS slot `slot`; returns false on a corrupt stream
    auto decode_one = [&](uint32_t i, int slot) -> bool {
        const uint32_t xslot = x & 0xffffu;
        uint32_t head = heads_sh[gi][slot][t];
        unsigned long long m1 = __ballot(head <= xslot);
        uint32_t seg_cnt = __popc((uint32_t)(m1 >> gshift) & 0xffffu);
        if (seg_cnt == 0) { err = -24; return false; }
        uint32_t seg = seg_cnt - 1;
        uint32_t ent = rows_sh[gi][slot][16 * seg + t];
        unsigned long long m2 = __ballot(ent <= xslot);
        uint32_t s = 16 * seg + __popc((uint32_t)(m2 >> gshift) & 0xffffu) - 1;
        uint32_t start = rows_sh[gi][slot][s];
        uint32_t end = (s == 255u) ? 0x10000u : (uint32_t)rows_sh[gi][slot][(s + 1) & 255];
        x = (end - start) * (x >> 16) + xslot - start;
        if (x < RANS_L) {
            if (wpos >= my_len) { err = -25; return false; }
            uint32_t w = __shfl(win0, (int)(wpos - wbase), GROUP);
            x = (x << 16) | w;
            ++wpos;
            if (wpos - wbase == 16) {                       // first half consumed: slide the window
                win0 = win1;
                wbase += 16;
                win1 = fetch(wbase + 16 + t);
            }
        }
        if (t == 0) dst[i] = (int16_t)s;
        return true;
    };

    if (b < e) {
        // validate the prior indices of this stream once (row_of clamps, this reports)
        for (uint32_t l = b / M + t; l < (e + M - 1) / M; l += GROUP) {
            int k = img_idx[l];
            if (k < 0 || k >= g.K) err = -23;
        }
        if (__ballot(err != 0) >> gshift & 0xffffu) err = -23;
    }
    if (!err && b < e) {
        RowRegs ra = load_row(next_row(), t);
        RowRegs rb = load_row(next_row(), t);
        stage(0, ra);
        ra = load_row(next_row(), t);
        uint32_t i = b;
        // invariant at the top: LDS slot (i-b)&3 holds row i; rb holds row i+1; ra holds row i+2
        while (i < e) {
            int s0 = (int)((i - b) & (RING - 1));
            stage((s0 + 1) & (RING - 1), rb);
            rb = load_row(next_row(), t);
            if (!decode_one(i, s0)) break;
            ++i;
            if (i >= e) break;
            int s1 = (int)((i - b) & (RING - 1));
            stage((s1 + 1) & (RING - 1), ra);
            ra = load_row(next_row(), t);
            if (!decode_one(i, s1)) break;
            ++i;
        }
    }
    if (!err && (x != RANS_L || wpos != my_len)) err = -26;
    if (err && t == 0) atomicMin(&status[img], err);
}

int check_geometry(mpc_ctx *ctx, Geometry &g, int K, int M, int L, int n_images, int64_t n_loc, int n_streams,
                   int locs_per_stream) {
    MPC_REQUIRE(ctx, K > 0 && M > 0 && L > 1 && n_images > 0 && n_loc > 0 && n_streams > 0 && locs_per_stream > 0,
                "rans: bad geometry");
    MPC_REQUIRE(ctx, (int64_t)n_streams * locs_per_stream >= n_loc, "rans: n_streams * locs_per_stream < n_loc");
    MPC_REQUIRE(ctx, (int64_t)locs_per_stream * M + 2 <= 65535, "rans: stream too long (locs_per_stream*M + 2 > 65535)");
    MPC_REQUIRE(ctx, n_loc * M < (1ll << 32), "rans: more than 2^32 symbols per image");
    g.n_images = n_images; g.n_loc = n_loc; g.M = M; g.L = L; g.K = K;
    g.n_streams = n_streams; g.locs_per_stream = locs_per_stream;
    g.sps = (uint32_t)locs_per_stream * M;
    g.n_symbols = (uint32_t)(n_loc * M);
    g.cap_words = g.sps + 2;
    return MPC_OK;
}

struct Workspace {
    uint32_t *params, *lens, *offs;
    uint16_t *scratch;
    size_t bytes;
};

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

Workspace carve(void *base, int n_images, int64_t n_loc, int M, int n_streams, int locs_per_stream) {
    Workspace w;
    size_t G = (size_t)n_images * n_streams;
    size_t sps = (size_t)locs_per_stream * M;
    size_t off = 0;
    uint8_t *p = (uint8_t *)base;
    w.params = (uint32_t *)(p + off); off += align256(sps * G * 4);
    w.scratch = (uint16_t *)(p + off); off += align256((sps + 2) * G * 2);
    w.lens = (uint32_t *)(p + off); off += align256(G * 4);
    w.offs = (uint32_t *)(p + off); off += align256(G * 4);
    w.bytes = off;
    (void)n_loc;
    return w;
}

}  // namespace

extern "C" {

size_t mpc_rans_workspace_bytes(int n_images, int64_t n_loc, int M, int n_streams) {
    if (n_images <= 0 || n_loc <= 0 || M <= 0 || n_streams <= 0) return 0;
    int lps = (int)mpc_cdiv(n_loc, n_streams);
    return carve(nullptr, n_images, n_loc, M, n_streams, lps).bytes;
}

int mpc_rans_encode(mpc_ctx *ctx, const int16_t *table, int K, int M, int L, const int32_t *idx, const int16_t *sym,
                    int n_images, int64_t n_loc, int n_streams, int locs_per_stream, void *workspace,
                    size_t workspace_bytes, uint8_t *out, size_t out_stride, uint64_t *nbytes_dev,
                    int32_t *status_dev) {
    MPC_REQUIRE(ctx, table && idx && sym && workspace && out && nbytes_dev && status_dev, "rans_encode: NULL argument");
    Geometry g;
    int rc = check_geometry(ctx, g, K, M, L, n_images, n_loc, n_streams, locs_per_stream);
    if (rc) return rc;
    MPC_REQUIRE(ctx, locs_per_stream == (int)mpc_cdiv(n_loc, n_streams), "rans_encode: locs_per_stream must be ceil(n_loc / n_streams)");
    MPC_REQUIRE(ctx, ((uintptr_t)out % 4) == 0 && out_stride % 4 == 0 && ((uintptr_t)workspace % 256) == 0,
                "rans_encode: out must be 4-byte aligned with a stride multiple of 4; workspace 256-byte aligned");
    Workspace w = carve(workspace, n_images, n_loc, M, n_streams, locs_per_stream);
    if (w.bytes > workspace_bytes) MPC_FAIL(ctx, MPC_ENOMEM, "rans_encode: workspace too small (%zu < %zu)", workspace_bytes, w.bytes);
    MPC_REQUIRE(ctx, out_stride >= 16 + 2 * (size_t)n_streams, "rans_encode: out_stride smaller than the header");
    int64_t G = (int64_t)n_images * n_streams;
    int64_t total = (int64_t)n_images * g.n_symbols;
    MPC_REQUIRE(ctx, mpc_cdiv(total, 256) < (1ll << 31), "rans_encode: too many symbols in one call");
    MPC_HIP(ctx, hipMemsetAsync(status_dev, 0, sizeof(int32_t) * n_images, ctx->stream));
    symbol_params_kernel<<<dim3((unsigned)mpc_cdiv(total, 256)), dim3(256), 0, ctx->stream>>>(
        g, (const uint16_t *)table, idx, sym, w.params, status_dev);
    MPC_CHECK_LAUNCH(ctx, "rans symbol_params");
    encode_streams_kernel<<<dim3((unsigned)mpc_cdiv(G, 64)), dim3(64), 0, ctx->stream>>>(g, w.params, w.scratch, w.lens);
    MPC_CHECK_LAUNCH(ctx, "rans encode_streams");
    header_kernel<<<dim3((unsigned)n_images), dim3(256), 0, ctx->stream>>>(g, w.lens, w.offs, out, out_stride, nbytes_dev, status_dev);
    MPC_CHECK_LAUNCH(ctx, "rans header");
    copy_streams_kernel<<<dim3((unsigned)G), dim3(256), 0, ctx->stream>>>(g, w.scratch, w.lens, w.offs, out, out_stride, nbytes_dev);
    MPC_CHECK_LAUNCH(ctx, "rans copy_streams");
    return MPC_OK;
}

int mpc_rans_decode(mpc_ctx *ctx, const int16_t *table, int K, int M, int L, const int32_t *idx, const uint8_t *in,
                    size_t in_stride, const uint64_t *nbytes_dev, int n_images, int64_t n_loc, int n_streams,
                    int locs_per_stream, int16_t *sym_out, int32_t *status_dev) {
    MPC_REQUIRE(ctx, table && idx && in && nbytes_dev && sym_out && status_dev, "rans_decode: NULL argument");
    Geometry g;
    int rc = check_geometry(ctx, g, K, M, L, n_images, n_loc, n_streams, locs_per_stream);
    if (rc) return rc;
    MPC_REQUIRE(ctx, ((uintptr_t)in % 4) == 0 && in_stride % 4 == 0, "rans_decode: in must be 4-byte aligned with a stride multiple of 4");
    int64_t G = (int64_t)n_images * n_streams;
    MPC_HIP(ctx, hipMemsetAsync(status_dev, 0, sizeof(int32_t) * n_images, ctx->stream));
    if (L == 257)
        decode_streams_lds_kernel<<<dim3((unsigned)mpc_cdiv(G, GROUPS_PER_BLOCK)), dim3(256), 0, ctx->stream>>>(
            g, (const uint16_t *)table, idx, in, in_stride, nbytes_dev, sym_out, status_dev);
    else
        decode_streams_kernel<<<dim3((unsigned)mpc_cdiv(G, GROUPS_PER_BLOCK)), dim3(256), 0, ctx->stream>>>(
            g, (const uint16_t *)table, idx, in, in_stride, nbytes_dev, sym_out, status_dev);
    MPC_CHECK_LAUNCH(ctx, "rans decode_streams");
    return MPC_OK;
}

}  // extern "C"
