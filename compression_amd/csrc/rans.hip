// MRS1 entropy coder: tile-interleaved rANS streams over the per-location CDF rows
// table[idx[loc]][c][:].  Replaces torchac.encode_cdf / decode_cdf
// (src/compression/models/manynets_compressor.py:274,328,724,727).  The byte format is specified by
// the scalar implementation in oracle/rans_ref.c; the encoder here must produce exactly those bytes.
//
// Encoder = three launches.
//   1. symbol_params: one thread per symbol, fully parallel: gathers (start, freq) of the symbol from
//      the int16 table (2 B symbol + two 2 B table entries read) and stores them with floor(2^32 / freq)
//      (8 B written), transposed, params[r][stream], so that step 2 reads coalesced.
//   2. encode_streams: one LANE per stream (rANS is sequential within a stream).  Walks the stream's
//      symbols backwards, emits 16-bit words backwards into the stream's scratch slot.  The state's
//      division by freq is a multiplication by the stored reciprocal and one correction: the division
//      sequence was most of a step's serial chain.
//   3. compact: per image, exclusive scan of the stream lengths, header, and a copy of every stream's
//      words to its final offset.
// Decoders.  Rows of any length (decode_streams_kernel): one 16-LANE GROUP per stream, lane t holds entries
// 16t..16t+15 of the symbol's CDF row, counts entries <= slot and the group sums the counts -> symbol.  The
// standard 256-symbol rows (decode_streams_wave_kernel) and the per-element Laplace rows of the hyperprior
// models (decode_streams_laplace_kernel): one WAVE per stream with the coder state in scalar registers; the row
// address (or the row's sigma) depends only on (idx, channel), never on decoded data, so whatever the search looks
// at is brought or evaluated ahead of the serial chain.
#include "mpc_internal.h"

// Scratch of the allocation-free decode entry point: the compact copy of the CDF table (256 bytes per row, see the fast
// decoder below), at most NOALLOC_MAX_ROWS rows (the standard model has K*M = 64*256 of them) = 4 MiB per GPU, held until
// the process ends.  In host code `scratch` is only a handle for hipGetSymbolAddress; kernels get the per-device address.
constexpr int NOALLOC_MAX_ROWS = 64 * 256;
constexpr int CROW = 128;                  // 16-bit halves of a compact row: 64 words of (window entry | head or w0 << 16), 256 bytes
constexpr size_t NOALLOC_SCRATCH_BYTES = (size_t)NOALLOC_MAX_ROWS * CROW * sizeof(uint16_t);
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];
static void *static_scratch_ptr[MPC_MAX_DEVICES] = {};

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
    int latent_f16 = 0;     // decoders: write fp16(symbol * lat_q + lat_min) (the latent g_s reads) instead of the symbol
    int lat_min = 0, lat_q = 1;
};

// what a decoder stores for symbol s: the symbol itself, or the bits of the latent value as fp16 (exact for |v| <= 2048)
__device__ __forceinline__ int16_t decoded_value(const Geometry &g, int s) {
    if (!g.latent_f16) return (int16_t)s;
    const _Float16 v = (_Float16)fminf(fmaxf((float)(s * g.lat_q + g.lat_min), -65504.0f), 65504.0f);
    return __builtin_bit_cast(int16_t, v);
}

// Stream geometry chosen on the device (mpc_rans_encode_auto): ov = {n_streams, locs_per_stream} replaces the
// host's values (which then only size the launch: grids cover the largest geometry, surplus threads leave).
__device__ __forceinline__ Geometry resolve(Geometry g, const int32_t *__restrict__ ov) {
    if (ov) {
        g.n_streams = ov[0];
        g.locs_per_stream = ov[1];
        g.sps = (uint32_t)ov[1] * (uint32_t)g.M;
        g.cap_words = g.sps + 2;
    }
    return g;
}

// One workgroup: est = sum(minbits) / 8 / n_images bytes per image, then the stream count that keeps the container's
// per-stream overhead (6 bytes) within `budget` of it -- the rule of ops.choose_streams, evaluated in double.
__global__ __launch_bounds__(1024) void choose_geometry_kernel(const float *__restrict__ minbits, int64_t n, int n_images,
                                                               int64_t n_loc, int M, double budget, int max_streams,
                                                               int32_t *__restrict__ ov) {
    __shared__ double part[16];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) acc += (double)minbits[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double total = 0.0;
        for (int w = 0; w < 16; ++w) total += part[w];
        const double est = total / 8.0 / (double)n_images;
        const int64_t per = 65533 / M > 0 ? 65533 / M : 1;
        const int64_t min_streams = (n_loc + per - 1) / per;
        double w = est * budget - 16.0;
        int64_t want = (w > 0.0) ? (int64_t)floor(w / 6.0) : 0;       // NaN compares false: no extra streams
        int64_t ns = want < max_streams ? want : max_streams;
        if (ns > n_loc) ns = n_loc;
        if (ns < min_streams) ns = min_streams;
        if (ns < 1) ns = 1;
        const int64_t lps = (n_loc + ns - 1) / ns;
        ns = (n_loc + lps - 1) / lps;                                   // drop streams that would be empty
        ov[0] = (int32_t)ns;
        ov[1] = (int32_t)lps;
    }
}

// floor(2^32 / freq) for 1 <= freq <= 2^16, saturated at 2^32 - 1 (freq = 1).  With it q = floor(x / freq) for any
// 32-bit x is mulhi(x, r) or one more: 2^32 = freq r + t with 0 <= t < freq, so x r / 2^32 lies in (x / freq - 1, x / freq].
// (The double quotient is exact enough to truncate: a quotient that is not an integer lies at least 1 / freq >= 2^-16
// below the next one, rounding moves it by at most 2^-21.)
__device__ __forceinline__ uint32_t reciprocal32(uint32_t freq) {
    return freq <= 1u ? 0xffffffffu : (uint32_t)(4294967296.0 / (double)freq);
}

__global__ __launch_bounds__(256) void symbol_params_kernel(Geometry g, const uint16_t *__restrict__ table,
                                                             const int32_t *__restrict__ idx,
                                                             const int16_t *__restrict__ sym,
                                                             uint2 *__restrict__ params,
                                                             int32_t *__restrict__ status, const int32_t *__restrict__ ov) {
    g = resolve(g, ov);
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
    params[(int64_t)r * G + (int64_t)img * g.n_streams + stream] = make_uint2(start | (freq << 16), reciprocal32(freq));
}

// symbol_params for per-element Laplace(0, sigma) rows (the hyperprior's main stream, abstract_model.py:402-423): the
// two row entries a symbol needs are evaluated from its sigma; no row is ever stored.  sigma is laid out like sym,
// [n_images][n_loc][M].
__global__ __launch_bounds__(256) void symbol_params_laplace_kernel(Geometry g, const float *__restrict__ sigma, float first,
                                                                     float factor, const int16_t *__restrict__ sym,
                                                                     uint2 *__restrict__ params,
                                                                     int32_t *__restrict__ status) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)g.n_images * g.n_symbols;
    if (i >= total) return;
    int img = (int)(i / g.n_symbols);
    uint32_t local = (uint32_t)(i % g.n_symbols);
    int s = sym[i];
    bool bad = (s < 0) | (s > g.L - 2);
    s = bad ? 0 : s;
    const float scale = mpc_laplace_scale(sigma[i]);
    uint32_t start = mpc_laplace_row_entry(scale, first, s, factor);
    uint32_t end = (s == g.L - 2) ? 0x10000u : mpc_laplace_row_entry(scale, first, s + 1, factor);
    if (end <= start) bad = true;
    uint32_t freq = bad ? 1u : end - start;
    if (bad) { start = 0; atomicMin(&status[img], -10); }
    uint32_t stream = local / g.sps, r = local % g.sps;
    int64_t G = (int64_t)g.n_images * g.n_streams;
    params[(int64_t)r * G + (int64_t)img * g.n_streams + stream] = make_uint2(start | (freq << 16), reciprocal32(freq));
}

__global__ __launch_bounds__(64) void encode_streams_kernel(Geometry g, const uint2 *__restrict__ params,
                                                             uint16_t *__restrict__ scratch,
                                                             uint32_t *__restrict__ lens, const int32_t *__restrict__ ov) {
    g = resolve(g, ov);
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
    const uint2 *p = params + gid;
    auto step = [&](uint2 sf) {
        uint32_t start = sf.x & 0xffffu, freq = sf.x >> 16;
        if (x >= (freq << 16)) { buf[--pos] = (uint16_t)(x & 0xffffu); x >>= 16; }
        uint32_t q = __umulhi(x, sf.y), rem = x - q * freq;  // q = floor(x / freq) or one less (reciprocal32)
        if (rem >= freq) { ++q; rem -= freq; }
        x = (q << 16) + rem + start;
    };
    // symbols are consumed backwards; the (start,freq) words of the next 8 are loaded while the current 8
    // are coded (the addresses do not depend on the coder state)
    constexpr int U = 8;
    uint32_t r = n;
    uint2 cur[U], nxt[U];
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
                                                      int32_t *__restrict__ status, const int32_t *__restrict__ ov) {
    g = resolve(g, ov);
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
                                                            const uint64_t *__restrict__ nbytes, const int32_t *__restrict__ ov) {
    g = resolve(g, ov);
    int64_t gid = blockIdx.x;
    if (gid >= (int64_t)g.n_images * g.n_streams) return;   // grid sized for the largest geometry
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
    const uint64_t hdr = 16 + 2 * (uint64_t)g.n_streams;   // the host has checked hdr <= in_stride < 2^32
    int err = 0;
    if (nb < hdr || nb > in_stride) err = -20;
    uint32_t my_len = 0, off = (uint32_t)hdr;
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
            if (t == 0) dst[i] = decoded_value(g, (int)s);
        }
        if (err) break;
    }
    if (!err && (x != RANS_L || wp != wend)) err = -26;
    if (err && t == 0) atomicMin(&status[img], err);
}

// Decoder for per-element Laplace(0, sigma) rows, S = L - 1 <= 256 symbols.  Nothing is read per symbol but its sigma;
// the row entries the search looks at are evaluated (mpc_laplace_row_entry, the function the encoder side uses, so both
// sides see the same integers; entries at index >= S count as 2^16, as in count_le).
//
// One wave decodes one stream.  The serial chain state -> symbol -> state is all there is to a stream, so the kernel
// is built to keep that chain short:
//   * whatever depends on sigma only is evaluated ahead of the chain, four symbols per pass: the 64 lanes are 4 symbols
//     x 16 entries.  Which 16 entries follows the row's shape.  scale <= LAPLACE_WINDOW_MAX_SCALE: entry(w0 + t), the
//     window round the mode (values -7 .. +7 for the standard table); a state that falls between the first and the last
//     of them (all but exp(-7.5 / scale) of the states: 99.8 % at scale 1.2, 61 % at 8) is decoded with no evaluation on
//     the chain.  Wider rows: the segment heads entry(16 t); the chain then evaluates entry(16 seg + t) of the segment
//     the state falls into (the two-level search of the fast decoder below).  A narrow row whose state falls outside
//     the window does the two-level search with both evaluations on the chain.  (Measured per symbol of a stream:
//     0.21 us decoded from the window, 0.42 us with one evaluation on the chain, 0.63 us with two: the window pays
//     up to a miss rate of one half, scale 10.)
//   * the coder state, the slot, the symbol and the interval live in scalar registers: a ballot gives the count, the
//     two interval ends are read from the lanes that hold them (v_readlane), the state update is scalar arithmetic.
//   * nothing on the chain waits for memory: the stream's words and the sigmas are held 64 at a time in a register
//     (lane l: element base + l), the following 64 are loaded while those are used; the next word / the scales of the
//     next four symbols are taken out of the register by lane index.  Symbols are stored 64 at a time.
constexpr float LAPLACE_WINDOW_MAX_SCALE = 8.0f;
constexpr int LAPLACE_WAVES_PER_BLOCK = 4;

__device__ __forceinline__ uint32_t lane_value(uint32_t v, uint32_t lane) {      // lane: the same in every lane
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane);
}

__global__ __launch_bounds__(256) void decode_streams_laplace_kernel(Geometry g, const float *__restrict__ sigma, float first,
                                                                      float factor, const uint8_t *__restrict__ in,
                                                                      size_t in_stride, const uint64_t *__restrict__ nbytes,
                                                                      int16_t *__restrict__ sym_out,
                                                                      int32_t *__restrict__ status) {
    const uint32_t lane = threadIdx.x & 63;
    const int t = (int)(lane & 15);
    // (the wave's index through readfirstlane: the compiler then knows that everything derived from it is the same in
    // all lanes and keeps it, the coder state included, in scalar registers)
    const int64_t gid = (int64_t)blockIdx.x * LAPLACE_WAVES_PER_BLOCK + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t G = (int64_t)g.n_images * g.n_streams;
    if (gid >= G) return;                                   // whole waves leave together
    const int img = (int)(gid / g.n_streams);
    const uint32_t stream = (uint32_t)(gid % g.n_streams);
    const uint8_t *src = in + (size_t)img * in_stride;
    const uint64_t nb = nbytes[img];
    const uint64_t hdr = 16 + 2 * (uint64_t)g.n_streams;   // the host has checked hdr <= in_stride < 2^32
    int err = 0;
    if (nb < hdr || nb > in_stride) err = -20;
    uint32_t my_len = 0, off = (uint32_t)hdr;
    if (!err) {
        const uint32_t *h4 = (const uint32_t *)src;
        if (h4[0] != MRS1_MAGIC || h4[1] != (uint32_t)g.n_streams || h4[2] != g.n_symbols || h4[3] != g.sps) err = -21;
    }
    if (!err) {
        const uint16_t *lens = (const uint16_t *)(src + 16);
        uint32_t acc = 0;
        for (uint32_t j = lane; j < stream; j += 64) acc += lens[j];
        acc = group_sum(acc);
        acc = lane_value(acc, 0) + lane_value(acc, 16) + lane_value(acc, 32) + lane_value(acc, 48);
        off += 2 * acc;
        my_len = lens[stream];
        if (my_len < 2 || (uint64_t)off + 2ull * my_len > nb) err = -22;
    }
    if (err) {
        if (lane == 0) atomicMin(&status[img], err);
        return;
    }
    off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
    my_len = (uint32_t)__builtin_amdgcn_readfirstlane((int)my_len);
    const uint16_t *words = (const uint16_t *)(src + off);
    // (indices clamped, not guarded: a guarded load is waited for where the guard ends; what lies past the end is never used)
    auto load_words = [&](uint32_t base) -> uint32_t { return (uint32_t)words[min(base + lane, my_len - 1)]; };
    uint32_t wbuf = load_words(0), wbuf_nx = load_words(64);
    uint32_t x = lane_value(wbuf, 0) | (lane_value(wbuf, 1) << 16);
    uint32_t wi = 2;                                        // words taken so far
    uint32_t wnext = lane_value(wbuf, 2);                   // (past the end: never used, a renormalisation there is an error)

    const uint32_t b = (uint32_t)min((uint64_t)stream * g.sps, (uint64_t)g.n_symbols);
    const uint32_t e = (uint32_t)min((uint64_t)b + g.sps, (uint64_t)g.n_symbols);
    const uint32_t len = e - b;
    const int S = g.L - 1;
    const float *sg = sigma + (int64_t)img * g.n_symbols + b;
    int16_t *dst = sym_out + (int64_t)img * g.n_symbols + b;
    // (evaluated for every index and selected afterwards: no branch round the evaluation)
    auto entry = [&](float scale, int s) -> uint32_t {
        const uint32_t v = mpc_laplace_row_entry(scale, first, min(s, S - 1), factor);
        return s < S ? v : 0x10000u;
    };
    auto load_sigmas = [&](uint32_t base) -> float { return len ? sg[min(base + lane, len - 1)] : 1.0f; };
    // first entry of the window round the mode: the symbol that holds the value 0 is -first - 0.5
    const int w0 = max(0, min((int)rintf(-first - 0.5f) - 7, S - 15));

    float sbuf = load_sigmas(0), sbuf_nx = load_sigmas(64);
    // lane (u, t) of a pass: symbol n0 + u; what it holds of that symbol's row, and the symbol's scale
    auto ahead = [&](uint32_t n0, float &scale_out) -> uint32_t {
        const float raw = __shfl(sbuf, (int)((n0 & 63) + (lane >> 4)), 64);
        const float scale = mpc_laplace_scale(raw);
        scale_out = scale;
        return entry(scale, scale <= LAPLACE_WINDOW_MAX_SCALE ? w0 + t : 16 * t);
    };
    float scale_v = 1.0f, scale_nx = 1.0f;
    uint32_t pre = ahead(0, scale_v);
    uint32_t mysym = 0;
    for (uint32_t n0 = 0; n0 < len && !err; n0 += 4) {
        // ---- ahead of the chain: the next four symbols ----
        if (((n0 + 4) & 63) == 0) {
            sbuf = sbuf_nx;
            sbuf_nx = load_sigmas(n0 + 4 + 64);
        }
        const uint32_t pre_nx = ahead(n0 + 4, scale_nx);
        const uint64_t narrow = __ballot(scale_v <= LAPLACE_WINDOW_MAX_SCALE);
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
            const uint32_t n = n0 + u;
            if (n >= len) break;
            // ---- the chain: state -> symbol -> new state, in scalar registers ----
            const uint32_t slot = x & 0xffffu;
            const uint32_t cnt = __popc((uint32_t)(__ballot(pre <= slot) >> (16 * u)) & 0xffffu);
            const bool window = (narrow >> (16 * u)) & 1;
            uint32_t sym, start, end;
            if (window && cnt >= 1 && cnt <= 15) {
                sym = (uint32_t)w0 + cnt - 1;
                start = lane_value(pre, 16 * u + cnt - 1);
                end = lane_value(pre, 16 * u + cnt);
            } else {
                const float scale = __uint_as_float(lane_value(__float_as_uint(scale_v), 16 * u));
                uint32_t seg_cnt = cnt;
                if (window) seg_cnt = __popc((uint32_t)__ballot(entry(scale, 16 * t) <= slot) & 0xffffu);
                if (seg_cnt == 0) { err = -24; break; }                 // slot below the row's first entry: corrupt stream
                const uint32_t seg = seg_cnt - 1;
                // lanes 0..16: the segment's entries and the head of the next one (2^16 behind the row's end)
                const uint32_t ent = entry(scale, (int)(16 * seg + lane));
                const uint32_t in_seg = __popc((uint32_t)__ballot(ent <= slot) & 0xffffu);
                if (in_seg == 0) { err = -24; break; }
                sym = 16 * seg + in_seg - 1;
                start = lane_value(ent, in_seg - 1);
                end = lane_value(ent, in_seg);
            }
            x = (end - start) * (x >> 16) + slot - start;
            if (x < RANS_L) {
                if (wi >= my_len) { err = -25; break; }
                x = (x << 16) | wnext;
                ++wi;
                if ((wi & 63) == 0) {
                    wbuf = wbuf_nx;
                    wbuf_nx = load_words(wi + 64);
                }
                wnext = lane_value(wbuf, wi & 63);
            }
            if (lane == (n & 63)) mysym = sym;
        }
        // ---- off the chain: symbols are stored 64 at a time, one per lane ----
        if (!err && ((n0 & 63) == 60 || n0 + 4 >= len)) {
            const uint32_t base = n0 & ~63u;
            if (base + lane < min(n0 + 4, len)) dst[base + lane] = decoded_value(g, (int)mysym);
        }
        pre = pre_nx;
        scale_v = scale_nx;
    }
    if (!err && (x != RANS_L || wi != my_len)) err = -26;
    if (err && lane == 0) atomicMin(&status[img], err);
}

// ---------------------------------------------------------------------------------------------
// Fast decoder for 256-symbol rows (L = 257, the standard table).
//
// The serial chain of a stream is: state -> symbol search in the symbol's CDF row -> new state; a stream is nothing
// but that chain, a few thousand symbols long, and the decoder's time is the time of one chain.  One wave decodes one
// stream and everything is arranged so that the chain is a handful of scalar instructions that never wait for memory:
//
//   * What is searched is not the 514-byte row but its COMPACT form, 256 bytes = one 32-bit word per lane, made once per
//     call from the table (compact_rows_kernel).  Lane l's word holds in its low half entry 16 w0 + l of the row: the
//     64 entries of the four consecutive segments of 16 that hold the most probability mass (the window, segments
//     w0 .. w0 + 3); in its high half segment head row[16 l] (lanes 0-15) and w0 (lane 16).  One compare and one
//     ballot over the low halves give the symbol's position in the window; the two interval ends are read from the
//     lanes that hold them (v_readlane); coder state, slot, symbol and interval live in scalar registers.  A state that
//     falls outside the window (the tails of the distribution: rare) finds its segment among the heads and reads the
//     segment's entries from the full table.  The compact table, 4 MB for the standard model, is served by the L2.
//   * The row of every symbol is known in advance (prior index of the location, channel), so the words of the next
//     WAVE_DEPTH symbols' rows are loaded while the current WAVE_DEPTH are decoded: one 4-byte load per lane and symbol.
//   * The stream's renormalisation words are held 64 at a time in a register (lane l: word base + l), the following 64
//     are loaded while those are used; the next word is taken out of the register by lane index.  Symbols are stored 64
//     at a time.
//
// No LDS and no LDS-DMA: a CU performs its LDS-DMA transfers in issue order across ALL of its waves, so a decoder that
// brought its rows that way made the first layer of the analysis transform, which runs beside this kernel in the codec
// pipeline and streams its weights by LDS-DMA, wait behind the rows at every step (1.28 -> 1.67 ms for that layer).
// The kernel stays under 48 vector registers: two of its waves then fit a SIMD beside two 208-register waves of that layer.
// ---------------------------------------------------------------------------------------------
constexpr int WINDOW_SEGS = 4;
constexpr int WAVE_DEPTH = 16;             // symbols whose row words are in registers ahead of the chain (and as many in flight)
constexpr int WAVE_STREAMS_PER_BLOCK = 4;

// table [rows][257] -> out [rows][64] words; one 16-lane group per row.
__global__ __launch_bounds__(256) void compact_rows_kernel(const uint16_t *__restrict__ table, int64_t rows, uint32_t *__restrict__ out) {
    const int t = threadIdx.x & 15;
    const int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (r >= rows) return;                                 // whole groups leave together
    const uint16_t *row = table + r * 257;
    uint32_t *o = out + r * (CROW / 2);
    const uint32_t head = row[16 * t];
    // mass of the window that starts at segment t: row[16 (t + 4)] - row[16 t], the row's end counting as 2^16
    const uint32_t upper = (t + WINDOW_SEGS <= 15) ? (uint32_t)row[16 * (t + WINDOW_SEGS)] : 0x10000u;
    uint32_t key = (t <= 16 - WINDOW_SEGS) ? (((upper - head) << 4) | (uint32_t)(15 - t)) : 0u;   // largest mass, then smallest t
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) key = max(key, (uint32_t)__shfl_xor((int)key, off, 16));
    const uint32_t w0 = 15u - (key & 15u);
#pragma unroll
    for (int q = 0; q < WINDOW_SEGS; ++q) {
        const uint32_t high = q == 0 ? head : (q == 1 && t == 0 ? w0 : 0u);
        o[16 * q + t] = (uint32_t)row[16 * (w0 + q) + t] | (high << 16);
    }
}

// ALIGNED: M is a multiple of WAVE_DEPTH.  A pass then lies inside one location (a stream starts on a location), its
// WAVE_DEPTH rows are consecutive in the compact table -- one address and WAVE_DEPTH loads at fixed offsets -- and a
// stream is a whole number of passes.  Otherwise every symbol's row is addressed on its own and the last pass may be short.
template <bool ALIGNED>
__global__ __launch_bounds__(256) void decode_streams_wave_kernel(Geometry g, const uint32_t *__restrict__ compact,
                                                                   const uint16_t *__restrict__ table,
                                                                   const int32_t *__restrict__ idx,
                                                                   const uint8_t *__restrict__ in, size_t in_stride,
                                                                   const uint64_t *__restrict__ nbytes,
                                                                   int16_t *__restrict__ sym_out,
                                                                   int32_t *__restrict__ status) {
    const uint32_t lane = threadIdx.x & 63;
    // (the wave's index through readfirstlane: the compiler then knows that everything derived from it is the same in
    // all lanes and keeps it, the coder state included, in scalar registers)
    const int64_t gid = (int64_t)blockIdx.x * WAVE_STREAMS_PER_BLOCK + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t G = (int64_t)g.n_images * g.n_streams;
    if (gid >= G) return;                                   // whole waves leave together
    const int img = (int)(gid / g.n_streams);
    const uint32_t stream = (uint32_t)(gid % g.n_streams);
    const uint8_t *src = in + (size_t)img * in_stride;
    const uint64_t nb = nbytes[img];
    const uint64_t hdr = 16 + 2 * (uint64_t)g.n_streams;   // the host has checked hdr <= in_stride < 2^32
    const uint32_t M = (uint32_t)g.M;
    int err = 0;
    if (nb < hdr || nb > in_stride) err = -20;
    uint32_t my_len = 0, off = (uint32_t)hdr;
    if (!err) {
        const uint32_t *h4 = (const uint32_t *)src;
        if (h4[0] != MRS1_MAGIC || h4[1] != (uint32_t)g.n_streams || h4[2] != g.n_symbols || h4[3] != g.sps) err = -21;
    }
    if (!err) {
        const uint16_t *lens = (const uint16_t *)(src + 16);
        uint32_t acc = 0;
        for (uint32_t j = lane; j < stream; j += 64) acc += lens[j];
        acc = group_sum(acc);
        acc = lane_value(acc, 0) + lane_value(acc, 16) + lane_value(acc, 32) + lane_value(acc, 48);
        off += 2 * acc;
        my_len = lens[stream];
        if (my_len < 2 || (uint64_t)off + 2ull * my_len > nb) err = -22;
    }
    const uint32_t b = (uint32_t)min((uint64_t)stream * g.sps, (uint64_t)g.n_symbols);
    const uint32_t e = (uint32_t)min((uint64_t)b + g.sps, (uint64_t)g.n_symbols);
    const uint32_t len = e - b;
    const int32_t *img_idx = idx + (int64_t)img * g.n_loc;
    if (!err && len) {                                      // validate the prior indices of this stream once
        bool bad = false;
        for (uint32_t l = b / M + lane; l < (e + M - 1) / M; l += 64) {
            const int k = img_idx[l];
            bad |= k < 0 || k >= g.K;
        }
        if (__ballot(bad)) err = -23;
    }
    if (err) {
        if (lane == 0) atomicMin(&status[img], err);
        return;
    }
    off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
    my_len = (uint32_t)__builtin_amdgcn_readfirstlane((int)my_len);
    const uint16_t *words = (const uint16_t *)(src + off);
    // The stream's words: wa holds words [wbase, wbase + 64), wb the 64 behind them, lane l word l of each.  The chain reads
    // only these two; the words behind them are loaded into wc and moved into place at the top of a later pass, so that
    // nothing the chain reads is the target of a load in flight (the compiler would wait for it at every renormalisation).
    // (indices clamped, not guarded: a guarded load is waited for where the guard ends; what lies past the end is never used)
    auto load_words = [&](uint32_t base) -> uint32_t { return (uint32_t)words[min(base + lane, my_len - 1)]; };
    uint32_t wa = load_words(0), wb = load_words(64), wc = 0;
    bool wc_loaded = false;
    uint32_t x = lane_value(wa, 0) | (lane_value(wa, 1) << 16);
    uint32_t wi = 2, wbase = 0;                             // words taken so far; first word of wa
    uint32_t wnext = lane_value(wa, 2);                     // (past the end: never used, a renormalisation there is an error)
    if (len == 0) {
        if (x != RANS_L || my_len != 2) { if (lane == 0) atomicMin(&status[img], -26); }
        return;
    }
    int16_t *dst = sym_out + (int64_t)img * g.n_symbols + b;

    // the cursor that runs WAVE_DEPTH .. 2 WAVE_DEPTH symbols ahead of the chain and stays on the last symbol (pass) at the end
    uint32_t pf_n = 0, pf_c = 0, pf_loc = b / M;            // (a stream starts on a location: b is a multiple of M)
    int64_t pf_row = (int64_t)img_idx[pf_loc] * M;
    auto next_row_word = [&]() -> uint32_t {
        const uint32_t w = compact[(pf_row + pf_c) * (CROW / 2) + lane];
        if (pf_n + 1 < len) {
            ++pf_n;
            if (++pf_c == M) {
                pf_c = 0;
                ++pf_loc;
                pf_row = (int64_t)img_idx[pf_loc] * M;
            }
        }
        return w;
    };
    auto next_pass = [&](uint32_t (&rows)[WAVE_DEPTH]) {
        if constexpr (ALIGNED) {
            const uint32_t *p = compact + (pf_row + pf_c) * (CROW / 2) + lane;
#pragma unroll
            for (int u = 0; u < WAVE_DEPTH; ++u) rows[u] = p[u * (CROW / 2)];
            if (pf_n + WAVE_DEPTH < len) {
                pf_n += WAVE_DEPTH;
                pf_c += WAVE_DEPTH;
                if (pf_c == M) {
                    pf_c = 0;
                    ++pf_loc;
                    pf_row = (int64_t)img_idx[pf_loc] * M;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < WAVE_DEPTH; ++u) rows[u] = next_row_word();
        }
    };
    uint32_t cur[WAVE_DEPTH], nxt[WAVE_DEPTH];
    next_pass(cur);
    uint32_t mysym = 0;
    // What went wrong, as flags: the chain does not stop inside a pass (every read it makes stays in bounds whatever the
    // state is), the loop ends behind the pass.  1: slot below a row's first entry (-24); 2: the stream's words ran out (-25).
    uint32_t bad = 0;
    for (uint32_t n0 = 0; n0 < len && !bad; n0 += WAVE_DEPTH) {
        // ---- off the chain: loads for later passes ----
        if (wc_loaded) {
            wb = wc;
            wc_loaded = false;
        }
        if (wi - wbase >= 64) {                             // (a pass takes at most WAVE_DEPTH words: wb is not needed before the next pass)
            wa = wb;
            wbase += 64;
            wc = load_words(wbase + 64);
            wc_loaded = true;
        }
        next_pass(nxt);
#pragma unroll
        for (int u = 0; u < WAVE_DEPTH; ++u) {              // (no early exit from this loop: it must unroll, cur[] lives in registers)
            const uint32_t n = n0 + u;
            if (ALIGNED || n < len) {
                const uint32_t w = cur[u];
                // ---- the chain: state -> symbol -> new state, in scalar registers ----
                const uint32_t slot = x & 0xffffu;
                const uint32_t cnt = (uint32_t)__popcll(__ballot((w & 0xffffu) <= slot));
                uint32_t sym, start, end;
                if (cnt >= 1 && cnt <= 63) {
                    sym = 16 * (lane_value(w, 16) >> 16) + cnt - 1;
                    start = lane_value(w, cnt - 1) & 0xffffu;
                    end = lane_value(w, cnt) & 0xffffu;
                } else {
                    // outside the window: the segment among the heads, its entries (and the head behind them) from the full table
                    const uint32_t seg_cnt = __popc((uint32_t)__ballot((w >> 16) <= slot) & 0xffffu);
                    const uint32_t seg = seg_cnt ? seg_cnt - 1 : 0;
                    const uint32_t sidx = b + n, loc = sidx / M, c = sidx - loc * M;
                    const uint16_t *full = table + ((int64_t)img_idx[loc] * M + c) * 257;
                    const uint32_t at = 16 * seg + lane;
                    const uint32_t ent = (lane <= 16 && at < 256) ? (uint32_t)full[at] : 0x10000u;
                    const uint32_t in_seg = __popc((uint32_t)__ballot(ent <= slot) & 0xffffu);
                    if (seg_cnt == 0 || in_seg == 0) bad |= 1;         // a row starts at 0: the table is corrupt
                    sym = 16 * seg + in_seg - 1;
                    start = lane_value(ent, (in_seg - 1) & 63);
                    end = lane_value(ent, in_seg);
                }
                x = (end - start) * (x >> 16) + slot - start;
                if (x < RANS_L) {
                    if (wi >= my_len) bad |= 2;
                    x = (x << 16) | wnext;
                    ++wi;
                    const uint32_t pos = wi - wbase;
                    wnext = pos < 64 ? lane_value(wa, pos) : lane_value(wb, (pos - 64) & 63);
                }
                if (lane == (n & 63)) mysym = sym;
            }
        }
        // ---- off the chain: symbols are stored 64 at a time, one per lane ----
        if (!bad && ((n0 & 63) == 64 - WAVE_DEPTH || n0 + WAVE_DEPTH >= len)) {
            const uint32_t base = n0 & ~63u;
            if (base + lane < min(n0 + WAVE_DEPTH, len)) dst[base + lane] = decoded_value(g, (int)mysym);
        }
#pragma unroll
        for (int u = 0; u < WAVE_DEPTH; ++u) cur[u] = nxt[u];
    }
    err = (bad & 1) ? -24 : (bad & 2) ? -25 : 0;
    if (!err && (x != RANS_L || wi != my_len)) err = -26;
    if (err && lane == 0) atomicMin(&status[img], err);
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
    uint2 *params;
    uint32_t *lens, *offs;
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
    w.params = (uint2 *)(p + off); off += align256(sps * G * sizeof(uint2));
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

}  // extern "C"

// Launches of both encode entry points.  g holds the geometry the grids are sized for; ov (device, may be null)
// the geometry the kernels actually use.
static int rans_encode_launch(mpc_ctx *ctx, const Geometry &g, const int16_t *table, const int32_t *idx, const int16_t *sym,
                              const Workspace &w, uint8_t *out, size_t out_stride, uint64_t *nbytes_dev, int32_t *status_dev,
                              const int32_t *ov) {
    const int64_t G = (int64_t)g.n_images * g.n_streams;
    const int64_t total = (int64_t)g.n_images * g.n_symbols;
    MPC_REQUIRE(ctx, mpc_cdiv(total, 256) < (1ll << 31), "rans_encode: too many symbols in one call");
    MPC_HIP(ctx, hipMemsetAsync(status_dev, 0, sizeof(int32_t) * g.n_images, ctx->stream));
    symbol_params_kernel<<<dim3((unsigned)mpc_cdiv(total, 256)), dim3(256), 0, ctx->stream>>>(
        g, (const uint16_t *)table, idx, sym, w.params, status_dev, ov);
    MPC_CHECK_LAUNCH(ctx, "rans symbol_params");
    encode_streams_kernel<<<dim3((unsigned)mpc_cdiv(G, 64)), dim3(64), 0, ctx->stream>>>(g, w.params, w.scratch, w.lens, ov);
    MPC_CHECK_LAUNCH(ctx, "rans encode_streams");
    header_kernel<<<dim3((unsigned)g.n_images), dim3(256), 0, ctx->stream>>>(g, w.lens, w.offs, out, out_stride, nbytes_dev, status_dev, ov);
    MPC_CHECK_LAUNCH(ctx, "rans header");
    copy_streams_kernel<<<dim3((unsigned)G), dim3(256), 0, ctx->stream>>>(g, w.scratch, w.lens, w.offs, out, out_stride, nbytes_dev, ov);
    MPC_CHECK_LAUNCH(ctx, "rans copy_streams");
    return MPC_OK;
}

// Workspace for any geometry with at most max_streams streams: n_streams * locs_per_stream < n_loc + n_streams.
static Workspace carve_max(void *base, int n_images, int64_t n_loc, int M, int max_streams) {
    Workspace w;
    const size_t G = (size_t)n_images * max_streams;
    const size_t sym_slots = (size_t)n_images * ((size_t)n_loc + max_streams) * M;     // >= G' * sps for every geometry
    size_t off = 0;
    uint8_t *p = (uint8_t *)base;
    w.params = (uint2 *)(p + off); off += align256(sym_slots * sizeof(uint2));
    w.scratch = (uint16_t *)(p + off); off += align256((sym_slots + 2 * G) * 2);
    w.lens = (uint32_t *)(p + off); off += align256(G * 4);
    w.offs = (uint32_t *)(p + off); off += align256(G * 4);
    w.bytes = off;
    return w;
}

extern "C" {

size_t mpc_rans_auto_workspace_bytes(int n_images, int64_t n_loc, int M, int max_streams) {
    if (n_images <= 0 || n_loc <= 0 || M <= 0 || max_streams <= 0) return 0;
    return carve_max(nullptr, n_images, n_loc, M, max_streams).bytes;
}

int mpc_rans_encode(mpc_ctx *ctx, const int16_t *table, int K, int M, int L, const int32_t *idx, const int16_t *sym,
                    int n_images, int64_t n_loc, int n_streams, int locs_per_stream, void *workspace,
                    size_t workspace_bytes, uint8_t *out, size_t out_stride, uint64_t *nbytes_dev,
                    int32_t *status_dev) {
    MPC_ENTER(ctx);
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
    return rans_encode_launch(ctx, g, table, idx, sym, w, out, out_stride, nbytes_dev, status_dev, nullptr);
}

int mpc_rans_encode_auto(mpc_ctx *ctx, const int16_t *table, int K, int M, int L, const int32_t *idx, const int16_t *sym,
                         const float *minbits, int n_images, int64_t n_loc, double budget, int max_streams,
                         void *workspace, size_t workspace_bytes, uint8_t *out, size_t out_stride, uint64_t *nbytes_dev,
                         int32_t *status_dev, int32_t *geometry_dev) {
    MPC_ENTER(ctx);
    MPC_REQUIRE(ctx, table && idx && sym && minbits && workspace && out && nbytes_dev && status_dev && geometry_dev,
                "rans_encode_auto: NULL argument");
    MPC_REQUIRE(ctx, max_streams > 0 && budget >= 0.0, "rans_encode_auto: bad budget / max_streams");
    if ((int64_t)max_streams > n_loc) max_streams = (int)n_loc;
    // the shortest streams any choice can have: every location count per stream from 1 up is legal for the grids below
    const int64_t per = 65533 / M > 0 ? 65533 / M : 1;
    const int64_t min_streams = mpc_cdiv(n_loc, per);
    MPC_REQUIRE(ctx, min_streams <= max_streams, "rans_encode_auto: max_streams is below what the 16-bit stream length needs (%lld)",
                (long long)min_streams);
    Geometry g;
    int rc = check_geometry(ctx, g, K, M, L, n_images, n_loc, max_streams, (int)mpc_cdiv(n_loc, max_streams));
    if (rc) return rc;
    MPC_REQUIRE(ctx, ((uintptr_t)out % 4) == 0 && out_stride % 4 == 0 && ((uintptr_t)workspace % 256) == 0,
                "rans_encode_auto: out must be 4-byte aligned with a stride multiple of 4; workspace 256-byte aligned");
    Workspace w = carve_max(workspace, n_images, n_loc, M, max_streams);
    if (w.bytes > workspace_bytes) MPC_FAIL(ctx, MPC_ENOMEM, "rans_encode_auto: workspace too small (%zu < %zu)", workspace_bytes, w.bytes);
    MPC_REQUIRE(ctx, out_stride >= 16 + 2 * (size_t)max_streams, "rans_encode_auto: out_stride smaller than the largest header");
    choose_geometry_kernel<<<dim3(1), dim3(1024), 0, ctx->stream>>>(minbits, (int64_t)n_images * n_loc, n_images, n_loc, M, budget,
                                                                   max_streams, geometry_dev);
    MPC_CHECK_LAUNCH(ctx, "rans choose_geometry");
    return rans_encode_launch(ctx, g, table, idx, sym, w, out, out_stride, nbytes_dev, status_dev, geometry_dev);
}

}  // extern "C"

// Shared body of the two decode entry points.  `compact_rows`: device buffer of K*M*CROW uint16 for the compact form
// of the table (L == 257 only; unused otherwise).
static int rans_decode_launch(mpc_ctx *ctx, const Geometry &g, const int16_t *table, const int32_t *idx, const uint8_t *in,
                              size_t in_stride, const uint64_t *nbytes_dev, int16_t *sym_out, int32_t *status_dev,
                              void *compact_rows) {
    const int64_t G = (int64_t)g.n_images * g.n_streams;
    MPC_HIP(ctx, hipMemsetAsync(status_dev, 0, sizeof(int32_t) * g.n_images, ctx->stream));
    if (g.L == 257) {
        // compact form of the table (rebuilt per call: the caller may have changed the table)
        const int64_t rows = (int64_t)g.K * g.M;
        compact_rows_kernel<<<dim3((unsigned)mpc_cdiv(rows, 16)), dim3(256), 0, ctx->stream>>>((const uint16_t *)table, rows,
                                                                                            (uint32_t *)compact_rows);
        MPC_CHECK_LAUNCH(ctx, "rans compact_rows");
        auto kern = g.M % WAVE_DEPTH == 0 ? decode_streams_wave_kernel<true> : decode_streams_wave_kernel<false>;
        kern<<<dim3((unsigned)mpc_cdiv(G, WAVE_STREAMS_PER_BLOCK)), dim3(256), 0, ctx->stream>>>(
            g, (const uint32_t *)compact_rows, (const uint16_t *)table, idx, in, in_stride, nbytes_dev, sym_out, status_dev);
    } else {
        decode_streams_kernel<<<dim3((unsigned)mpc_cdiv(G, GROUPS_PER_BLOCK)), dim3(256), 0, ctx->stream>>>(
            g, (const uint16_t *)table, idx, in, in_stride, nbytes_dev, sym_out, status_dev);
    }
    MPC_CHECK_LAUNCH(ctx, "rans decode_streams");
    return MPC_OK;
}

static int rans_decode_check(mpc_ctx *ctx, Geometry &g, const int16_t *table, int K, int M, int L, const int32_t *idx,
                             const uint8_t *in, size_t in_stride, const uint64_t *nbytes_dev, int n_images, int64_t n_loc,
                             int n_streams, int locs_per_stream, int16_t *sym_out, int32_t *status_dev) {
    MPC_REQUIRE(ctx, table && idx && in && nbytes_dev && sym_out && status_dev, "rans_decode: NULL argument");
    int rc = check_geometry(ctx, g, K, M, L, n_images, n_loc, n_streams, locs_per_stream);
    if (rc) return rc;
    MPC_REQUIRE(ctx, ((uintptr_t)in % 4) == 0 && in_stride % 4 == 0, "rans_decode: in must be 4-byte aligned with a stride multiple of 4");
    // the stream geometry comes from an untrusted container header: a stream holds at least one location, and the
    // header (16 bytes + one 16-bit length per stream) must lie inside one image's slot
    MPC_REQUIRE(ctx, (int64_t)n_streams <= n_loc, "rans_decode: more streams than locations (corrupt header?)");
    MPC_REQUIRE(ctx, in_stride < (1ull << 32) && 16 + 2 * (size_t)n_streams <= in_stride,
                "rans_decode: in_stride smaller than the container header (or >= 4 GiB)");
    return MPC_OK;
}

extern "C" {

int mpc_rans_decode(mpc_ctx *ctx, const int16_t *table, int K, int M, int L, const int32_t *idx, const uint8_t *in,
                    size_t in_stride, const uint64_t *nbytes_dev, int n_images, int64_t n_loc, int n_streams,
                    int locs_per_stream, int16_t *sym_out, int32_t *status_dev) {
    MPC_ENTER(ctx);
    Geometry g;
    int rc = rans_decode_check(ctx, g, table, K, M, L, idx, in, in_stride, nbytes_dev, n_images, n_loc, n_streams,
                               locs_per_stream, sym_out, status_dev);
    if (rc) return rc;
    void *compact = nullptr;
    if (L == 257) {
        compact = mpc_ctx_scratch(ctx, (size_t)K * M * CROW * sizeof(uint16_t));
        if (!compact) return MPC_EHIP;
    }
    rc = rans_decode_launch(ctx, g, table, idx, in, in_stride, nbytes_dev, sym_out, status_dev, compact);
    if (L == 257) mpc_ctx_scratch_release(ctx);
    return rc;
}

// The decoder's output as the fp16 latent g_s consumes (value = symbol * q_intv + min_feat, what
// mpc_symbols_to_latent_f16 would make of mpc_rans_decode's symbols in one more pass over them).
int mpc_rans_decode_latent_f16(mpc_ctx *ctx, const int16_t *table, int K, int M, int L, const int32_t *idx, const uint8_t *in,
                               size_t in_stride, const uint64_t *nbytes_dev, int n_images, int64_t n_loc, int n_streams,
                               int locs_per_stream, int min_feat, int q_intv, void *latent_f16_out, int32_t *status_dev) {
    MPC_ENTER(ctx);
    Geometry g;
    int rc = rans_decode_check(ctx, g, table, K, M, L, idx, in, in_stride, nbytes_dev, n_images, n_loc, n_streams,
                               locs_per_stream, (int16_t *)latent_f16_out, status_dev);
    if (rc) return rc;
    MPC_REQUIRE(ctx, q_intv > 0, "rans_decode_latent_f16: q_intv must be positive");
    g.latent_f16 = 1;
    g.lat_min = min_feat;
    g.lat_q = q_intv;
    void *compact = nullptr;
    if (L == 257) {
        compact = mpc_ctx_scratch(ctx, (size_t)K * M * CROW * sizeof(uint16_t));
        if (!compact) return MPC_EHIP;
    }
    rc = rans_decode_launch(ctx, g, table, idx, in, in_stride, nbytes_dev, (int16_t *)latent_f16_out, status_dev, compact);
    if (L == 257) mpc_ctx_scratch_release(ctx);
    return rc;
}

// ---- decode without any allocation ------------------------------------------------------------------------------
// mpc_rans_decode keeps its compact copy of the table in context scratch that it allocates on demand.  Callers that
// may not allocate (code under HIP graph capture, tight serving loops) use mpc_rans_decode_noalloc instead: the same
// kernels, with the copy in the file-scope device array below.  Every GPU holds its own instance of the array at its
// own address, found once per device by mpc_rans_static_init (called from mpc_create with that device current).
int mpc_rans_decode_noalloc(mpc_ctx *ctx, const int16_t *table, int K, int M, int L, const int32_t *idx, const uint8_t *in,
                            size_t in_stride, const uint64_t *nbytes_dev, int n_images, int64_t n_loc, int n_streams,
                            int locs_per_stream, int16_t *sym_out, int32_t *status_dev) {
    MPC_ENTER(ctx);
    Geometry g;
    int rc = rans_decode_check(ctx, g, table, K, M, L, idx, in, in_stride, nbytes_dev, n_images, n_loc, n_streams,
                               locs_per_stream, sym_out, status_dev);
    if (rc) return rc;
    MPC_REQUIRE(ctx, L == 257, "rans_decode_noalloc: only tables with 257 entries per row (256 symbols) are supported");
    MPC_REQUIRE(ctx, (int64_t)K * M <= NOALLOC_MAX_ROWS,
                "rans_decode_noalloc: K*M = %lld table rows exceed the static scratch (%d rows of %d bytes); use mpc_rans_decode",
                (long long)K * M, NOALLOC_MAX_ROWS, CROW * 2);
    void *compact = ctx->device >= 0 && ctx->device < MPC_MAX_DEVICES ? static_scratch_ptr[ctx->device] : nullptr;
    MPC_REQUIRE(ctx, compact != nullptr, "rans_decode_noalloc: the static scratch of device %d was not initialised", ctx->device);
    // compact_rows_kernel rewrites every byte of the K*M rows this call reads, so the previous call's contents never show
    return rans_decode_launch(ctx, g, table, idx, in, in_stride, nbytes_dev, sym_out, status_dev, compact);
}

}  // extern "C"

// ---- per-element Laplace rows: the hyperprior's main stream without materialised rows ---------------------------------
extern "C" {

int mpc_rans_encode_laplace(mpc_ctx *ctx, const float *sigma, const int16_t *sym, float first_target, int L, int precision,
                            int n_images, int64_t n_loc, int M, int n_streams, int locs_per_stream, void *workspace,
                            size_t workspace_bytes, uint8_t *out, size_t out_stride, uint64_t *nbytes_dev,
                            int32_t *status_dev) {
    MPC_ENTER(ctx);
    MPC_REQUIRE(ctx, sigma && sym && workspace && out && nbytes_dev && status_dev, "rans_encode_laplace: NULL argument");
    MPC_REQUIRE(ctx, L > 1 && L <= 257 && precision >= 8 && precision <= 16, "rans_encode_laplace: L must be 2..257, precision 8..16");
    Geometry g;
    int rc = check_geometry(ctx, g, 1, M, L, n_images, n_loc, n_streams, locs_per_stream);
    if (rc) return rc;
    MPC_REQUIRE(ctx, locs_per_stream == (int)mpc_cdiv(n_loc, n_streams), "rans_encode_laplace: locs_per_stream must be ceil(n_loc / n_streams)");
    MPC_REQUIRE(ctx, ((uintptr_t)out % 4) == 0 && out_stride % 4 == 0 && ((uintptr_t)workspace % 256) == 0,
                "rans_encode_laplace: out must be 4-byte aligned with a stride multiple of 4; workspace 256-byte aligned");
    Workspace w = carve(workspace, n_images, n_loc, M, n_streams, locs_per_stream);
    if (w.bytes > workspace_bytes) MPC_FAIL(ctx, MPC_ENOMEM, "rans_encode_laplace: workspace too small (%zu < %zu)", workspace_bytes, w.bytes);
    MPC_REQUIRE(ctx, out_stride >= 16 + 2 * (size_t)n_streams, "rans_encode_laplace: out_stride smaller than the header");
    const int64_t G = (int64_t)g.n_images * g.n_streams;
    const int64_t total = (int64_t)g.n_images * g.n_symbols;
    MPC_REQUIRE(ctx, mpc_cdiv(total, 256) < (1ll << 31), "rans_encode_laplace: too many symbols in one call");
    const float factor = (float)(1 << precision) - (float)(L - 1);
    MPC_HIP(ctx, hipMemsetAsync(status_dev, 0, sizeof(int32_t) * g.n_images, ctx->stream));
    symbol_params_laplace_kernel<<<dim3((unsigned)mpc_cdiv(total, 256)), dim3(256), 0, ctx->stream>>>(g, sigma, first_target, factor, sym,
                                                                                                   w.params, status_dev);
    MPC_CHECK_LAUNCH(ctx, "rans symbol_params_laplace");
    encode_streams_kernel<<<dim3((unsigned)mpc_cdiv(G, 64)), dim3(64), 0, ctx->stream>>>(g, w.params, w.scratch, w.lens, nullptr);
    MPC_CHECK_LAUNCH(ctx, "rans encode_streams");
    header_kernel<<<dim3((unsigned)g.n_images), dim3(256), 0, ctx->stream>>>(g, w.lens, w.offs, out, out_stride, nbytes_dev, status_dev, nullptr);
    MPC_CHECK_LAUNCH(ctx, "rans header");
    copy_streams_kernel<<<dim3((unsigned)G), dim3(256), 0, ctx->stream>>>(g, w.scratch, w.lens, w.offs, out, out_stride, nbytes_dev, nullptr);
    MPC_CHECK_LAUNCH(ctx, "rans copy_streams");
    return MPC_OK;
}

int mpc_rans_decode_laplace(mpc_ctx *ctx, const float *sigma, float first_target, int L, int precision, const uint8_t *in,
                            size_t in_stride, const uint64_t *nbytes_dev, int n_images, int64_t n_loc, int M, int n_streams,
                            int locs_per_stream, int16_t *sym_out, int32_t *status_dev) {
    MPC_ENTER(ctx);
    MPC_REQUIRE(ctx, sigma != nullptr, "rans_decode_laplace: NULL argument");
    MPC_REQUIRE(ctx, L > 1 && L <= 257 && precision >= 8 && precision <= 16, "rans_decode_laplace: L must be 2..257, precision 8..16");
    Geometry g;
    int rc = rans_decode_check(ctx, g, (const int16_t *)sigma, 1, M, L, (const int32_t *)sigma, in, in_stride, nbytes_dev, n_images,
                               n_loc, n_streams, locs_per_stream, sym_out, status_dev);
    if (rc) return rc;
    const int64_t G = (int64_t)g.n_images * g.n_streams;
    const float factor = (float)(1 << precision) - (float)(L - 1);
    MPC_HIP(ctx, hipMemsetAsync(status_dev, 0, sizeof(int32_t) * g.n_images, ctx->stream));
    decode_streams_laplace_kernel<<<dim3((unsigned)mpc_cdiv(G, LAPLACE_WAVES_PER_BLOCK)), dim3(256), 0, ctx->stream>>>(
        g, sigma, first_target, factor, in, in_stride, nbytes_dev, sym_out, status_dev);
    MPC_CHECK_LAUNCH(ctx, "rans decode_streams_laplace");
    return MPC_OK;
}

}  // extern "C"

int mpc_rans_static_init(mpc_ctx *ctx) {
    const int d = ctx->device;
    if (d < 0 || d >= MPC_MAX_DEVICES) return MPC_OK;            // such a device gets no static scratch (the entry point says so)
    if (static_scratch_ptr[d]) return MPC_OK;
    void *p = nullptr;
    MPC_HIP(ctx, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    static_scratch_ptr[d] = p;
    return MPC_OK;
}
