// Prior competition and per-symbol CDF row gather.
//
// prior_argmin: one wavefront per latent location, lane k = prior k.  For each of the M channels the
// wave reads one 64-float LUT row lut[c][sym][0..63] (256 B, coalesced) and adds it to a per-lane
// running total; the 64 totals are then reduced to (min, first argmin).  Compulsory HBM traffic is the
// 2 B symbol; the 4*K = 256 B/symbol of LUT traffic is served by L2 / Infinity Cache (the 16.8 MB LUT
// is shared by every wave and real latents concentrate on a few symbols per channel).
//
// The fp32 association of the channel sum reproduces ATen's cascade sum over dim 2 of the
// [K,B,M,h,w] tensor that the reference reduces (src/compression/models/manynets_compressor.py:551-553):
// 16 consecutive channels are added sequentially into a level-0 accumulator, which is then added into
// a level-1 accumulator (and every 256 channels level 1 into level 2).  With the same LUT values the
// totals are therefore bit-identical to the reference's, and so is the argmin.
//
// gather_cdf: cdf[loc][c][:] = table[idx[loc]][c][:].  table[k] is one contiguous M*L*2-byte block, so
// a location's rows are a straight block copy.  Pure HBM-write streaming: 2*L bytes per symbol; the only reads
// are the table's.  The locations are first ordered by prior index (a counting sort, three small launches), then
// one 1024-thread workgroup per location, in that order, loads the prior's block into registers and writes it
// with 16-byte non-temporal stores: workgroups that run together copy the same few priors, so the table reads
// are L2 hits instead of 128 KB per location from beyond L2 (measured at 36 864 locations: 6.0 TB/s against
// 4.5 TB/s for the unordered copy; a device memset reaches 6.4 TB/s).
#include "mpc_internal.h"

namespace {

struct Cand {
    float v;
    int k;
};

// torch.min semantics: NaN beats everything, otherwise smaller value, ties -> lower index.
__device__ __forceinline__ Cand better(Cand a, Cand b) {
    bool an = a.v != a.v, bn = b.v != b.v;
    bool a_wins;
    if (an || bn) a_wins = (an && bn) ? (a.k < b.k) : an;
    else a_wins = (a.v < b.v) || (a.v == b.v && a.k < b.k);
    return a_wins ? a : b;
}

constexpr int WAVES_PER_BLOCK = 4;

__global__ __launch_bounds__(256) void prior_argmin_kernel(const int16_t *__restrict__ sym, int64_t n_loc, int M,
                                                            int K, int S, const float *__restrict__ lut,
                                                            int32_t *__restrict__ idx, float *__restrict__ minbits,
                                                            float *__restrict__ total_bits) {
    const int lane = threadIdx.x & 63;
    const int64_t loc = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (loc >= n_loc) return;                       // whole wave exits together
    const int16_t *srow = sym + loc * M;
    const bool active = lane < K;
    const float *lane_lut = lut + (active ? lane : 0);

    float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;
    const unsigned smax = (unsigned)(S - 1);
    int c = 0;
    if ((M & 15) == 0 && (((uintptr_t)sym) & 15) == 0) {
        for (; c + 16 <= M; c += 16) {
            // 16 symbols of this location: 32 contiguous bytes, identical for every lane of the wave
            const uint4 *sp = (const uint4 *)(srow + c);
            uint4 q0 = sp[0], q1 = sp[1];
            unsigned w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            float v[16];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                unsigned ww = (unsigned)__builtin_amdgcn_readfirstlane((int)w[j]);
                unsigned s0 = min(ww & 0xffffu, smax), s1 = min(ww >> 16, smax);
                v[2 * j] = lane_lut[((int64_t)(c + 2 * j) * S + s0) * K];
                v[2 * j + 1] = lane_lut[((int64_t)(c + 2 * j + 1) * S + s1) * K];
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) acc0 = __fadd_rn(acc0, v[j]);
            acc1 = __fadd_rn(acc1, acc0);
            acc0 = 0.0f;
            if (((c + 16) & 0xF0) == 0) {           // every 256 channels level 1 drains into level 2
                acc2 = __fadd_rn(acc2, acc1);
                acc1 = 0.0f;
            }
        }
    }
    for (; c < M; ++c) {
        // generic tail / unaligned path: same association (16-channel blocks), one symbol at a time
        unsigned s = min((unsigned)(uint16_t)__builtin_amdgcn_readfirstlane((int)srow[c]), smax);
        acc0 = __fadd_rn(acc0, lane_lut[((int64_t)c * S + s) * K]);
        if (((c + 1) & 15) == 0 && c + 1 + 0 <= (M & ~15)) {
            acc1 = __fadd_rn(acc1, acc0);
            acc0 = 0.0f;
            if (((c + 1) & 0xF0) == 0) {
                acc2 = __fadd_rn(acc2, acc1);
                acc1 = 0.0f;
            }
        }
    }
    float total = __fadd_rn(__fadd_rn(acc0, acc1), acc2);

    if (total_bits && active) total_bits[(int64_t)lane * n_loc + loc] = total;

    Cand best;
    best.v = active ? total : __builtin_inff();
    best.k = active ? lane : 0x7fffffff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Cand o;
        o.v = __shfl_xor(best.v, off, 64);
        o.k = __shfl_xor(best.k, off, 64);
        best = better(best, o);
    }
    if (lane == 0) {
        idx[loc] = best.k;
        minbits[loc] = best.v;
    }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int SORT_MAX_K = 4096;           // priors the counting sort handles (LDS histogram)
constexpr int SORT_CHUNK = 1024;           // locations per workgroup of the sort kernels
constexpr int GATHER_T = 1024;             // threads of the copy workgroup

__device__ __forceinline__ int clamp_prior(int k, int K) { return k < 0 ? 0 : (k >= K ? K - 1 : k); }

// counts[k] = number of locations with prior k
__global__ __launch_bounds__(SORT_CHUNK) void prior_histogram_kernel(const int32_t *__restrict__ idx, int64_t n_loc, int K,
                                                                      int32_t *__restrict__ counts) {
    extern __shared__ int32_t lds_cnt[];
    for (int k = threadIdx.x; k < K; k += SORT_CHUNK) lds_cnt[k] = 0;
    __syncthreads();
    const int64_t loc = (int64_t)blockIdx.x * SORT_CHUNK + threadIdx.x;
    if (loc < n_loc) atomicAdd(&lds_cnt[clamp_prior(idx[loc], K)], 1);
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += SORT_CHUNK)
        if (lds_cnt[k]) atomicAdd(&counts[k], lds_cnt[k]);
}

// cursor[k] = number of locations with a prior below k (one workgroup)
__global__ __launch_bounds__(256) void prior_scan_kernel(const int32_t *__restrict__ counts, int K, int32_t *__restrict__ cursor) {
    __shared__ int32_t part[256];
    const int t = threadIdx.x, per = (K + 255) / 256;
    int32_t sum = 0;
    for (int j = 0; j < per; ++j) {
        int k = t * per + j;
        if (k < K) sum += counts[k];
    }
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        int32_t a = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += a;
        __syncthreads();
    }
    int32_t run = part[t] - sum;
    for (int j = 0; j < per; ++j) {
        int k = t * per + j;
        if (k < K) {
            cursor[k] = run;
            run += counts[k];
        }
    }
}

// order[...] = the locations grouped by prior (any order inside a group)
__global__ __launch_bounds__(SORT_CHUNK) void prior_scatter_kernel(const int32_t *__restrict__ idx, int64_t n_loc, int K,
                                                                    int32_t *__restrict__ cursor, int32_t *__restrict__ order) {
    extern __shared__ int32_t lds[];                       // [K] counts of this chunk, then [K] their bases
    int32_t *cnt = lds, *base = lds + K;
    for (int k = threadIdx.x; k < K; k += SORT_CHUNK) cnt[k] = 0;
    __syncthreads();
    const int64_t loc = (int64_t)blockIdx.x * SORT_CHUNK + threadIdx.x;
    int k = 0, rank = 0;
    if (loc < n_loc) {
        k = clamp_prior(idx[loc], K);
        rank = atomicAdd(&cnt[k], 1);
    }
    __syncthreads();
    for (int q = threadIdx.x; q < K; q += SORT_CHUNK)
        if (cnt[q]) base[q] = atomicAdd(&cursor[q], cnt[q]);
    __syncthreads();
    if (loc < n_loc) order[base[k] + rank] = (int32_t)loc;
}

// one workgroup per location, taken in prior order; R 16-byte vectors per thread
__global__ __launch_bounds__(GATHER_T) void gather_cdf_sorted_kernel(const u32x4 *__restrict__ table, int64_t vec_per_prior,
                                                                      const int32_t *__restrict__ idx, const int32_t *__restrict__ order,
                                                                      int K, u32x4 *__restrict__ out) {
    const int64_t loc = order[blockIdx.x];
    const u32x4 *src = table + (int64_t)clamp_prior(idx[loc], K) * vec_per_prior;
    u32x4 *dst = out + loc * vec_per_prior;
    for (int64_t i0 = 0; i0 < vec_per_prior; i0 += 12 * GATHER_T) {
        u32x4 r[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int64_t i = i0 + (int64_t)j * GATHER_T + threadIdx.x;
            if (i < vec_per_prior) r[j] = src[i];
        }
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int64_t i = i0 + (int64_t)j * GATHER_T + threadIdx.x;
            if (i < vec_per_prior) __builtin_nontemporal_store(r[j], dst + i);
        }
    }
}

// unordered form (more priors than the sort handles): one 256-thread workgroup per location
__global__ __launch_bounds__(256) void gather_cdf_vec_kernel(const u32x4 *__restrict__ table, int64_t vec_per_prior,
                                                              const int32_t *__restrict__ idx, int K,
                                                              u32x4 *__restrict__ out) {
    const int64_t loc = blockIdx.x;
    const u32x4 *src = table + (int64_t)clamp_prior(idx[loc], K) * vec_per_prior;
    u32x4 *dst = out + loc * vec_per_prior;
    int64_t i = threadIdx.x;
    for (; i + 3 * 256 < vec_per_prior; i += 4 * 256) {
        u32x4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
        __builtin_nontemporal_store(a, dst + i);
        __builtin_nontemporal_store(b, dst + i + 256);
        __builtin_nontemporal_store(c, dst + i + 512);
        __builtin_nontemporal_store(d, dst + i + 768);
    }
    for (; i < vec_per_prior; i += 256) __builtin_nontemporal_store(src[i], dst + i);
}

__global__ __launch_bounds__(256) void gather_cdf_scalar_kernel(const int16_t *__restrict__ table, int64_t per_prior,
                                                                 const int32_t *__restrict__ idx, int K,
                                                                 int16_t *__restrict__ out) {
    const int64_t loc = blockIdx.x;
    const int16_t *src = table + (int64_t)clamp_prior(idx[loc], K) * per_prior;
    int16_t *dst = out + loc * per_prior;
    for (int64_t i = threadIdx.x; i < per_prior; i += 256) dst[i] = src[i];
}

}  // namespace

extern "C" {

int mpc_prior_argmin(mpc_ctx *ctx, const int16_t *sym, int64_t n_loc, int M, int K, int S, const float *lut,
                     int32_t *idx, float *minbits, float *total_bits) {
    MPC_ENTER(ctx);
    MPC_REQUIRE(ctx, sym && lut && idx && minbits && n_loc > 0 && M > 0 && M < 4096 && K > 0 && K <= 64 && S > 0,
                "prior_argmin: bad arguments (need 0 < K <= 64, 0 < M < 4096)");
    int64_t blocks = mpc_cdiv(n_loc, WAVES_PER_BLOCK);
    MPC_REQUIRE(ctx, blocks < (1ll << 31), "prior_argmin: too many locations");
    prior_argmin_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(sym, n_loc, M, K, S, lut, idx, minbits, total_bits);
    MPC_CHECK_LAUNCH(ctx, "prior_argmin");
    return MPC_OK;
}

int mpc_gather_cdf(mpc_ctx *ctx, const int16_t *table, int K, int M, int L, const int32_t *idx, int64_t n_loc,
                   int16_t *out) {
    MPC_ENTER(ctx);
    MPC_REQUIRE(ctx, table && idx && out && K > 0 && M > 0 && L > 1 && n_loc > 0 && n_loc < (1ll << 31),
                "gather_cdf: bad arguments");
    int64_t per_prior = (int64_t)M * L;
    bool vec_ok = (per_prior * 2) % 16 == 0 && ((uintptr_t)table % 16) == 0 && ((uintptr_t)out % 16) == 0;
    if (vec_ok && K <= SORT_MAX_K) {
        // scratch: order[n_loc], counts[K], cursor[K]
        int32_t *order = (int32_t *)mpc_ctx_scratch(ctx, sizeof(int32_t) * ((size_t)n_loc + 2 * (size_t)K));
        if (!order) return MPC_ENOMEM;
        int32_t *counts = order + n_loc, *cursor = counts + K;
        const unsigned chunks = (unsigned)mpc_cdiv(n_loc, SORT_CHUNK);
        MPC_HIP(ctx, hipMemsetAsync(counts, 0, sizeof(int32_t) * K, ctx->stream));
        prior_histogram_kernel<<<dim3(chunks), dim3(SORT_CHUNK), sizeof(int32_t) * K, ctx->stream>>>(idx, n_loc, K, counts);
        prior_scan_kernel<<<dim3(1), dim3(256), 0, ctx->stream>>>(counts, K, cursor);
        prior_scatter_kernel<<<dim3(chunks), dim3(SORT_CHUNK), sizeof(int32_t) * 2 * K, ctx->stream>>>(idx, n_loc, K, cursor, order);
        gather_cdf_sorted_kernel<<<dim3((unsigned)n_loc), dim3(GATHER_T), 0, ctx->stream>>>(
            (const u32x4 *)table, per_prior * 2 / 16, idx, order, K, (u32x4 *)out);
        mpc_ctx_scratch_release(ctx);
    } else if (vec_ok) {
        gather_cdf_vec_kernel<<<dim3((unsigned)n_loc), dim3(256), 0, ctx->stream>>>(
            (const u32x4 *)table, per_prior * 2 / 16, idx, K, (u32x4 *)out);
    } else {
        gather_cdf_scalar_kernel<<<dim3((unsigned)n_loc), dim3(256), 0, ctx->stream>>>(table, per_prior, idx, K, out);
    }
    MPC_CHECK_LAUNCH(ctx, "gather_cdf");
    return MPC_OK;
}

}  // extern "C"
