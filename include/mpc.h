/*
 * mpc.h -- C ABI of libmpc.so, the MI355X (gfx950) implementation of the Manypriors / JDC codec
 * inference path.
 *
 * The reference (trougnouf/compression) is pure Python over PyTorch and has no FFI of its own; the
 * native seams on its inference path are the ATen operators it calls and the third-party `torchac`
 * extension.  Each entry point below replaces one such call site (cited as file:line relative to the
 * reference root).  Conventions:
 *   - every function returns 0 on success or a negative MPC_E* code; mpc_last_error(ctx) gives text;
 *   - no C++ types, no torch types, no exceptions cross this boundary;
 *   - all `*_dev` / unmarked data pointers are DEVICE pointers on the context's GPU unless the
 *     parameter is documented as host memory; the caller owns every buffer it passes;
 *   - work is enqueued on the context's HIP stream and is asynchronous unless stated otherwise;
 *   - a context is not thread-safe; distinct contexts may be used from distinct threads;
 *   - a context owns one device scratch buffer that grows on demand (the decoder's compact copy of the CDF
 *     table); calls on one context must therefore not overlap in time on different streams.
 *
 * Activations are NHWC ([B][H][W][C], C contiguous): fp32 on the exact path, fp16 between MPC_F16 layers, the
 * split format described at MPC_F16X3 between split-precision layers.  The quantised latent in NHWC IS the
 * reference's symbol order (location-major, channel-minor,
 * src/compression/models/manynets_compressor.py:264-269).
 */
#ifndef MPC_H
#define MPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpc_ctx mpc_ctx;
typedef struct mpc_conv mpc_conv;

enum {
    MPC_OK = 0,
    MPC_EINVAL = -1,   /* bad argument / unsupported shape */
    MPC_EHIP = -2,     /* a HIP runtime call failed */
    MPC_ENOMEM = -3,   /* device allocation failed or caller buffer too small */
    MPC_EDATA = -4     /* data-dependent failure reported by a kernel (e.g. corrupt bitstream) */
};

/* ---- context ------------------------------------------------------------------------------- */
/* `stream` is a hipStream_t (NULL = the default stream).  Replaces the reference's implicit
 * torch.cuda.set_device(device) (src/compression/libs/initfun.py:265-268). */
int mpc_create(int device_id, void *stream, mpc_ctx **out);
int mpc_destroy(mpc_ctx *ctx);
int mpc_set_stream(mpc_ctx *ctx, void *stream);
int mpc_synchronize(mpc_ctx *ctx);
const char *mpc_last_error(const mpc_ctx *ctx);
/* Library build string ("libmpc gfx950 ..."). */
const char *mpc_version(void);

/* Launches a one-thread kernel named mpc_profile_mark_kernel on the context's stream: a mark in a rocprofv3 kernel
 * trace (bench.py brackets its timed region with two of them).  No counterpart in the reference, whose timing is
 * time.perf_counter on the host (src/compression/tests.py:356-372). */
int mpc_profile_mark(mpc_ctx *ctx, int id);

/* ---- layout helpers ------------------------------------------------------------------------ */
/* NCHW fp32 [B,C,H,W] -> NHWC [B,H,W,Cpad] (channels C..Cpad-1 zero).  The reference keeps NCHW
 * throughout (e.g. Balle2018PT_compressor.py:98-102); this is the entry conversion. */
int mpc_nchw_to_nhwc(mpc_ctx *ctx, const float *in, int B, int C, int H, int W, int Cpad, float *out);
/* NHWC [B,H,W,C] -> NCHW [B,C,H,W]. */
int mpc_nhwc_to_nchw(mpc_ctx *ctx, const float *in, int B, int C, int H, int W, float *out);
/* sym[i] = clamp(rint((latent[i] - min_feat) / q_intv), 0, n_sym_values-1) as int16;
 * *n_clamped_dev (int32, device, may be NULL) receives the number of clamped values.
 * manynets_compressor.py:264-269 (the reference widens its tables instead of clamping, :212-227;
 * the caller can reproduce that from n_clamped). */
int mpc_latent_to_symbols(mpc_ctx *ctx, const float *latent, int64_t n, int min_feat, int q_intv,
                          int n_sym_values, int16_t *sym, int32_t *n_clamped_dev);
/* latent[i] = sym[i] * q_intv + min_feat.  manynets_compressor.py:329-330. */
int mpc_symbols_to_latent(mpc_ctx *ctx, const int16_t *sym, int64_t n, int min_feat, int q_intv,
                          float *latent);
/* Double-precision sum of n floats into *out_dev (device double). Used for sum(minbits) -> bpp_feature
 * (manynets_compressor.py:604-607, 634). */
int mpc_sum_f32(mpc_ctx *ctx, const float *x, int64_t n, double *out_dev);

/* xhat[i] = clamp(recon[i], 0, 1) (xhat may be NULL) and *sse_dev += sum (recon[i]-target[i])^2 in double
 * (sse_dev may be NULL; zeroed by this call).  manynets_compressor.py:537 and the MSE visual loss of
 * :543 (F.mse_loss = sse / n). */
int mpc_clamp_sse(mpc_ctx *ctx, const float *recon, const float *target, int64_t n, float *xhat,
                  double *sse_dev);

/* The same for n_images images of n_per_image elements each, with one sum per image: sse_dev[i].  score_clamped != 0
 * takes the error on the clamped value: the per-image MSE / PSNR that the reference's test harness logs from
 * "reconstructed_image" (src/compression/tests.py:393-398); 0 takes it on the raw reconstruction (the visual loss). */
int mpc_clamp_sse_batched(mpc_ctx *ctx, const float *recon, const float *target, int n_images,
                          int64_t n_per_image, int score_clamped, float *xhat, double *sse_dev);

/* latent[i] as fp16 (the input type of an MPC_F16 layer stack), same arithmetic as mpc_symbols_to_latent. */
int mpc_symbols_to_latent_f16(mpc_ctx *ctx, const int16_t *sym, int64_t n, int min_feat, int q_intv,
                              void *latent_f16);
/* Asynchronous copy of `rows` rows of `width` bytes from device memory (row pitch src_pitch) to HOST memory (row pitch
 * dst_pitch; page-locked memory keeps it asynchronous) on the context's stream: brings the used part of every
 * image's worst-case-sized bitstream slot to the host in one call (the reference copies the whole CDF tensor to
 * the host instead, manynets_compressor.py:274). */
int mpc_download_rows(mpc_ctx *ctx, const void *src_dev, size_t src_pitch, size_t width, size_t rows,
                      void *dst_host, size_t dst_pitch);

/* ---- image-quality statistics of the reference's test harness (src/compression/tests.py:393-418) ------------- */
/* Sums over the valid region ((H-win_size+1) x (W-win_size+1)) of the SSIM map and of the contrast-structure map
 * of every plane of x and y ([planes][H][W] fp32, planes = batch*channels), with the separable window `window`
 * (host array of win_size floats, odd, <= 15) and the stabilisers C1 = (K1*data_range)^2, C2 = (K2*data_range)^2.
 * The caller divides by the region size.  Replaces _ssim (src/common/extlibs/pt_ms_ssim.py:41-84). */
int mpc_ssim_stats(mpc_ctx *ctx, const float *x, const float *y, int planes, int H, int W, const float *window,
                   int win_size, float C1, float C2, double *ssim_sum_dev, double *cs_sum_dev);
/* F.avg_pool2d(kernel_size=2, padding=(H%2, W%2)) between the MS-SSIM scales (pt_ms_ssim.py:205-207):
 * out [planes][Ho][Wo], Ho = (H + 2*(H%2) - 2)/2 + 1. */
int mpc_avg_pool2(mpc_ctx *ctx, const float *in, int64_t planes, int H, int W, float *out);

/* Adds one batch to the running sums that the reference's test harness keeps per evaluated directory
 * (src/compression/tests.py:439-460) and that the multi-GPU driver all-reduces.  vec_dev: 16 doubles on the device,
 * [0] bits, [1] pixels, [2] squared error, [3] sum of per-image PSNR, [4] images, [5] bytes, [6] seconds,
 * [7] side-string bits, [8] sum of per-image MS-SSIM, [9] sum of per-image MS-SSIM in dB, [10..15] unused.
 * sse_dev[i]: squared error of image i over values_per_image (= 3*H*W) values (mpc_clamp_sse_batched);
 * msssim_dev[i] (may be NULL): its MS-SSIM; host_sums (host array) = {bits, pixels, bytes, seconds, side_bits} of the
 * batch; reset != 0 starts the sums from zero instead of adding to vec_dev.  PSNR = 10 log10(1 / mse), MS-SSIM dB =
 * -10 log10(1 - msssim) (tests.py:393-412). */
int mpc_stats_accumulate(mpc_ctx *ctx, const double *sse_dev, const double *msssim_dev, int n_images,
                         int64_t values_per_image, const double host_sums[5], int reset, double *vec_dev);

/* ---- learned CDF tables -------------------------------------------------------------------- */
/* Float CDF table of a (multi-head) factorised prior evaluated at targets first_target + s, s < L:
 * table_f[row][s], row < n_rows, with the 4-layer Bitparm chain of bitEstimator.py:136-141.
 * params[11] = {h1,b1,a1,h2,b2,a2,h3,b3,a3,h4,b4}, each n_rows contiguous floats (the flattened
 * [K,1,M,1,1] tensors, row = k*M + c).  manynets_compressor.py:142-152; abstract_model.py:277-284. */
int mpc_bitparm_cdf_table(mpc_ctx *ctx, const float *const params[11], int64_t n_rows,
                          float first_target, int L, float *table_f);
/* Per-element evaluation of the same chain: out[i] = CDF_row(i)(x[i] + offset).  x is [n_rows][n_per_row]
 * (row_minor = 0) or [n_per_row][n_rows] (row_minor = 1: the channel-minor NHWC layout).
 * bitEstimator.py:41-45 as used by abstract_model.py:238. */
int mpc_bitparm_cdf_eval(mpc_ctx *ctx, const float *const params[11], int64_t n_rows,
                         const float *x, int64_t n_per_row, int row_minor, float offset, float *out);

#define MPC_ORDER_ROUND_THEN_ADD 0 /* torchac._renorm_cast_cdf_ (call site manynets_compressor.py:153) */
#define MPC_ORDER_ADD_THEN_ROUND 1 /* abstract_model.py:59-73 */
/* int16 CDF rows from float CDF rows, both rounding orders, wrap-around cast. */
int mpc_renorm_cast_cdf(mpc_ctx *ctx, const float *cdf, int64_t n_rows, int L, int precision, int order,
                        int16_t *out);

#define MPC_BITS_LOG2 0 /* clamp(-log2(p + 1e-10), 0, 50)        manynets_compressor.py:551-553 */
#define MPC_BITS_LN 1   /* clamp(-1.0*log(p + 1e-10)/log(2), 0, 50)  manynets_compressor.py:247-248 */
/* Code-length LUT in competition layout: lut[(c*S + s)*K + k] = bits of symbol s of channel c under
 * prior k, S = L-1, from table_f[K][M][L]. */
int mpc_bits_lut(mpc_ctx *ctx, const float *table_f, int K, int M, int L, int form, float *lut);

/* ---- prior competition --------------------------------------------------------------------- */
/* For every location: total[k] = sum_c lut[c][sym[loc][c]][k] (channel order and fp32 association of
 * ATen's sum over dim 2 of a [K,B,M,h,w] tensor), idx = first argmin_k, minbits = min_k.
 * sym is [n_loc][M] int16.  total_bits (may be NULL) is [K][n_loc].
 * manynets_compressor.py:548-554 and :238-249.  K must be 64 or less. */
int mpc_prior_argmin(mpc_ctx *ctx, const int16_t *sym, int64_t n_loc, int M, int K, int S,
                     const float *lut, int32_t *idx, float *minbits, float *total_bits);
/* The same competition in its network form, for latents that are not integers: total[k] = sum_c
 * clamp(-log2(P_kc(v + .5) - P_kc(v - .5) + 1e-10), 0, 50) with v = y[loc][c] (+ noise[loc][c] when noise is given),
 * the Bitparm chains of params[11] (K*M rows, row = k*M + c) evaluated at v itself.  y and noise are [n_loc][M] fp32
 * (NHWC); y_entropy (may be NULL) receives v; total_bits [K][n_loc], idx, minbits as in mpc_prior_argmin (each may be
 * NULL).  Training-mode forward: manynets_compressor.py:514-530 (uniform-noise quantisation) and :547-554. */
int mpc_prior_bits_network(mpc_ctx *ctx, const float *const params[11], int K, int M, const float *y,
                           const float *noise, int64_t n_loc, float *y_entropy, float *total_bits, int32_t *idx,
                           float *minbits);

/* ---- per-symbol CDF rows ("CDF build") ------------------------------------------------------ */
/* out[loc][c][:] = table[idx[loc]][c][:]   (int16 [n_loc][M][L]).  manynets_compressor.py:252-262,
 * :317-327, :707-710.  Only parity tests and the reference-style torchac hand-off need the
 * materialised rows; the rANS coder below reads the table directly.  Orders the locations by prior index first
 * (4*(n_loc + 2K) bytes of context scratch), so that concurrent copies read the same table blocks. */
int mpc_gather_cdf(mpc_ctx *ctx, const int16_t *table, int K, int M, int L, const int32_t *idx,
                   int64_t n_loc, int16_t *out);

/* ---- entropy coder (replaces torchac.encode_cdf / decode_cdf, manynets_compressor.py:274,328) -- */
/* MRS1 containers: see oracle/rans_ref.c for the byte-exact format.  A batch of n_images
 * independent images is coded in one call; image i uses idx + i*n_loc, sym + i*n_loc*M, writes its
 * container at out + i*out_stride and its size to nbytes_dev[i].  Each image is cut into n_streams
 * streams of locs_per_stream locations (last stream shorter).  (locs_per_stream*M + 2) <= 65535. */
size_t mpc_rans_workspace_bytes(int n_images, int64_t n_loc, int M, int n_streams);
int mpc_rans_encode(mpc_ctx *ctx, const int16_t *table, int K, int M, int L, const int32_t *idx,
                    const int16_t *sym, int n_images, int64_t n_loc, int n_streams, int locs_per_stream,
                    void *workspace, size_t workspace_bytes, uint8_t *out, size_t out_stride,
                    uint64_t *nbytes_dev, int32_t *status_dev);
/* mpc_rans_encode with the stream geometry chosen ON THE DEVICE, so that the host does not have to read the bit
 * estimate back before it can launch the coder: est = sum(minbits) / 8 / n_images bytes per image (minbits: the
 * n_images*n_loc minima of mpc_prior_argmin), n_streams = the largest count whose 6 bytes per stream stay within
 * `budget` (e.g. 0.004) of est, at least what the 16-bit stream length needs, at most max_streams and n_loc;
 * locs_per_stream = ceil(n_loc / n_streams), empty streams dropped.  geometry_dev (device int32[2]) receives
 * {n_streams, locs_per_stream}; the containers' headers carry them too.  out_stride must hold the largest header
 * (16 + 2*max_streams) plus the payload; the workspace is sized by mpc_rans_auto_workspace_bytes. */
size_t mpc_rans_auto_workspace_bytes(int n_images, int64_t n_loc, int M, int max_streams);
int mpc_rans_encode_auto(mpc_ctx *ctx, const int16_t *table, int K, int M, int L, const int32_t *idx,
                         const int16_t *sym, const float *minbits, int n_images, int64_t n_loc, double budget,
                         int max_streams, void *workspace, size_t workspace_bytes, uint8_t *out,
                         size_t out_stride, uint64_t *nbytes_dev, int32_t *status_dev, int32_t *geometry_dev);
/* Decodes n_images containers (in + i*in_stride, nbytes_dev[i] bytes; nbytes_dev is a device array)
 * into sym_out + i*n_loc*M.  n_streams / locs_per_stream are the values the containers were written
 * with (fields 1 and 3 of the header; every image of the batch must share them).  status_dev[i]
 * (device int32) is 0 or a negative detail code of MPC_EDATA.  Tables with L = 257 (256-symbol rows, the standard
 * model) take the fast path: one wave per stream, the coder state in scalar registers; it works on a compact copy
 * of the table made by this call in the context's scratch (K*M*256 bytes: per row the 16 segment heads and the 64
 * entries of the four heaviest consecutive segments; other segments are read from `table` itself). */
int mpc_rans_decode(mpc_ctx *ctx, const int16_t *table, int K, int M, int L, const int32_t *idx,
                    const uint8_t *in, size_t in_stride, const uint64_t *nbytes_dev, int n_images,
                    int64_t n_loc, int n_streams, int locs_per_stream, int16_t *sym_out,
                    int32_t *status_dev);

/* mpc_rans_decode with the output written as the fp16 latent the synthesis transform reads: value =
 * symbol * q_intv + min_feat (manynets_compressor.py:330 turns the decoded symbols into the latent with one more
 * pass; here the decoder stores it directly).  latent_f16_out: n_images * n_loc * M fp16 values, NHWC. */
int mpc_rans_decode_latent_f16(mpc_ctx *ctx, const int16_t *table, int K, int M, int L, const int32_t *idx,
                               const uint8_t *in, size_t in_stride, const uint64_t *nbytes_dev, int n_images,
                               int64_t n_loc, int n_streams, int locs_per_stream, int min_feat, int q_intv,
                               void *latent_f16_out, int32_t *status_dev);

/* Same as mpc_rans_decode, for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc): it calls no allocating routine.  It has no counterpart in the
 * reference.  The compact table copy lives in a static device array of the library instead of context scratch:
 * L must be 257 and K*M at most 16384 rows (4 MiB per GPU, held until the process ends), otherwise MPC_EINVAL.
 * All calls on one GPU share that array: calls that overlap in time on one GPU (two streams, two contexts) race. */
int mpc_rans_decode_noalloc(mpc_ctx *ctx, const int16_t *table, int K, int M, int L, const int32_t *idx,
                            const uint8_t *in, size_t in_stride, const uint64_t *nbytes_dev, int n_images,
                            int64_t n_loc, int n_streams, int locs_per_stream, int16_t *sym_out,
                            int32_t *status_dev);

/* The coder over per-element rows that are never materialised: the hyperprior's main stream, whose symbol i has the
 * int16 CDF row of Laplace(0, clamp(sigma[i], 1e-10, 1e10)) at the targets first_target + s, s < L, "add arange, then
 * round" order -- the rows mpc_laplace_cdf_rows would write (abstract_model.py:402-416), handed to torchac.encode_cdf
 * at abstract_model.py:423.  sigma and sym are [n_images][n_loc][M] (NHWC); the encoder evaluates the two row entries
 * of every symbol, the decoder the 32 entries its search looks at; the only per-symbol scratch is the encoder's 8-byte
 * (start, frequency, reciprocal) entry in `workspace` (mpc_rans_workspace_bytes).  L at most 257.  Containers, geometry arguments
 * and status codes as in mpc_rans_encode / mpc_rans_decode. */
int mpc_rans_encode_laplace(mpc_ctx *ctx, const float *sigma, const int16_t *sym, float first_target, int L,
                            int precision, int n_images, int64_t n_loc, int M, int n_streams, int locs_per_stream,
                            void *workspace, size_t workspace_bytes, uint8_t *out, size_t out_stride,
                            uint64_t *nbytes_dev, int32_t *status_dev);
int mpc_rans_decode_laplace(mpc_ctx *ctx, const float *sigma, float first_target, int L, int precision,
                            const uint8_t *in, size_t in_stride, const uint64_t *nbytes_dev, int n_images,
                            int64_t n_loc, int M, int n_streams, int locs_per_stream, int16_t *sym_out,
                            int32_t *status_dev);

/* ---- transform layers ---------------------------------------------------------------------- */
enum { MPC_CONV = 0, MPC_DECONV = 1 };
enum {
    MPC_ACT_NONE = 0,
    MPC_ACT_GDN = 1,  /* y / sqrt(beta + gamma y^2)          GDN.py:83-91 */
    MPC_ACT_IGDN = 2, /* y * sqrt(beta + gamma y^2) */
    MPC_ACT_RELU = 3,
    MPC_ACT_EXP = 4   /* SynthesisTFCodeExp_prior_net, Balle2018PT_compressor.py:248-251 */
};
enum {
    MPC_F16 = 16,          /* layer flag: fp16 operands / fp32 accumulate; activations in and out are fp16 NHWC
                              (the small-cout RGB layer still writes fp32 NCHW).  For transforms whose output
                              tolerance allows it (g_s: x_hat within 1e-2); g_a must stay fp32. */
    MPC_F16X3 = 32,        /* layer flag: split-precision fp16 MFMA (a = hi + lo, three MFMAs per product, fp32
                              accumulate) -- fp32-grade results on the fp16 matrix pipe, for g_a.  The INPUT
                              is in the split format "S16": [pixel][C/8][hi x8 | lo x8] fp16, scaled by 2^8
                              (4 bytes per element); the output is fp32 NHWC or, with MPC_OUT_SPLIT, S16. */
    MPC_OUT_SPLIT = 128,   /* forward flag: write the output in the S16 split format (fp32 GDN layers and
                              MPC_F16X3 layers), for a following MPC_F16X3 layer */
    MPC_WIDE_BLOCK = 64,   /* forward flag (MPC_F16 / MPC_F16X3 layers): 8-wave (256-pixel) workgroups.  Chosen
                              automatically for large inputs; the flag forces it.  Results are identical. */
    MPC_IN_ABS = 1,        /* take |x| of the input (Analysis_prior_net.forward, :123-124) */
    MPC_OUT_ROUND = 2,     /* also emit rint(out) (torch.round, manynets_compressor.py:533) */
    MPC_OUT_CLAMP01 = 4,   /* clamp(0,1) (manynets_compressor.py:537) */
    MPC_OUT_NCHW = 8       /* write the output as NCHW instead of NHWC */
};

typedef struct {
    int32_t kind;     /* MPC_CONV (nn.Conv2d) or MPC_DECONV (nn.ConvTranspose2d) */
    int32_t cin, cout;
    int32_t ksize;    /* 3 or 5 */
    int32_t stride;   /* 1 or 2 */
    int32_t pad;      /* (ksize-1)/2 */
    int32_t out_pad;  /* stride-1 for MPC_DECONV, else 0 */
    int32_t act;      /* MPC_ACT_* applied to (conv + bias) */
    int32_t flags;    /* MPC_IN_ABS | MPC_F16 | MPC_F16X3 */
    float in_bound;   /* MPC_F16X3 with GDN/IGDN: an upper bound of |input| (1 for an image in [0,1],
                         1/sqrt(min_i gamma_ii) after a GDN).  > 0 lets the channel mix run on the split-precision
                         fp16 MFMA as well; 0 keeps it on the exact f32 MFMA. */
} mpc_conv_desc;

/* Builds one layer from PyTorch-layout device tensors: weight [cout,cin,k,k] (MPC_CONV) or
 * [cin,cout,k,k] (MPC_DECONV); bias [cout]; for GDN/IGDN the raw parameters beta [cout] and gamma
 * [cout,cout] (lower bound + square reparametrisation of GDN.py:75-81 is applied here).  Weights are
 * repacked into the MFMA operand layout in library-owned device memory.  Synchronous.
 * Replaces nn.Conv2d / nn.ConvTranspose2d / GDN (Balle2018PT_compressor.py:82-102, :105-127,
 * :223-251, :286-308; testolina_compressor.py:91-100). */
int mpc_conv_create(mpc_ctx *ctx, const mpc_conv_desc *desc, const float *weight, const float *bias,
                    const float *beta, const float *gamma, mpc_conv **out);
int mpc_conv_destroy(mpc_conv *conv);
/* Output spatial size for an input of H x W. */
int mpc_conv_out_hw(const mpc_conv *conv, int H, int W, int *Ho, int *Wo);
/* in: NHWC [B,H,W,cin_stride] with cin_stride = cin rounded up to a multiple of 4 (only cin = 3 is
 * padded).  out: NHWC [B,Ho,Wo,cout] (or NCHW with MPC_OUT_NCHW).  With MPC_OUT_ROUND, `out` (may be
 * NULL) receives the unrounded values and `out_round` the rounded ones. */
int mpc_conv_forward(mpc_ctx *ctx, const mpc_conv *conv, const void *in, int B, int H, int W,
                     int flags, void *out, float *out_round);
/* GDN / IGDN as a layer of its own on an NCHW fp32 tensor x [B, C, HW] (a 5-D input [B, C, D, W, H] is the same
 * memory with HW = D*W*H, GDN.py:67-70): out = x / sqrt(beta_eff + gamma_eff . x^2) (inverse != 0: x * sqrt(...)), the
 * raw parameters beta [C], gamma [C, C] reparametrised as in GDN.py:75-81, the channel mix on the exact fp32 matrix
 * pipe.  C a multiple of 32.  Replaces GDN.forward (src/compression/models/GDN.py:65-95); inside the transforms the
 * normalisation is fused into the producing convolution (mpc_conv_create with MPC_ACT_GDN / MPC_ACT_IGDN). */
int mpc_gdn_forward(mpc_ctx *ctx, const float *x, const float *beta, const float *gamma, int B, int C, int64_t HW,
                    int inverse, float *out);
/* The last layer of g_s with the evaluation's clamp and squared error fused into its epilogue: xhat_nchw =
 * clamp(layer output, 0, 1) as fp32 NCHW and sse_dev[b] (device float64 [B], zeroed by this call) = sum over image b of
 * (xhat - target)^2 -- recon.clamp(0, 1) and the per-image MSE of the test harness (manynets_compressor.py:537-543,
 * tests.py:439-460) without a second pass over the image.  Available for the fp16 small-cout transposed layer at input
 * widths that are multiples of 32 (mpc_conv_scored_supported returns 1); otherwise MPC_EINVAL: use mpc_conv_forward
 * followed by mpc_clamp_sse_batched. */
int mpc_conv_scored_supported(const mpc_conv *conv, int H, int W);
int mpc_conv_forward_scored(mpc_ctx *ctx, const mpc_conv *conv, const void *in, int B, int H, int W, float *xhat_nchw,
                            const float *target_nchw, double *sse_dev);
/* The last layer of g_a with quantisation and symbolisation fused into its epilogue: a layer without activation on
 * the f32 or MPC_F16X3 path writes sym[pixel][c] = clamp(rint(conv + bias) - min_feat, 0, n_sym_values - 1) as int16
 * (NHWC = the reference's symbol order) and, if out_round is not NULL, rint(conv + bias) as fp32 NHWC.
 * *n_clamped_dev (device int32, may be NULL; zeroed by this call) counts clamped values.
 * torch.round + ((y_hat - min_feat)/q_intv).round().short(), manynets_compressor.py:204, :264-269, with q_intv = 1. */
int mpc_conv_forward_symbols(mpc_ctx *ctx, const mpc_conv *conv, const void *in, int B, int H, int W, int min_feat,
                             int n_sym_values, float *out_round, int16_t *sym, int32_t *n_clamped_dev);
/* fp32 <-> fp16 conversion of n contiguous elements (saturating to +-65504), for the boundaries of an
 * MPC_F16 layer stack. */
int mpc_f32_to_f16(mpc_ctx *ctx, const float *in, int64_t n, void *out_f16);
int mpc_f16_to_f32(mpc_ctx *ctx, const void *in_f16, int64_t n, float *out);

/* ---- hyperprior likelihood (abstract_model.py:226-234) -------------------------------------- */
/* bits[i] = clamp(-ln(F(y+.5) - F(y-.5) + 1e-10)/ln 2, 0, 50) with F = Laplace(0, clamp(sigma,1e-10,1e10));
 * also accumulates the double-precision total into *total_dev (may be NULL; must be zeroed by caller).
 * bits may be NULL. */
int mpc_laplace_bits(mpc_ctx *ctx, const float *yhat, const float *sigma, int64_t n, float *bits,
                     double *total_dev);
/* Same expression on p[i] = hi[i] - lo[i] (factorised z prior, abstract_model.py:236-243). */
int mpc_prob_bits(mpc_ctx *ctx, const float *cdf_hi, const float *cdf_lo, int64_t n, float *bits,
                  double *total_dev);
/* Per-element int16 CDF rows of Laplace(0, sigma[i]) at targets first_target + s (in-repo rounding
 * order): out [n][L].  abstract_model.py:404-416. */
int mpc_laplace_cdf_rows(mpc_ctx *ctx, const float *sigma, int64_t n, float first_target, int L,
                         int precision, int16_t *out);

#ifdef __cplusplus
}
#endif
#endif /* MPC_H */
