/*
 * mmlf_hip.h -- C ABI of libmmlf_hip.so, the MI355X (gfx950) kernels behind the EPI-stack CNN
 * forward/backward path of titus-leistner/mmlf.
 *
 * The reference has no FFI: its hot path reaches arithmetic through stock torch.nn modules.  Each
 * entry point below therefore names the reference call site whose arithmetic it replaces
 * (paths relative to the reference repo root).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer on the current HIP device; the library owns no memory,
 *    allocates nothing, never synchronises and launches on the `stream` it is given
 *    (a hipStream_t passed as void*).  Re-entrant; no global mutable state except the
 *    thread-local last-error string.
 *  - return value: 0 = ok, nonzero = error (bad argument or launch failure); the text is
 *    available from mmlf_last_error() on the calling thread.
 *
 * Activation layout ("padded grid"): one image of spatial size H x W lives on a
 * grid of R = H+2 rows and pitch P = W+2 positions, G = R*P positions per image, NHWC with a
 * channel stride `cs` (multiple of 4 floats).  A tensor of extent (H, W) is stored at grid offset
 * (1,1) (position shift P+1) with a zero border; a tensor of extent (H+1, W+1) (the output of the
 * k=2, pad=1 convolution) is stored at grid offset (0,0).  With this layout every 2x2 convolution,
 * its data gradient and its weight gradient are 4-tap correlations over the flat position index
 * q = (b*R + y)*P + x with tap offsets {0, 1, P, P+1}.
 * Buffers must hold mmlf_grid_alloc_positions(B,H,W) positions; positions outside the stored
 * extent must be zero (the kernels that write a buffer maintain this; the caller zeroes the
 * head/tail slack once per allocation).
 */
#ifndef MMLF_HIP_H
#define MMLF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMLF_TILE_POSITIONS 256

const char *mmlf_last_error(void);
/* Bumped whenever an entry point's arguments or a layout they share changes.  mmlf_abi_version() returns the value the
 * library was BUILT with: a binding compares it with the header it was written against (mmlf_amd/_lib.py does, and reads
 * the number from this line) before making any other call. */
#define MMLF_ABI_VERSION 8
int mmlf_abi_version(void);

/* What the binary is: one line with the ABI version, the source revision it was built from and the value of every build
 * switch that changes behaviour ("abi=7 git=... MMLF_ABL_TERMS=3 ... ablation=0").  mmlf_build_is_ablation() is nonzero for a
 * build that computes WRONG results by construction (the timing ablations, switches in csrc/conv_device.h): a binding must refuse such a
 * library unless its user asked for it (mmlf_amd/_lib.py: MMLF_ALLOW_ABLATION=1). */
const char *mmlf_build_info(void);
int mmlf_build_is_ablation(void);
/* compute units the persistent conv / weight-gradient launches size their grids by: the device's count, or the cap the
 * environment variable MMLF_CONV_CUS sets (leaves CUs free for a collective's kernels under data parallelism) */
int mmlf_conv_cus(void);

/* Bounds audit: END (largest byte offset + 1) of what ONE launch of mmlf_conv2x2_h2 / mmlf_conv2x2_wgrad_h2 of the given
 * shape may touch behind each of its pointer arguments, derived from the launch geometry (end of csrc/conv.hip and csrc/wgrad.hip).
 * A caller that allocates what the size queries below say is inside every end: tests/test_bounds_audit.py asserts it. */
enum { MMLF_AUDIT_IN = 0, MMLF_AUDIT_PACKED = 1, MMLF_AUDIT_BIAS = 2, MMLF_AUDIT_OUT = 3, MMLF_AUDIT_REF = 4,
       MMLF_AUDIT_IN_AMAX = 5, MMLF_AUDIT_OUT_AMAX = 6, MMLF_AUDIT_BN_PARTIAL = 7, MMLF_AUDIT_MASK = 8, MMLF_AUDIT_CONV_N = 9 };
int mmlf_audit_conv_h2(int cs_in, int K, int N, int cs_out, int N_store, int out_shift, int cs_ref, int B, int H, int W,
                       int64_t *ends);
enum { MMLF_AUDIT_WG_IN = 0, MMLF_AUDIT_WG_G = 1, MMLF_AUDIT_WG_GW = 2, MMLF_AUDIT_WG_GB = 3, MMLF_AUDIT_WG_WORKSPACE = 4,
       MMLF_AUDIT_WG_IN_AMAX = 5, MMLF_AUDIT_WG_G_AMAX = 6, MMLF_AUDIT_WGRAD_N = 7 };
int mmlf_audit_wgrad_h2(int cs_in, int Cin, int cs_g, int Cout, int g_shift, int B, int H, int W, int64_t *ends);

/* number of positions a grid buffer must provide for batch B and image extent H x W */
int64_t mmlf_grid_alloc_positions(int B, int H, int W);
/* number of floats of a packed filter for K input and N output channels (see mmlf_pack_filter) */
int64_t mmlf_packed_filter_floats(int K, int N);
/* floats of workspace mmlf_conv2x2_wgrad* needs for (Cin, Cout) on a (B, H, W) grid */
int64_t mmlf_wgrad_workspace_floats(int Cin, int Cout, int B, int H, int W);

/* filter variants: the H / I streams run the shared stream net on a transposed / transposed+flipped
 * image (reference mmlf/model/feed_forward.py:236-256); the same result is obtained on the
 * un-transformed image with the filter taps permuted. */
enum { MMLF_VAR_IDENTITY = 0, MMLF_VAR_TRANSPOSE = 1, MMLF_VAR_TRANSPOSE_FLIPH = 2 };

/* OIHW master filter (Cout,Cin,2,2) -> packed K-major MFMA operand.
 * dgrad = 0: K = Cin, N = Cout (forward).  dgrad = 1: K = Cout, N = Cin, taps reversed (data grad).
 * replaces: the implicit filter layout transforms inside nn.Conv2d (feed_forward.py:123,125). */
int mmlf_pack_filter(const float *w_oihw, float *packed, int Cout, int Cin, int variant, int dgrad,
                     void *stream);

/* 4-tap correlation = nn.Conv2d(k=2) forward (feed_forward.py:123,125) or its data gradient.
 *   out[q + out_shift][n] = valid(q) ? act(bias[n] + sum_t sum_k in[q + off_t][k] * Wp[t][k][n]) : 0
 * valid(q): q < B*G and y < vh and x < vw.  relu: fuse nn.ReLU (feed_forward.py:124).
 * relu_ref (nullable): multiply by (relu_ref[q + out_shift][n] > 0) -- ReLU backward fused into dgrad.
 * N_store channels are written per position (N_store <= cs_out; pad channels get 0). */
int mmlf_conv2x2(const float *in, int cs_in, int K, const float *packed, const float *bias, int N,
                 float *out, int cs_out, int N_store, int out_shift, int vh, int vw,
                 int B, int H, int W, int relu, const float *relu_ref, int cs_ref, void *stream);

/* 3-way bf16 split variant of mmlf_pack_filter / mmlf_conv2x2 (same arguments and semantics; MMLF_CONV_MODE=bf16x6).
 * Every f32 operand is split exactly into three bf16 (hi+mid+lo); each product is evaluated as its six
 * leading cross terms on v_mfma_f32_16x16x32_bf16 with f32 accumulation.  Measured error vs a double
 * reference is at the level of the f32 MFMA fma chain, at 2.67x its rate; needs no operand scaling
 * (DESIGN.md section 4.4).  `packed` holds mmlf_packed_filter_split_bytes(K, N) bytes. */
int64_t mmlf_packed_filter_split_bytes(int K, int N);
int mmlf_pack_filter_split(const float *w_oihw, void *packed, int Cout, int Cin, int variant, int dgrad,
                           void *stream);
int mmlf_conv2x2_split(const float *in, int cs_in, int K, const void *packed, const float *bias, int N,
                       float *out, int cs_out, int N_store, int out_shift, int vh, int vw,
                       int B, int H, int W, int relu, const float *relu_ref, int cs_ref, void *stream);

/* 2-way f16 split variant ("f16x3", the default arithmetic): an operand x is multiplied by a power of two s and
 * rounded to two f16, x*s ~ hi + lo (2 x 11 = 22 significant bits, against float32's 24; NOT an exact split);
 * each product is evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 with f32 accumulation and the
 * powers of two are undone exactly in the epilogue.  Measured error vs a double reference: at the level of the
 * exact-f32 MFMA chain (tools/f16x2_accuracy.hip, tests/test_gpu_precision.py), at half the MFMA passes of the
 * bf16 split.  The scales are LOCAL, so that small-magnitude regions keep their relative precision:
 *  - activations: one scale per wave (32 output positions), from the maxima of the grid rows its taps read
 *    (2-3 image rows of one patch); elements within 2^-18 of that maximum keep all 22 bits;
 *  - weights: one scale per packed column (= output channel of the launch), chosen by mmlf_pack_filter_h2.
 * "amax array" of a grid tensor (mmlf_amax_entries(B,H,W) floats): a head of mmlf_amax_head() floats that holds 64
 * partial maxima of |x| over the tensor, mmlf_amax_shard_stride() floats apart (the tensor's maximum is the largest of
 * them; one slot would serialise a hundred thousand atomics per launch at the memory side), then [head + r] = max |x|
 * of grid row r = q / P, all channels.  Every kernel that writes a grid tensor raises the entries of what it wrote by
 * atomic max (`amax_out` arguments; upper bounds are as good as exact values); mmlf_zero_slack zeroes them together
 * with the buffer's slack, before the tensor's first producer.
 * `packed` holds mmlf_packed_filter_h2_bytes(K, N) bytes (the columns' 1/scale factors sit behind the planes). */
int64_t mmlf_amax_entries(int B, int H, int W);
int mmlf_grid_pad_w(void);    /* grid pitch P = W + pad_w (2; a build option of the library, so the binding asks) */
int mmlf_grid_pad_h(void);    /* grid rows per image R = H + pad_h (2) */
int mmlf_amax_head(void);
int mmlf_amax_shard_stride(void);
int64_t mmlf_packed_filter_h2_bytes(int K, int N);
int mmlf_pack_filter_h2(const float *w_oihw, void *packed, int Cout, int Cin, int variant, int dgrad, void *stream);
/* The same for MANY filters in one launch (a training step packs every filter of the net twice -- forward and data
 * gradient -- from weights that Adam has just rewritten: 76 launches of ~6 us at the default hyper-parameters, which is
 * 1 % of a 64-patch step).  `table` is a DEVICE array of n descriptors; its pointers stay owned by the caller.
 * replaces: the same filter layout transforms inside nn.Conv2d (feed_forward.py:123,125), per optimisation step. */
typedef struct mmlf_pack_desc {
    const float *w_oihw;   /* OIHW master filter (Cout, Cin, 2, 2) */
    void *packed;          /* mmlf_packed_filter_h2_bytes(K, N) bytes */
    int32_t Cout, Cin, variant, dgrad;
    int32_t col0;          /* first packed column of this filter in the launch: running sum of the filters' packed widths */
    int32_t np;            /* packed width of this filter, = mmlf_packed_filter_h2_columns(N) */
} mmlf_pack_desc;
int mmlf_packed_filter_h2_columns(int N);
int mmlf_pack_filters_h2(const mmlf_pack_desc *table, int n, int total_columns, void *stream);
int mmlf_conv2x2_h2(const float *in, int cs_in, int K, const void *packed, const float *bias, int N,
                    float *out, int cs_out, int N_store, int out_shift, int vh, int vw,
                    int B, int H, int W, int relu, const float *relu_ref, int cs_ref,
                    const float *in_amax /* amax array of `in` */, float *out_amax /* nullable: amax array of `out` */,
                    double *bn_partial /* nullable: see mmlf_bn_stats_finalize */,
                    uint32_t *relu_mask_out /* nullable */, const uint32_t *relu_mask_in /* nullable */, void *stream);
/* ReLU masks as bits (mmlf_relu_mask_words(B,H,W) words): relu_mask_out receives (out > 0) of every element of
 * this launch; relu_mask_in -- the mask a launch with the same (B,H,W) and N wrote -- replaces relu_ref in the data
 * gradient (nn.ReLU backward, feed_forward.py:124): the layer's activations are not read again.  The word layout is private
 * to the library (it depends on the kernel and orientation a (B,H,W,N) launch takes: csrc/conv.hip); a caller only carries the
 * buffer from the producing launch to the consuming one. */
int64_t mmlf_relu_mask_words(int B, int H, int W);
/* number of workgroups mmlf_conv2x2_h2 launches for K input and N output channels on this grid (= rows of bn_partial) */
int mmlf_conv2x2_blocks(int K, int N, int B, int H, int W);

/* "Thin" convolution: N <= 2 output channels over a wide input (first convolution of the BASE / UPR head,
 * feed_forward.py:179-182): a matrix-vector product bound by reading the input once, evaluated with plain float32
 * FMAs straight from the OIHW master filter (no packing, no operand split, any MMLF_CONV_MODE).  Same semantics as
 * mmlf_conv2x2 for the written extent: every position of the output grid is written (zeros outside the valid
 * extent and in the pad channels), the output's amax array is raised.  workspace:
 * mmlf_conv2x2_thin_workspace_floats(B,H,W) floats; the weight gradient's: mmlf_conv2x2_wgrad_thin_workspace_floats(Cin). */
int64_t mmlf_conv2x2_thin_workspace_floats(int B, int H, int W);
int mmlf_conv2x2_thin(const float *in, int cs_in, int K, const float *w_oihw, const float *bias, int N,
                      float *out, int cs_out, int out_shift, int vh, int vw, int B, int H, int W, int relu,
                      int variant, float *workspace, float *out_amax, void *stream);
int64_t mmlf_conv2x2_wgrad_thin_workspace_floats(int Cin);
int mmlf_conv2x2_wgrad_thin(const float *in, int cs_in, int Cin, const float *g, int cs_g, int Cout, int g_shift,
                            float *gw_oihw, float *gb, int variant, int accumulate, float *workspace,
                            int B, int H, int W, void *stream);

/* Weight + bias gradient of the convolution above (autograd of feed_forward.py:123,125 reached
 * from train/cli.py:257):  gw[co][ci][tap] (+)= sum_q in[q + off_t][ci] * g[q + g_shift][co],
 * gb[co] (+)= sum_q g[q + g_shift][co].  g must be zero outside its stored extent.
 * variant maps taps back to the OIHW master; accumulate != 0 adds into gw/gb (shared stream nets). */
int mmlf_conv2x2_wgrad(const float *in, int cs_in, int Cin, const float *g, int cs_g, int Cout,
                       int g_shift, float *gw_oihw, float *gb, int variant, int accumulate,
                       float *workspace, int B, int H, int W, void *stream);

/* split-precision variant of mmlf_conv2x2_wgrad (same arguments; see mmlf_conv2x2_split) */
int mmlf_conv2x2_wgrad_split(const float *in, int cs_in, int Cin, const float *g, int cs_g, int Cout,
                             int g_shift, float *gw_oihw, float *gb, int variant, int accumulate,
                             float *workspace, int B, int H, int W, void *stream);

/* f16-split variant of mmlf_conv2x2_wgrad (see mmlf_conv2x2_h2): in_amax / g_amax are the operands' amax
 * arrays.  The sum runs over all positions, so every 32-position chunk carries the same PRODUCT of operand
 * scales (the tensors' global ones), split between the two operands per chunk by the maxima of the grid rows it
 * reads: in * (sA * 2^x), g * (sG * 2^-x), x = half the difference of the operands' headroom.  A chunk keeps all
 * 22 bits of both operands while its products are within 2^-36 of the launch's largest (smaller ones are below
 * float32's own accumulation error of the sum). */
int mmlf_conv2x2_wgrad_h2(const float *in, int cs_in, int Cin, const float *g, int cs_g, int Cout,
                          int g_shift, float *gw_oihw, float *gb, int variant, int accumulate,
                          float *workspace, int B, int H, int W, const float *in_amax, const float *g_amax,
                          void *stream);

/* nn.BatchNorm2d training statistics (feed_forward.py:134): per-channel mean / biased variance of
 * the extent-(H,W) tensor z (zero border), running-stat update (unbiased var), and the affine
 * coefficients scale = gamma*invstd, shift = beta - mean*scale.  partial: 2*C*nblocks doubles. */
int mmlf_bn_stats_train(const float *z, int cs, int C, const float *gamma, const float *beta,
                        float *running_mean, float *running_var, double momentum, double eps,
                        float *save_mean, float *save_invstd, float *scale, float *shift,
                        double *partial, int nblocks, int B, int H, int W, void *stream);
/* eval-mode coefficients from the running statistics */
int mmlf_bn_coeffs_eval(const float *gamma, const float *beta, const float *running_mean,
                        const float *running_var, double eps, float *scale, float *shift, int C,
                        void *stream);
/* eval-mode BatchNorm folded into the convolution in front of it (inference only):
 * w_out[co,...] = w[co,...]*scale[co], bias_out = bias*scale + shift, so that conv + fused ReLU
 * replaces conv, nn.BatchNorm2d(eval), nn.ReLU (feed_forward.py:125,134-135). */
int mmlf_fold_bn_eval(const float *w_oihw, const float *bias, const float *scale, const float *shift,
                      float *w_out, float *bias_out, int Cout, int Cin, void *stream);
/* The second half of mmlf_bn_stats_train for statistics that the convolution already accumulated:
 * mmlf_conv2x2_h2(bn_partial = partial) writes per-workgroup sums of z and z^2 per channel
 * ([mmlf_conv2x2_blocks()][2][C] doubles) from its epilogue, so the pad-0 convolution's output is not
 * read again for nn.BatchNorm2d's training statistics (feed_forward.py:134). */
int mmlf_bn_stats_finalize(const double *partial, int nblocks, int C, const float *gamma, const float *beta,
                           float *running_mean, float *running_var, double momentum, double eps,
                           float *save_mean, float *save_invstd, float *scale, float *shift,
                           int B, int H, int W, void *stream);
/* y[q][c_off + c] = interior(q) ? relu(z[q][c]*scale[c] + shift[c]) : 0   (BN apply + nn.ReLU,
 * feed_forward.py:134-135; writing a channel slice implements torch.cat, feed_forward.py:266-267) */
int mmlf_bn_apply_relu(const float *z, int cs_z, int C, const float *scale, const float *shift,
                       float *y, int cs_y, int c_off, int C_store, int B, int H, int W,
                       float *amax_out /* nullable: amax array of y */, void *stream);
/* The same for the four stream nets' last blocks at once: y[q][k*C + c] = interior(q) ? relu(z[k][q][c]*scale[k][c] +
 * shift[k][c]) : 0 for k = 0..3 -- ONE pass that writes whole rows of the concat buffer (cs_y == 4*C, C even) instead of
 * four passes that each write a quarter of every row (torch.cat of the streams, feed_forward.py:266-267). */
int mmlf_bn_apply_relu4(const float *const z[4], int cs_z, int C, const float *const scale[4],
                        const float *const shift[4], float *y, int cs_y, int B, int H, int W,
                        float *amax_out /* nullable: amax array of y */, void *stream);
/* BatchNorm2d + ReLU backward, pass 1: per-channel sums of g and g*zhat with
 * g = gy * (z*scale+shift > 0); emits dgamma, dbeta (accumulating) and coefficients k[3*C]. */
int mmlf_bn_bwd_reduce(const float *gy, int cs_gy, int c_off, const float *z, int cs_z, int C,
                       const float *scale, const float *shift, const float *gamma,
                       const float *save_mean, const float *save_invstd,
                       float *dgamma, float *dbeta, int accumulate, float *coef,
                       double *partial, int nblocks, int B, int H, int W, void *stream);
/* pass 2: dz[q][c] = interior(q) ? k1*g - k2 - k3*(z - mean) : 0 */
int mmlf_bn_bwd_apply(const float *gy, int cs_gy, int c_off, const float *z, int cs_z, int C,
                      const float *scale, const float *shift, const float *save_mean,
                      const float *coef, float *dz, int cs_dz, int B, int H, int W,
                      float *amax_out /* nullable: amax array of dz */, void *stream);

/* (B, C, H, W) NCHW  <->  padded-grid NHWC (extent (H,W), grid offset (1,1), zero border, zero pad
 * channels).  replaces the .view / layout handling of feed_forward.py:226-232 and output[:, 0]. */
int mmlf_pack_nchw(const float *nchw, int C, float *grid, int cs, int B, int H, int W,
                   float *amax_out /* nullable: amax array of grid */, void *stream);
int mmlf_unpack_nchw(const float *grid, int cs, float *nchw, int C, int B, int H, int W, void *stream);
/* zero the slack of a freshly allocated grid buffer: positions [0, P+1) and [B*G, mmlf_grid_alloc_positions),
 * and (amax nullable) the buffer's amax array */
int mmlf_zero_slack(float *grid, int cs, int B, int H, int W, float *amax, void *stream);
/* the same for up to four buffers of one (B, H, W) geometry in one launch (null grid pointers are skipped) */
int mmlf_zero_slack4(float *const grid[4], const int cs[4], float *const amax[4], int B, int H, int W, void *stream);

/* UPR head (feed_forward.py:292-302, laplacian :9-12): posterior[b,k,y,x] from output[:,0:2]. */
int mmlf_head_upr(const float *output_nchw, const float *grid108, float *posterior, int steps,
                  int B, int H, int W, void *stream);
/* DPP head (feed_forward.py:276-290, dl.py:160-182). */
int mmlf_head_dpp(const float *scores_nchw, const float *grid_torch, const float *grid_np,
                  float *one_hot, float *posterior, float *mean, float *logvar, int steps,
                  int B, int H, int W, void *stream);
/* Backward of the heads: gradients of `posterior` (UPR) and of `posterior` / `logvar` (DPP) with respect to the network
 * output, for callers that put a loss on them (the reference's graph has these edges, feed_forward.py:276-302; its own
 * losses do not use them).  grad_output (B,2,H,W) / grad_scores (B,steps,H,W) are WRITTEN.  DPP: `mean` is the forward's
 * arg-max depth (a constant of the graph, as in the reference); grad_posterior or grad_logvar may be null. */
int mmlf_head_upr_bwd(const float *output_nchw, const float *grid108, const float *grad_posterior, float *grad_output,
                      int steps, int B, int H, int W, void *stream);
int mmlf_head_dpp_bwd(const float *scores_nchw, const float *grid_np, const float *mean, const float *grad_posterior,
                      const float *grad_logvar, float *grad_scores, int steps, int B, int H, int W, void *stream);

/* Losses (value + gradient w.r.t. the raw out_net output, NCHW), reference mmlf/model/loss.py.
 * kind 0: MaskedL1Loss (:70-77)  1: ImprovedUncertaintyL1Loss without mask_padding (:264-294)
 *      2: MaskedCrossEntropy (:146-160) with the target built as reg_to_class(gt) (dl.py:109-131).
 * scratch: >= 2*nblocks+2 doubles.  loss_out: one float.  grad may be NULL (value only).
 * den_override (nullable, device): use *den_override instead of the local mask count as the
 * denominator -- global_count/world_size under data parallelism, so that the rank-averaged
 * gradient is the gradient of the GLOBAL masked mean (train/cli.py:245-255 computes the loss on
 * the gathered batch). */
int mmlf_loss_fwd_bwd(int kind, const float *output_nchw, int oc, const float *gt, const int32_t *mask,
                      const float *grid_torch, double half_step, float *loss_out, float *grad_nchw,
                      double *scratch, int nblocks, const double *den_override, int B, int H, int W,
                      void *stream);

/* Multimodal and padded training losses (value + gradient w.r.t. the raw out_net output), reference loss.py:
 * kind 3: MultiMaskedL1Loss (:80-103)   4: ImprovedMultiUncertaintyL1Loss (:336-372)
 *      5: MaskedCrossEntropy (:146-160) on the target mpi_to_weights(mpi) (mmlf/utils/dl.py:134-157)
 *      6: ImprovedUncertaintyL1Loss WITH mask_padding (:264-294; --train_loss_padding, train/cli.py:217-222)
 * target: kinds 3-5 the multi-plane tensor mpi (B, P, 5, H, W) ([:, :, 3] alpha, [:, :, 4] disparity); kind 6
 * gt (B, H, W) with mask_padding int32 (B, H, W).  scratch: mmlf_loss_multi_scratch_doubles(nblocks) doubles.
 * aux_override (nullable, device double[2]): the whole-batch sums kinds 4 / 6 normalise by (sum of the total
 * alpha and count of surface-less pixels; count of in-range pixels) -- the all-reduced ones under data
 * parallelism, since the reference evaluates the loss on the gathered batch. */
int64_t mmlf_loss_multi_scratch_doubles(int nblocks);
int mmlf_loss_multi_fwd_bwd(int kind, const float *output_nchw, int oc, const float *target, int P,
                            const int32_t *mask, const int32_t *mask_padding, const float *grid_torch,
                            double half_step, float *loss_out, float *grad_nchw, double *scratch, int nblocks,
                            const double *den_override, const double *aux_override, int B, int H, int W,
                            void *stream);

/* torch.optim.Adam step with default hyper-parameters (train/cli.py:117-118,258) on a flat buffer.
 * g is multiplied by grad_scale first (1/world_size after the RCCL sum all-reduce). */
int mmlf_adam_step(float *p, const float *g, float *m, float *v, int64_t n, double lr, double beta1,
                   double beta2, double eps, int64_t step, double grad_scale, void *stream);

/* hci4d.Shift on device (mmlf/data/hci4d.py:907-990): sub-pixel circular shear of each view, for S
 * shift values at once.  in: (views,3,H,W) per stack; out: (S,views,3,H,W) per stack.
 * tab_s[2*(s*views+k)+{0,1}] = integer shifts (shift0, shift1), tab_w[...] = weights (1-alpha, alpha)
 * computed on the host exactly as hci4d.py:934-938 (math.modf / copysign on doubles). */
int mmlf_shift_views(const float *h, const float *v, const float *i, const float *d,
                     float *oh, float *ov, float *oi, float *od, const int32_t *tab_s, const float *tab_w,
                     int S, int views, int H, int W, void *stream);
/* The same shear of ONE view stack for S shift values, written straight into the padded NHWC grid the trunk reads (the member
 * batch of mmlf/model/ensamble.py:61-76 without the (S,views,3,H,W) intermediate: mmlf_shift_views + mmlf_pack_nchw in one
 * pass, the same bits).  kind: 0 = horizontal stack (along W), 1 = vertical (along H), 2 = increasing diagonal (W, then H with
 * the negated shift, hci4d.py:971-975), 3 = decreasing diagonal (W, then H).  in: (views,3,H,W); grid: S images of the grid
 * layout, channel stride cs >= 3 views (pad channels and borders written as zeros); amax_out: the grid's amax array or NULL. */
int mmlf_shift_pack(const float *in, int kind, float *grid, int cs, const int32_t *tab_s, const float *tab_w,
                    int S, int views, int H, int W, float *amax_out, void *stream);
/* Ensamble reduction (mmlf/model/ensamble.py:78-101): means/logvars (S,B,H,W) ->
 * mean, logvar (arg-min logvar member), posterior (B,S,H,W) = mean of S Laplacians on grid[S]. */
int mmlf_ensamble_reduce(const float *means, const float *logvars, const float *grid, float *mean,
                         float *logvar, float *posterior, int S, int B, int H, int W, void *stream);

/* Discretised Laplace mixture of the ensemble members (mmlf/validate/cli.py:74-118, called at :302,318):
 * out[b][k][p] (float64) = (1/S) sum_s cdf(edges[k+1]) - cdf(edges[k]) of Laplace(means[s][b][p],
 * exp(logvars[s][b][p])); edges = n_bins+1 float64 bin edges.  S = 1 is laplace_to_discrete (:90-103). */
int mmlf_lmm_to_discrete(const float *means, const float *logvars, const double *edges, double *out,
                         int S, int B, int n_bins, long long HW, void *stream);

/* Training patch pipeline: the transform chain the reference's training CLI
 * composes (mmlf/train/cli.py:72-91) -- RandomDownSampling, RandomShift, RandomCrop + CenterCrop,
 * RandomRotate, RedistColor, Brightness (mmlf/data/hci4d.py:483-510, 907-990, 532-575, 1041-1071,
 * 698-715, 773-782) -- on scenes cached in device memory, one gather per output pixel.
 *   stacks (S,4,V,3,Hf,Wf), center (S,3,Hf,Wf), gt (S,Hf,Wf), mpi (S,P,5,Hf,Wf) or NULL with P=0,
 *   mask (S,Hf,Wf) int32.  Per sample b (parameters drawn on the host in the reference's call order):
 *   iparam[b] = {scene, factor, y0, x0, rot, flags (1 shift | 2 colour | 4 brightness), 0, 0},
 *   fparam[b] = {float(factor), disp, brightness, 0}, tab_s/tab_w[b][view] as mmlf_shift_views,
 *   mat[b] = 3x3 RedistColor matrix (double), rot_src[rot][stack][view] = source stack*V + view after
 *   `rot` Rotate90 steps.  Outputs: o_stacks (4,B,V,3,ps,ps), o_center (B,3,ps,ps), o_gt (B,ps,ps),
 *   o_mpi (B,P,5,ps,ps), o_mask (B,ps,ps) (the reference does not rotate the mask, hci4d.py:1056);
 *   mean_sum[b] (zeroed by the caller) accumulates the sum of the horizontal stack for Contrast. */
int mmlf_patch_gather(const float *stacks, const float *center, const float *gt, const float *mpi,
                      const int32_t *mask, int S, int V, int P, int Hf, int Wf, const int32_t *iparam,
                      const float *fparam, const int32_t *tab_s, const float *tab_w, const double *mat,
                      const int32_t *rot_src, float *o_stacks, float *o_center, float *o_gt, float *o_mpi,
                      int32_t *o_mask, double *mean_sum, int B, int ps, void *stream);
/* Contrast (hci4d.py:740-751) in place on o_stacks / o_center: x*alpha[b][0] + mean_b*alpha[b][1] with
 * alpha[b] = {float(alpha), float(1 - alpha)} and mean_b = mean_sum[b] / (V*3*ps*ps). */
int mmlf_patch_contrast(float *o_stacks, float *o_center, const double *mean_sum, const float *alpha,
                        int B, int V, int ps, void *stream);

#ifdef __cplusplus
}
#endif
#endif
