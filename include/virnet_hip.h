/*
 * virnet_hip.h -- C ABI of the MI355X (gfx950) kernels behind VIRNet's convolutional forward.
 *
 * The reference (zsyOAOA/VIRNet) is pure Python on PyTorch: it has no FFI, plugin or operator
 * registry.  Its boundary for this path is the nn.Module surface of networks/VIRNet.py
 * (VIRAttResUNet.forward :42-46, VIRAttResUNetSR.forward :80-97) and the arithmetic sits in
 * torch.nn.functional call sites.  Each entry point below therefore cites the reference CALL SITE
 * (file:line under /root/reference) whose arithmetic it replaces; the modules under virnet_amd/networks/ is the
 * Python host side that mirrors the reference classes and binds these symbols through ctypes
 * (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless stated otherwise; no torch types cross the ABI
 *   - activations between kernels are NHWC fp32 ("pixel records"), channel count a multiple of 16
 *   - images enter and leave as NCHW fp32 exactly as the reference modules receive/return them
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous
 *   - return value: 0 on success, non-zero on error; virnet_last_error() gives the message
 */
#ifndef VIRNET_HIP_H
#define VIRNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): round 4's additions -- virnet_t_emit / virnet_knet_layer, the emitting and persistent entry points, the convT weight image
 * that keeps cin when cin % 32 == 0 -- were shipped under version 1; a stale library now fails the version check instead of an
 * AttributeError / a mis-sized packing.
 * 4 (round 6): virnet_conv_wx4_last_plan; the packed entry image (virnet_pack_entry_weight / virnet_entry_weight_floats) carries a four-word
 * trailer {cin, k-steps per row, n_pad, tag} that virnet_conv_entry's kernel checks against the launch (a mismatch gives NaN). */
#define VIRNET_ABI_VERSION 4

int virnet_abi_version(void);
const char* virnet_last_error(void);
/* number of visible HIP devices, or -1 when the runtime cannot be initialised */
int virnet_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * Weight packing.  Reference layouts in, MFMA-fragment ("stage") layout out.
 *   kind 0: nn.Conv2d weight  [Cout][Cin][KS][KS]            (AttResUNet.py:43,46,67,117,139; DnCNN.py:22-29;
 *                                                             KNet.py:32,34,49)
 *   kind 1: nn.ConvTranspose2d(k=2,s=2) weight [Cin][Cout][2][2] (AttResUNet.py:80), packed as the equivalent
 *           1x1 GEMM to N = 4*Cout columns ordered (a*2+b)*Cout + co
 * `cin_pad` (multiple of 16) and `n_pad` (multiple of 32*nrep) give the zero-padded GEMM extents; `nrep` is the
 * number of 32-column blocks one workgroup owns (see virnet_conv_plan).
 * ---------------------------------------------------------------------------------------------- */
/*   kind 2: the input-gradient ("dgrad") weights of a 3x3 conv, W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx]: pass the forward
 *           OIHW tensor, cout/cin are the FORWARD counts; the packed GEMM has cin rows (n_pad covers cin) and cout_fwd
 *           contraction channels (cin_pad covers cout).  Used with stride 1 (a stride-2 conv's dgrad runs on the zero-stuffed
 *           gradient, virnet_zero_stuff2).
 *   kind 3: dgrad weights of the 2x2/s2 transposed conv: a 1x1 GEMM over the space-to-depth gradient
 *           (virnet_space_to_depth2): W'[ci][ab*cout+co] = Wt[ci][co][a][b]; n_pad covers cin, cin_pad covers 4*cout. */
size_t virnet_packed_weight_floats(int ks, int cin_pad, int n_pad);
int virnet_pack_weight(const float* w, int kind, int cout, int cin, int ks, int cin_pad, int n_pad, int nrep,
                       float* packed, void* stream);

/* Tile plan chosen by the library for one conv shape (so the host can pack with the right nrep). */
typedef struct virnet_conv_plan {
  int nrep;    /* 32-column blocks per workgroup (GEMM-N block = 32*nrep) */
  int n_pad;   /* padded GEMM-N extent */
  int cin_pad; /* padded Cin (multiple of 16) */
} virnet_conv_plan;
/* ks in {1,3}; stride in {1,2}; gemm_n = Cout (conv) or 4*Cout (transposed conv) */
int virnet_conv_get_plan(int ks, int stride, int cin, int gemm_n, virnet_conv_plan* plan);

/* ------------------------------------------------------------------------------------------------
 * The MFMA implicit-GEMM convolution (v_mfma_f32_32x32x2_f32, exact fp32).
 * Replaces: AttResBlock.conv1/conv2 (AttResUNet.py:55,58 with the residual add :59), DownBlock.downsampler
 * (AttResUNet.py:67), UpBlock.upsampler + bridge add (AttResUNet.py:84-87), AttResUNet.head/tail (:153-155,:173),
 * DnCNN.conv1/mid_layer/conv_last (DnCNN.py:38-41), RB_Layer convs and KernelNet.tail conv (KNet.py:32-34,49).
 * ---------------------------------------------------------------------------------------------- */
enum { VIRNET_EPI_NHWC = 0, VIRNET_EPI_CONVT = 1, VIRNET_EPI_NCHW = 2 };
enum { VIRNET_NCHW_PLAIN = 0, VIRNET_NCHW_ADD = 1, VIRNET_NCHW_EXPCLAMP = 2 };

typedef struct virnet_conv_desc {
  const float* x;      /* NHWC [n][h][w][cin_pad] */
  const float* wpack;  /* from virnet_pack_weight */
  const float* bias;   /* [cout] or NULL */
  const float* res;    /* EPI_NHWC: NHWC [n][oh][ow][cout] residual (AttResUNet.py:59) or NULL
                          EPI_CONVT: NHWC [n][2h][2w][cout] bridge (AttResUNet.py:87) or NULL
                          EPI_NCHW : NCHW [n][cout][crop_h][crop_w] (the `+ x_in` of AttResUNet.py:173) or NULL */
  const float* mul;    /* [n][cout] SFT scale applied to y_act (AttResUNet.py:57-58) or NULL (=1) */
  const float* add;    /* [n][cout] SFT shift applied to y_act or NULL (=0) */
  const float* mask;   /* EPI_NHWC only, backward passes: tensor of the output's shape; acc is multiplied by lrelu'(mask) =
                          (mask > 0 ? 1 : mask_slope) BEFORE res is added (d_pre = d_act * lrelu'(saved), AttResUNet.py:55,58) */
  const float* in_mul; /* [n][cin_pad] SFT scale applied to x while it is staged (AttResUNet.py:54-55) or NULL (=1) */
  const float* in_add; /* [n][cin_pad] SFT shift applied to x while it is staged or NULL (=0) */
  float* y_raw;        /* conv + bias (+res); NULL = not stored */
  float* y_act;        /* leaky_relu(y_raw*mul+add, slope); NULL = not stored (EPI_NHWC / EPI_CONVT only) */
  int n, h, w;         /* input batch / spatial size */
  int cin_pad;         /* channels of x (multiple of 16) */
  int cout;            /* real output channels (EPI_CONVT: channels of the up-sampled tensor) */
  int n_pad;           /* padded GEMM-N extent used when packing */
  int nrep;            /* from the plan used when packing */
  int ks, stride;      /* {3,1}: s1 or s2, pad 1 ; {1,1}: pointwise */
  int epi;             /* VIRNET_EPI_* */
  int nchw_op;         /* VIRNET_NCHW_* (EPI_NCHW only) */
  int crop_h, crop_w;  /* EPI_NCHW: stored extent (<= oh, ow) */
  int res_sf;          /* EPI_NCHW + VIRNET_NCHW_ADD: res is [n][cout][crop_h/res_sf][crop_w/res_sf] and is read through a
                          nearest x res_sf up-sampling (the x_up of VIRNet.py:83 added at AttResUNet.py:173); 0/1 = none */
  int in_act;          /* 1: the conv consumes leaky_relu(x*in_mul+in_add, in_slope) -- the pre-activation of AttResUNet.py:55 --
                          applied to the pixels on their way into LDS; padding stays zero AFTER the activation. 0: plain x */
  float in_slope;      /* LeakyReLU slope of the input activation */
  float slope;         /* LeakyReLU slope of y_act */
  float mask_slope;    /* slope of the LeakyReLU whose derivative `mask` selects */
  float clamp_lo, clamp_hi; /* VIRNET_NCHW_EXPCLAMP: y = exp(clamp(v, lo, hi))  (VIRNet.py:43) */
} virnet_conv_desc;

int virnet_conv_mfma(const virnet_conv_desc* d, void* stream);
/* Which instantiation virnet_conv_mfma would launch for `d`: out = {KS, STRIDE, MREP, NREP} of
 * conv_mfma_kernel<KS,STRIDE,MREP,NREP> (the name rocprofv3 reports).  Used by bench.py to attribute event timings. */
int virnet_conv_mfma_variant(const virnet_conv_desc* d, int out[4]);

/* ------------------------------------------------------------------------------------------------
 * The same stride-1 3x3 convolution in Winograd F(2x2,3x3) form (16 instead of 36 multiplies per 2x2 output tile and channel
 * pair; fp32 on v_mfma_f32_32x32x2_f32).  Call sites: AttResBlock.conv1/conv2 (AttResUNet.py:55,58 + residual :59), DnCNN
 * mid_layer (DnCNN.py:25-28,39-40) and their input-gradient GEMMs in the training step.  Takes the SAME descriptor as
 * virnet_conv_mfma with ks = 3, stride = 1, epi = VIRNET_EPI_NHWC, cout a multiple of 32 (n_pad = cout; nrep is ignored) and
 * `wpack` from virnet_pack_wino_weight: U = G g G^T per channel pair, [cout/32][cin_pad/4][16 positions][2][32][2] floats.
 * `dgrad` = 1 packs the input-gradient GEMM of the layer (W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx]; pass the forward OIHW tensor
 * and the FORWARD cout/cin; n_pad then covers cin and cin_pad covers cout, as for kind 2 of virnet_pack_weight).
 * ---------------------------------------------------------------------------------------------- */
size_t virnet_wino_weight_floats(int cin_pad, int n_pad);
int virnet_pack_wino_weight(const float* w_oihw, int dgrad, int cout, int cin, int cin_pad, int n_pad, float* packed, void* stream);
int virnet_conv_wino(const virnet_conv_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The same stride-1 3x3 convolution on the f16 matrix pipe with SPLIT fp32 operands (csrc/conv_f16.hip): every fp32 weight and
 * activation is written exactly as hi + lo in fp16 and w*x is evaluated as w_lo*x_hi + w_hi*x_lo + w_hi*x_hi on
 * v_mfma_f32_32x32x16_f16 with fp32 accumulation -- fp32-class results (measured error vs fp64 no larger than the fp32 MFMA
 * chain's, profiles/r02_probes.md) at 3/16 of the fp32 matrix pipe's cycles.  Call sites as virnet_conv_wino: AttResBlock.conv1/
 * conv2 (AttResUNet.py:55,58 + residual :59), DnCNN mid_layer (DnCNN.py:25-28,39-40), RB_Layer convs (KNet.py:32,34) and their
 * input-gradient GEMMs.  Takes the SAME descriptor as virnet_conv_mfma with ks = 3, stride = 1, epi = VIRNET_EPI_NHWC, cout a
 * multiple of 32 (n_pad = cout; nrep is ignored) and `wpack` from virnet_pack_f16_weight: n_pad inverse row scales (fp32) followed
 * by the split image [n_pad/32][cin_pad/16][9 taps, column-major][hi|lo][64 lanes][8 x fp16].  Weights are scaled per output
 * channel by a power of two (exact).  `dgrad` as for virnet_pack_wino_weight.  Activations must stay below 65504 in magnitude.
 * ---------------------------------------------------------------------------------------------- */
size_t virnet_f16_weight_floats(int cin_pad, int n_pad);
int virnet_pack_f16_weight(const float* w_oihw, int dgrad, int cout, int cin, int cin_pad, int n_pad, float* packed, void* stream);
int virnet_conv_f16(const virnet_conv_desc* d, void* stream);
/* The other two dense layers of the U-Net on the same pipe, through virnet_conv_f16:
 *   - ks = 3, stride = 2, epi = VIRNET_EPI_NHWC (DownBlock.downsampler, AttResUNet.py:67,74): `wpack` from virnet_pack_f16_weight,
 *     bias / single-store epilogue only (csrc/conv_f16_s2.hip);
 *   - ks = 1, epi = VIRNET_EPI_CONVT (UpBlock.upsampler + bridge add, AttResUNet.py:80,84-87): `wpack` from
 *     virnet_pack_f16_convt_weight (the IOHW [cin][cout][2][2] tensor as the pointwise GEMM to rows (a*2+b)*cout + co, contraction
 *     kept as is when cin is a multiple of 32 -- K then runs in 2-chunk stages with two workgroups per CU -- else zero-padded to a
 *     multiple of 48: virnet_f16_convt_weight_floats sizes it), n_pad = 4*cout, bias + `res` (bridge) / single-store epilogue (csrc/conv_f16_pw.hip). */
/* bf16-OPERAND variant of the stride-1 3x3 NHWC convolution (BASELINE configs[4]'s training precision; reduced accuracy, ~4e-3 per
 * operand): ONE v_mfma_f32_32x32x16_bf16 per k-step, operands rounded to bf16 (activations while staged, weights when packed), fp32
 * accumulation, bias / mask / residual / activation in fp32.  Same descriptor; `wpack` from virnet_pack_bf16_weight (same size and
 * layout as the f16 image: virnet_f16_weight_floats). */
int virnet_pack_bf16_weight(const float* w_oihw, int dgrad, int cout, int cin, int cin_pad, int n_pad, float* packed, void* stream);
int virnet_conv_bf16(const virnet_conv_desc* d, void* stream);
size_t virnet_f16_convt_weight_floats(int cin, int cout);
int virnet_pack_f16_convt_weight(const float* w_iohw, int cout, int cin, float* packed, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The same stride-1 3x3 NHWC convolution in Winograd F(4,3) form ALONG X (direct along y) on the f16 matrix pipe with split fp32
 * operands (csrc/conv_f16_wx4.hip): 6 transform positions x 3 row taps = 18 k-steps per 4 output pixels instead of 36, i.e. 1.5
 * executed FLOP per algorithmic FLOP instead of conv_f16's 3, same three-product fp32-class arithmetic per position product.
 * Call sites as virnet_conv_f16: AttResBlock.conv1/conv2 (AttResUNet.py:55,58 + residual :59), DnCNN mid_layer (DnCNN.py:25-28,39-40)
 * and their input-gradient GEMMs.  SAME descriptor (ks = 3, stride = 1, epi = VIRNET_EPI_NHWC, cout % 32 == 0, n_pad = cout) with
 * `wpack` from virnet_pack_wx4_weight: n_pad inverse row scales followed by U[dy][j] = sum_b G[j][b] w[dy][b] (formed in fp64) as
 * the split image [n_pad/32][cin_pad/16][6 positions][3 dy][hi|lo][64 lanes][8 x fp16].  Accuracy: the Winograd transform adds
 * rounding relative to the largest magnitude inside a 6-pixel input window (not per output): whole denoise-syn network 1.1e-5 max-abs
 * against fp64 (conv_f16 8.0e-6, fp32 direct 8.9e-6).  Transformed activations must stay below 65504 (|x| < 6.5e3).
 * ---------------------------------------------------------------------------------------------- */
size_t virnet_wx4_weight_floats(int cin_pad, int n_pad);
int virnet_pack_wx4_weight(const float* w_oihw, int dgrad, int cout, int cin, int cin_pad, int n_pad, float* packed, void* stream);
int virnet_conv_wx4(const virnet_conv_desc* d, void* stream);
/* What the calling thread's most recent virnet_conv_wx4 / virnet_conv_wx4_emit call launched (the library picks the tile form per launch
 * size, csrc/conv_f16_wx4.hip): out[0] = tile rows of its first launch (16: conv_f16_wx4.hip / conv_f16_wx4p.hip, 8: conv_f16_wx4h.hip),
 * out[1] = 1 when that launch was the persistent form (VIRNET_WX4_PERSIST=1), out[2] = 32-channel slabs per workgroup, out[3] = launches. */
void virnet_conv_wx4_last_plan(int* out4);

/* Few-output-channel exits with planar store (csrc/conv_exit.hip): AttResUNet.tail + crop + `+ x_in` (AttResUNet.py:139,173),
 * DnCNN.conv_last + exp(clamp) (DnCNN.py:29,41; VIRNet.py:43), KernelNet.tail (KNet.py:49) -- for cout * 9 <= 32 the (channel, tap) pairs
 * are the ROWS of one pointwise split-fp16 GEMM and the taps are summed afterwards: the input is read once, straight into MFMA
 * fragments (HBM-bound; algorithmic bytes n*h*w*cin_pad*4 in + 4 per output value).  `d` as for virnet_conv_f16 with epi =
 * VIRNET_EPI_NCHW (nchw_op, crop, res / res_sf, clamp honoured; in_act / in_slope optional); wpack from virnet_pack_exit_weight:
 * 32 inverse row scales, then [cin_pad/16][hi|lo][64 lanes][8 x fp16]. */
size_t virnet_exit_weight_floats(int cin_pad);
int virnet_pack_exit_weight(const float* w_oihw, int cout, int cin, int cin_pad, float* packed, void* stream);
int virnet_conv_exit(const virnet_conv_desc* d, void* stream);

/* Range guard of the split-fp16 kernels (virnet_conv_f16 incl. its stride-2 / transposed forms, virnet_conv_wx4).  An operand of magnitude
 * >= 65520 (transformed magnitude for virnet_conv_wx4: up to 10x the activation) does not fit fp16: the product turns Inf / NaN, which is
 * loud in a tensor but clamped away by the exp(clamp(.)) / tanh epilogues (VIRNet.py:43, KNet.py:56-58).  Register a zeroed device int per
 * device (NULL = off); the kernels OR 1 into it when a staged operand leaves the range.  The host side (virnet_amd/engine.py) reads it at
 * the end of a forward and re-runs the image in the fp32 Winograd form.  The reference's fp32 path has no such limit. */
int virnet_set_range_flag(int* device_flag);
/* (the registration is per (device, calling host thread): forwards driven from several threads keep separate flags)
 * virnet_poison_on_flag: fills y[0..n) with NaN when *device_flag != 0 -- the last node of a replayed hipGraph (virnet_amd/graph.py), so
 * that an out-of-range forward is loud even before the host has read the flag. */
int virnet_poison_on_flag(const int* device_flag, float* y, size_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * 3x3 convolution to 1..4 output channels with planar store (HBM/LDS-bound VALU kernel, not MFMA work):
 * AttResUNet.tail + crop + `+ x_in` (AttResUNet.py:139,173), DnCNN.conv_last + exp(clamp) (DnCNN.py:29,41; VIRNet.py:43),
 * KernelNet.tail conv (KNet.py:49).  Weights: virnet_pack_thin_weight of the OIHW tensor.
 * ---------------------------------------------------------------------------------------------- */
typedef struct virnet_thin_desc {
  const float* x;      /* NHWC [n][h][w][c], c a multiple of 16 */
  const float* wpack;  /* [c/16][9][16][4] from virnet_pack_thin_weight */
  const float* bias;   /* [cout] or NULL */
  const float* res;    /* VIRNET_NCHW_ADD: NCHW [n][cout][crop_h/res_sf][crop_w/res_sf] */
  float* y;            /* NCHW [n][cout][crop_h][crop_w] */
  int n, h, w, c, cout;
  int crop_h, crop_w;  /* stored extent (<= h, w) */
  int op;              /* VIRNET_NCHW_* */
  int res_sf;          /* nearest up-sampling factor of res (0/1 = none) */
  float clamp_lo, clamp_hi;
} virnet_thin_desc;
size_t virnet_thin_weight_floats(int c_pad);
int virnet_pack_thin_weight(const float* w_oihw, int cout, int c, int c_pad, float* packed, void* stream);
int virnet_conv3x3_thin(const virnet_thin_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Image entry: NCHW -> 16-channel NHWC pixel records, fusing
 *   - the nearest x`sf` up-sampling of VIRNet.py:83 (sf = 1 for denoising),
 *   - the bottom/right reflect pad of utils/util_net.py:20-25 (hp >= h*sf, wp >= w*sf),
 *   - the channel concat of AttResUNet.py:153 with per-image vectors (kinfo / sqrt(sigma) repeated, VIRNet.py:89,92)
 *     and/or a per-pixel map (sigma map; VIRNet.py:44 sqrt, VIRNet.py:94 nearest x msf).
 * Channel order: [c0 image channels][ev vector channels][em map channels][zeros up to 16].
 * ---------------------------------------------------------------------------------------------- */
typedef struct virnet_pack_desc {
  const float* x;    /* NCHW [n][c0][h][w] */
  const float* vec;  /* [n][ev] or NULL */
  const float* map;  /* NCHW [n][em][mh][mw] or NULL; source pixel = (reflect(y)/msf, reflect(x)/msf) */
  float* out;        /* NHWC [n][hp][wp][16] */
  int n, c0, h, w, sf;
  int ev;
  int em, mh, mw, msf, map_sqrt;
  int hp, wp;
  int zero_pad;      /* 1: positions beyond h*sf, w*sf are zero instead of reflected (gradient records of the training step) */
} virnet_pack_desc;
int virnet_pack_input(const virnet_pack_desc* d, void* stream);
/* The same entry folded into the first convolution (AttResUNet.head AttResUNet.py:153-155, DnCNN.conv1 DnCNN.py:38): virnet_conv_f16 on
 * ONE 16-channel chunk (d->cin_pad = 16, d->h x d->w = e->hp x e->wp, plain single-store epilogue) whose staging gathers each pixel's
 * record [image | vector | map | 0] from the NCHW sources of `e` -- the packed tensor never exists (e->out is ignored; c0 + ev + em <= 8). */
int virnet_conv_f16_entry(const virnet_conv_desc* d, const virnet_pack_desc* e, void* stream);
/* Round 5: the entries as their own STORE-bound kernel (csrc/conv_entry.hip; same call sites: AttResUNet.py:153-155, DnCNN.py:38).  K is
 * walked one kernel ROW per MFMA k-step (slot j = dx * cin + ch), the B fragments are gathered from a planar fp32 copy of the input tile in
 * LDS, every output row leaves as lane-linear 1-KB stores.  cin = c0 + ev + em <= 8, cout a multiple of 32 up to 96, plain epilogue
 * (exactly one of y_raw / y_act).  Weight image: virnet_entry_weight_floats(cin, n_pad) floats from virnet_pack_entry_weight (OIHW
 * [cout][cin][3][3] in; n_pad inverse scales first).  Agrees with virnet_conv_f16_entry to fp32 rounding (other summation order). */
size_t virnet_entry_weight_floats(int cin, int n_pad);
int virnet_pack_entry_weight(const float* w, int cout, int cin, int n_pad, float* packed, void* stream);
int virnet_conv_entry(const virnet_conv_desc* d, const virnet_pack_desc* e, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training step: weight / bias gradients and the layout helpers of the input-gradient convs.
 * Input gradients themselves are virnet_conv_mfma launches with kind-2 / kind-3 packed weights and the `mask` epilogue.
 * ---------------------------------------------------------------------------------------------- */
typedef struct virnet_wgrad_desc {
  const float* x;       /* NHWC [n][h][w][cx]: the conv's forward input */
  const float* dy;      /* NHWC [n][oh][ow][cy]: gradient of the conv's output (transposed: space-to-depth gradient, cy = 4*cout) */
  const float* in_mul;  /* the forward conv's staging transform (lrelu(x*in_mul+in_add)), or NULL */
  const float* in_add;
  float* dw;            /* += : [cout][cin][ks][ks] (OIHW) or, transposed, [cin][cout][2][2]; zero it first (fp32 atomics) */
  int* counters;        /* zeroed int32 scratch, ceil(rows/32)*ceil(cin/32) entries (rows = cout, or 4*cout transposed): tile hand-out */
  int n, h, w, cx, cy;
  int cin, cout;        /* real channel counts (<= cx, cy) */
  int ks, stride;       /* {3,1} {3,2} {1,1} */
  int transposed;       /* 1: weight of ConvTranspose2d(k2,s2) */
  int in_act;
  float in_slope;
} virnet_wgrad_desc;
int virnet_conv_wgrad(const virnet_wgrad_desc* d, void* stream);
/* The same gradient for the stride-1 3x3 convs on the f16 matrix pipe (csrc/wgrad_f16.hip; backward of networks/AttResUNet.py:43,46,
 * DnCNN.py:22-29).  The contraction runs over pixels, so both operands are first re-laid CHANNEL-major as fp16 planes:
 *   T[n][h+2][ceil(c/32)][hi|lo][seg][32 ch][8 px], pixel x at index x+8 of its row, zero rows / pads around the image.
 * virnet_chsplit writes T from an NHWC fp32 tensor (c % 4 == 0), applying lrelu(x*in_mul+in_add) (the forward conv's staging transform)
 * and the hi/lo split of virnet_conv_f16 (bf16 != 0: one bf16 plane, single product); `out` holds virnet_chsplit_bytes() bytes.
 * virnet_conv_wgrad_f16: dw[cout][cin][3][3] = sum_p dy[p][co] * a[p + tap][ci] from the two T tensors (cx / cy = their stored channels);
 * the pixel range is split over workgroups whose partial sums go to `scratch` (virnet_conv_wgrad_f16_scratch_bytes() bytes, need not be
 * zeroed) and are then reduced in a fixed order: dw is overwritten, and bitwise reproducible. */
size_t virnet_chsplit_bytes(int n, int h, int w, int c);
size_t virnet_conv_wgrad_f16_scratch_bytes(int n, int h, int w, int cx, int cy);
size_t virnet_chsplit_colsum_bytes(int n, int h, int w, int c);
/* db != NULL: the bias gradient as a by-product of the dY pass, db[ch] += sum over pixels of x[..][ch] for ch < cvalid (zero db first;
 * replaces a virnet_colsum pass over the tensor); col_scratch = virnet_chsplit_colsum_bytes() bytes of per-block partial sums. */
int virnet_chsplit(const float* x, int n, int h, int w, int c, int in_act, float in_slope, const float* in_mul, const float* in_add,
                   int bf16, void* out, float* col_scratch, float* db, int cvalid, void* stream);
int virnet_conv_wgrad_f16(const void* xt, const void* yt, float* dw, float* scratch, int n, int h, int w, int cx, int cy, int cin, int cout,
                          int bf16, void* stream);
/* ... and, in the same reduction launch, the bias gradient from the column partials an emitting convolution left for yt's tensor
 * (virnet_t_emit.col, nblk from virnet_conv_emit_ok): db[c] += sum over blocks, c < cvalid (zero db first). */
int virnet_conv_wgrad_f16_db(const void* xt, const void* yt, float* dw, float* scratch, int n, int h, int w, int cx, int cy, int cin, int cout,
                             int bf16, const float* col, float* db, long nblk, int cvalid, void* stream);
/* The weight gradients of the two stride-2 layers on the same kernel (backward of DownBlock.downsampler networks/AttResUNet.py:67 and
 * UpBlock.upsampler :80).  The HIGH-resolution operand [n][h][w][c] (c % 32 == 0, w even) is re-laid by virnet_chsplit_s2 into a
 * column-phase T: one row per source row, [h+2][2*c/32 blocks][hi|lo][seg(w/2)][32 ch][8 px], block par*c/32 + k = the pixels
 * x = 2*ox + par of channel block k -- every tap of a stride-2 window is then an aligned 16-pixel run (or one shifted by a single T
 * pixel), and a step of the contraction (low-res row oy) reads the high-res rows 2oy-1 .. 2oy+1: two new ring rows per step.
 * The LOW-resolution operand [n][oh][ow][clo] is a plain virnet_chsplit T.
 *   mode 0: 3x3 stride-2 pad-1 conv, hi = the conv's input (cin real channels), lo = its output gradient (cout): dw[cout][cin][3][3]
 *   mode 1: 2x2 stride-2 transposed conv, hi = its output gradient (cout), lo = its input (cin):              dw[cin][cout][2][2]
 * dw is overwritten (fixed summation order: bitwise reproducible); scratch = virnet_conv_wgrad_f16_s2_scratch_bytes() bytes. */
size_t virnet_chsplit_s2_bytes(int n, int h, int w, int c);
size_t virnet_chsplit_s2_colsum_bytes(int n, int h, int w, int c);
int virnet_chsplit_s2(const float* x, int n, int h, int w, int c, int in_act, float in_slope, const float* in_mul, const float* in_add,
                      int bf16, void* out, float* col_scratch, float* db, int cvalid, void* stream);
size_t virnet_conv_wgrad_f16_s2_scratch_bytes(int n, int oh, int ow, int chi, int clo);
int virnet_conv_wgrad_f16_s2(const void* hi_t, const void* lo_t, float* dw, float* scratch, int n, int oh, int ow, int chi, int clo,
                             int cin, int cout, int mode, int bf16, void* stream);
/* T emission (training step): a stride-1 3x3 NHWC convolution with a single-store epilogue (y_raw XOR y_act; residual and / or mask
 * allowed; no output SFT, no in_mul for the Winograd form) can write, besides its NHWC tensor, the channel-major T image of that
 * tensor -- exactly what a virnet_chsplit pass over it would produce, without reading it back (the 74 re-layout passes of a training
 * step were 13 % of it) -- and per-workgroup channel sums of it (the bias gradient of the conv whose output gradient this tensor is).
 *   t_out : virnet_chsplit_bytes(n, h, w, cout) bytes whose rows 0 / h+1 and pad segments are ALREADY zero (the kernel writes the image
 *           rows and, inside them, zeros for tile pixels beyond w; it never touches the pads) -- bf16 != 0: one bf16 plane
 *   act   : T holds lrelu(y, slope) instead of y (the staging transform of the NEXT conv, AttResUNet.py:55, whose weight gradient reads it)
 *   col   : NULL, or virnet_conv_emit_ok()'s nblk * cout floats: partial sums col[(cb * nblk + blk) * 32 + ch] of y over the pixels of
 *           workgroup-wave blk; virnet_colpart_reduce adds them into db (zero db first)
 * virnet_conv_emit_ok: 1 when `d` can run with emission in that form (0: virnet_conv_f16 / virnet_conv_bf16, 1: virnet_conv_wx4 with 16-row
 * tiles, 2: with 8-row tiles). */
typedef struct virnet_t_emit {
  void* t_out;
  float* col;
  int act;
  float slope;
  int bf16;
  int rows;    /* virnet_conv_wx4_emit: 0 / 16 = 16-row tiles (one workgroup per CU), 8 = 8-row tiles (two per CU: the image's stores
                  run beside the other workgroup's K loop); virnet_conv_emit_ok form 1 / 2 */
} virnet_t_emit;
int virnet_conv_emit_ok(const virnet_conv_desc* d, int form, int* nblk);
int virnet_conv_f16_emit(const virnet_conv_desc* d, const virnet_t_emit* te, int bf16_operands, void* stream);
int virnet_conv_wx4_emit(const virnet_conv_desc* d, const virnet_t_emit* te, void* stream);
int virnet_colpart_reduce(const float* col, float* db, long nblk, int ncb, int cvalid, void* stream);
/* Backward of the SFT pre-activation a = lrelu(x*mul + add, slope) with per-image [n][c] vectors (AttResUNet.py:54-58), given da = dL/da:
 * du = da * lrelu'(x*mul+add);  dx = du*mul (+ res, the skip gradient, may be NULL);  dmul[n][c] += sum_p du*x;  dadd[n][c] += sum_p du
 * (zero dmul / dadd first).  NHWC tensors of n images x hw pixels x c channels (c % 4 == 0). */
int virnet_sft_backward(const float* da, const float* x, const float* mul, const float* add, const float* res, float slope, float* dx,
                        float* dmul, float* dadd, int n, long hw, int c, void* stream);
/* db[c] += sum over pixels of dy[p][c], c < cvalid (NHWC rows of `c` stored channels, c % 4 == 0); zero db first */
int virnet_colsum(const float* dy, float* db, long npix, int c, int cvalid, void* stream);
/* z[n][2h][2w][c] = dy at even positions, 0 elsewhere: the stride-2 conv's dgrad is a stride-1 conv of z (AttResUNet.py:67) */
int virnet_zero_stuff2(const float* dy, float* z, int n, int h, int w, int c, void* stream);
/* out[n][h][w][(a*2+b)*c + k] = dy[n][2h+a][2w+b][k]: the transposed conv's dgrad / wgrad operand (AttResUNet.py:80) */
int virnet_space_to_depth2(const float* dy, float* out, int n, int h, int w, int c, void* stream);
/* Adjoint of virnet_pack_input for ONE map channel: dmap[n][h][w] (+)= sum over padded positions that read (y,x) of
 * drec[n][hp][wp][crec][chan] * (map_sqrt ? 0.5/sqrt(map[y][x]) : 1)   (reflect pad of util_net.py:20-25, sqrt of VIRNet.py:44) */
int virnet_pack_input_backward(const float* drec, int crec, int chan, const float* map, float* dmap, int n, int h, int w, int hp,
                               int wp, int map_sqrt, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * KNet pieces (networks/KNet.py) and the SFT generator (networks/AttResUNet.py:11-32).  Small, latency-bound kernels.
 * ---------------------------------------------------------------------------------------------- */

/* KernelNet.head: Conv2d(cin -> cout, k=9, s=4, p=4, bias=False) (KNet.py:45,53).  x NCHW [n][cin][h][w], w OIHW,
 * out NHWC [n][oh][ow][cout] with oh = (h-1)/4+1, ow = (w-1)/4+1; cout a multiple of 64. */
int virnet_conv_head_s4(const float* x, const float* w, float* out, int n, int cin, int h, int w_, int cout, void* stream);
/* Its weight gradient (SISR training step, train_SISR.py:207-224): dw[cout][cin][9][9] = sum_{n,oy,ox} dy[n][oy][ox][co] *
 * x[n][ci][4oy+ky-4][4ox+kx-4]; x NCHW, dy NHWC [n][oh][ow][cout]; dw is overwritten. */
int virnet_conv_head_s4_wgrad(const float* x, const float* dy, float* dw, int n, int cin, int h, int w_, int cout, void* stream);

/* Global average pool of a planar tensor [n][c][h][w] -> out[n][c], with the finishing op of its call site:
 *   VIRNET_GAP_MEAN      mean                                  (DnCNN.py:31,42)
 *   VIRNET_GAP_EXPCLAMP  exp(clamp(mean, lo, hi))              (VIRNet.py:81 on the pooled SNet output)
 *   VIRNET_GAP_KINFO     channels 0..c-2 exp(clamp(mean, lo, hi)), channel c-1 tanh(mean)   (KNet.py:50,56-59) */
enum { VIRNET_GAP_MEAN = 0, VIRNET_GAP_EXPCLAMP = 1, VIRNET_GAP_KINFO = 2 };
int virnet_gap_nchw(const float* x, float* out, int n, int c, int h, int w, int finish, float lo, float hi, void* stream);

/* CALayer gate (KNet.py:15-25): gate[n][c] = sigmoid(W2 lrelu0.2(W1 mean_hw(x[n]) + b1) + b2); x NHWC [n][h][w][c];
 * W1 [cr][c], W2 [c][cr] (the 1x1 conv weights); c <= 256 and a divisor of 256. */
int virnet_ca_gate(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* gate,
                   int n, int h, int w, int c, int cr, void* stream);

/* RB_Layer tail (KNet.py:26,38): out = hcv * gate[n][c] + skip, all NHWC [n][h][w][c], c % 4 == 0. */
int virnet_scale_add(const float* hcv, const float* gate, const float* skip, float* out, int n, int hw, int c, void* stream);

/* CALayer + RB_Layer tail in one launch (KNet.py:15-26,38): out = hcv * sigmoid(W2 lrelu0.2(W1 mean_hw(hcv) + b1) + b2) + skip, for
 * maps of at most 16384 float4 items per image (h*w*c/4; KernelNet's 16x16x64 map has 4096): one workgroup holds an image in
 * registers.  Same arguments and limits as virnet_ca_gate / virnet_scale_add, which remain for larger maps. */
int virnet_ca_scale_add(const float* hcv, const float* w1, const float* b1, const float* w2, const float* b2, const float* skip,
                        float* out, int n, int h, int w, int c, int cr, void* stream);

/* KernelNet body as ONE persistent kernel (KNet.py:28-39,46-48,54: `nlayers` RB_Layers = conv3x3 + LeakyReLU(0.2) + conv3x3 + CALayer +
 * skip, each), one workgroup per image holding the h x w x 64 map on the CU (csrc/knet_body.hip): for maps of at most 16 x 16 pixels
 * (LR images up to 64 x 64 behind the stride-4 head).  x, y: NHWC [n][h][w][64] fp32 (y may alias x).  Per layer: w1pack / w2pack =
 * virnet_pack_f16_weight images of the two 64->64 convs (cin_pad = n_pad = 64), b1 / b2 their biases (NULL = none), caw1 [cr][64], cab1
 * [cr], caw2 [64][cr], cab2 [64] the CALayer's 1x1 convs.  Same split-fp16 arithmetic per product as virnet_conv_f16; the channel means
 * are summed in a fixed order (bitwise reproducible).  Larger maps: virnet_conv_f16 + virnet_ca_scale_add per layer. */
typedef struct virnet_knet_layer {
  const float *w1pack, *b1, *w2pack, *b2;
  const float *caw1, *cab1, *caw2, *cab2;
} virnet_knet_layer;
int virnet_knet_body(const float* x, float* y, const virnet_knet_layer* layers, int nlayers, int n, int h, int w, int c, int cr, void* stream);

/* AttLayer weights (AttResUNet.py:18-25), all 1x1 convs stored [cout][cin]. */
typedef struct virnet_sft_weights {
  const float *w1, *b1;   /* [nf1][e]   */
  const float *w2, *b2;   /* [nf2][nf1] */
  const float *wm, *bm;   /* [nf][nf2]  mul_conv (sigmoid) */
  const float *wa, *ba;   /* [nf][nf2]  add_conv */
  int e, nf1, nf2, nf;
} virnet_sft_weights;

/* Spatially constant conditioning: mul[n][nf], add[n][nf] from vec[n][e] (AttLayer.forward, AttResUNet.py:27-32). */
int virnet_sft_vec(const float* vec, const virnet_sft_weights* wt, float* mul, float* add, int n, void* stream);
/* The same for `nlayers` (1..16) AttLayers on ONE vector in one launch (every AttResBlock.sft1 / sft2 of the down path, AttResUNet.py:50,57,
 * depends on nothing but the conditioning vector): wts[l] -> muls[l][n][nf_l], adds[l][n][nf_l]; all layers take the same `e`.  (ABI 3) */
int virnet_sft_vec_multi(const float* vec, const virnet_sft_weights* wts, int nlayers, float* const* muls, float* const* adds, int n, void* stream);

/* Per-pixel conditioning: act = lrelu0.2(raw * mul(e) + add(e)) (AttResUNet.py:54-58) where e = channels
 * [chan0, chan0+wt->e) of the full-resolution 16-channel records rec[n][hp][wp][16] sampled at (y*step, x*step)
 * -- the nearest resize of AttResUNet.py:168.  raw/act NHWC [n][h][w][nf], hp = h*step, wp = w*step. */
int virnet_sft_apply(const float* raw, const float* rec, const virnet_sft_weights* wt, float* act, int n, int h, int w,
                     int step, int chan0, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VIRNET_HIP_H */
