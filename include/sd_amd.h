/* sd_amd.h — C ABI of libsdamd.so: MI355X (gfx950) kernels for the contrastive training hot path of
 * SeanNobel/speech-decoding (BrainEncoder + CLIPLoss + Classifier).
 *
 * The reference is pure Python/PyTorch and has no FFI of its own: what a binding
 * would replace are the aten calls made by
 *     speech_decoding/models.py:45-65    SpatialAttention.forward      -> sda_sa_weights_*, sda_conv_gemm
 *     speech_decoding/models.py:111-117  SubjectBlock.forward          -> sda_conv_gemm (widx = subject)
 *     speech_decoding/models.py:152-166  ConvBlock.forward             -> sda_conv_gemm, sda_bn_*, sda_glu_*
 *     speech_decoding/models.py:191-196  BrainEncoder.forward          -> sda_conv_gemm (GELU epilogue)
 *     speech_decoding/utils/loss.py:58-79 CLIPLoss.forward (fast path) -> sda_rows_sumsq, sda_conv_gemm (split-K),
 *                                                                         sda_clip_logits_stats, sda_clip_grad,
 *                                                                         sda_wgrad_gemm (typed output)
 *     speech_decoding/models.py:208-248  Classifier.forward            -> sda_clip_ranks
 * and their autograd backward.  Every entry point takes plain device pointers, sizes and a
 * hipStream_t (passed as void*); no torch types cross this boundary.  All functions return 0 on
 * success and a negative code on failure; sda_last_error() gives the message.
 *
 * Activation "row layout" (RL): a (B, C, T) tensor is stored channels-last as rows of Cp elements,
 *     row(b, t) = b * (T + SDA_ROW_PAD) + SDA_ROW_PAD + t,        Cp = C rounded up to 64,
 * with all padding rows / channels equal to zero, in a buffer of sda_rows_alloc(B, T) rows.
 */
#ifndef SD_AMD_H
#define SD_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDA_ABI_VERSION 4   /* 2: sda_conv_args gained glu_out / glu_gate, sda_pack_desc gained glu_tile, flag 16384 = SDA_CONV_FLAT_TILES;
                               (still 2, should have been bumped: sda_wgrad_args.acc_scale, new arguments of sda_bn_finalize,
                               sda_clip_logits_stats and sda_clip_grad, new entries sda_clip_dz / sda_param_gemm / sda_copy3d)
                               3: sda_wgrad_args.flags (SDA_WGRAD_FLAT_ROWS), sda_stream_create_cumask / sda_stream_create_priority / sda_stream_destroy, sda_sim_gemm / sda_sim_gemm_ksplit,
                               conv3_flat takes x_pitch == w_pitch only
                               4: sda_fill_zero, sda_gather_samples, sda_clip_merge_rows; SDA_WGRAD_FLAT_ROWS with a sample permutation;
                                  sda_clip_dz serves more than 256 speech rows (256 x 256 tiles); SDA_CONV_WIDE_TILES */
#define SDA_ROW_PAD 16
#define SDA_CH_ALIGN 64

enum { SDA_F32 = 0, SDA_BF16 = 1, SDA_F16 = 2 };   /* storage + MFMA operand type; accumulation is always fp32 */

/* conv_gemm epilogue flags */
enum { SDA_EPI_GELU = 1,
       SDA_EPI_GLU_BWD = 4,         /* the conv's output is the gradient dy entering an F.glu whose forward kept (out, gate) (SDA_EPI_GLU):
                                       y [rows][2 * Cout_p] = [dy * sig(gate) | dy * out * (1 - sig(gate))] (the GLU backward, written
                                       instead of dy), `stats` rows = per-tile column sums of the two halves ([tile][0] value half,
                                       [tile][1] gate half: the bias gradient of the conv that fed the GLU).  Needs glu_out, glu_gate and
                                       stats; the tile-per-workgroup kernels only (not SDA_CONV_FLAT_TILES); no GELU / bn_x */
       SDA_EPI_GLU = 2,             /* with SDA_CONV_FLAT_TILES: the conv's Cout_p = 2 * Hp channels are [value | gate] pairs
                                       laid out per 160-channel tile as 80 value + 80 gate channels (weights and bias packed with
                                       glu_tile = 80); y [rows][Hp] = value * sigmoid(gate) (models.py:164, F.glu), both as
                                       rounded to the storage type; y_pre [rows][Hp] = gate, or NULL.  No res / stats / bn_x */
       SDA_CONV_SINGLE_TILE = 4096, /* force one 128-row tile per workgroup */
       SDA_CONV_PAIR_TILES = 8192,  /* two tiles per workgroup sharing one weight slab (default: one) */
       SDA_CONV_ONE_PER_CU = 32768, /* with SDA_CONV_FLAT_TILES: at most one workgroup per CU (half the LDS stays free) */
       SDA_CONV_FLAT_TILES = 16384, /* kernel size 3 with Cout_p % 160 == 0: 256-row x 160-channel tiles cut from the
                                       flat row space, two workgroups per CU (conv3_flat.hip); kernel size 1 with
                                       Cout_p % 160 == 0 or Cout_p % 128 == 0, shared weights, no residual / statistics:
                                       the same tiling (conv1_flat.hip); other shapes ignore the flag.  `stats` then has
                                       sda_conv_stats_rows(...) rows instead of B * n_t_tiles */
       SDA_CONV_WIDE_TILES = 524288,/* kernel size 1, 16-bit storage, Cout_p % 256 == 0 or % 320 == 0, shared weights, no residual: 256-row x
                                       256- / 320-channel tiles of the flat row space, eight waves, one persistent workgroup per CU
                                       (conv1_wide.hip); takes SDA_EPI_GELU (+ y_pre), SDA_EPI_GELU_BWD (`stats` = one row per 256-row
                                       tile: sda_conv_stats_rows), SDA_EPI_ROW_SUMSQ (Cout_p % 256 == 0).  Any other shape with this
                                       flag is an error, not a fallback */
       SDA_CONV_WAVE_PRIO = 1048576,/* the tile-per-workgroup kernels: the launch's waves run at s_setprio 3 (they win a SIMD's issue
                                       arbitration against co-resident waves of other kernels) */
       SDA_EPI_GELU_BWD = 65536,    /* kernel size 1 (flat tiles or one tile per workgroup): the conv's output is the gradient entering a GELU whose input u = bn_x ([rows][Cout_p],
                                       same layout as y) the forward kept: y = round(conv) * GELU'(u) (what sda_gelu_backward_colsum
                                       computes from the stored gradient), `stats` rows (sda_conv_stats_rows) = per-unit column
                                       sums of the products in plane 0 (the bias gradient of the layer that fed the GELU), plane 1
                                       zero.  No bias / y_pre / SDA_EPI_GELU */
       SDA_EPI_BN_STORE_DG = 262144,/* with bn_x (the BatchNorm-backward statistics epilogue): y = dg = round(dy) * GELU'(gamma * xhat + beta),
                                       rounded to the storage type, INSTEAD of dy, and the statistics are those of dg as stored —
                                       sda_bn_gelu_backward_from_stats_dg / _apply_dg then finish the BatchNorm backward without
                                       evaluating GELU' a second time */
       SDA_EPI_ROW_SUMSQ = 131072   /* the flat kernel-size-1 form only (SDA_CONV_FLAT_TILES must be set, else the call fails), Cout_p % 128 == 0: `stats` = float [>= B * (T + SDA_ROW_PAD) rows][Cout_p / 128]; entry
                                       [r][j] = sum of squares of buffer row r's output channels [128 j, 128 j + 128) as stored
                                       (rows that are padding are not written); sda_rows_sumsq_from_row_parts turns them into
                                       per-sample norms */ };

int sda_abi_version(void);
const char* sda_last_error(void);
long sda_rows_alloc(int B, int T);
int sda_pad_channels(int C);

/* (B, C, T) fp32 contiguous  ->  RL rows of Cp elements of `dtype` (channel padding zero-filled; pad
 * rows are NOT touched: the caller zero-initialises the buffer once). */
int sda_pack_rows(const float* src, void* dst, int B, int C, int T, int Cp, int dtype, void* stream);
/* the same with padding channel `ones_channel` (C <= ones_channel < Cp) set to 1 on every valid row: a bias folded into a
 * per-sample weight matrix rides on it (the composed SubjectBlock, models.py:111-117) */
int sda_pack_rows_ones(const float* src, void* dst, int B, int C, int T, int Cp, int ones_channel, int dtype, void* stream);
/* RL -> (B, C, T) fp32 contiguous */
int sda_unpack_rows(const void* src, float* dst, int B, int C, int T, int Cp, int dtype, void* stream);
/* (B, C, T) of `dtype`... not provided: gradients enter/leave in RL. */

/* per-sample sum of squares over an RL tensor viewed as B rows of `row_elems` contiguous elements with
 * pitch `pitch` (elements). out[b] fp32. `scratch` holds B*64 floats. */
int sda_rows_sumsq(const void* x, float* out, float* scratch, int B, long row_elems, long pitch,
                   int dtype, void* stream);
/* the same norms for a tensor a conv just produced, from that conv's per-tile statistics ([B * tiles_per_sample][2][Cp],
 * plane 1 = sum of squares of the stored values): no second pass over the tensor */
int sda_rows_sumsq_from_stats(const float* stats, int tiles_per_sample, int Cp, float* out, int B, void* stream);
/* the same from SDA_EPI_ROW_SUMSQ's per-row partial sums ([rows][n_parts]): out[b] = sum over sample b's T rows and
 * n_parts entries, fp64, fixed order */
int sda_rows_sumsq_from_row_parts(const float* parts, int n_parts, float* out, int B, int T, void* stream);

/* fp32 conv weight [nW][Cout][Cin][KS]  ->  packed `dtype` operand.
 * mode 0 (forward):  dst[n][tap][co'][ci]  = w[n][co][ci][tap]
 * mode 1 (dgrad):    dst[n][tap][ci][co']  = w[n][co][ci][KS-1-tap]
 * co' = co if glu_half == 0, else (co < glu_half ? co : glu_half_p + (co - glu_half)).
 * Rows/cols beyond the source extent are zero. Cout_p/Cin_p are the padded extents of co'/ci. */
int sda_pack_conv_weight(const float* w, void* dst, int nW, int Cout, int Cin, int KS, int Cout_p,
                         int Cin_p, int mode, int glu_half, int glu_half_p, int dtype, void* stream);
/* All operand packs of a step in one launch.  `descs` is a DEVICE array of n descriptors: a weight pack as in
 * sda_pack_conv_weight (dst of `dtype`), or with is_vector != 0 a bias pack as in sda_pack_vector (Cout = C,
 * Cout_p = Cp, dst fp32).  total = number of destination elements; max_total = the largest of them. */
typedef struct sda_pack_desc {
  const float* src;
  void* dst;
  int nW, Cout, Cin, KS, Cout_p, Cin_p, mode, glu_half, glu_half_p, is_vector;
  int glu_tile;   /* 0: the two GLU halves stay contiguous ([0, glu_half_p) values, then gates); 80: SDA_EPI_GLU's layout —
                   * packed output channel 160 j + w is value channel 80 j + w (w < 80) or gate channel 80 j + w - 80 */
  long total;
} sda_pack_desc;
int sda_pack_multi(const sda_pack_desc* descs, int n, long max_total, int dtype, void* stream);
/* dst fp32 [Cout][Cin][KS] = ordered sum over nslabs K-split slabs [KS][Cout_p][Cin_p] (sda_wgrad_gemm output),
 * i.e. sda_reduce_slabs + sda_unpack_conv_wgrad in one pass */
int sda_reduce_unpack_wgrad(const float* slabs, int nslabs, float* dst, int Cout, int Cin, int KS, int Cout_p,
                            int Cin_p, int glu_half, int glu_half_p, void* stream);
/* Adam step (torch.optim.Adam defaults: no weight decay, no amsgrad; train.py:161-163) over n tensors in one
 * launch.  descs: DEVICE array; n = number of fp32 elements (complex parameters count their real view);
 * aligned != 0 when all four pointers are 16-byte aligned.  step = 1-based update count (bias correction). */
typedef struct sda_adam_desc {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  long n;
  int aligned;
} sda_adam_desc;
int sda_adam_multi(const sda_adam_desc* descs, int n, long max_n, float lr, float beta1, float beta2, float eps,
                   long step, void* stream);
/* fp32 vector [C] -> padded fp32 [Cp] with the same GLU remap */
int sda_pack_vector(const float* v, float* dst, int C, int Cp, int glu_half, int glu_half_p, void* stream);
/* inverse of mode 0 for gradients: fp32 [nW][KS][Cout_p][Cin_p] -> fp32 [nW][Cout][Cin][KS] */
int sda_unpack_conv_wgrad(const float* g, float* dst, int nW, int Cout, int Cin, int KS, int Cout_p,
                          int Cin_p, int glu_half, int glu_half_p, void* stream);
int sda_unpack_vector(const float* g, float* dst, int C, int Cp, int glu_half, int glu_half_p, void* stream);

/* Implicit-GEMM 1-D convolution, kernel size 1 or 3, stride 1, "same" zero padding, dilation <= 16:
 *   y[b, t, co] = bias[co] + sum_{tap, ci} x[b, t + (tap - KS/2) * dil, ci] * w[widx[b]][tap][co][ci]  (+ res[b, t, co])
 * Serves the forward convolutions, the data gradients (weights packed with mode 1) and, with
 * ksplit > 1, the B x B similarity matmul of the loss (raw fp32 partial slabs, no epilogue). */
typedef struct sda_conv_args {
  const void* x;        /* RL [rows][x_pitch]; contraction over Cin_p elements of each row */
  const void* w;        /* packed [nW][KS][Cout_p][w_pitch] */
  const float* bias;    /* [Cout_p] or NULL */
  const void* res;      /* RL [rows][Cout_p] residual added in the epilogue, or NULL */
  void* y;              /* RL [rows][Cout_p] output (post-activation when SDA_EPI_GELU; [rows][Cout_p / 2] with SDA_EPI_GLU) */
  void* y_pre;          /* with SDA_EPI_GELU: pre-activation output, or NULL; with SDA_EPI_GLU: the gate, or NULL */
  const int32_t* widx;  /* [B] weight selector per sample (device), or NULL */
  float* stats;         /* [B * n_t_tiles][2][Cout_p] per-tile (sum, sum of squares) over valid rows, or NULL */
  float* partial;       /* ksplit > 1: [ksplit][T][Cout_p] fp32 raw accumulators (B must be 1) */
  const void* bn_x;     /* optional RL [rows][Cout_p]: input of the BatchNorm+GELU whose OUTPUT gradient this conv
                         * produces (data-gradient convs).  With bn_x, `stats` receives the per-tile sums of the
                         * BatchNorm backward instead: [tile][0][c] = sum dg, [tile][1][c] = sum dg * xhat, with
                         * dg = y * GELU'(gamma * xhat + beta), xhat = (bn_x - mean) * rstd, y as stored */
  const float* bn_coef; /* with bn_x: [4][Cout_p] = gamma, beta, mean, rstd (zero on padded channels) */
  const void* glu_out;  /* with SDA_EPI_GLU_BWD: RL [rows][Cout_p] forward output of the GLU (value * sigmoid(gate)) ... */
  const void* glu_gate; /* ... and its gate, RL [rows][Cout_p] */
  int B, T, Cin_p, Cout_p, KS, dil;
  long x_pitch, w_pitch; /* elements per row of x / per output-channel row of w.  With KS == 1, x_pitch < Cin_p is allowed:
                          * rows then overlap — a frame-major buffer read with pitch stride*C and row length k*C is the
                          * im2col matrix of a strided Conv1d (the wav2vec2 feature encoder) */
  long x_row0;           /* first row of sample 0 (SDA_ROW_PAD for RL, 0 for plain matrices) */
  long x_sample_rows;    /* rows between consecutive samples (T + SDA_ROW_PAD for RL) */
  long x_rows_limit;     /* rows >= limit read as zero */
  int w_rows_limit;      /* output channels >= limit read as zero (Cout_p when fully padded) */
  int ksplit;            /* >= 1 */
  int flags, dtype;
} sda_conv_args;
int sda_conv_gemm(const sda_conv_args* a, void* stream);
int sda_conv_n_t_tiles(int T);
/* number of [2][Cout_p] rows of `stats` a launch with these parameters writes */
int sda_conv_stats_rows(int B, int T, int KS, int Cout_p, int flags);

/* BatchNorm1d (training statistics) over valid rows.
 * finalize: per-tile partial (sum, sumsq) -> mean/rstd, fused affine scale/shift, running-stat update
 * (momentum, unbiased variance) when running_mean != NULL.  In eval mode call with ntiles = 0 and the
 * running stats: scale/shift are derived from them. */
int sda_bn_finalize(const float* partial, int ntiles, double count, const float* gamma, const float* beta,
                    float eps, float momentum, float* running_mean, float* running_var, float* mean,
                    float* rstd, float* scale, float* shift, float* bwd_coef /* optional [4][Cp]: gamma, beta,
                    mean, rstd for sda_conv_args.bn_coef */, int C, int Cp, int training,
                    long* batches_tracked /* optional: the module's num_batches_tracked, += 1 in training mode */, void* stream);
/* y = GELU(x * scale[c] + shift[c]) on valid rows */
int sda_bn_gelu_forward(const void* x, void* y, const float* scale, const float* shift, int B, int T,
                        int Cp, int dtype, void* stream);
/* backward of y = GELU(BN(x)) (training statistics), two calls so that data-parallel ranks can
 * all-reduce the sums in between:  dg = dy * GELU'(u);
 * reduce: dbeta = sum dg, dgamma = sum dg*xhat over this rank's valid rows (fp32 [Cp]; ordered
 *         two-stage reduction through `partial`, sda_reduce_scratch_floats(Cp) floats);
 * apply:  dx = gamma*rstd*(dg - dbeta/count - xhat*dgamma/count), count = GLOBAL number of rows. */
int sda_bn_gelu_backward_reduce(const void* dy, const void* x, const float* mean, const float* rstd,
                                const float* gamma, const float* beta, int C, float* partial, float* dgamma,
                                float* dbeta, int B, int T, int Cp, int dtype, void* stream);
/* out0[c] = sum_k partial[k][0][c] (and out1 from [k][1][c] when out1 != NULL): fixed-order fp64 reduction of
 * per-tile statistics written by sda_conv_gemm (`stats`), e.g. the BatchNorm backward sums of bn_x mode. */
int sda_reduce_stats(const float* partial, int nrows, float* out0, float* out1, int Cp, void* stream);
/* Single-process shortcut: per-tile sums (sda_conv_gemm bn_x mode) -> dgamma/dbeta + coefficient table in ONE
 * launch, then the apply pass.  (Data-parallel runs use sda_reduce_stats, all-reduce the two sums, then ..._apply.) */
int sda_bn_gelu_backward_from_stats(const float* partial, int nrows, const void* dy, const void* x, const float* mean,
                                    const float* rstd, const float* gamma, const float* beta, int C, double count,
                                    float* dgamma, float* dbeta, float* coef, void* dx, int B, int T, int Cp, int dtype,
                                    void* stream);
int sda_bn_gelu_backward_apply(const void* dy, const void* x, const float* mean, const float* rstd,
                               const float* gamma, const float* beta, int C, const float* dgamma,
                               const float* dbeta, double count, float* coef /* 6*Cp floats scratch */, void* dx,
                               int B, int T, int Cp, int dtype, void* stream);
/* the same two for a gradient buffer that already holds dg = dy * GELU'(gamma * xhat + beta) (written by sda_conv_gemm with
 * SDA_EPI_BN_STORE_DG, whose statistics rows are those of dg as stored): no second GELU' evaluation */
int sda_bn_gelu_backward_from_stats_dg(const float* partial, int nrows, const void* dy, const void* x, const float* mean,
                                    const float* rstd, const float* gamma, const float* beta, int C, double count,
                                    float* dgamma, float* dbeta, float* coef, void* dx, int B, int T, int Cp, int dtype,
                                    void* stream);
int sda_bn_gelu_backward_apply_dg(const void* dy, const void* x, const float* mean, const float* rstd,
                               const float* gamma, const float* beta, int C, const float* dgamma,
                               const float* dbeta, double count, float* coef /* 6*Cp floats scratch */, void* dx,
                               int B, int T, int Cp, int dtype, void* stream);

int sda_reduce_scratch_floats(int Cp);

/* Zero the rows of an RL buffer that kernels never write: the SDA_ROW_PAD rows in front of each of the B samples and the
 * slack behind the last one (a buffer whose valid rows a kernel is about to fill needs nothing else initialised) */
int sda_zero_pad_rows(void* buf, int B, int T, int Cp, int dtype, void* stream);
/* nbytes zero bytes at p (16-byte aligned): the fresh zero gradients a backward hands out (conv biases in front of a training-mode
 * BatchNorm: autograd's zero, speech_decoding/models.py:135,143) without a framework fill kernel in the step */
int sda_fill_zero(void* p, long nbytes, void* stream);
/* dst sample b = table sample idx[b] (idx: B int64 on the device), samples of sample_bytes contiguous bytes (a multiple of 16).
 * The resident-feed form of `Y = dataset.Y[batch indices]` (gwilliams2022.py:129-142 hands out one embedding per item): with the
 * embedding table kept in row layout (a sample = (T + SDA_ROW_PAD) * Cp elements, pad rows included) the gathered batch IS the
 * packed row-layout operand of the loss — no index_select + sda_pack_rows pair per step */
int sda_gather_samples(const void* table, const long* idx, void* dst, int B, long sample_bytes, void* stream);
/* out[i] = a[i] * b[0], i < n (device scalars: e.g. d loss / d temp times the incoming gradient) */
int sda_scalar_mul(const float* a, const float* b, float* out, int n, void* stream);

/* GLU over channels: y[:, c] = x[:, c] * sigmoid(x[:, Ch + c]), x has 2*Ch channels (models.py:164) */
int sda_glu_forward(const void* x, void* y, int B, int T, int Ch, int dtype, void* stream);
int sda_glu_backward(const void* x, const void* dy, void* dx, int B, int T, int Ch, int dtype, void* stream);
/* du = dz * GELU'(u) */
int sda_gelu_backward(const void* u, const void* dz, void* du, int B, int T, int Cp, int dtype, void* stream);
/* the same two backward stages fused with the column sums of their own output (= the bias gradient of the
 * layer that produced x / u): colsum fp32 [2*Ch] resp. [Cp]; scratch sda_reduce_scratch_floats(.) floats */
/* (colsum == NULL: the final reduction is left to the caller — scratch then holds sda_reduce_scratch_rows(B, T) rows of
 * [2][Ch] (resp. [2][Cp], slot 1 unused) partial sums for sda_reduce_stats, e.g. on another stream) */
int sda_reduce_scratch_rows(int B, int T);
int sda_glu_backward_colsum(const void* x, const void* dy, void* dx, float* colsum, float* scratch, int B, int T,
                            int Ch, int dtype, void* stream);
int sda_gelu_backward_colsum(const void* u, const void* dz, void* du, float* colsum, float* scratch, int B, int T,
                             int Cp, int dtype, void* stream);
/* GLU backward after a forward with SDA_EPI_GLU (the value half was never stored): out = value * sigmoid(gate) and gate,
 * both RL [rows][Ch]; dx RL [rows][2*Ch] = [d value | d gate] = [dy * sig(g) | dy * out * (1 - sig(g))]; colsum as above */
int sda_glu_backward_colsum_og(const void* out, const void* gate, const void* dy, void* dx, float* colsum, float* scratch,
                               int B, int T, int Ch, int dtype, void* stream);
/* column sums of an RL tensor over valid rows: out[c] = sum_{b,t} x[b,t,c]  (bias gradients) */
int sda_colsum(const void* x, float* out, float* scratch /* sda_reduce_scratch_floats(Cp) */, int B, int T,
               int Cp, int dtype, void* stream);

/* Weight-gradient GEMM:  g[seg][tap][co][ci] = sum_{b in seg} sum_t dy[b,t,co] * x[b, t+(tap-KS/2)*dil, ci]
 * Samples are visited through `perm` (device int32 [B]); segment s covers perm[seg_start[s]..seg_start[s+1]).
 * Output fp32 slabs [nseg][KS][Cout_p][Cin_p].  With out_e != NULL (KS must be 1, nseg 1) the result is
 * written as `dtype` rows instead:  out_e[co][ci] = out_scale * (acc_scale[co] * acc - rscale[co] * sub[co][ci])   (loss backward dZ). */
/* sda_wgrad_args.flags */
enum { SDA_WGRAD_FLAT_ROWS = 1  /* dy is a row-layout buffer whose rows between samples (the SDA_ROW_PAD rows in front of every
                                   sample, the slack behind the last) are zero.  perm == NULL: a segment is contracted as ONE run of
                                   rows, pad rows included (they contribute nothing), in whole K-chunks — no partial chunk per
                                   sample.  perm != NULL (ABI 4): every SAMPLE is extended into the zero rows around it to whole
                                   K-chunks (when (T rounded up to the chunk) - T <= 2 * SDA_ROW_PAD; ignored otherwise) */ };
typedef struct sda_wgrad_args {
  const void* dy;       /* RL [rows][dy_pitch] */
  const void* x;        /* RL [rows][x_pitch] */
  float* g;             /* [nseg][KS][Cout_p][Cin_p] fp32 */
  void* out_e;          /* optional typed output [Cout rows][out_pitch] */
  const void* sub;      /* optional [Cout rows][out_pitch] */
  const float* rscale;  /* optional [Cout] */
  const float* out_scale; /* optional device scalar multiplying the typed output (incoming dloss) */
  const int32_t* perm;
  const int32_t* seg_start; /* device int32 [nseg + 1] */
  int nseg, B, T, Cout_p, Cin_p, KS, dil;
  long dy_pitch, x_pitch, out_pitch;
  long row0, sample_rows; /* as in sda_conv_args */
  long rows_limit;        /* x rows are clamped into [0, rows_limit) (they only meet zero dy rows out there) */
  long dy_zero_row;       /* index of a row of dy that is all zero (row 0 of any RL buffer); stands in for t >= T */
  int co_valid;           /* rows of out_e to write (out_e mode) */
  int dtype;
  const float* acc_scale; /* optional [Cout]: multiplies the fp32 accumulator of typed-output row co before `sub` is taken off */
  int flags;              /* SDA_WGRAD_FLAT_ROWS */
} sda_wgrad_args;
int sda_wgrad_gemm(const sda_wgrad_args* a, void* stream);
/* dst[i] = sum_s src[s][i] in fixed order */
int sda_reduce_slabs(const float* src, float* dst, int nslabs, long n, void* stream);
/* The loss's similarity matmul (loss.py:68) on 256 x 256 output tiles, 16-bit storage (csrc/sim_gemm.hip):
 *     partial[ks][i][j] = sum_{k in K slice ks} X[i][k] * W[j][k],   i < M, j < Np, ks < ksplit      (fp32)
 * X (M rows) and W (N rows) are K-contiguous rows `pitch` elements apart (K % 32 == 0, 16-byte aligned); columns [N, Np) are
 * don't-care.  The caller sums the K slices with sda_reduce_slabs (fixed order).  sda_sim_gemm_ksplit returns the number of
 * K slices to launch with for this shape on the current device, 0 if the shape is not served (fp32 storage: use sda_conv_gemm's
 * split-K matrix mode). */
int sda_sim_gemm_ksplit(int M, int N, long K, int dtype);
int sda_sim_gemm(const void* X, const void* W, float* partial, int M, int N, int Np, long K, long pitch, int ksplit, int dtype,
                 void* stream);

/* SpatialAttention weights (models.py:49-58 with SpatialDropout 81-84 folded in):
 * a = Re(z) cos + Im(z) sin; W = softmax_c(a); Wd = W * mask.  z is complex64 interleaved (re, im).
 * Outputs: W fp32 [D1][C] (saved for backward) and the packed operand Wp `dtype` [D1p][Cp]. */
int sda_sa_weights_forward(const float* z, const float* cos_t, const float* sin_t, const float* mask,
                           float* W, void* Wp, float* scratch /* sda_sa_scratch_floats(D1, K2, C) */, int D1, int K2,
                           int C, int D1p, int Cp, int dtype, void* stream);
int sda_sa_scratch_floats(int D1, int K2, int C);
/* The same weight build with its two contractions on the matrix cores (sda_conv_gemm in split-K matrix mode, fp32):
 * forward  a (D1 x C, pitch a_pitch) = [Re z | Im z] . [cos | sin]^T, then sda_sa_softmax_pack = softmax over sensors,
 *          dropout mask, W fp32 and the packed operand Wp (as sda_sa_weights_forward);
 * backward sda_sa_softmax_backward: da = W (dWd*mask - <dWd*mask, W>) (D1 x da_pitch, padding zeroed), then
 *          dz = da . [cos ; sin] by the same GEMM. */
int sda_sa_softmax_pack(const float* a, int a_pitch, const float* mask, float* W, void* Wp, int D1, int C, int D1p,
                        int Cp, int dtype, void* stream);
int sda_sa_softmax_backward(const float* dWd, int dwd_pitch, const float* W, const float* mask, float* da,
                            int da_pitch, int D1, int C, void* stream);
/* dWd fp32 [D1p][Cp] (from sda_wgrad_gemm) -> dz complex64 interleaved [D1][K2].
 * cosT/sinT are the transposed tables [C][K2] (constant buffers, transposed once by the host). */
int sda_sa_weights_backward(const float* dWd, const float* W, const float* mask, const float* cosT,
                            const float* sinT, float* dz, int D1, int K2, int C, int Cp, void* stream);

/* CLIP loss tail on the raw similarity S[i][j] = <Y_i, Z_j> (rows = speech, loss.py:60-79):
 * logits = S / (|Y_i| |Z_j|) * exp(temp); per-row (max, sumexp) over the local column block, per-column
 * lse over all rows, and diag[i] = logits[i][i - col0] for rows whose positive lives in this block.
 * Multi-GPU: rows are global (Bm), columns are this rank's Bn samples starting at global index col0;
 * zsq holds the Bn local norms. */
int sda_clip_logits_stats(const float* S, long s_pitch, const float* ysq, const float* zsq, const float* temp,
                          float* logits, float* row_max, float* row_sum, float* col_lse, float* diag /* every row written: 0 where
                          the positive lives elsewhere */, float* row_lse /* optional: max + log(sum) of the local block */,
                          int Bm, int Bn, int col0, void* stream);
/* Given the final row lse (after any cross-rank merge), with D_ij = p_row + p_col - 2 delta (dloss/dlogits = inv_norm * D):
 * G[i][j] = D_ij * ymax / |Y_i| (ymax = max_i |Y_i|: O(1) entries, safe in 16-bit) stored as `dtype` with pitch g_pitch —
 * [Bm + 1][g_pitch], the extra row and the columns >= Bn are written as zeros —, cscale[j] = inv_norm * exp(temp) /
 * (ymax |Z_j|) (the factor taken out of G: sda_wgrad_args.acc_scale of the dZ product), rscale[j] = inv_norm * sum_i D_ij
 * logits_ij / |Z_j|^2, so that dZ_j = cscale_j * sum_i G_ij Y_i - rscale_j Z_j; scalars[0] = this block's share of the loss,
 * scalars[1] = its share of dloss/dtemp.  inv_norm = 1/(2*B_global) for reduction="mean", 1/2 for "sum".  colpart: 2*Bn floats. */
int sda_clip_grad(const float* logits, const float* row_lse, const float* col_lse, const float* ysq,
                  const float* zsq, const float* temp, float inv_norm, int col0, void* G, long g_pitch,
                  float* rscale, float* cscale, float* colpart, float* scalars, int Bm, int Bn, int dtype, void* stream);
/* The embedding gradient of the loss on one GPU as a streaming kernel (loss_gemm.hip):
 *     out[j][k] = out_scale[0] * (cscale[j] * sum_{i < Bm} G[i][j] * Y[i][k] - rscale[j] * Z[j][k]),   j < Bn, k < row_elems
 * G [Bm][g_pitch], Y [Bm][row_elems], Z and out [Bn][row_elems] of `dtype` (16-bit types only), cscale / rscale fp32 [Bn]
 * (cscale may be NULL = 1), out_scale a device scalar or NULL.  Same result as sda_wgrad_gemm's typed-output mode, which
 * serves what this one does not: sda_clip_dz_supported() says whether a shape is served — bf16 / fp16 and either
 * Bm <= 256 with row_elems % 64 == 0 (the coefficient matrix in registers) or, ABI 4, Bm >= 256 with Bm % 32 == 0,
 * row_elems % 256 == 0 and g_pitch % 8 == 0 (256 x 256 tiles: any number of speech rows — a rank's block under data
 * parallelism contracts over the GLOBAL batch, loss.py:68 with x all-gathered). */
int sda_clip_dz_supported(int Bm, int Bn, long row_elems, int dtype);
int sda_clip_dz(const void* G, long g_pitch, const void* Y, const void* Z, void* out, const float* cscale, const float* rscale,
                const float* out_scale, int Bm, int Bn, long row_elems, int dtype, void* stream);
/* cnt[i] = #{local j : logits[i][j] beats diag[i]} (ties: lower global index wins) — Classifier ranks */
int sda_clip_ranks(const float* logits, const float* diag, int32_t* cnt, int Bm, int Bn, int col0, void* stream);
/* data parallelism: merge of the per-rank row statistics of the loss (all = the all-gathered [world][3][Bg] table of
 * (row max, row sum exp(l - max), positive's logit or 0) over each rank's block of brain columns): lse[i] = log-sum-exp of global
 * speech row i over the columns of ALL ranks, diag[i] = its positive's logit (utils/loss.py:79's two cross-entropies at the
 * global batch) */
int sda_clip_merge_rows(const float* all, int world, int Bg, float* lse, float* diag, void* stream);
int sda_device_count(void);
/* CU partitions (diagnostic / scheduling experiments): a HIP stream whose kernels run only on the CUs set in `mask` (bit i of
 * word i / 32 = CU i in the runtime's numbering; hipExtStreamCreateWithCUMask), and the CU count persistent grids launched
 * from THIS thread should size themselves for (0 = the device's; returns the previous limit).  The stream is created in
 * the calling process and destroyed with sda_stream_destroy. */
int sda_stream_create_cumask(const uint32_t* mask, int nwords, void** stream);
/* A non-blocking HIP stream of the given priority (hipStreamCreateWithPriority; the device's range on gfx950 is -1 = high,
 * 0 = normal, 1 = LOW — PyTorch's stream pool offers only the first two).  Out-of-range priorities are an error. */
int sda_stream_create_priority(int priority, void** stream);
int sda_stream_destroy(void* stream);
int sda_set_cu_limit(int cus);

/* Batched fp32 matrix product on parameter-sized operands with arbitrary element strides (exact-fp32 MFMA):
 *     C[b][i][j] = sum_{k < K} A[b][i][k] * B[b][k][j],    i < M, j < N, b < batch
 * with A[b][i][k] at A + b * a_b + i * a_i + k * a_k (B, C alike).  Nothing needs padding or alignment.  C is written as
 * c_dtype (SDA_F32 / SDA_BF16 / SDA_F16); elements outside [0, M) x [0, N) are not touched.  Replaces the parameter-space
 * products of the composed SubjectBlock (models.py:111-117: W_subj[s] . (W_sb . W_sa | b_sb)) and of its chain rule. */
typedef struct sda_pgemm_args {
  const float* A;
  const float* B;
  void* C;
  int M, N, K, batch;
  long a_i, a_k, a_b;
  long b_k, b_j, b_b;
  long c_i, c_j, c_b;
  int c_dtype;
} sda_pgemm_args;
int sda_param_gemm(const sda_pgemm_args* a, void* stream);
/* dst[i * d0 + j * d1 + k * d2] = src[i * s0 + j * s1 + k * s2] for (i, j, k) < (n0, n1, n2): fp32, element strides */
int sda_copy3d(float* dst, long d0, long d1, long d2, const float* src, long s0, long s1, long s2, int n0, int n1, int n2,
               void* stream);
/* Asynchronous upload of a small host table (nwords 32-bit words) carried in kernel arguments: no memcpy, no
 * host<->stream synchronisation. The host buffer is read before the call returns. */
int sda_upload_words(void* dst, const void* src_host, long nwords, void* stream);

/* Batch collate (gwilliams2022.py:651-661): per (sample, channel) row of T fp32 samples: subtract the mean of
 * the first baseline_len samples (preproc_utils.py:128-142), RobustScaler over time (median / inter-quartile
 * range, sklearn semantics; preproc_utils.py:69-90), clamp to +-clamp_lim when `clamp`.  rows = B*C, T <= 1024. */
int sda_collate_rows(const float* src, float* dst, long rows, int T, int baseline_len, float clamp_lim, int clamp,
                     void* stream);
/* Same, fused with the segment gather of gwilliams2022.py:129-142: sample b is the window
 * X_session[:, onset : onset + T] of a session recording resident in HBM — win_ptr[b] points at channel 0 of the
 * window, win_cstride[b] is that session's channel stride in elements (device arrays of length B). dst (B, C, T). */
int sda_collate_windows(const float* const* win_ptr, const long* win_cstride, float* dst, int B, int C, int T,
                        int baseline_len, float clamp_lim, int clamp, void* stream);

/* ---- Frozen wav2vec 2.0 speech embedder (utils/wav2vec_util.py:14-32 calls the third-party HF Wav2Vec2Model; this is that
 * model's published forward for the layer-norm / stable-layer-norm variant xlsr-53 uses).  Every Linear, strided Conv1d and
 * the grouped positional conv run on sda_conv_gemm (kernel size 1 on overlapping-row views); these are the other stages.
 * Activations are row-layout buffers of ONE chunk: row SDA_ROW_PAD + t = frame t. ---- */
/* Feature-encoder layer 0: y = GELU(LayerNorm_C(Conv1d(1 -> C, K, stride)(wave) + bias)); wave fp32 [n_samples] on the
 * device, w fp32 [C][K], T = (n_samples - K) / stride + 1 frames */
int sda_w2v_conv0(const float* wave, long n_samples, const float* w, const float* bias, const float* gamma,
                  const float* beta, void* y, int T, int C, int Cp, int K, int stride, float eps, int dtype, void* stream);
/* y = LayerNorm over the C valid channels of each of T rows (biased variance, eps inside the root), affine, then GELU
 * when `gelu`; pad channels of y are written as zero.  Cp * sizeof(element) <= 4096 */
int sda_layernorm_rows(const void* x, void* y, const float* gamma, const float* beta, int T, int C, int Cp, float eps,
                       int gelu, int dtype, void* stream);
/* Grouped positional conv, input side: channels [g*gw, (g+1)*gw) of frame t -> xg[g][lead + t][0..gw) (G buffers of
 * group_rows rows x gwp channels; rows outside [lead, lead + T) and channels >= gw must already be zero) */
int sda_w2v_group_split(const void* h, void* xg, int T, int Hp, int gw, int gwp, int G, long group_rows, int lead,
                        int dtype, void* stream);
/* ... output side: out[t][g*gw + c] = h[t][g*gw + c] + f(yg[g][SDA_ROW_PAD + t][c] + bias[g*gw + c]), f = GELU when `gelu`
 * (bias fp32 [G*gw] or NULL).  The G per-group GEMMs between the two are ONE sda_conv_gemm launch: the groups are its
 * "samples" (B = G, sample stride group_rows, widx = 0..G-1 selects the group's weights) */
int sda_w2v_group_merge_add(const void* h, const void* yg, void* out, int T, int Hp, int gw, int gwp, int G, long yg_rows,
                            const float* bias, int gelu, int dtype, void* stream);
/* out = softmax(q k^T * scale) v per head (no mask): q, k row layout with pitch qk_pitch, head h in columns
 * [64h, 64h+64); vt = V transposed, plain matrix [heads*64][vt_pitch >= T rounded up to 64] (columns >= T finite);
 * out row layout with pitch out_pitch.  head_dim must be 64 */
int sda_w2v_attention(const void* q, const void* k, const void* vt, void* out, int T, int heads, int head_dim,
                      long qk_pitch, long vt_pitch, long out_pitch, float scale, int dtype, void* stream);
/* Epilogue of a split-K sda_conv_gemm (partial = its raw fp32 slabs [ksplit][T][Cp], B = 1): y[SDA_ROW_PAD + t][c] =
 * f(sum_s partial[s][t][c] + bias[c]) + res[SDA_ROW_PAD + t][c]; bias / res may be NULL, f = GELU when `gelu` */
int sda_splitk_epilogue(const float* partial, int ksplit, const float* bias, const void* res, void* y, int T, int Cp,
                        int gelu, int dtype, void* stream);
/* out fp32 dense [T][C] = mean of four row-layout hidden states (wav2vec_util.py:18-20) */
int sda_w2v_mean4(const void* a, const void* b, const void* c, const void* d, float* out, int T, int C, int Cp,
                  int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SD_AMD_H */
