/* libvitae_hip.so — C ABI of the MI355X (gfx950) ViT-AE++ pre-training hot path.
 *
 * The reference (chinmay5/vit_ae_plus_plus) is pure Python/PyTorch and has no FFI of its own
 * (SURVEY §2.2, §8b): its hot path lowers to stock ATen ops.  This header therefore defines the
 * boundary a maintainer would bind instead of those ops — one launcher per fused op, each citing the
 * reference lines (relative to the reference repo root) whose arithmetic it replaces.
 *
 * Conventions
 *   - every tensor argument is a raw DEVICE pointer to contiguous fp32 (unless stated), sizes are
 *     plain ints/longs; no torch types cross this boundary;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - nothing allocates, frees or synchronises: every call only enqueues kernels (graph-capture
 *     safe); scratch space is passed in by the caller;
 *   - return value: 0 = enqueued; VITAE_ERR_INVALID_ARG (-1), VITAE_ERR_UNSUPPORTED_SHAPE (-2),
 *     VITAE_ERR_LAUNCH (-3).  Nothing throws across the boundary;
 *   - gradients of parameters ACCUMULATE into their destination (+=) where noted, so the caller
 *     zeroes the gradient arena once per optimisation step (reference: optimizer.zero_grad(),
 *     utils/train_one_epoch.py:35,73-74).
 */
#ifndef VITAE_HIP_H
#define VITAE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define VITAE_ABI_VERSION 48

/* matrix-core arithmetic of the dense contractions */
#define VITAE_PREC_F32 0  /* v_mfma_f32_32x32x2_f32: exact fp32 (the reference's precision, autocast off at utils/train_one_epoch.py:50) */
#define VITAE_PREC_BF16 1 /* v_mfma_f32_32x32x16_bf16: bf16 operands, fp32 accumulate */
#define VITAE_PREC_BF16X3 2 /* fp32 operands split into bf16 hi + lo while staged, hi.hi + hi.lo + lo.hi on the bf16 MFMA, fp32
                             * accumulate: products exact to 2^-16 — the fast mode that holds the reference's fp32 losses to 1e-4 */

/* GEMM epilogues */
#define VITAE_EPI_NONE 0
#define VITAE_EPI_GELU 1      /* aux <- pre-activation, out <- exact-erf GELU (nn.GELU, model/vit.py:85-92) */
#define VITAE_EPI_DGELU 2     /* out <- acc * GELU'(aux) */
#define VITAE_EPI_RELU_MASK 3 /* out <- aux > 0 ? acc : 0 */
#define VITAE_EPI_RELU 4      /* out <- max(acc + bias, 0) (vitae_gemm_glds only; no aux) */
#define VITAE_EPI_AUX_BF16 16 /* OR-ed into epi (vitae_gemm_glds*, vitae_linear_bwd_pair_glds*): aux points at bf16, not fp32 */
#define VITAE_EPI_AUX_DERIV 32 /* OR-ed into epi (every GEMM entry point): aux holds the DERIVATIVE GELU'(pre-activation) instead of the
                                * pre-activation — VITAE_EPI_GELU saves GELU'(acc + bias) (one more FMA on the gate it computes anyway),
                                * VITAE_EPI_DGELU multiplies by aux.  Same value as evaluating GELU' in the backward (fp32 aux: bit for
                                * bit), without 20 VALU operations per element in the fc2 input-gradient epilogue (ABI 41) */

/* device-resident hyper-parameter block `hp` (float[VITAE_HP_COUNT]) */
#define VITAE_HP_LR 0
#define VITAE_HP_BETA1 1
#define VITAE_HP_BETA2 2
#define VITAE_HP_EPS 3
#define VITAE_HP_BC1 4       /* 1 - beta1^t; <= 0: computed on the device from hp[VITAE_HP_STEP] (< 0: the slot holds -(1 - beta1)) */
#define VITAE_HP_BC2 5       /* 1 - beta2^t; same convention */
#define VITAE_HP_GRAD_MUL 6  /* multiplier applied to grads inside AdamW (1/loss_scale; 1 here) */
#define VITAE_HP_G_RECON 7   /* d total / d recon_loss  */
#define VITAE_HP_G_EDGE 8    /* d total / d raw_edge_mse (= edge_map_weight * upstream) */
#define VITAE_HP_G_CONTR 9   /* d total / d (mean-cosine term), includes contr_weight */
#define VITAE_HP_EDGE_W 10   /* edge_map_weight (model/vit_autoenc.py:225) */
#define VITAE_HP_CONTR_W 11  /* args.contr_weight (utils/train_one_epoch.py:114) */
#define VITAE_HP_HOST_COUNT 13 /* slots [0, 13) are the HOST's (uploaded every step); the rest of the block is device-owned state */
#define VITAE_HP_STEP 13     /* number of AdamW steps APPLIED so far (a float holding an integer): bumped on the device by the last
                              * AdamW launch of a step unless the gradient norm was not finite (GradScaler.step's skip).  With
                              * hp[BC1] == 0 the AdamW kernels derive the bias corrections 1 - beta^t from t = hp[STEP] + 1, so a
                              * skipped step never advances them and the host needs no late correction */
#define VITAE_HP_NOISE_KEEP 12 /* host slot: != 0 = the masking noise buffer was filled by the caller (injected noise): vitae_step_prologue leaves it alone */
#define VITAE_HP_COUNT 16

/* device-resident scalar accumulators `acc` (double[VITAE_ACC_COUNT]); caller zeroes them per step */
#define VITAE_ACC_RECON 0
#define VITAE_ACC_EDGE 1
#define VITAE_ACC_COS 2
#define VITAE_ACC_GRADSQ 3
#define VITAE_ACC_TICKET_A 4  /* first 4 bytes: workgroup arrival counter of vitae_opt_tail's norm pass (zero after the per-step zeroing) */
#define VITAE_ACC_TICKET_B 5  /* ... of its AdamW pass */
#define VITAE_ACC_TICKET_C 6  /* ... of vitae_cosine_loss_fwd (its last workgroup writes the scalar and leaves the counter at zero) */
#define VITAE_ACC_NONFINITE 7 /* the first 4 bytes of this slot are a FLOAT: 0 after the per-step zeroing, NaN once the loss
                                * backward produced a non-finite gradient (the early form of GradScaler.step's inf check:
                                * a `grad_norm` pointer for vitae_adamw_step before the global norm exists) */
/* Round 6: the squared gradient norm is accumulated in VITAE_ACC_SQ_SLOTS spread slots, acc[VITAE_ACC_SQ_BASE + s * VITAE_ACC_SQ_STRIDE]
 * (one 128-byte line each), workgroup b adding to slot b mod SLOTS: double atomics on ONE address retire one per ~10 ns, and a
 * weight-gradient launch of the batch-4 step has 400-600 workgroups (tools/pair_bench.py: 18.2 -> 14.1 us for the encoder's fc2 pair).
 * acc[VITAE_ACC_GRADSQ] stays a valid place to add to; every reader (vitae_grad_norm_finalize, vitae_opt_tail, vitae_grad_sqnorm's own
 * finalisation) takes acc[GRADSQ] + the sum of the slots. */
#define VITAE_ACC_SQ_BASE 16
#define VITAE_ACC_SQ_SLOTS 64
#define VITAE_ACC_SQ_STRIDE 16
#define VITAE_ACC_COUNT 1040

#define VITAE_MAX_TAPS 33

int vitae_abi_version(void);
const char* vitae_build_arch(void);
int vitae_memset_zero(void* ptr, long bytes, void* stream);

/* ---- dense contractions -------------------------------------------------------------------------
 * Generic C[M,N] (+)= epi(sum_k A(m,k) B(n,k) + bias[n]) (+ residual).  *_kcontig = 1: element
 * (row,k) at ptr[row*ld + k]; 0: at ptr[k*ld + row].  split_k > 1 needs
 * vitae_gemm_workspace_floats() floats of scratch. */
int vitae_gemm(int prec, int a_kcontig, int b_kcontig, const float* A, long lda, const float* B, long ldb,
               float* C, long ldc, int M, int N, int K, const float* bias, const float* residual, long ldr,
               int epi, float* aux, long ldaux, int accumulate, int split_k, float* splitk_ws, void* stream);
long vitae_gemm_workspace_floats(int M, int N, int K, int split_k);
int vitae_gemm_pick_split_k(int M, int N, int K);

/* Throughput-mode variant (bf16 MFMA, fp32 accumulate): A fp32, B fp32 or a bf16 shadow (b_is_bf16);
 * 64 x {64,128} tiles with deep k-phases, transpose reads for row-contiguous operands, XCD-aware order.
 * a_colsum_accum (optional, A k-contiguous): a_colsum_accum[k] += sum_m A(m,k) — the bias gradient of a
 * Linear computed while its dgrad streams dy. */
int vitae_gemm_bf16(int a_kcontig, int b_kcontig, const float* A, long lda, const void* B, long ldb, int b_is_bf16,
                    float* C, long ldc, int M, int N, int K, const float* bias, const float* residual, long ldr,
                    int epi, float* aux, long ldaux, int accumulate, int split_k, float* splitk_ws,
                    float* a_colsum_accum, void* stream);
int vitae_gemm_bf16_pick_split_k(int M, int N, int K);
/* Split-operand variant (what vitae_gemm and vitae_linear_{fwd,bwd_input,bwd_weight} run at prec = VITAE_PREC_BF16X3): the
 * contract of vitae_gemm, both operands fp32. */
int vitae_gemm_bf16x3(int a_kcontig, int b_kcontig, const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                      int M, int N, int K, const float* bias, const float* residual, long ldr, int epi, float* aux,
                      long ldaux, int accumulate, int split_k, float* splitk_ws, void* stream);
int vitae_gemm_bf16x3_pick_split_k(int M, int N, int K);
/* The same product on wave-specialised workgroups (csrc/gemm_bt.hip: gemm_wsx3_kernel — four MFMA waves, four producer waves that
 * split the fp32 operand tiles into bf16 hi + lo while they stage them): K any multiple of 4 (zero-filled tail), in-launch
 * split-K with the workspace contract of vitae_gemm_glds (vitae_gemm_glds_ws_floats(M, N, split_k) floats, the first
 * VITAE_GLDS_TICKETS words zero before the first use), exact-erf GELU epilogues, optional out_colsum_accum[n] += column sums of
 * the result and, in the weight-gradient form (a_kcontig = b_kcontig = 0), a_rowsum_accum[m] += sum_k A(m, k) = colsum(dy), the
 * bias gradient beside dW = dy^T x.  VITAE_ERR_UNSUPPORTED_SHAPE (misaligned operands, N % 4): keep vitae_gemm_bf16x3. */
int vitae_gemm_wsx3(int a_kcontig, int b_kcontig, const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                    int M, int N, int K, const float* bias, const float* residual, long ldr, int epi, float* aux, long ldaux,
                    int accumulate, int split_k, float* splitk_ws, float* out_colsum_accum, float* a_rowsum_accum, void* stream);
int vitae_gemm_wsx3_pick_split_k(int M, int N, int K);
/* Backward of one nn.Linear in a single launch (dgrad + wgrad + bias grad), bf16 MFMA:
 * dx[M,K] (+)= epi(dy[M,N] W[N,K]) (W from its bf16 shadow), dW[N,K] (+)= dy^T x, db[N] += colsum(dy). */
int vitae_linear_bwd_pair_bf16(const float* dy, const void* w_bf16, const float* x, float* dx, float* dw,
                               float* db_accum, int M, int N, int K, int epi, float* aux, int dx_accumulate,
                               int dw_accumulate, void* stream);
/* bf16 x bf16 variant with a 3-stage LDS-DMA (global_load_lds) pipeline; A16/B16 bf16, K % 64 == 0;
 * C (fp32) and/or C16 (bf16 copy of the result) may be given; out_colsum_accum[n] += sum_m result(m,n).
 * split_k > 1 reduces inside the launch (last workgroup of a tile sums the partials in split order and runs the
 * epilogue): splitk_ws must hold vitae_gemm_glds_ws_floats(M, N, split_k) floats whose first VITAE_GLDS_TICKETS
 * words are ZERO before the first use (the kernel leaves them zero again). */
#define VITAE_GLDS_TICKETS 4096
long vitae_gemm_glds_ws_floats(int M, int N, int split_k);
int vitae_gemm_glds(int a_kcontig, int b_kcontig, const void* A16, long lda, const void* B16, long ldb, float* C,
                    long ldc, void* C16, long ldc16, int M, int N, int K, const float* bias, const float* residual,
                    long ldr, int epi, float* aux, long ldaux, int accumulate, int split_k, float* splitk_ws,
                    float* out_colsum_accum, void* stream);
int vitae_gemm_glds_pick_split_k(int M, int N, int K);
/* ... for a given operand form (vitae_gemm_glds_pick_split_k plans the forward form; the weight-gradient form may want another split:
 * pass the split of the SAME form to vitae_gemm_glds, or the launch falls back to the 64-row tiles) */
int vitae_gemm_glds_pick_split_k_form(int a_kcontig, int b_kcontig, int M, int N, int K);
/* The forward form of vitae_gemm_glds (both operands k-contiguous) with a TWO-PLANE weight operand: y = x16 (W_hi + W_lo)^T, the
 * planes side by side in B16_hilo[N][2 K] (vitae_cast_bf16_hilo).  The weight's rounding enters at ~2^-17 instead of 2^-9, the
 * activations stay bf16: nn.Linear forward of the layers whose weight rounding carries the bf16 schedule's loss error
 * (model/vit.py:85-92 of the decoder blocks: tools/bf16_rounding_ablation.py).  Few rows: the wave-specialised 64 x 64 workgroup
 * stages both planes per k-tile and issues two MFMAs per k-slice; many rows: any big tile runs the reduction over 2 K with the A
 * operand wrapping.  Epilogue, split-K workspace (sized for (M, N, split_k)) and errors as vitae_gemm_glds; split_k from
 * vitae_gemm_glds_w2_pick_split_k. */
int vitae_gemm_glds_w2(const void* A16, long lda, const void* B16_hilo, float* C, long ldc, void* C16,
                       long ldc16, int M, int N, int K, const float* bias, const float* residual, long ldr, int epi, float* aux,
                       long ldaux, int accumulate, int split_k, float* splitk_ws, float* out_colsum_accum, void* stream);
int vitae_gemm_glds_w2_pick_split_k(int M, int N, int K);
/* Big-tile kernels (csrc/gemm_bt.hip; 0: 256x256 on 8 waves, 3: 128x128 on 4 waves, in-launch split-K) behind vitae_gemm_glds and
 * vitae_linear_bwd_pair_glds (whose halves then go out as two launches), plus the wave-specialised tiles (4: 128x128, 5: 64x64 —
 * four MFMA waves + four LDS-DMA producer waves per workgroup; with tile 5 both halves of vitae_linear_bwd_pair_glds stay ONE
 * launch).  mode -1 (default): picked per problem by the cost
 * model together with the split (vitae_gemm_glds_pick_split_k returns the split of the plan: pass it on unchanged); -2: never;
 * 0 / 3 / 4 / 5 / 6: that tile for every eligible problem (tests, tools; 6 = the wave-specialised 128 x 256 tile, weight-gradient form only).  vitae_gemm_glds_bt_choice = the tile a problem would get (-1 = none).
 * vitae_linear_bwd_pair_glds / vitae_wgrad_group_bt plan their own splits: they are told what the workspace holds per call
 * (splitk_ws_floats) and never make a plan that needs more. */
int vitae_gemm_glds_set_bt_tile(int mode);
int vitae_gemm_glds_bt_choice(int a_kcontig, int b_kcontig, int M, int N, int K);
/* profiling hook (tools/gemm_phase_probe.py): 8 long long per workgroup; NULL = off */
int vitae_gemm_glds_set_debug(void* buf);
/* Gradient norm without a pass over the gradients: while `slot` is set (NULL clears), every weight-gradient launch
 * (vitae_linear_bwd_pair_glds' wgrad half; vitae_gemm_glds called with a_kcontig = b_kcontig = 0) adds the sum of squares
 * of the tile it stores to *slot (double; the step's acc[VITAE_ACC_GRADSQ]).  Process-global, launch-time. */
int vitae_gemm_glds_set_wgrad_sqnorm(double* slot);
/* The same over n_slots (a power of two) addresses `stride` doubles apart: workgroup b adds to slots[(b mod n_slots) * stride].
 * The step passes acc + VITAE_ACC_SQ_BASE, VITAE_ACC_SQ_SLOTS, VITAE_ACC_SQ_STRIDE. */
int vitae_gemm_glds_set_wgrad_sqnorm_spread(double* slots, int n_slots, int stride);
/* norm_out[0] = sqrt(acc[VITAE_ACC_GRADSQ] + the VITAE_ACC_SQ_SLOTS spread slots) (the finalisation vitae_grad_sqnorm appends, on its own) */
int vitae_grad_norm_finalize(const double* acc, float* norm_out, void* stream);
/* Backward of one nn.Linear on bf16 operands in one launch: dx / dx16 [M,K] = epi(dy16 W16), optional
 * dx_colsum_accum[k] += sum_m dx(m,k); dW[N,K] (+)= dy16^T x16 reduced over Mpad (>= M, multiple of 64) token
 * rows — rows M..Mpad-1 of dy16 and x16 must be zero. */
int vitae_linear_bwd_pair_glds(const void* dy16, const void* w16, const void* x16, float* dx, void* dx16, float* dw,
                               void* dw16 /* optional bf16 copy of the (accumulated) dW: the wire buffer of a bf16
                               data-parallel gradient exchange */, int M, int Mpad, int N, int K, int epi, float* aux,
                               float* dx_colsum_accum, float* dy_colsum_accum /* optional: [N] += colsum(dy16) = the bias
                               gradient of this Linear (one extra MFMA against ones in the wgrad workgroups when both halves use 64x64
                               tiles, a separate bf16 column-sum launch otherwise) */,
                               int dx_accumulate /* dx += instead of = (fp32 dx only) */, int dw_accumulate, int split_k,
                               float* splitk_ws, long splitk_ws_floats /* floats splitk_ws holds (first VITAE_GLDS_TICKETS zero):
                               the planner of either half never exceeds it; >= vitae_gemm_glds_ws_floats(M, K, split_k) */,
                               void* stream);
/* (dw == NULL: the input gradient only — the weight gradient is deferred to vitae_wgrad_group_bt.)
 * Weight gradients of up to four Linears with the same token count in ONE launch of 128x128 tiles (ping-pong workgroups, two
 * per CU, or wave-specialised ones, one per CU: the cheaper by the planner's clocks; a forced tile 3 / 4 forces the kind): for i < n,
 * dw[i][N[i], K[i]] (+)= dy16[i][Mpad, N[i]]^T x16[i][Mpad, K[i]] (rows M..Mpad-1 of every operand zero), optional bf16 copies
 * dw16[i] (NULL array or entries), optional dy_colsum[i][N[i]] += column sums of dy16[i] (bias gradients).  The pointer arrays
 * and N / K live on the HOST.  Together the problems of a transformer block have enough tiles that the reduction needs no
 * split (or two) where each alone wanted 3-8; splitk_ws as for vitae_gemm_glds, holding splitk_ws_floats floats (the split is
 * shrunk to fit). */
int vitae_wgrad_group_bt(int n, const void* const* dy16, const void* const* x16, float* const* dw, void* const* dw16,
                         float* const* dy_colsum, const int* N, const int* K, int M, int Mpad, int dw_accumulate,
                         float* splitk_ws, long splitk_ws_floats, void* stream);
/* split of the dgrad reduction for the call above (1 = none); workspace as for vitae_gemm_glds with (M, K) */
int vitae_linear_bwd_pair_pick_split_k(int M, int Mpad, int N, int K);
/* dst_bf16[i] = bf16(src[i]) (round to nearest even) */
int vitae_cast_bf16(const float* src, void* dst_bf16, long n, void* stream);
/* hi | lo planes of `count` equally spaced fp32 [rows, K] tensors (tensor t starts at src + t * stride), side by side in
 * out_bf16[count][rows][2 K]: columns [0, K) = bf16(x) (the shadow), [K, 2K) = bf16(x - float(bf16(x))) — what the shadow drops;
 * together they carry a weight to ~2^-17.  The B operand of vitae_gemm_glds_w2. */
int vitae_cast_bf16_hilo(const float* src, void* out_bf16, long rows, int K, long stride, int count, void* stream);

/* nn.Linear forward  y = x W^T + b  (model/vit.py:85-96 fc1/fc2, :107-114 qkv, :109,122 proj;
 * model/vit_autoenc.py:41 decoder_embed, :53 decoder_pred, :263-268 predictor; and the Conv3d patch
 * embedding of model/vit.py:65,72 as a GEMM over gathered patches).  x[M,K], w[N,K], y[M,N]. */
int vitae_linear_fwd(int prec, const float* x, const float* w, const float* bias, float* y, int M, int N, int K,
                     int epi, float* aux, const float* residual, int split_k, float* ws, void* stream);
/* dx[M,K] (+)= epi(dy[M,N] W[N,K]) */
int vitae_linear_bwd_input(int prec, const float* dy, const float* w, float* dx, int M, int N, int K, int epi,
                           float* aux, int accumulate, int split_k, float* ws, void* stream);
/* dW[N,K] (+)= dy[M,N]^T x[M,K] */
int vitae_linear_bwd_weight(int prec, const float* dy, const float* x, float* dw, int M, int N, int K,
                            int accumulate, int split_k, float* ws, void* stream);
/* out[n] += sum_m dy[m*ld + n]  (bias gradients) */
int vitae_colsum_accum(const float* dy, long ld, float* out, int M, int N, void* stream);

/* ---- LayerNorm (partial(nn.LayerNorm, eps=1e-6): model/vit_autoenc.py:292,300,308; used at
 * model/vit.py:132,135,142-143, model/vit_autoenc.py:36,51,175,195) ------------------------------ */
/* y (fp32) and/or y_bf16 (bf16 copy for the next GEMM's operand) */
int vitae_layernorm_fwd(const float* x, const float* w, const float* b, float* y, void* y_bf16, float* mean,
                        float* rstd, int M, int D, float eps, void* stream);
/* dw, db accumulate (+=); dx overwritten, or += when dx_accumulate; optional dx_bf16 = bf16(final dx) and
 * dx_colsum_accum[c] += sum_m final dx(m,c) (the bias gradient of the Linear whose output gradient dx is) */
int vitae_layernorm_bwd(const float* dy, const float* x, const float* w, const float* mean, const float* rstd,
                        float* dx, float* dw, float* db, void* dx_bf16, float* dx_colsum_accum, int M, int D,
                        int dx_accumulate, void* stream);
/* The same without atomics (round 4): a launch of vitae_layernorm_bwd_part_records(M) workgroups, each leaving its column partials
 * [d gamma | d beta | colsum(dx)] as one 3 D-float record in part[records][3][D]; vitae_ln_grad_reduce adds the records of n
 * LayerNorm instances (HOST arrays of pointers / counts; dx_colsum entries may be NULL) into dw / db / dx_colsum in ONE launch,
 * in a fixed order (bitwise reproducible).  D in {256, 512, 768, 1024}, 16-byte aligned operands. */
int vitae_layernorm_bwd_part_records(int M);
int vitae_layernorm_bwd_part(const float* dy, const float* x, const float* w, const float* mean, const float* rstd,
                             float* dx, float* part, void* dx_bf16, int M, int D, int dx_accumulate, void* stream);
int vitae_ln_grad_reduce(int n, const float* const* part, float* const* dw, float* const* db, float* const* dx_colsum,
                         const int* records, const int* D, void* stream);

/* ---- attention core  softmax(q k^T / sqrt(hd)) v  (model/vit.py:117-121) -------------------------
 * qkv [B,N,3,H,hd] (output of the qkv Linear, model/vit.py:114), o [B,N,H*hd], lse/delta [B,H,N]. */
int vitae_sdpa_fwd(const float* qkv, float* o, float* lse, int B, int N, int H, int head_dim, void* stream);
int vitae_sdpa_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv,
                   float* delta_ws, int B, int N, int H, int head_dim, void* stream);

/* bf16-MFMA implementation of the same two entry points (head_dim 32 or 64; operands rounded to bf16,
 * fp32 softmax / accumulation); VITAE_ERR_UNSUPPORTED_SHAPE for other head dims. */
int vitae_sdpa_mfma_fwd(const float* qkv, float* o, void* o_bf16, float* lse, int B, int N, int H, int head_dim,
                        void* stream);
int vitae_sdpa_mfma_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* dqkv /* may be NULL when
                        dqkv_bf16 is given */,
                        void* dqkv_bf16, float* dqkv_colsum_accum /* [3*H*hd] += column sums of dqkv, or NULL */,
                        float* delta_ws, int B, int N, int H, int head_dim, void* stream);
/* The same two ops with q | k | v read from the bf16 copy the qkv GEMM writes ([B*N, 3*H*hd] bf16: no fp32 qkv in HBM):
 * operands are what the fp32-input kernels round to while staging, except that q takes the softmax scale after its
 * rounding.  vitae_sdpa_bwd_fused_fits(N, hd): 1 when a head fits LDS and the backward is ONE launch; otherwise two streaming
 * kernels (dQ + delta, then dK / dV) run and delta_ws [B * H * N] floats is required (may be NULL when it fits). */
int vitae_sdpa_mfma_fwd_bf16in(const void* qkv_bf16, float* o, void* o_bf16, float* lse, int B, int N, int H, int head_dim,
                               void* stream);
int vitae_sdpa_bwd_fused_fits(int N, int head_dim);
int vitae_sdpa_mfma_bwd_bf16in(const void* qkv_bf16, const float* o, const float* d_o, const float* lse, float* dqkv,
                               void* dqkv_bf16, float* dqkv_colsum_accum, float* delta_ws, int B, int N, int H, int head_dim,
                               void* stream);

/* ---- masking and sequence assembly ---------------------------------------------------------------
 * random_masking (model/vit_autoenc.py:141-153) from a caller-supplied noise[B,L] (the torch.rand of
 * :139): ids_shuffle/ids_restore int32 [B,L], mask fp32 [B,L] (0 keep, 1 remove); ids_restore_i64
 * optional (the int64 tensor the reference returns). */
int vitae_random_masking(const float* noise, int* ids_shuffle, int* ids_restore, float* mask,
                         long long* ids_restore_i64, int B, int L, int len_keep, void* stream);
/* rows of the patch-embedding GEMM for the kept patches only: out[B*keep, C*p^3] in Conv3d weight
 * order (model/vit.py:65,72-74 + model/vit_autoenc.py:147-148) */
int vitae_gather_patches(const float* vol, const int* ids_shuffle, float* out, void* out_bf16, int B, int C, int Lz,
                         int Hy, int Wx, int p, int keep, void* stream);
/* both views of the contrastive model (model/vit_autoenc.py:272,277) in one launch: ids_shuffle [2B, L], out [2B*keep, C*p^3] */
int vitae_gather_patches_2views(const float* vol1, const float* vol2, const int* ids_shuffle, float* out, void* out_bf16, int B,
                                int C, int Lz, int Hy, int Wx, int p, int keep, void* stream);
/* x[B,keep+1,D]: + pos_embed, cls token (model/vit_autoenc.py:162-170) */
int vitae_encoder_assemble_fwd(const float* tok, const float* cls_token, const float* pos_embed,
                               const int* ids_shuffle, float* x, int B, int L, int keep, int D, void* stream);
int vitae_encoder_assemble_bwd(const float* dx, float* dtok, void* dtok_bf16, float* dcls_accum, int B, int keep, int D,
                               void* stream);
/* out[B,D] = mean_{n >= first} x[B,N,D]: global average pool over the patch tokens of the encoder-only model
 * (model/vit.py:277-278, first = 1 skips the cls token) */
int vitae_mean_pool_tokens(const float* x, float* out, int B, int N, int D, int first, void* stream);
/* xd[B,L+1,Dd]: mask-token fill + unshuffle + decoder_pos_embed (model/vit_autoenc.py:184-190) */
int vitae_decoder_assemble_fwd(const float* e, const float* mask_token, const float* dpos, const int* ids_restore,
                               float* xd, int B, int L, int keep, int Dd, void* stream);
/* de[B,keep+1,Dd] = kept rows of dxd (optional bf16 copy de_bf16: the dy operand of decoder_embed's backward on the
 * LDS-DMA GEMM); dmask_token_accum[Dd] += sum of dxd over the masked positions.  One launch. */
int vitae_decoder_assemble_bwd(const float* dxd, const int* ids_shuffle, float* de, void* de_bf16, float* dmask_token_accum,
                               int B, int L, int keep, int Dd, void* stream);

/* ---- perceptual-loss hook, forward only (model/model_utils/perceptual_loss.py:46-77; a no-gradient logging term,
 * model/vit_autoenc.py:229-230).  VGG16's convolutions run as vitae_gemm_glds (bias + VITAE_EPI_RELU) over im2col matrices in
 * NHWC bf16; these are the index kernels around them.
 * im2col_first: every (batch, z) slice of ONE channel of the predicted and of the target volume [B, C, Z, H, W] -> rows
 *   [2 * n_img * H * W, 64] holding the 3x3 neighbourhood (9 taps, zero padded to K = 64), images img0 .. img0 + n_img - 1 of
 *   the prediction first, then the same images of the target;
 * im2col: NHWC bf16 [n_img, H, W, Cin] -> [n_img * H * W, 9 * Cin], tap-major then channel;
 * maxpool2: 2x2 stride 2; sqdiff: acc[0] += sum (a - b)^2 (double). */
int vitae_percep_im2col_first(const float* vol_pred, const float* vol_target, void* A16, int B, int C, int channel, int Z, int H,
                              int W, int img0, int n_img, void* stream);
int vitae_percep_im2col(const void* in16, void* A16, long n_img, int H, int W, int Cin, void* stream);
int vitae_percep_maxpool2(const void* in16, void* out16, long n_img, int H, int W, int C, void* stream);
int vitae_percep_sqdiff(const void* a16, const void* b16, long n, double* acc, void* stream);

/* ---- data-parallel gradient exchange (the reference itself only all-reduces logging scalars,
 * utils/misc.py:332-340) -------------------------------------------------------------------------------------------------
 * RCCL all-reduce (SUM) of one gradient bucket on a side stream, forked from / joined to the compute stream with events
 * created at init — capturable inside a HIP graph, so a data-parallel optimisation step is ONE graph replay.  RCCL is bound
 * at run time (the librccl.so already in the process, e.g. PyTorch's).  Rendezvous belongs to the caller: rank 0 obtains the
 * 128-byte id, ships it to the other ranks (torch.distributed broadcast, file, MPI ...), every rank calls vitae_ddp_init.
 * One communicator per process.  The caller folds 1/world_size into the loss-gradient multipliers (VITAE_HP_G_*), so the
 * SUM is the mean. */
int vitae_ddp_available(void);                       /* 1 when RCCL could be bound */
int vitae_ddp_unique_id(void* out128);               /* HOST buffer of 128 bytes */
int vitae_ddp_init(const void* unique_id128, int world_size, int rank);
int vitae_ddp_world_size(void);                      /* 0 before init */
/* buf[0..count) (fp32, or bf16 when is_bf16) <- sum over ranks, in place, on comm_stream, ordered after everything enqueued
 * on compute_stream so far */
int vitae_ddp_allreduce_bucket(void* buf, long count, int is_bf16, void* compute_stream, void* comm_stream);
/* compute_stream waits for every bucket launched on comm_stream so far */
int vitae_ddp_wait(void* compute_stream, void* comm_stream);
int vitae_ddp_destroy(void);

/* ---- input normalisation (dataset/brats_dataset/brats.py:26-37, dataset/egd_dataset/egd.py:44-55) ---------------
 * `groups` contiguous runs of n elements, each normalised on its own: z-score with the unbiased variance, min-max
 * to [-1, 1], or min-max to [0, 1].  ws: 3 doubles per group (scratch).  In place allowed (y == x). */
#define VITAE_NORM_ZSCORE 0
#define VITAE_NORM_MINMAX_PM1 1
#define VITAE_NORM_MINMAX_01 2
int vitae_normalize_volumes(const float* x, float* y, double* ws, int groups, long n, int mode, void* stream);
/* ---- augmentations of the pre-training scripts as batch ops (k_fold_training_scripts/
 * k_fold_cross_valid_combined_brats.py:93-97: tio.RandomAffine(), tio.RandomNoise(std=0.1), tio.RandomGamma(log_gamma=
 * (-0.3, 0.3)) applied to the raw item in dataset/brats_dataset/brats.py:39-44).  torchio 0.18.73 / SimpleITK 2.2.1 are
 * not in /root/reference: their published behaviour is restated (oracle/augment_ref.py, parity unpinned); random
 * parameters are drawn on the host (utils/augment.py), the kernels are deterministic.
 * vitae_volume_minmax: ws[3*g+2] <- {min, max} of each group of n elements (the 'minimum' pad value of RandomAffine).
 * vitae_affine_resample: y[b,c,o] = linear interpolation of x[b,c,:] at the continuous index mats[b] (3x4, row-major,
 *   axis order l,h,w) applied to (o,1); points outside [-0.5, n-0.5) get the pad value (the group minimum from minmax_ws
 *   when given, else pad_value).  Not in place.
 * vitae_noise_gamma: y = sign(v)|v|^gammas[b], v = x + stds[b]*noise  (noise / gammas may be NULL). */
int vitae_volume_minmax(const float* x, double* ws, int groups, long n, void* stream);
int vitae_affine_resample(const float* x, float* y, const float* mats, const double* minmax_ws, float pad_value, int B, int C,
                          int Lz, int Hy, int Wx, void* stream);
int vitae_noise_gamma(const float* x, const float* noise, float* y, const float* stds, const float* gammas, int B, long n,
                      void* stream);

/* ---- loss chain ----------------------------------------------------------------------------------
 * pred element (b,l,e) lives at pred[b*pred_bstride + l*P + e] (P = p^3*C), so the decoder output
 * with its cls row is read in place (model/vit_autoenc.py:198-201). */
int vitae_recon_loss_fwd(const float* pred, long pred_bstride, const float* imgs, const float* mask, double* acc,
                         int B, int C, int Lz, int Hy, int Wx, int p, void* stream);   /* :205-227 */
int vitae_recon_loss_bwd(const float* pred, long pred_bstride, const float* imgs, const float* mask, const float* hp,
                         float* dpred, float mask_sum, int B, int C, int Lz, int Hy, int Wx, int p, void* stream);
int vitae_unpatchify(const float* pred, long pred_bstride, float* vol, int B, int C, int Lz, int Hy, int Wx, int p,
                     void* stream);                                                     /* :115-128 */
/* separable 3-pass blur == dense k(x)k(x)k conv3d of model/model_utils/gaussian_filter.py:16-26;
 * taps_host is a HOST array (copied into the launch), ntaps odd */
int vitae_gauss_blur_fwd(const float* vol, float* tmp, float* out, const float* taps_host, int ntaps, int BC, int Lz,
                         int Hy, int Wx, void* stream);
/* edge[B,Lz,Hy,Wx] = sum_c |sobel(vol[:,c])| (model/model_utils/sobel_filter.py:37-45); with edge_ref
 * also acc[VITAE_ACC_EDGE] += sum (edge-edge_ref)^2 (model/vit_autoenc.py:224) */
int vitae_sobel_edge_fwd(const float* vol, float* edge, const float* edge_ref, double* acc, int B, int C, int Lz,
                         int Hy, int Wx, void* stream);
/* dpred += d(edge mse)/d pred ; dG_ws = B*C*3*Lz*Hy*Wx floats of scratch */
int vitae_sobel_edge_bwd(const float* pred_vol, const float* edge_pred, const float* edge_tgt, const float* hp,
                         float* dG_ws, float* dpred, void* dpred_bf16, long pred_bstride, int B, int C, int Lz, int Hy,
                         int Wx, int p, void* stream);
/* forward loss terms on the prediction in one pass (model/vit_autoenc.py:221-227): pred_vol = unpatchify(pred),
 * edge_pred = Sobel magnitude of it, acc[VITAE_ACC_RECON] += masked MSE sum / P, acc[VITAE_ACC_EDGE] += sum
 * (edge_pred - edge_tgt)^2.  C = 4: one LDS-tiled kernel; other C: recon_loss_fwd + unpatchify + sobel_edge_fwd. */
int vitae_loss_fwd_fused(const float* pred, long pred_bstride, const float* imgs, const float* mask, const float* edge_tgt,
                         float* pred_vol, float* edge_pred, double* acc, int B, int C, int Lz, int Hy, int Wx, int p,
                         void* stream);
/* whole loss backward in one pass (model/vit_autoenc.py:224-232 differentiated):
 * dpred = mask*2*g_recon*(pred-target)/(P*mask.sum()) + d(edge mse)/d pred, fp32 and optionally bf16.
 * C in {1,4}: one LDS-tiled kernel; other C: recon_bwd + the two Sobel backward kernels (needs dG_ws). */
int vitae_loss_bwd_fused(const float* pred, const float* pred_vol, const float* imgs, const float* mask,
                         const float* edge_pred, const float* edge_tgt, const float* hp, float* dG_ws, float* dpred,
                         void* dpred_bf16, float* nonfinite_flag /* optional: set to NaN when a non-finite gradient is written
                         (C in {1,4} only) */, long pred_bstride, float mask_sum, int B, int C, int Lz, int Hy, int Wx, int p,
                         void* stream);
/* The target's edge map in one launch (csrc/loss_fused.hip; C = 4, 11 taps): edge_tgt = sum_c |Sobel(gauss(imgs[:, c]))| =
 * vitae_gauss_blur_fwd followed by vitae_sobel_edge_fwd without the blurred intermediates (model/vit_autoenc.py:221-223).
 * vitae_target_edge_supported: 1 when the geometry is served, else VITAE_ERR_UNSUPPORTED_SHAPE. */
int vitae_target_edge_supported(int C, int ntaps, int Lz, int Hy, int Wx);
int vitae_target_edge(const float* imgs, float* edge_tgt, const float* taps_host, int ntaps, int B, int C, int Lz, int Hy, int Wx,
                      void* stream);
/* Both of the above in ONE pass over the prediction (csrc/loss_fused.hip; C = 4): acc[VITAE_ACC_RECON] / acc[VITAE_ACC_EDGE] as
 * vitae_loss_fwd_fused, dpred / dpred_bf16 / nonfinite_flag as vitae_loss_bwd_fused; the unpatchified prediction and its edge map
 * are not produced.  Both MSE terms are linear in their upstream gradient (hp[VITAE_HP_G_RECON], hp[VITAE_HP_G_EDGE]), so the
 * gradient needs no result of the forward.  dpred may be NULL when dpred_bf16 is given (the bf16 copy only).  vitae_loss_fwd_bwd_supported: 1 when the geometry is served (else
 * VITAE_ERR_UNSUPPORTED_SHAPE and the caller keeps the two calls above). */
int vitae_loss_fwd_bwd_supported(int C, int Lz, int Hy, int Wx, int p);
int vitae_loss_fwd_bwd(const float* pred, long pred_bstride, const float* imgs, const float* mask, const float* edge_tgt,
                       const float* hp, float* dpred, void* dpred_bf16, float* nonfinite_flag, double* acc, float mask_sum,
                       int B, int C, int Lz, int Hy, int Wx, int p, void* stream);
/* out4 = [loss, raw_edge_mse, recon, percep=0] (model/vit_autoenc.py:231-232) */
int vitae_loss_finalize(const double* acc, const float* hp, float* out4, float mask_sum, long edge_count, void* stream);

/* ---- contrastive head ----------------------------------------------------------------------------
 * BatchNorm1d (training) + ReLU of the predictor (model/vit_autoenc.py:263-268); running stats updated
 * in place (momentum 0.1, unbiased variance). */
int vitae_bn1d_relu_fwd(const float* x, const float* w, const float* b, float* y, void* y_bf16 /* optional bf16 copy of y: the next
                        Linear's GEMM operand */, float* save_mean, float* save_rstd,
                        float* running_mean, float* running_var, long long* num_batches_tracked, int R, int D,
                        float eps, float momentum, void* stream);
/* the same in eval mode (model.eval()): y = relu((x - running_mean) / sqrt(running_var + eps) * w + b) */
int vitae_bn1d_relu_eval(const float* x, const float* w, const float* b, const float* running_mean,
                         const float* running_var, float* y, int R, int D, float eps, void* stream);
int vitae_bn1d_relu_bwd(const float* dy, const float* x, const float* y, const float* w, const float* save_mean,
                        const float* save_rstd, float* dx, void* dx_bf16 /* optional bf16 copy of dx */, float* dw_accum, float* db_accum,
                        int R, int D, void* stream);
/* (D % 4 == 0 and 16-byte aligned operands: the kernels move float4 column groups) */
/* The same two ops with the ROWS split over workgroups, two launches each (partial statistics / sums, then the normalisation):
 * for many rows — the forms above give one 64-column strip all R rows, i.e. D / 64 workgroups whatever R is.  Same results to
 * fp32 round-off (the statistics are merged with Chan's update, not from E[x^2] - E[x]^2); deterministic.  ws: scratch of
 * vitae_bn1d_split_ws_floats(R, D) floats (16-byte aligned), used between the two launches of one call only. */
long vitae_bn1d_split_ws_floats(int R, int D);
int vitae_bn1d_relu_fwd_split(const float* x, const float* w, const float* b, float* y, void* y_bf16, float* save_mean, float* save_rstd,
                              float* running_mean, float* running_var, long long* num_batches_tracked, int R, int D,
                              float eps, float momentum, float* ws, void* stream);
int vitae_bn1d_relu_bwd_split(const float* dy, const float* x, const float* y, const float* w, const float* save_mean,
                              const float* save_rstd, float* dx, void* dx_bf16, float* dw_accum, float* db_accum, int R, int D,
                              float* ws, void* stream);
/* contr = contr_w * (-(mean cos(p1,z2) + mean cos(p2,z1))/2)  (utils/train_one_epoch.py:113-114) */
int vitae_cosine_loss_fwd(const float* p1, const float* z2, const float* p2, const float* z1, double* acc,
                          const float* hp, float* out1, int R, int D, void* stream);
int vitae_cosine_loss_bwd(const float* p1, const float* z2, const float* p2, const float* z1, const float* hp,
                          float* dp1, float* dp2, int R, int D, void* stream);
/* the same with bf16 copies of dp1 / dp2 (the predictor's bf16 GEMM operands: no cast launch in between); dp1 / dp2 may both be NULL */
int vitae_cosine_loss_bwd_bf16(const float* p1, const float* z2, const float* p2, const float* z1, const float* hp,
                               float* dp1, float* dp2, void* dp1_bf16, void* dp2_bf16, int R, int D, void* stream);

/* ---- optimiser -----------------------------------------------------------------------------------
 * utils/misc.py:265-266,280-292 (global grad L2 norm) and torch.optim.AdamW
 * (k_fold_training_scripts/k_fold_cross_valid_combined_brats.py:168-169). */
int vitae_grad_sqnorm(const float* grads, long n, double* acc, float* norm_out, void* stream);
/* shadow_bf16 (optional): bf16 copy of the updated parameters, written in the same pass */
int vitae_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, long n,
                     const float* hp, const float* grad_norm, float weight_decay, void* stream);
/* the same two passes reading the gradients as bf16 — the wire copy of a bf16 data-parallel all-reduce, consumed
 * in place instead of being converted back into the fp32 arena first */
int vitae_grad_sqnorm_bf16(const void* grads_bf16, long n, double* acc, float* norm_out, void* stream);
int vitae_adamw_step_bf16g(float* params, const void* grads_bf16, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                           long n, const float* hp, const float* grad_norm, float weight_decay, void* stream);
/* Round 6: the same pass with BOTH moments stored in bf16 (22 instead of 30 bytes of HBM traffic per parameter).  The step is computed
 * in fp32 from the stored values and m_new / v_new enter the parameter update unrounded; only what is written back is rounded
 * (tools/opt_state_ablation.py: the reference's pinned ViT-B trajectory moves by 2e-7..2e-6).  The caller must keep 1 - beta >= 2^-6
 * for both betas (a smaller update would stall under round-to-nearest); grads: fp32, or the bf16 wire copy when grads_bf16 != 0.
 * vitae_opt_tail takes the same storage through its state_bf16 flag. */
int vitae_adamw_step_s16(float* params, const void* grads, int grads_bf16, void* exp_avg_bf16, void* exp_avg_sq_bf16,
                         void* shadow_bf16, long n, const float* hp, const float* grad_norm, float weight_decay, void* stream);
/* ... gated by the accumulator block itself instead of a finalised norm: the pass is skipped when acc[VITAE_ACC_GRADSQ] + the
 * VITAE_ACC_SQ_* slots (the squares collected so far) are not finite — the verdict vitae_grad_norm_finalize + grad_norm would give,
 * without that launch in front of every gradient bucket (round 6: a node of a replayed step costs 4-5 us whatever it does). */
int vitae_adamw_step_s16_acc(float* params, const void* grads, int grads_bf16, void* exp_avg_bf16, void* exp_avg_sq_bf16,
                             void* shadow_bf16, long n, const float* hp, const double* acc, float weight_decay, void* stream);
/* hp[VITAE_HP_STEP] += 1 unless grad_norm[0] is not finite: the end of an optimiser step issued as separate vitae_adamw_step calls */
int vitae_opt_count_bump(float* hp, const float* grad_norm, void* stream);
/* The tail of one optimisation step in TWO launches (was: norm pass, finalisation, two AdamW launches): the last n_decay +
 * n_plain elements of the arena — tokens (weight decay) then vectors (none), whose gradients are only final when the whole
 * backward is — (1) add their squares to acc[VITAE_ACC_GRADSQ]; the last workgroup to arrive (acc[VITAE_ACC_TICKET_A]) writes
 * norm_out[0] = sqrt(acc[GRADSQ]): the global gradient norm of utils/misc.py:265-266; (2) AdamW over both segments, skipped
 * when that norm is not finite; the last workgroup (acc[VITAE_ACC_TICKET_B]) bumps hp[VITAE_HP_STEP].  grads: fp32, or bf16 when
 * grads_bf16 != 0 (the wire copy of a bf16 gradient exchange).  All pointers at the first of the n_decay elements. */
int vitae_opt_tail(float* params, const void* grads, int grads_bf16, void* exp_avg, void* exp_avg_sq, int state_bf16, void* shadow_bf16,
                   long n_decay, long n_plain, float* hp, double* acc, float* norm_out, float weight_decay, void* stream);
/* The head of one optimisation step as ONE launch inside the captured step (was, between two graph replays: torch's uniform_,
 * a host-to-device copy of hp, and two zeroing launches): hp[0 .. VITAE_HP_HOST_COUNT) <- hp_ring[(*step_seq) % ring_slots] (a
 * pinned host ring the caller filled for this step), noise[0 .. n_noise) <- U[0, 1) from Philox4x32-10 keyed by (seed, *step_seq)
 * unless the ring slot says VITAE_HP_NOISE_KEEP, acc[0 .. VITAE_ACC_COUNT) <- 0, zero_ptr[0 .. zero_bytes) <- 0 (the token / vector
 * gradient segment; may be NULL).  *step_seq is advanced by vitae_step_epilogue (the last launch of the step), never here: every
 * workgroup of this launch reads it. */
int vitae_step_prologue(float* hp, const float* hp_ring, int ring_slots, const long long* step_seq, float* noise, long n_noise,
                        long long seed, double* acc, void* zero_ptr, long zero_bytes, void* stream);
int vitae_step_epilogue(long long* step_seq, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VITAE_HIP_H */
