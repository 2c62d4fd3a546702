/* diffusion_amd.h - C ABI of libdiffusion_amd.so: the MI355X (gfx950) kernels behind the Stable Diffusion 2
 * U-Net training step.
 *
 * Boundary contract.  The reference (fanzhongyi/diffusion) is pure Python and has no
 * FFI of its own: its hot path enters native code through torch / diffusers / xformers calls made from
 *   diffusion/models/stable_diffusion.py:177-183   (timestep draw, add_noise, unet(...)['sample'])
 *   diffusion/models/stable_diffusion.py:185-187   (F.mse_loss)
 *   diffusion/train.py:33 + yamls/hydra-yamls/SD-2-base-256.yaml:55-58  (torch.optim.AdamW)
 *   diffusion/models/models.py:109-111             (xformers memory-efficient attention)
 *   diffusion/train.py:91-108                      (Composer low-precision GroupNorm / LayerNorm)
 * Every entry point below replaces one of those native ops; the comment on each names the call site.
 *
 * Conventions
 *   - plain pointers + sizes only; all pointers are DEVICE pointers unless said otherwise
 *   - activations are bf16, "NHWC": a tensor [B,H,W,C] is a row-major matrix [B*H*W rows][C] whose row
 *     stride (ld*, in ELEMENTS) may exceed C, so column slices of wider buffers (fused QKV, concat) are
 *     addressed without copies.  C, ld* and every column offset are multiples of 8 (16-byte vectors).
 *   - statistics, biases, norm affine parameters, master weights and gradients are fp32
 *   - no hidden allocation, no host synchronisation: scratch buffers are passed in, work is enqueued on
 *     `stream` (a hipStream_t) and the call returns immediately; safe under hipGraph stream capture
 *   - return value: 0 ok, DA_ERR_SHAPE (1) rejected arguments (nothing launched), DA_ERR_LAUNCH (2) HIP error
 */
#ifndef DIFFUSION_AMD_H
#define DIFFUSION_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* da_stream_t; /* == hipStream_t */

/* C[M,N] = alpha * gather(A)[M,K] . W[N,K]^T (+bias[N]) (+rowbias[image(m)][N]) (+R[M,N]).
 * Replaces torch.nn.Conv2d 3x3/1x1 (cuDNN) and torch.nn.Linear (cuBLAS) inside diffusers ResnetBlock2D /
 * Transformer2DModel, reached from stable_diffusion.py:183; with the transposed weight shadow it is also
 * their dgrad in backward.  K = ksize*ksize*Cin; W is [N][kh][kw][Cin]; M = B*Hout*Wout.
 * mode 0: stride 1 (pad 1 when ksize 3); 1: stride 2 pad 1 (Downsample2D); 2: dgrad of mode 1 (A = dY at
 * Hin x Win = half resolution); 3: 3x3 conv over the nearest-2x upsampled A (Upsample2D), Hout = 2*Hin; 4: stride 2 with
 * zero padding at the bottom / right only (AutoencoderKL encoder Downsample2D: F.pad(x, (0,1,0,1)) + conv stride 2).
 * out_fp32: C is float* (else bf16).  splitk_ws (may be NULL): fp32 workspace of splitk_ws_floats elements; when the
 * tile grid alone would leave most CUs idle (small M) the K loop is split over up to 8 workgroups per tile whose
 * partial slabs (splits*M*N floats) are summed by a fused finalize pass. */
int da_gemm_nt(const void* A, long lda, const void* W, void* C, long ldc, const float* bias, const void* rowbias,
               long ldrb, const void* R, long ldr, int M, int N, int K, int Cin, int Hin, int Win, int Hout, int Wout,
               int ksize, int mode, int out_fp32, float alpha, float* splitk_ws, long splitk_ws_floats,
               da_stream_t stream);

/* Feed-forward input projection with its GEGLU activation fused (diffusers FeedForward.net.0 = GEGLU.proj + gelu gate,
 * reached from stable_diffusion.py:183): F[M][2*inner] = A[M][K] . W[2*inner][K]^T + bias (bf16, kept for backward) and
 * G[M][inner] = F[:, :inner] * gelu_erf(F[:, inner:]) in one launch; bit-identical to da_gemm_nt followed by
 * da_geglu_fwd.  Requires inner % 160 == 0 and K % 64 == 0 (DA_ERR_SHAPE otherwise: use the two calls). */
int da_gemm_nt_geglu(const void* A, long lda, const void* W, void* F, long ldf, void* G, long ldg, const float* bias,
                     int M, int inner, int K, da_stream_t stream);

/* Backward counterpart on the dgrad of FeedForward.net.2: dG[M][inner] = dY[M][K] . Wt[inner][K]^T is gated against the
 * saved pre-activation F[M][2*inner] in the epilogue and leaves as dF[M][2*inner] (= da_geglu_bwd(F, dG)) without dG
 * ever reaching HBM; bit-identical to da_gemm_nt followed by da_geglu_bwd.  Requires inner % 320 == 0, K % 64 == 0. */
int da_gemm_nt_geglu_bwd(const void* dY, long lddy, const void* Wt, const void* F, long ldf, void* dF, long lddf, int M,
                         int inner, int K, da_stream_t stream);

/* tuning / test hooks (process-wide; 0 is always the shipped behaviour unless noted).  Returns DA_ERR_SHAPE for unknown keys.
 *   "gemm_nt_variant"   0 auto | 1 the 128x128 register-staged kernel | 4 / 5 / 10 / 14 / 12 force the 256x128 / 256x160 /
 *                       256x320 (8-wave) / 256x256 / 256x320 (16-wave) LDS-DMA form where eligible (Cin % 64 == 0) |
 *                       11 the 4-wave 128x320x32 form | 15 / 16 the 256x320 tile on 32x32x16 MFMAs (16 waves as 8x2 / 8 waves as 4x2) |
 *                       18 the 16-wave 384x128 form (N <= 128: VAE encoder)
 *   "gemm_nt_mfma32"    0 (default) never | -1 variant 12 becomes 15 for K <= 320 | 1 always
 *   "gemm_nt_dispatch"  1 (default) cost model over (tile form, split-K) | 0 the round-1 fill thresholds
 *   "gemm_nt_korder"    1 (default) 3x3 K loop walks a 64-channel chunk through its 9 taps | 0 tap-major
 *   "gemm_nt_splitk"    1 (default) split-K allowed | 0 never split
 *   "gemm_nt_persist"   -1 (default) linears / fused GEGLU run as one resident workgroup per CU walking the tile list |
 *                       n > 0 that many resident workgroups | 0 one workgroup per tile
 *   "gemm_nt_persist_conv" 1 (default) 3x3 convolutions with more tiles than CUs also run as a resident tile walk (next tile's
 *                       descriptors + first K-step ahead of the epilogue; +0...1.4 %, bit-identical) | 0 one workgroup per tile
 *   "gemm_nt_de"        1 (default) the convolution forms leave through the direct register -> HBM epilogue | 0 the LDS strip
 *                       epilogue everywhere | 2 / 3 also the linears without / with a residual (measured slower).  Bit-identical.
 *   "gemm_nt_ws"        bit mask, default 1: bit 0 the K = 320 linears (N = 320 ... 1280, M % 32 == 0, >= 8 row tiles per CU) run in
 *                       the weight-stationary kernel (W in the registers of four waves; gemm_nt_ws.hip) | bit 1 the K = 640 linears
 *                       too | bit 2 the fused GEGLU forward at K = 320 too | bit 3 the K = 640 form as two unpipelined workgroups per
 *                       CU (all three measured +-0 in the step).  Bit-identical.
 *   "gemm_nt_stream"    0 (default) | 1 / 2 the streaming short-K linear kernel (gemm_nt_v3.hip; slower) where it measured best /
 *                       wherever eligible; "gemm_nt_stream_lw" 4 | 16 its loader waves.  Bit-identical.
 *   "gemm_tn_ring"      0 (default) | 4 | 5: linear-layer weight gradients with a ring of 32-pixel half-stages (+-3 %).  Bit-identical.
 *   "attn_fused_bwd"    1 (default) da_attn_bwd runs as ONE kernel for Nk <= 128 (cross-attention, the 64-token level) | 0 the
 *                       dK/dV + dQ pair | 2 also 129 ... 256 keys on an 8-wave form (slower)
 *   "gemm_tn_variant"   0 auto | 1 the 128x128x32 wgrad kernel | 2 the 320x192x64 LDS-DMA wgrad kernel
 *   "gn_resident"       n (default 192): da_groupnorm_fwd / _bwd run as ONE kernel that holds a workgroup's (image, whole groups)
 *                       slab in registers - x (and dy) are read once - when the slab fits and the launch has >= n workgroups;
 *                       0 never (always the reduce / finalize / apply passes), 1 whenever the slab fits
 *   "gn_resident_min_slab" bytes (default 65536) of one tensor per workgroup below which the multi-pass form runs
 *   "gn_resident_form"  0 (default) auto | 1 16-wave backward forms only | 2 the 12-wave backward form above 8 vectors/thread
 *   "grad_overwrite"    0 (default) the gradient-producing entry points ADD to their outputs, as documented below | 1 they WRITE
 *                       them (da_gemm_tn_wgrad's dW and dbias, da_colsum_accum, da_image_colsum's db, dgamma / dbeta of
 *                       da_groupnorm_bwd / da_layernorm_bwd): set by the host around the first backward of an optimizer step,
 *                       which then needs no zero fill of the gradient buffer
 *   "reserve_cus"       R in [0, 128] (default 0): every grid sized to one round of the chip (persistent GEMM tile walks,
 *                       weight-gradient pixel splits, the dispatch cost model) uses #CUs - R, leaving R CUs to the RCCL
 *                       channels of an overlapping gradient all-reduce (set by the trainer when world size > 1) */
int da_set_option(const char* key, int value);
/* which kernel da_gemm_nt dispatches to for (M, N, K, Cin) given a split-K workspace of that many floats: 1 = gemm_nt_kernel (128x128 tile), 4 / 5 / 10 =
 * gemm_nt2_kernel with a 256x128 / 256x160 / 256x320 tile, 12 = the 16-wave 256x320 form, 14 = the 16-wave 256x256 form,
 * 15 = the 16-wave 256x320 form on 32x32x16 MFMAs (profiling labels only) */
int da_gemm_nt_variant_for(int M, int N, int K, int Cin, long splitk_ws_floats);

/* dW[N][ksize*ksize*Cin] += sum_m dY[m][n] * gather(X)[m][k]   (fp32).
 * Replaces the cuDNN/cuBLAS wgrad kernels autograd runs for the same layers (loss.backward() driven by
 * Composer).  modes 0, 1, 3 as above.  If dbias != NULL, dbias[n] += sum_m dY[m][n] as well
 * (fused into the large-tile kernel; the small-shape path uses da_colsum_accum with `scratch`, >= 256*N*2 floats).
 * When the pixel range is split over workgroups, the partial tiles AND the partial bias gradients are stored in split_ws
 * (fp32, caller-owned, may be shared with da_gemm_nt's split-K workspace on the same stream; ~128 MiB covers every split
 * grid of both wgrad kernels) and summed in a fixed order: dW and dbias are then bitwise reproducible run to run.  With
 * split_ws == NULL or too small they are accumulated with fp32 atomics instead (order-dependent last bits). */
int da_gemm_tn_wgrad(const void* dY, long lddy, const void* X, long ldx, float* dW, float* dbias, float* scratch, int M,
                     int N, int Cin, int Hin, int Win, int Hout, int Wout, int ksize, int mode, float* split_ws,
                     long split_ws_floats, da_stream_t stream);

/* which kernel da_gemm_tn_wgrad dispatches to (test / profiling label): 1 = gemm_tn_kernel (128x128x32, register-staged),
 * 2 = gemm_tn2_kernel<192, generic gather>, 3 = gemm_tn2_kernel<192, FAST> (uniform source stride + periodic border mask) */
int da_gemm_tn_variant_for(int M, int N, int Cin, int Hin, int Win, int Hout, int Wout, int ksize, int mode);

/* softmax(Q K^T * scale) V for head_dim 64, heads at column offsets h*64 of Q/K/V/O; L2[B][H][Nq] receives the
 * per-row log2-sum-exp.  Replaces xformers memory_efficient_attention (models.py:109-111) / diffusers
 * attention for attn1 (self) and attn2 (cross, Nk = 77). */
int da_attn_fwd(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv, void* O, long ldo,
                float* L2, int B, int H, int Nq, int Nk, float scale, da_stream_t stream);
/* the same with a causal mask (key j <= query q), Nq == Nk == N: the self-attention of the frozen text encoder
 * (transformers CLIPTextModel behind stable_diffusion.py:168,172); forward only. */
int da_attn_fwd_causal(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv, void* O, long ldo,
                       float* L2, int B, int H, int N, float scale, da_stream_t stream);
/* backward of da_attn_fwd; Delta[B][H][Nq] is scratch (rowsum(dO*O)). */
int da_attn_bwd(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv, const void* O, long ldo,
                const void* dO, long lddo, const float* L2, float* Delta, void* dQ, long lddq, void* dK, long lddk,
                void* dV, long lddv, int B, int H, int Nq, int Nk, float scale, da_stream_t stream);

/* floats of scratch the norm / colsum entry points need for (B, HW, C) */
long da_norm_scratch_floats(int B, int HW, int C);

/* GroupNorm (+ optional fused SiLU) over [B][HW][C], G groups.  Replaces torch.nn.GroupNorm / Composer
 * LPGroupNorm (train.py:91-99) + F.silu in ResnetBlock2D / Transformer2DModel / conv_norm_out.
 * mean_rstd[B][G][2] is saved for backward; scale_shift[B][C][2] is scratch (untouched when the single-pass form runs).
 * Single pass (see "gn_resident"): a 1024-thread workgroup owns HW pixels x CW channels (whole groups) of one image in
 * registers, so the tensor crosses HBM once in and once out; otherwise statistics, finalize and apply are three launches. */
int da_groupnorm_fwd(const void* X, long ldx, void* Y, long ldy, const float* gamma, const float* beta,
                     float* mean_rstd, float* scale_shift, float* scratch, int B, int HW, int C, int G, float eps,
                     int silu, da_stream_t stream);
/* dX = GN(+SiLU) backward (+ Radd if non-null); dgamma/dbeta accumulated (+=); coef[B][G][2] scratch.  In the single-pass
 * form x and dy stay in registers between the group sums and the dx sweep, and Radd is added to the bf16-rounded dX (as a
 * separate add of two bf16 tensors would); the multi-pass form adds it in fp32 before rounding. */
int da_groupnorm_bwd(const void* X, long ldx, const void* dY, long lddy, const void* Radd, long ldr, void* dX,
                     long lddx, const float* gamma, const float* beta, const float* mean_rstd, float* dgamma,
                     float* dbeta, float* coef, float* scratch, int B, int HW, int C, int G, int silu,
                     da_stream_t stream);

/* LayerNorm over rows of C.  Replaces torch.nn.LayerNorm / Composer LPLayerNorm (train.py:100-108) in
 * BasicTransformerBlock.  mean_rstd[M][2] saved for backward. */
int da_layernorm_fwd(const void* X, long ldx, void* Y, long ldy, const float* gamma, const float* beta,
                     float* mean_rstd, int M, int C, float eps, da_stream_t stream);
int da_layernorm_bwd(const void* X, long ldx, const void* dY, long lddy, const void* Radd, long ldr, void* dX,
                     long lddx, const float* gamma, const float* mean_rstd, float* dgamma, float* dbeta,
                     float* scratch, int M, int C, da_stream_t stream);

/* out[c] += sum_m X[m][c]  (bias gradients of every conv / linear) */
int da_colsum_accum(const void* X, long ldx, float* out, float* scratch, int M, int C, da_stream_t stream);

/* out[b][c] = sum over the HW pixels of image b of X (bf16, row stride ldo) and db[c] += sum_b out[b][c]:
 * gradient of the broadcast timestep-FiLM add (h + time_emb_proj(temb)[:, :, None, None]) and of conv1's bias. */
int da_image_colsum(const void* X, long ldx, void* out, long ldo, float* db, float* scratch, int B, int HW, int C,
                    da_stream_t stream);

/* GEGLU feed-forward gate: in = [a | g] (2*Cout columns), out = a * gelu_erf(g).  Replaces diffusers GEGLU. */
int da_geglu_fwd(const void* in, long ldi, void* out, long ldo, int M, int Cout, da_stream_t stream);
int da_geglu_bwd(const void* in, long ldi, const void* dout, long lddo, void* din, long lddi, int M, int Cout,
                 da_stream_t stream);

/* SiLU on the timestep embedding (ResnetBlock2D nonlinearity(temb), TimestepEmbedding act) */
int da_silu_fwd(const void* x, long ldx, void* y, long ldy, int M, int C, da_stream_t stream);
/* erf-GELU (the text encoder's MLP activation, CLIPTextConfig.hidden_act = "gelu"); forward only */
int da_gelu_fwd(const void* x, long ldx, void* y, long ldy, int M, int C, da_stream_t stream);
int da_silu_bwd(const void* x, long ldx, const void* dy, long lddy, void* dx, long lddx, int M, int C,
                da_stream_t stream);

/* strided add / copy: residual gradient sums and torch.cat([h, skip], dim=1) of the up blocks */
int da_add(const void* a, long lda, const void* b, long ldb, void* o, long ldo, int M, int C, da_stream_t stream);
int da_copy2d(const void* a, long lda, void* o, long ldo, int M, int C, da_stream_t stream);

/* nearest-neighbour 2x upsample of [B,H,W,C] (diffusers Upsample2D / F.interpolate) and its backward */
int da_upsample2x_fwd(const void* x, void* y, int B, int H, int W, int C, da_stream_t stream);
int da_upsample2x_bwd(const void* dy, void* dx, int B, int H, int W, int C, da_stream_t stream);

/* diffusers Timesteps(flip_sin_to_cos=True, freq_shift=0): out[B][dim] = [cos | sin] (bf16); t is int64 */
int da_timestep_embed(const long long* t, void* out, int B, int dim, da_stream_t stream);

/* DDPMScheduler.add_noise (stable_diffusion.py:180) fused with the NCHW fp32 -> NHWC(8) bf16 relayout and
 * the training target (eps, or get_velocity when v_pred - pixel_diffusion.py:90-91).
 * x0, eps: [B][4][HW] fp32; xt: [B*HW][8] bf16 (channels 4..7 zero); target: [B*HW][8] fp32. */
int da_add_noise(const float* x0, const float* eps, const long long* t, const float* sqrt_ac,
                 const float* sqrt_1mac, void* xt, float* target, int B, int HW, int v_pred, da_stream_t stream);

/* F.mse_loss(pred, target) (stable_diffusion.py:187) over the 4 valid channels of NHWC(8) fp32 tensors and its
 * gradient dpred = grad_coef * (pred - target) (bf16, NHWC(8)).  loss[0] (+)= weight * mean.  scratch >= 1024 floats */
int da_mse_loss(const float* pred, const float* target, void* dpred, float* loss, float* scratch, long total_pix,
                float grad_coef, float weight, int accumulate, da_stream_t stream);

/* torch.optim.AdamW step (train.py:33; SD-2-base-256.yaml:55-58) on flat fp32 master/moment buffers, gradient
 * pre-scaled by grad_scale; also writes the bf16 compute shadow.  If ema != NULL the exponential moving average of
 * the weights (diffusion/algorithms/ema.py:26-76 compute_ema: ema = s*ema + (1-s)*w) is updated in the same pass. */
int da_adamw(float* p, const float* g, float* m, float* v, void* shadow, float* ema, float ema_smoothing, long n,
             float lr, float beta1, float beta2, float eps, float wd, int step, float grad_scale, da_stream_t stream);

int da_cast_f32_bf16(const float* src, void* dst, long n, da_stream_t stream);

/* dst[c][T-1-t][n] = src[n][t][c]: the weight layout da_gemm_nt needs for dgrad */
int da_transpose_weight(const void* src, void* dst, int N, int T, int C, da_stream_t stream);

/* the same for many tensors in one launch.  desc: device array of ntensors records
 * {long src_off, long dst_off (elements from the bases), int N, int T, int C, int first_block}; tensor i owns blocks
 * [first_block_i, first_block_i + T*ceil(N/64)*ceil(C/64)); total_blocks = their sum. */
int da_transpose_weights_batched(const void* src_base, void* dst_base, const void* desc, int ntensors, int total_blocks,
                                 da_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
