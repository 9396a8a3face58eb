/*
 * vaa.h — C-ABI of libvaa_hip.so: the MI355X (gfx950) replacement for the UADA/UPA/TMA inner attack loop's
 * patch operators of William-wAng618/roboticAttack.
 *
 * The reference is pure Python and has no FFI; the seams this library replaces are the Python operator calls
 * listed per entry point below (file:line under the reference tree). INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add to call them.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only. Every pointer marked `dev` is a DEVICE pointer owned by the
 *     caller (e.g. a PyTorch-ROCm tensor's data_ptr()); `host` pointers are ordinary host memory read during
 *     the call. Nothing is allocated or freed by the library; scratch comes from the caller via *_ws_bytes().
 *   - Every function returns VAA_OK (0) or a negative VAA_E_* code; vaa_last_error() returns a thread-local
 *     message for the last failure on the calling thread.
 *   - Kernels are enqueued on the caller's `stream` (a hipStream_t passed as void*; NULL = default stream)
 *     and the call returns without synchronising. Re-entrant per stream; no global mutable state.
 *   - Images are 224x224 (the reference hard-codes this size: UADA.py:60, modeling_prismatic.py:120).
 */
#ifndef VAA_H_
#define VAA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VAA_OK 0
#define VAA_E_INVALID (-1)     /* bad argument (null pointer, size out of range, unknown mode) */
#define VAA_E_UNSUPPORTED (-2) /* valid request the kernels do not cover (e.g. patch larger than 224) */
#define VAA_E_LAUNCH (-3)      /* HIP runtime reported an error at launch */
#define VAA_E_WORKSPACE (-4)   /* workspace missing or too small */
#define VAA_E_NODEVICE (-5)    /* no gfx950 device visible */

#define VAA_IMG 224

/* paste-mask rule */
#define VAA_MASK_LT_M20 0 /* keep patch where !(canvas < -20)  : apply_random_patch_batch, appply_random_transform.py:131 */
#define VAA_MASK_NE_M100 1 /* keep patch where canvas != -100   : paste_patch_fix / random_paste_patch, :153 :179; geometry=0 only
                              (with a warp the rule would hinge on the rounding of -100*(sum of weights): VAA_E_UNSUPPORTED) */

/* loss modes (K3) */
#define VAA_LOSS_UADA 0     /* w^2*mean((r-t)^2) + 1/CE                       UADA.py:145-148,381-406 */
#define VAA_LOSS_UADA_DDP 1 /* w^2*mean((r-t)^2)                              UADA_ddp.py:99-124,203-206 */
#define VAA_LOSS_UPA 2      /* alpha*mean(cos+1) + beta/(mean||e-l|| + 1e-3)  UPA.py:367-387 */
#define VAA_LOSS_CE 3       /* scale*CE (TMA target CE, UPA guide / -CE)      TMA.py:148, UPA.py:143-150 */

/* logits element type */
#define VAA_DTYPE_F32 0
#define VAA_DTYPE_BF16 1

/* logits layout */
#define VAA_LAYOUT_FULL 0 /* [B,S,V], S = 256 + L: row (b, S-L+k) predicts labels[b,k+1] (HF shift; UADA.py:385) */
#define VAA_LAYOUT_ROWS 1 /* [R,V]: only the labelled rows, in (b,k) row-major order of labels[b,k+1] != -100 */

/* gradient storage of vaa_loss_rows_fwd_bwd */
#define VAA_GRAD_FULL 0  /* [R,V], the logits' dtype */
#define VAA_GRAD_SLICE 1 /* [R,256]: the action columns 31744..31999 only (UADA_DDP / UPA: the gradient is zero elsewhere) */

/* optimiser modes (K4) */
#define VAA_OPT_ADAMW_HF 0 /* transformers==4.40.1 AdamW.step (eps outside bias correction) + clamp(0,1)  UADA.py:155-156 */
#define VAA_OPT_PGD_SIGN 1 /* p = clamp(p - lr*sign(g), 0, 1)                                            TMA.py:171-175 */

const char* vaa_last_error(void);
int vaa_version(void);
/*
 * PROCESS-WIDE STATE — the two places where this library is NOT the stateless, per-stream re-entrant operator set SURVEY.md section 8b sketched
 * (everything else is: caller-owned memory and streams, no allocation, no synchronisation, thread-local error text):
 *   (1) the device-failure word behind vaa_async_error() below: one pinned host word per process, STICKY, not attributed to a stream;
 *   (2) the per-dispatch profiler vaa_prof_*: one record table per process behind a mutex.
 * Also per process, but invisible to callers: a launch-tag counter and "one stream at a time has a waiting grid in flight" bookkeeping for the
 * launches whose workgroups wait for each other (vaa_head_slice_fwd_bwd; the opt-in one-launch form of vaa_loss_rows_fwd_bwd) — they only decide
 * between the one-launch and the two-launch form of those calls, never their results.
 *
 * Device-side failures surface here and NEVER as a silent NaN: a kernel that has to give up (vaa_head_slice_fwd_bwd's one-launch form and the opt-in
 * one-launch K3, when a hand-over runs out of polls) NaN-poisons its outputs AND sets a bit in a pinned host word. The word is process-wide and STICKY:
 * from then on EVERY library call on any stream or thread returns VAA_E_LAUNCH (reason in vaa_last_error()) — the failure is not attributed
 * to a stream, and no call consumes it — until vaa_async_error() is called: the explicit poll reports the failure (VAA_E_LAUNCH) and clears
 * the word. The attack loops poll it behind their once-per-outer-iteration read-back, before anything is written to disk (UADA.py:257-275).
 * Returns VAA_OK or VAA_E_LAUNCH.
 */
int vaa_async_error(void);
/* VAA_OK when a gfx950 device is visible to the HIP runtime, else VAA_E_NODEVICE. */
int vaa_device_check(void);

/*
 * Per-dispatch timing of the library's own kernels (measurement aid; no reference counterpart). While armed, every kernel the library
 * launches is dispatched with its own start/stop event pair bound to THAT dispatch (hipExtLaunchKernel): its elapsed time is the
 * dispatch's begin-to-end GPU time, the quantity `rocprofv3 --kernel-trace` reports, read inside a running step without marker packets.
 *   vaa_prof_start(capacity): allocate (once) `capacity` event pairs, drop earlier records, arm; capacity 0 disarms. Dispatches beyond
 *             the capacity run unprofiled. Do not arm during stream capture.
 *   vaa_prof_stop(): disarm; returns the number of records.
 *   vaa_prof_get(i, &name, &usec): waits for record i's dispatch and returns the kernel's name (static string) and duration in us.
 * Process-wide state behind a mutex; the launching thread arms / disarms between steps.
 */
int vaa_prof_start(int capacity);
int vaa_prof_stop(void);
int vaa_prof_get(int i, const char** name, float* usec);

/*
 * K1 — replaces RandomPatchTransform.apply_random_patch_batch (appply_random_transform.py:104-136) followed by the
 * caller's `.to(torch.bfloat16)` (UADA.py:142), and paste_patch_fix (:160-188) with mask_mode=VAA_MASK_NE_M100,
 * geometry=0. The random draws (:120-128) stay on the host so the RNG streams match; their results come in as xy/theta.
 *   img_u8   dev  [B,224,224,3] uint8 HWC (what PIL/ToTensor sees, :108)
 *   patch    dev  [3,ph,pw] float32
 *   xy       dev  [B,2] int32 (x,y) paste position (:123-125)
 *   theta    dev  [B,6] float32 row-major 2x3 affine (rows 0-1 of combined_transform_matrix(), :88-96); ignored if !geometry
 *   mean6/std6 host [6] float32: channels 0-2 -> first normalisation, 3-5 -> second (UADA.py:56-57)
 *   out_bf16 dev  [B,6,224,224] bfloat16 bits: the model's pixel_values
 *   keep_bits dev [B,3,224*224/8] uint8 or NULL: bit (p&7) of byte p>>3 is 1 where channel c of pixel p=i*224+j
 *             shows the patch (the `torch.where` mask, :131); consumed by vaa_patch_grad_gather
 */
int vaa_patch_apply_fwd(const uint8_t* img_u8, const float* patch, const int32_t* xy, const float* theta, int B, int ph,
                        int pw, int geometry, int mask_mode, const float* mean6, const float* std6, uint16_t* out_bf16,
                        uint8_t* keep_bits, void* stream);

/*
 * K1 in TILE-MAJOR form — the same values as vaa_patch_apply_fwd, laid out as the operands of the two ViT patch-embed GEMMs
 * (modeling_prismatic.py:120-123 -> timm PatchEmbed, Conv2d(3, D, 14, stride 14) == a GEMM over 588-pixel tiles), for callers that own the
 * patch-embed weights (K2' below): the [B,6,224,224] tensor and its two 19 MB im2col copies per step are never made.
 *   pdesc     dev [B,4] or NULL: per-image patches as in vaa_patch_apply_fwd_multi (ph/pw are then upper bounds)
 *   out0,out1 dev bf16 [B,256,588]: tile t = ty*16 + tx, element c*196 + y*14 + x; out0 = first normalisation (channels 0-2 of
 *             pixel_values), out1 = second (channels 3-5). e_k = out_k @ W_k^T + b_k with W_k = conv weight [D,3,14,14] flattened to [D,588]
 *   keep_tiles dev u16 [B,3,256,14] or NULL: bit x of word (c, t, y) = channel c of pixel (14 ty + y, 14 tx + x) shows the patch
 *   tile_flags dev u32 [B,256] or NULL: != 0 when tile t holds a kept pixel in any channel (the tile list K2' evaluates)
 */
int vaa_patch_apply_fwd_tiles(const uint8_t* img_u8, const float* patch, const int32_t* pdesc, const int32_t* xy, const float* theta, int B,
                              int ph, int pw, int geometry, int mask_mode, const float* mean6, const float* std6, uint16_t* out0,
                              uint16_t* out1, uint16_t* keep_tiles, uint32_t* tile_flags, void* stream);

/*
 * K2 — replaces the autograd backward of K1 to the patch (implicit in `.backward()`, UADA.py:148): bf16->f32 cast,
 * /std of both normalisations summed, where-mask, grid_sample-backward scatter, slice, sum over the batch.
 *   gout_bf16 dev [B,6,224,224] bfloat16 bits: dL/d pixel_values from the model
 *   patch, xy, theta, geometry, mask_mode: as given to K1 for the same step
 *   keep_bits dev: K1's mask output, or NULL to recompute the mask from `patch`
 *   std6     host [6]
 *   gpatch   dev  [3,ph,pw] float32, overwritten with dL/d patch (sum over the B images); NULL leaves the final fixed-order sum to
 *             vaa_step_epilogue[_update]: the vaa_patch_grad_partials(B) partial tiles [parts][3*ph*pw] f32 then sit at the start of ws
 *   ws       dev  scratch of at least vaa_patch_grad_ws_bytes(B,ph,pw) bytes (contents undefined on entry and exit)
 * Numerics: every bilinear contribution fl(G*w) is accumulated as an integer (quantum 2^-30 of the largest |G| a workgroup meets), the
 * partial tiles are added in a fixed order: the same arguments give the same bits, for every patch size up to 224x224 (no global or
 * floating-point atomics); the result is the exact sum of the reference's fp32 products rounded once. A non-finite upstream value makes
 * (at least) the patch rows it touches NaN.
 */
size_t vaa_patch_grad_ws_bytes(int B, int ph, int pw);
int vaa_patch_grad_gather(const uint16_t* gout_bf16, const float* patch, const int32_t* xy, const float* theta,
                          const uint8_t* keep_bits, int B, int ph, int pw, int geometry, int mask_mode, const float* std6,
                          float* gpatch, void* ws, size_t ws_bytes, void* stream);

/*
 * resize_patch=True (BASELINE config 5) — replaces, for the whole batch at once, `patch = transforms.Resize((height, width))(patch)`
 * with `scale = random.uniform(0.61, 1.39)` (appply_random_transform.py:113-116; semantics of SURVEY.md Appendix A-D2: every image
 * scales the BASE patch), the paste/warp of every image's own resized patch (:118-136) and their autograd backward.
 * The draws stay on the host; their results travel as a per-image descriptor:
 *   pdesc    dev  [B,4] int32 = {h_b, w_b, offset_b, 0}: image b's patch is [3,h_b,w_b] float32 at packed + offset_b
 *             (offsets in floats; regions must not overlap; vaa_patch_*_multi read sizes from here, max_h/max_w bound them)
 * vaa_patch_resize_fwd: packed[offset_b ...] = antialiased bilinear resize of patch [3,ph,pw] to (h_b, w_b) — torchvision Resize on a
 *             tensor == torch F.interpolate(mode='bilinear', antialias=True, align_corners=False); bit-exact against torch's CPU kernel.
 * vaa_patch_resize_bwd: gpatch [3,ph,pw] = sum_b adjoint(resize_b)(gpacked_b), overwritten; ws >= vaa_patch_resize_ws_bytes(B,ph,pw)
 *             (one partial per image group — one image per group until the batch alone fills the chip; 0 for a single group).
 * vaa_patch_apply_fwd_multi / vaa_patch_grad_gather_multi: K1 / K2 with per-image patches. K2's output gpacked has the layout of
 *             packed and holds d L / d (every image's own resized patch); elements between the patches are not written.
 * Launch counts do not depend on B: forward = resize + K1, backward = K2 + resize adjoint + fixed-order sum over the images.
 */
int vaa_patch_resize_fwd(const float* patch, int ph, int pw, const int32_t* pdesc, int B, float* packed, void* stream);
size_t vaa_patch_resize_ws_bytes(int B, int ph, int pw);
int vaa_patch_resize_bwd(const float* gpacked, int ph, int pw, const int32_t* pdesc, int B, float* gpatch, void* ws, size_t ws_bytes,
                         void* stream);
int vaa_patch_apply_fwd_multi(const uint8_t* img_u8, const float* packed, const int32_t* pdesc, const int32_t* xy, const float* theta,
                              int B, int max_h, int max_w, int geometry, int mask_mode, const float* mean6, const float* std6,
                              uint16_t* out_bf16, uint8_t* keep_bits, void* stream);
int vaa_patch_grad_gather_multi(const uint16_t* gout_bf16, const float* packed, const int32_t* pdesc, const int32_t* xy,
                                const float* theta, const uint8_t* keep_bits, int B, int max_h, int max_w, int geometry, int mask_mode,
                                const float* std6, float* gpacked, void* stream);

/*
 * K3 — replaces HF Llama's `.loss` (via modeling_prismatic.py:404-415) + OpenVLAAttacker.weighted_loss
 * (UADA.py:381-406, UADA_ddp.py:99-124, UPA.py:367-387) and their autograd backward to the logits.
 *   logits   dev  f32|bf16, layout FULL [B,S,V] or ROWS [R,V]. For ROWS pass S = R (the number of rows, which must equal the number
 *             of labelled positions): the call then builds the row map in `ws` and runs vaa_loss_rows_fwd_bwd (below); S = 0 (unknown)
 *             keeps the label-driven schedule
 *   labels   dev  [B,L] int64, already masked by the caller (mask_labels, UADA.py:371-379)
 *   params   host [4] float32: {w (MSE weight: 5 or --MSE_weights), alpha, beta, scale (1/accumulate_steps)}
 *   scalars  dev  [8] float32 out: {total, CE, w^2*MSE, aux0 (UPA angle), aux1 (UPA dist), #CE rows, #action rows, UAD}
 *   pred_tokens dev [B*(L-1)] int32 or NULL: argmax action token (31744 + argmax of the 256-slice, UADA.py:395) per
 *             labelled row in (b,k) order (rows whose label is <= 2 get -1)
 *   glogits  dev or NULL: d total / d logits in the logits' dtype and layout. FULL: only labelled rows are written
 *             (zero the buffer once; rows never change while labels keep their shape). ROWS: every row is written.
 *   ws       dev scratch >= vaa_loss_ws_bytes(B,L)
 */
size_t vaa_loss_ws_bytes(int B, int L);
int vaa_loss_fwd_bwd(const void* logits, int dtype, int layout, const int64_t* labels, int B, int S, int L, int V, int mode,
                     const float* params, float* scalars, int32_t* pred_tokens, void* glogits, void* ws, size_t ws_bytes,
                     void* stream);
/* The same with one more output: pred_full_tokens dev [B*(L-1)] or NULL = argmax over ALL V logits of every labelled row
 * (`action_logits.argmax(dim=2)`, UADA.py:165-167, TMA.py:148-149: what the reference's relative-distance / L1 / ASR metrics and the
 * best-patch selection read), -1 on unlabelled positions. pred_tokens stays the action-slice argmax that feeds UAD (UADA.py:395). */
int vaa_loss_fwd_bwd_ex(const void* logits, int dtype, int layout, const int64_t* labels, int B, int S, int L, int V, int mode,
                        const float* params, float* scalars, int32_t* pred_tokens, int32_t* pred_full_tokens, void* glogits, void* ws,
                        size_t ws_bytes, void* stream);

/*
 * K3 on the labelled rows with a prebuilt row map (LM head + loss on labelled rows only) — what the attack
 * loops use: labels are fixed during the innerLoop steps of an outer iteration (UADA.py:130-133), so the map is built once per outer
 * iteration and a step never touches the label matrix.
 *   vaa_loss_rowmap_build: rowmap dev (>= vaa_loss_rowmap_bytes(B,L)) = int32 {R, #action rows, 0, 0} followed by {b, k, label, ord}
 *             per labelled position in (b,k) row-major order of labels[b,k+1] != -100 (the order of VAA_LAYOUT_ROWS).
 *   vaa_loss_rows_fwd_bwd: logits dev [R,V] f32|bf16 in that order; R (host) must equal the map's count (else scalars[0] = NaN).
 *             pred_tokens      dev [B*(L-1)] or NULL: 31744 + argmax of the 256 action logits (UADA.py:395, feeds UAD); -1 elsewhere
 *             pred_full_tokens dev [B*(L-1)] or NULL: argmax over ALL V logits (`action_logits.argmax(dim=2)`, UADA.py:165-167,
 *                              TMA.py:148-149: relative distance, L1 / ASR metrics, best-patch selection); -1 on unlabelled positions
 *             grad dev or NULL, grad_kind VAA_GRAD_FULL [R,V] | VAA_GRAD_SLICE [R,256] (only for UADA_DDP / UPA, whose gradient is
 *                              confined to the action columns: the LM-head backward then contracts over 256 columns, not 32,064)
 *             ws >= vaa_loss_rows_ws_bytes(R). Rows are split over 2-4 workgroups so that 128 rows fill the 256 CUs; in UADA_DDP
 *             mode the gradient is written by the same pass that reads the logits. Full-row gradients whose scale needs the folded
 *             scalars (UADA's 1/CE^2 term, UADA.py:145-148; CE, TMA.py:148) can be ONE launch too — statistics, grid-wide hand-over,
 *             gradient from the registers, every row read once — ONLY with VAA_K3_ONE_PASS=1 in the environment (opt-in since
 *             round 4: 2.8 us per call against a residency assumption no launch API guarantees) and when the grid takes at most half
 *             of the device's resident slots, the stream is not being captured and no other stream of the process has such a launch
 *             in flight; else two launches with the same bits in every output. A hand-over that times out NaN-poisons the gradient AND
 *             raises vaa_async_error() (every later library call fails with VAA_E_LAUNCH until the word is polled).
 */
size_t vaa_loss_rowmap_bytes(int B, int L);
int vaa_loss_rowmap_build(const int64_t* labels, int B, int L, void* rowmap, size_t rowmap_bytes, void* stream);
size_t vaa_loss_rows_ws_bytes(int R);
int vaa_loss_rows_fwd_bwd(const void* logits, int dtype, const void* rowmap, int R, int B, int L, int V, int mode, const float* params,
                          float* scalars, int32_t* pred_tokens, int32_t* pred_full_tokens, void* grad, int grad_kind, void* ws,
                          size_t ws_bytes, void* stream);

/*
 * Split form of vaa_loss_rows_fwd_bwd for the data-parallel step (UADA_ddp.py:199-209), whose backward does not wait for the loss scalars:
 *   vaa_loss_rows_stats : the statistics pass alone; in VAA_LOSS_UADA_DDP mode it also writes the gradient (grad may be NULL; other modes
 *             need the folded scalars for their gradient and must use vaa_loss_rows_fwd_bwd). ws as in vaa_loss_rows_fwd_bwd.
 *   vaa_step_epilogue   : ONE launch between the backward and the gradient exchange — replaces K2's final reduce launch, K3's finishing
 *             launch and the packing of the DDP message (dist.PatchGradSync):
 *               msg[0..n)    = sum of K2's `nparts` partial tiles [nparts][n] (n = 3*ph*pw), the fixed order of
 *                              vaa_patch_grad_gather: bitwise the gradient that call would have written
 *               rowmap != NULL: the statistics in loss_ws (left by vaa_loss_rows_stats with the same R, B, L, V, mode, params) are folded:
 *                              scalars[8], pred_tokens, pred_full_tokens as vaa_loss_rows_fwd_bwd writes them
 *               rowmap == NULL: scalars is an INPUT (already final)
 *               msg[n..n+4)  = {CE, w^2*MSE, UAD, total}: the logging scalars that travel with the gradient (C3 + C4 in one all-reduce)
 */
int vaa_loss_rows_stats(const void* logits, int dtype, const void* rowmap, int R, int B, int L, int V, int mode, const float* params, void* grad,
                        int grad_kind, void* ws, size_t ws_bytes, void* stream);
int vaa_step_epilogue(const float* partials, int nparts, int n, const void* rowmap, int R, int B, int L, int V, int mode,
                      const float* params, const void* loss_ws, size_t loss_ws_bytes, float* scalars, int32_t* pred_tokens,
                      int32_t* pred_full_tokens, float* msg, void* stream);
/* Single-GPU form (UADA.py:148-157: nothing sits between the gradient and optimizer.step()): the epilogue also applies K4 — vaa_patch_update
 * with grad_scale = 1 and no L1 clip — to every gradient element as it is produced (same per-element arithmetic: patch / m / v get the
 * bits the separate launch would write). stat_part dev f64 [ceil(n/64)][2] or NULL: per-block {sum |g|, sum g}; their sums give K4's
 * logged {sum|g|, mean g = sum g / n}. */
int vaa_step_epilogue_update(const float* partials, int nparts, int n, const void* rowmap, int R, int B, int L, int V, int mode,
                             const float* params, const void* loss_ws, size_t loss_ws_bytes, float* scalars, int32_t* pred_tokens,
                             int32_t* pred_full_tokens, float* msg, float* patch, float* m, float* v, int opt_mode, float lr, float beta1,
                             float beta2, float eps, int step, double* stat_part, void* stream);

/*
 * LM head FUSED with K3's statistics (SURVEY.md section 8f-2 as the survey wrote it; for callers that own the LM-head weight) — replaces
 * `logits = lm_head(hidden)` on the labelled rows (modeling_prismatic.py:404-415 -> HF Llama's bf16 lm_head) followed by vaa_loss_rows_stats:
 * the [R,V] logits are never written. The head weight is streamed from HBM once (full 128-byte lines, global -> LDS by LDS-DMA, into MFMA fragments), every
 * workgroup reduces its 128 vocabulary columns to per-row {max, sum exp, argmax, label logit}; a second small launch folds them per row,
 * computes the action-slice statistics (UADA.py:384-389) and — VAA_LOSS_UADA_DDP — writes the gradient slice.
 *   hidden   dev bf16 [R,D]: final-norm hidden states of the labelled rows, in the row map's order;  w_head dev bf16 [V,D]
 *   rowmap, R, B, L, V, mode, params: as vaa_loss_rows_stats;  grad_slice dev bf16 [R,256] or NULL (only VAA_LOSS_UADA_DDP)
 *   loss_ws  dev >= vaa_loss_rows_ws_bytes(R): left exactly as vaa_loss_rows_stats leaves it — vaa_step_epilogue[_update] folds it unchanged
 *   head_ws  dev >= vaa_head_loss_ws_bytes(R, V): scratch + the rows' 256 action logits (bf16) for vaa_head_loss_rows_finish;  logits_dbg dev bf16 [R,V] or NULL (tests: the bf16 logits the statistics were made of)
 * Covers vaa_head_loss_rows_applies(R, D, V) == 1: R <= 128 rows (one pass over the weight), D a multiple of 64; else VAA_E_UNSUPPORTED and the
 * caller keeps its GEMM + vaa_loss_rows_stats. Logits are the fp32 MFMA sums rounded to bf16 (the reference's bf16 head); for the same bf16
 * logits the slice statistics and the gradient slice are bit for bit those of vaa_loss_rows_stats, CE agrees to fp32 summation order.
 */
size_t vaa_head_loss_ws_bytes(int R, int V);
int vaa_head_loss_rows_applies(int R, int D, int V);
int vaa_head_loss_rows_stats(const uint16_t* hidden, const uint16_t* w_head, int D, const void* rowmap, int R, int B, int L, int V, int mode,
                             const float* params, void* grad_slice, void* loss_ws, size_t loss_ws_bytes, void* head_ws, size_t head_ws_bytes,
                             uint16_t* logits_dbg, void* stream);

/*
 * The finishing pass behind vaa_head_loss_rows_stats for callers that do not run vaa_step_epilogue: folds the rows into scalars[8] and the
 * prediction maps exactly like the second launch of vaa_loss_rows_fwd_bwd (same kernel), reading a row's 256 action logits from head_ws
 * instead of [R,V] logits — and, VAA_LOSS_UPA (UPA.py:367-387: the gradient needs the batch means), writes the gradient slice.
 * Together the two calls evaluate every mode WITHOUT logits in memory (validation passes; UADA_DDP and UPA also with their gradient);
 * the gradients of the modes with a cross-entropy term (VAA_LOSS_UADA, VAA_LOSS_CE) need every logit: VAA_E_UNSUPPORTED, use the LM-head
 * GEMM + vaa_loss_rows_fwd_bwd there.
 *   rowmap, R, B, L, V, mode, params: as given to vaa_head_loss_rows_stats;  loss_ws, head_ws: as it left them
 *   scalars dev f32[8]; pred_tokens / pred_full_tokens dev i32 [B,L-1] or NULL;  grad_slice dev bf16 [R,256] or NULL (only VAA_LOSS_UPA)
 */
int vaa_head_loss_rows_finish(const void* rowmap, int R, int B, int L, int V, int mode, const float* params, void* loss_ws, size_t loss_ws_bytes,
                              const void* head_ws, size_t head_ws_bytes, float* scalars, int32_t* pred_tokens, int32_t* pred_full_tokens,
                              void* grad_slice, void* stream);

/*
 * K3s — the SLICE-ONLY head: LM head on the labelled rows restricted to the 256 action columns + slice statistics + loss gradient + the head's
 * backward, ONE launch per inner step for the modes whose loss lives in those columns (VAA_LOSS_UADA_DDP: UADA_ddp.py:99-124; VAA_LOSS_UPA:
 * UPA.py:367-387). Replaces vaa_head_loss_rows_stats (+ finish) and the `grad_slice @ W[31744:32000]` GEMM on the steps whose full-vocabulary CE
 * and argmax nobody reads: the reference reads `celoss` once per OUTER iteration (the last inner step's value, UADA_ddp.py:214-221) and never in
 * UPA's reverse-direction mode (UPA.py:145-186), so the 263 MB weight stream runs on those steps only and this kernel (2.1 MB of weights) on all.
 *   hidden    dev bf16 [R,D];  w_head dev bf16 [V,D] (rows 31744..31999 are read);  w_slice_t dev bf16 [D,256] = the slice transposed, written
 *             ONCE per weight by vaa_head_slice_pack (frozen weights: pack at model load, keep resident); may be NULL when dhidden is NULL
 *   rowmap, R, B, L, V, mode, params: as vaa_loss_rows_stats (mode UADA_DDP or UPA, else VAA_E_UNSUPPORTED)
 *   dhidden   dev bf16 [R,D] out or NULL (forward only): d total / d hidden = g [R,256] x W[31744:32000] with g rounded to bf16 (fp32 MFMA sums)
 *   grad_slice dev bf16 [R,256] out or NULL: g itself (tests; bit for bit the slice vaa_head_loss_rows_stats / _finish write for the same rows)
 *   loss_ws   dev >= vaa_loss_rows_ws_bytes(R): the SliceStats in K3's layout and NEUTRAL full-vocabulary parts {m = -inf, s = 0}: the fold
 *             (vaa_step_epilogue[_update], or this kernel's own when scalars != NULL) then reports CE = 0 ("not evaluated on this step") and
 *             pred_full = -1; calling vaa_head_loss_rows_stats AFTERWARDS on the same loss_ws (the steps whose CE is read) replaces the parts
 *             with real ones and rewrites the same SliceStat bits
 *   scalars   dev f32[8] or NULL; pred_tokens / pred_full_tokens dev i32 [B,L-1] or NULL: folded and published by the launch itself (NULL:
 *             left to vaa_step_epilogue)
 *   ws        dev >= vaa_head_slice_ws_bytes(R): the [ceil16(R),128] action logits as 64-bit words {launch tag, logit 2q+1, logit 2q} (bf16) — the
 *             SAME MFMA sequence per element as vaa_head_loss_rows_stats (same k order), hence bit for bit its slice logits, statistics and
 *             gradient slice
 * One launch needs workgroups that wait for each other (<= 256 — 32 per block of 16 rows, 8 action columns each; VAA_K3S_COLS=16: 16 per block —; they
 * poll the tagged words they are going to use): admitted when the device keeps twice the grid resident, the stream is not being
 * captured and no other stream of the process has a waiting grid in flight; otherwise (or VAA_K3S_ONE_LAUNCH=0) the same kernel runs as two
 * launches with the same bits. A hand-over that times out NaN-poisons dhidden / the statistics AND raises vaa_async_error().
 * Covers vaa_head_slice_applies(R, D, V) == 1: R <= 128, D a multiple of 64 up to 4096, V <= 32768.
 */
int vaa_head_slice_applies(int R, int D, int V);
size_t vaa_head_slice_ws_bytes(int R);
int vaa_head_slice_pack(const uint16_t* w_head, int D, int V, uint16_t* w_slice_t, void* stream);
int vaa_head_slice_fwd_bwd(const uint16_t* hidden, const uint16_t* w_head, const uint16_t* w_slice_t, int D, const void* rowmap, int R, int B, int L,
                           int V, int mode, const float* params, uint16_t* dhidden, uint16_t* grad_slice, void* loss_ws, size_t loss_ws_bytes,
                           float* scalars, int32_t* pred_tokens, int32_t* pred_full_tokens, void* ws, size_t ws_bytes, void* stream);

/*
 * K2' (for callers that own the model's patch-embed weights) — K2 fed by the gradient of the ViT patch-embed OUTPUTS instead of the pixel gradient: the
 * patch-embed backward (modeling_prismatic.py:120-123 -> timm PatchEmbed, Conv2d(3, D, 14, stride 14) == a GEMM over 588-pixel
 * tiles) is evaluated by MFMA only for the 14x14 tiles that carry kept patch pixels and consumed in place by the gather.
 *   dy0, dy1  dev bf16 [B,256,D0], [B,256,D1]: dL/d(patch-embed output) of the DINOv2 and SigLIP towers, tokens in tile order
 *   wp0, wp1  dev bf16 [vaa_patch_embed_packed_elems(D)]: the conv weights [D,3,14,14] of a tower, flattened to [D,588], transposed to
 *             wt [588,D] and re-ordered ONCE into MFMA fragment order by vaa_patch_embed_pack_weights (frozen weights: pack at model load,
 *             keep resident; 592 x D elements — 588 columns + 4 of zero padding)
 *   keep_bits dev: K1's keep mask (required); round_bf16 != 0 rounds each tower's pixel gradient to bf16 before the 1/std scaling
 *             (what the unfused path does: the model hands back a bf16 pixel gradient); D0, D1 multiples of 64
 *   other arguments as vaa_patch_grad_gather; ws >= vaa_patch_embed_grad_ws_bytes(B, ph, pw)
 */
size_t vaa_patch_embed_packed_elems(int D);
int vaa_patch_embed_pack_weights(const uint16_t* wt, int D, uint16_t* packed, void* stream);
size_t vaa_patch_embed_grad_ws_bytes(int B, int ph, int pw);
int vaa_patch_embed_grad_gather(const uint16_t* dy0, int D0, const uint16_t* dy1, int D1, const uint16_t* wp0, const uint16_t* wp1,
                                const float* patch, const int32_t* xy, const float* theta, const uint8_t* keep_bits, int B, int ph, int pw,
                                int geometry, int mask_mode, const float* std6, int round_bf16, float* gpatch, void* ws, size_t ws_bytes,
                                void* stream);
/* the same fed by the tile-major mask of vaa_patch_apply_fwd_tiles (keep_tiles, tile_flags: both required). gpatch == NULL leaves the final
 * fixed-order sum to the caller: the vaa_patch_grad_partials(B) partial tiles [parts][3*ph*pw] f32 then sit at the start of ws. */
int vaa_patch_grad_partials(int B);
int vaa_patch_embed_grad_gather_tiles(const uint16_t* dy0, int D0, const uint16_t* dy1, int D1, const uint16_t* wp0, const uint16_t* wp1,
                                      const float* patch, const int32_t* xy, const float* theta, const uint16_t* keep_tiles,
                                      const uint32_t* tile_flags, int B, int ph, int pw, int geometry, int mask_mode, const float* std6,
                                      int round_bf16, float* gpatch, void* ws, size_t ws_bytes, void* stream);
/* the same with one patch per image (resize_patch=True; packed / pdesc / gpacked as in vaa_patch_grad_gather_multi):
 * ws >= vaa_patch_embed_grad_multi_ws_bytes(B); the resize adjoint then folds gpacked into the base patch's gradient */
size_t vaa_patch_embed_grad_multi_ws_bytes(int B);
int vaa_patch_embed_grad_gather_multi(const uint16_t* dy0, int D0, const uint16_t* dy1, int D1, const uint16_t* wp0, const uint16_t* wp1,
                                      const float* packed, const int32_t* pdesc, const int32_t* xy, const float* theta,
                                      const uint8_t* keep_bits, int B, int max_h, int max_w, int geometry, int mask_mode, const float* std6,
                                      int round_bf16, float* gpacked, void* ws, size_t ws_bytes, void* stream);
/* ... fed by the tile-major mask of vaa_patch_apply_fwd_tiles (called with pdesc) */
int vaa_patch_embed_grad_gather_multi_tiles(const uint16_t* dy0, int D0, const uint16_t* dy1, int D1, const uint16_t* wp0, const uint16_t* wp1,
                                            const float* packed, const int32_t* pdesc, const int32_t* xy, const float* theta,
                                            const uint16_t* keep_tiles, const uint32_t* tile_flags, int B, int max_h, int max_w, int geometry,
                                            int mask_mode, const float* std6, int round_bf16, float* gpacked, void* ws, size_t ws_bytes, void* stream);

/*
 * K4 — replaces transformers.AdamW.step + `patch.data.clamp(0,1)` + zero_grad (UADA.py:155-157; UADA_ddp.py:208-209),
 * optional `clip_grad_norm_([patch], l1_clip, norm_type=1)` (UPA.py:157) and the PGD sign step (TMA.py:171-175).
 *   patch,m,v dev [n] float32 in place; g dev [n] float32 (sum over ranks when grad_scale = 1/world, DDP mean); g must not alias
 *            patch, m or v (VAA_E_INVALID): several workgroups read the whole gradient while others already write their elements
 *   step     1-based Adam step count t; lr already multiplied by the cosine schedule (UADA.py:109-115,162-164)
 *   stats    dev [2] float32 out or NULL: {sum|g*grad_scale|, mean(g*grad_scale)} (the logged TRAIN_patch_gradient)
 */
int vaa_patch_update(float* patch, const float* g, float* m, float* v, int n, int mode, float lr, float beta1, float beta2,
                     float eps, int step, float l1_clip, float grad_scale, float* stats, void* stream);

/*
 * Eval-time paste — replaces RandomPatchTransform.simulation_random_patch
 * (appply_random_transform.py:43-78), called per simulator frame by the LIBERO evaluation
 * (experiments/robot/libero/run_libero_eval_args_geo_batch.py:207): patch quantised to uint8 (ToPILImage), fixed
 * rotation+shear warp when geometry[b] != 0, composite where canvas >= 0, uint8 HWC in and out. Batched over B frames.
 *   img_u8, out_u8 dev [B,224,224,3] uint8;  patch dev [3,ph,pw] float32 in [0,1];  xy dev [B,2] int32 (position);
 *   theta dev [B,6] float32 (rows 0-1 of shear_matrix(shx,shy) @ rotation_matrix(angle), :69-74);  geometry dev [B] int32
 */
int vaa_patch_apply_eval(const uint8_t* img_u8, const float* patch, const int32_t* xy, const float* theta, const int32_t* geometry,
                         int B, int ph, int pw, uint8_t* out_u8, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VAA_H_ */
