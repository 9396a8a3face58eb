/*
 * vaa_model_ops.h — OPTIONAL fused elementwise operators for the PyTorch-ROCm model that surrounds the hot path
 * (same libvaa_hip.so, same conventions as vaa.h). They are NOT part of the reference's interface nor of the SURVEY.md
 * section-8 contract; the model runs without them (VAA_NO_FUSED_MODEL_OPS=1).
 */
#ifndef VAA_MODEL_OPS_H_
#define VAA_MODEL_OPS_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* HF rotate_half rotary embedding on bf16 x[B,T,H,hd] (element strides sb,st,sh; last dim contiguous) -> contiguous out.
 * cos_t/sin_t: float32 [T,hd/2]. sin_sign = +1 forward, -1 backward (the adjoint is the inverse rotation). */
int vaa_model_rope(const uint16_t* x, long sb, long st, long sh, const float* cos_t, const float* sin_t, int B, int T, int H, int hd,
                   float sin_sign, uint16_t* out, void* stream);
/* y = silu(gate) * up over n contiguous bf16 elements (n % 8 == 0), and its backward. */
int vaa_model_swiglu_fwd(const uint16_t* gate, const uint16_t* up, uint16_t* y, long n, void* stream);
int vaa_model_swiglu_bwd(const uint16_t* dy, const uint16_t* gate, const uint16_t* up, uint16_t* dgate, uint16_t* dup, long n, void* stream);
/* LayerScale + residual: out[r, :] = x[r, :] + a[r, :] * ls (ls bf16 [D], D % 8 == 0; x may be NULL: out = a * ls). */
int vaa_model_scale_add(const uint16_t* x, const uint16_t* a, const uint16_t* ls, uint16_t* out, long rows, int D, void* stream);
/* RMSNorm over rows of D bf16 (weight bf16 [D], HF LlamaRMSNorm rounding) saving rstd float32 [rows]; the backward adds the
 * residual pass-through gradient gpass (may be NULL): gx = gpass + rstd*(gh*w - xhat*mean(gh*w*xhat)). */
int vaa_model_rmsnorm_fwd(const uint16_t* x, const uint16_t* w, uint16_t* h, float* rstd, long rows, int D, float eps, void* stream);
int vaa_model_rmsnorm_bwd(const uint16_t* gh, const uint16_t* gpass, const uint16_t* x, const uint16_t* w, const float* rstd, uint16_t* gx,
                          long rows, int D, void* stream);
/* LayerNorm over rows of D bf16 (weight, bias bf16 [D]; fp32 statistics, one rounding) saving {mean, rstd} float32 [rows,2]; the
 * backward adds the residual pass-through gradient gpass (may be NULL). Weight and bias are frozen (no gradients). */
int vaa_model_layernorm_fwd(const uint16_t* x, const uint16_t* w, const uint16_t* b, uint16_t* h, float* stats, long rows, int D, float eps,
                            void* stream);
int vaa_model_layernorm_bwd(const uint16_t* gh, const uint16_t* gpass, const uint16_t* x, const uint16_t* w, const float* stats, uint16_t* gx,
                            long rows, int D, void* stream);
/* Softmax attention on the matrix cores for short sequences. q,k,v,o: bf16 views [B,T,H,hd] given by element strides
 * {batch, token, head} (last dim contiguous, all strides % 8 == 0), hd % 8 == 0, hd <= 128. lse: float32 [B,H,T] (natural log).
 * causal != 0: query t sees keys <= t.
 * cu_seqlens (int32 [B+1], may be NULL): the B sequences are PACKED back to back along the token axis (the batch stride is unused),
 * sample b owning tokens cu[b] .. cu[b+1]-1; T is then the maximum length (lse and dsum stay [B,H,T]). */
int vaa_model_attention_fwd(const uint16_t* q, const int64_t* q_str, const uint16_t* k, const int64_t* k_str, const uint16_t* v,
                            const int64_t* v_str, uint16_t* o, const int64_t* o_str, float* lse, const int32_t* cu_seqlens, int B, int H,
                            int T, int hd, int causal, float scale, void* stream);
/* Backward of vaa_model_attention_fwd: o, lse from the forward; dsum: float32 [B,H,T] workspace; dq/dk/dv: bf16 [B,T,H,hd] views
 * given by strides (e.g. the three slices of one packed [B,T,3,H,hd] gradient buffer). Two launches (dq, then dk/dv).
 * rope_cos/rope_sin (both or neither; float32 [T,hd/2], hd in {64,128}): q and k had vaa_model_rope applied before the forward;
 * dq and dk are then returned w.r.t. the un-rotated tensors (the adjoint rotation runs in the kernels' epilogues); with cu_seqlens the
 * tables are indexed by the PACKED token (one row per token, i.e. already gathered by position id). */
int vaa_model_attention_bwd(const uint16_t* q, const int64_t* q_str, const uint16_t* k, const int64_t* k_str, const uint16_t* v,
                            const int64_t* v_str, const uint16_t* o, const int64_t* o_str, const uint16_t* dout, const int64_t* do_str,
                            const float* lse, float* dsum, uint16_t* dq, const int64_t* dq_str, uint16_t* dk, const int64_t* dk_str,
                            uint16_t* dv, const int64_t* dv_str, const float* rope_cos, const float* rope_sin, const int32_t* cu_seqlens, int B,
                            int H, int T, int hd, int causal, float scale, void* stream);
#ifdef __cplusplus
}
#endif
#endif
