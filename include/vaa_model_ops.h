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
#ifdef __cplusplus
}
#endif
#endif
