// Internal interface of conv_vox.hip (small-channel 3x3x3 Conv3d on the 4x4x1 MFMA); called from the C ABI
// entry points in conv_gemm.hip.
#pragma once
#include "common.h"

bool vox_fwd_applicable(const muvo_conv_desc* d);
bool vox_dgrad_applicable(const muvo_conv_desc* d);
bool vox_wgrad_applicable(const muvo_conv_desc* d);
long vox_pack_floats(const muvo_conv_desc* d);
bool vox_bf3_shape_ok(const muvo_conv_desc* d, int dgrad);   // bf16x3 variant (16x16x32 MFMA) available for this direction
int vox_pack(const muvo_conv_desc* d, const float* w, float* wp, int dgrad, hipStream_t st, bool bf3);
int vox_forward(const muvo_conv_desc* d, const float* x, const float* wp, const float* bias, float* y, int act, float slope,
                hipStream_t st, bool bf3, double* moments = nullptr, const float* aff = nullptr);
bool vox_affine_ok(const muvo_conv_desc* d);   // forward / weight gradient can apply a per-(n, channel) scale / shift while staging
int vox_dgrad(const muvo_conv_desc* d, const float* dy, const float* wp, float* dx, hipStream_t st, bool bf3);
bool vox_bf3_wgrad_shape_ok(const muvo_conv_desc* d);
bool vox_wgrad_ps_only(const muvo_conv_desc* d);      // shapes only the plane-streaming bf16x3 weight gradient serves (Z = 16, 32 produced channels)
int vox_wgrad(const muvo_conv_desc* d, const float* x, const float* dz, float* dw, float* dbias, hipStream_t st, bool bf3,
              const float* aff = nullptr);
