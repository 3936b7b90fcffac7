// Internal interface of conv_pw.hip (1x1 convolutions with <= 4 output channels: the decoder heads).
// The "packed" weight of this path is the PyTorch weight itself ([Cout][Cin] floats).
#pragma once
#include "common.h"

bool pw_applicable(const muvo_conv_desc* d);
int pw_forward(const muvo_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int act, float slope,
               hipStream_t st);
int pw_dgrad(const muvo_conv_desc* d, const float* dy, const float* w, float* dx, hipStream_t st);
int pw_dgrad_acc(const muvo_conv_desc* d, const float* dy, const float* w, float* dx, hipStream_t st);
int pw_wgrad(const muvo_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias, hipStream_t st);
