// Internal interface of conv_bf3.hip: the implicit-GEMM convolution on bf16 MFMA with fp32-equivalent split products.
#pragma once
#include "conv_plan.h"

// Layout rule of a bf16x3 phase: Cp = roundup(C, 8), Kp = roundup(T*Cp, 32), Mp = roundup(M, 64 or 128).
void bf3_finish_phase(ConvPhase& g);
int bf3_pack_phase(const ConvPhase& g, const float* w, float* wp, hipStream_t st);
// workspace = channels-last bf16 hi/lo copy of the activation operand (N*S*roundup(C,8)*4 bytes)
long bf3_workspace_bytes(int N, int C, long S);
// yact/act/slope/dbias (optional): write the planes of x * act'(yact) and add its per-channel sums to dbias
// dhead / head_w / CO: optional data gradient of a 1x1 head on the same tensor, added on the fly (x may then be NULL);
// dhead_w / dhead_b: the head's weight / bias gradient accumulated in the same pass (needs yact = the head's input)
int bf3_split_input(const float* x, void* ws, int N, int C, long S, hipStream_t st, const float* yact = nullptr, int act = 0,
                    float slope = 0.f, float* dbias = nullptr, const float* dhead = nullptr, const float* head_w = nullptr, int CO = 0,
                    float* dhead_w = nullptr, float* dhead_b = nullptr);
int bf3_launch_fwd_phase(const ConvPhase& g, const void* ws, const float* wp, const float* bias, float* out, int act,
                         float slope, hipStream_t st, int ksplit = 1);
// fused 1x1 head of the NEXT forward launches on the eight-wave tiles (thread-local; co = 0 clears it): logits = b + W y, W [co][Msub]
void bf3_set_fused_head(const float* w, const float* b, float* out, int co);
void bf3_set_products(int n);   // 3: bf16x3 split products (default); 1: hi * hi only (plain bf16 arithmetic)
int bf3_get_products();
// split-K factor the caller should use for this phase (1: none; > 1: zero the output first, finish bias / activation after)
int bf3_fwd_ksplit(const ConvPhase& g);
// weight gradient of one forward-form phase on the split planes (ws_x: planes of x with Cin_total channels, ws_dz: planes
// of dy with Cout_total channels); wg = zeroed scratch, g.wp_off = float offset of the phase's [T][M][C] slab in it
int bf3_wgrad_phase(const ConvPhase& g, const void* ws_x, int Cin_total, const void* ws_dz, int Cout_total, float* wg,
                    float* dw, hipStream_t st);
// token-major fp32 [rows][C] (C % 8 == 0) -> the same split planes ((rows * C * 4 + 16) bytes)
int bf3_split_rows(const float* x, void* ws, long rows, int C, hipStream_t st);
// tile class of a bf16x3 phase: eight-wave ping-pong tiles (true) or the four-wave small tiles (false)
bool bf3_fwd_uses_pp(const ConvPhase& g);
bool bf3_wgrad_uses_pp(const ConvPhase& g);
// one entry of the batched pack table (muvo_pack_table_*): a phase, its PyTorch-layout source and its packed destination
struct PackItem {
  ConvPhase g;
  const float* w;
  float* dst;
  long blk0;      // first workgroup of this item
  int nblk, pad_;
};
int pack_table_launch(const PackItem* dev_items, int n, long n_blocks, hipStream_t st);
