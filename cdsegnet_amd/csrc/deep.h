// Deep-stage (C = 128 / 256) Block head and tail: internal entry points of csrc/deep.hip, reached through the public
// cdseg_block_rr_pack / cdseg_cpe_head_rr / cdseg_attn_tail_rr (csrc/blockrr.hip dispatches on the channel count).
#pragma once
#include "common.h"

bool deep_supported(int channels);
int deep_pack(int channels, const void* wl, const void* wqkv, void* head_img, const void* wp, const void* w1, const void* w2,
              void* tail_img, hipStream_t s);
// x / x_in: the residual rows read; x_out / x: the rows written (may alias the rows read: the in-place form, which never splits
// the head).  ws: optional fp32 workspace (16-byte aligned) that allows the tail's few-row hidden-chunk split.
int deep_head(const void* y, int ldy, const void* head_img, const float* bl, const float* lnp_g, const float* lnp_b,
              const float* x, int ldx, float* x_out, int ldxo, const float* colbias, const float* ln1_g, const float* ln1_b,
              float eps, const float* bqkv, void* qkv, int ldqkv, long n, int channels, int qkv_flags, hipStream_t s,
              // y given as `ysplits` (> 1) raw split-K partial planes (ysplits, n, channels) fp32 + the conv's bias instead of the
              // finished 16-bit rows (gemm_leave_partials): summed in slice order, bias added, rounded - the tile load does what
              // the conv's second pass would have
              const float* ypart = nullptr, int ysplits = 1, const float* ybias = nullptr);
struct cdseg_gemm_args;
int gemm_leave_partials(const cdseg_gemm_args* a, int* splits, void* stream);  // csrc/gemm.hip
int deep_tail(const void* o, int ldo, const void* tail_img, const float* bp, const float* ln_g, const float* ln_b, float eps,
              const float* b1, const float* b2, const float* x_in, int ldxi, float* x, int ldx, void* xc, int ldxc, long n,
              int channels, void* ws, size_t ws_bytes, hipStream_t s);
