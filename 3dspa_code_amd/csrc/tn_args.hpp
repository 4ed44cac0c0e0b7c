// tn_args.hpp -- arguments of the dW (TN) kernels, shared by gemm_fast.hip (8-wave kernels, dispatch) and gemm_tnb.hip (large register tile).
#pragma once
#include "common.hpp"

namespace SPA_NS {

// C[Ki][N] += sum_m A[m][Ki] * B[m][N]   (f32 C, split over m with atomics)
struct TnArgs {
  const bf16_t* A; const bf16_t* B; float* C; const bf16_t* zero;
  int64_t M; int Ki; int N; int64_t lda, ldb, ldc;
  int tiles_i, tiles_n, splits; int64_t rows_per_split;
  int brow_group, brow_skip;
  float* colsum;  // 8-phase / large-tile kernels only: colsum[n] += sum_m B[m][n] (the bias gradient), nullptr = off
  // 8-phase / large-tile kernels only: the N output columns are seg_n-wide segments that live in different buffers (the q / k / v kernels of a fused
  // projection are separate leaves): columns [s seg_n, (s+1) seg_n) go to Cseg[s - 1] for s >= 1, row stride ldc in each.  0 = one buffer.
  int seg_n; float* Cseg[2];
};

// large-register-tile kernel (gemm_tnb.hip): true when the shape is covered (Ki % 384 == 0 and N % 256 == 0, or Ki % 256 == 0 and N % 384 == 0;
// M >= 256) and the launches were issued.  Handles M % 32 rows with a small tail launch.
bool gemm_tnb(spa3d_ctx* c, TnArgs g);

}  // namespace SPA_NS
