// gemm_fast.hip -- LDS-tiled bf16 MFMA GEMMs for the hot shapes of the 3DSPA train step (gfx950).
//
// gemm_nt_bf16 :  C[M,N] (op)= epi(A[M,K] . Bt[N,K]^T + bias)            (Y = X.W and dX = dY.W^T; 2/3 of all FLOPs)
//   128x128x64 tile, 256 threads = 4 waves (2x2), 64x64 per wave as 4x4 v_mfma_f32_16x16x32_bf16,
//   operands staged HBM->LDS with global_load_lds_dwordx4 (no VGPR round trip), two LDS buffers, the next
//   K-tile in flight under the current tile's MFMAs.  LDS rows are 128 B; the 16-B chunk index is XORed with
//   (row & 7) -- on the SOURCE address, because LDS-DMA writes lane-linear (cdna_hip_programming.md rule 21) --
//   which makes every ds_read_b128 fragment read conflict-free.  MFMA operands are swapped (D = Bt.A^T) so a lane
//   ends up with 4 consecutive columns of one output row: 8-B bf16 / 16-B f32 stores, vector bias/residual loads.
//   Workgroup ids are remapped so all N-tiles of one 128-row A panel run on one XCD (its L2 then serves the panel).
//
// gemm_tn_bf16 :  C[Ki,N] += A[M,Ki]^T . B[M,N]   (dW = X^T.dY; 1/3 of all FLOPs; reduction over the huge M)
//   128x128 output tile, 64 reduction rows per LDS tile, v_mfma_f32_32x32x16_bf16; both operands are read from
//   row-major [m][*] LDS images with ds_read_b64_tr_b16 (hardware transpose), images XOR-swizzled per
//   cdna_hip_programming.md T10 layout (b).  M is split across workgroups; partial tiles are added with
//   global_atomic_add_f32 -- one accumulator register of a 32x32 tile is two 128-B row segments per
//   wave-instruction, the full-rate atomic shape (MI355X_MICROARCH.md "Global float atomics").
#include "common.hpp"
#include "tn_args.hpp"

namespace SPA_NS {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef mfma16x8 bf16x8;  // 8 activation elements: the MFMA A/B operand of one lane
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;

#define GLDS16(gptr, lptr) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)
// same with the non-temporal cache policy (aux bit 1): an operand that is read exactly once
#define GLDS16_NT(gptr, lptr) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lptr), 16, 0, 2)

// LDS-DMA with a wave-uniform 64-bit base in SGPRs and a 32-bit lane offset: keeps per-lane addressing at one VGPR per load
// (the builtin form is free to widen the offsets to 64 bits, which in the persistent kernel spilled them into the K-loop)
__device__ __forceinline__ void glds16_s(const void* base_uniform, unsigned off, const void* lds_dst) {
  const unsigned l = (unsigned)(uintptr_t)lds_dst;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base_uniform), "s"(l) : "memory", "m0");
}

struct NtArgs {
  const bf16_t* A; const bf16_t* Bt; void* C;
  int64_t M; int N; int K; int64_t lda, ldb, ldc;
  const float* bias; const bf16_t* aux; bf16_t* pre_out; int epi; int out_f32; int accumulate; float alpha;
  int tiles_m, tiles_n, crow_group, crow_skip;
#if SPA3D_ABL_NT
  int ablate;
#endif
  // one-pass input embedding (non-persistent 128x384 8-phase kernel only; GemmDesc::A2 ...): K columns [0, K1) come from A, [K1, K) from A2 (row
  // stride lda2); input rows are gathered through arow_idx (both sources), output rows scattered through crow_idx (< 0 = dropped); the
  // epilogue adds the rank-1 term r1_x[input row] * r1_w[column] in f32
  const bf16_t* A2; int64_t lda2; int K1; const int32_t* arow_idx; const int32_t* crow_idx; const bf16_t* r1_x; const float* r1_w;
  int nt_store;  // bf16 output with non-temporal stores: a streamed output far larger than the caches (+3-6 % measured at K = 384)
};

__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(NtArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A 16 KiB | B 16 KiB]
  // XCD-aware id: blocks b, b+8, b+16.. share an XCD; give each XCD whole A panels (all N-tiles of an M-tile)
  const int bid = blockIdx.x;
  const int xcd = bid & 7, idx = bid >> 3;
  const int tm = (idx / g.tiles_n) * 8 + xcd, tn = idx % g.tiles_n;
  if (tm >= g.tiles_m) return;
  const int64_t m0 = (int64_t)tm * 128;
  const int n0 = tn * 128;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  const int fr = lane & 15, fq = lane >> 4;

  // ---- staging addresses: wave w fills 8-row groups 4w..4w+3 of each operand tile; lane -> (row r = lane>>3, chunk cp = lane&7)
  const int sr = lane >> 3, scp = lane & 7;
  const int sc = (scp ^ sr) * 8;  // source chunk (elements) that lands in physical chunk scp of a row with (row&7)==sr
  const bf16_t* ga[4]; const bf16_t* gb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = (w * 4 + i) * 8 + sr;
    int64_t am = m0 + row; if (am > g.M - 1) am = g.M - 1;
    int bn = n0 + row; if (bn > g.N - 1) bn = g.N - 1;
    ga[i] = g.A + am * g.lda + sc;
    gb[i] = g.Bt + (int64_t)bn * g.ldb + sc;
  }
  auto stage = [&](int buf, int kt) {
    char* sa = smem + buf * 32768 + (w * 4) * 1024;
    char* sb = sa + 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) GLDS16(ga[i] + kt * 64, sa + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) GLDS16(gb[i] + kt * 64, sb + i * 1024);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nt = g.K / 64;
  stage(0, 0);
  __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and publishes tile 0
  // fragment byte offsets inside a tile: row*128 + ((ks*4 + fq) ^ (row&7))*16, row&7 == fr&7
  const int a_off = (wm + fr) * 128, b_off = (wn + fr) * 128;
  const int x0 = ((fq) ^ (fr & 7)) * 16, x1 = ((4 + fq) ^ (fr & 7)) * 16;
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) stage(cur ^ 1, t + 1);  // next tile in flight under this tile's MFMAs
    const char* sa = smem + cur * 32768;
    const char* sb = sa + 16384;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int xo = ks ? x1 : x0;
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(sa + a_off + i * 2048 + xo);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(sb + b_off + j * 2048 + xo);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = MFMA16(bfr[j], af[i], acc[i][j]);
    }
    __syncthreads();  // next tile landed (vmcnt(0)) and everyone is done reading `cur`
  }

  // ---- epilogue.  A lane holds C[m = wm+16i+fr][n = wn+16j+4fq .. +3]; the tile goes through LDS (f32, 16-B chunks XORed
  // with the row so neither side conflicts badly) and leaves as whole 256-B row segments: 16-byte coalesced bias /
  // residual / pre-activation / output accesses instead of 8-byte pieces scattered over 16 rows per instruction.
  float* cs = (float*)smem;  // [128][128] f32 = the 64 KiB the K-loop no longer needs (last barrier already passed)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wm + i * 16 + fr, ch = (wn >> 2) + j * 4 + fq;
      *(f32x4*)(cs + row * 128 + ((ch ^ (row & 31)) << 2)) = acc[i][j];
    }
  __syncthreads();
  const int er = tid >> 4, ec = tid & 15;
  const int gn = n0 + ec * 8;
  if (gn >= g.N) return;  // N % 8 == 0: an 8-column group is in or out as a whole
  float b8[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) b8[r] = 0.f;
  if (g.bias) { const float4 b0 = *(const float4*)(g.bias + gn), b1 = *(const float4*)(g.bias + gn + 4);
    b8[0] = b0.x; b8[1] = b0.y; b8[2] = b0.z; b8[3] = b0.w; b8[4] = b1.x; b8[5] = b1.y; b8[6] = b1.z; b8[7] = b1.w; }
#pragma unroll 2
  for (int it = 0; it < 8; ++it) {
    const int row = it * 16 + er;
    const int64_t gm = m0 + row;
    if (gm >= g.M) break;
    int64_t crow = gm;
    if (g.crow_group > 0) crow = gm + (gm / g.crow_group + 1) * (int64_t)g.crow_skip;
    const f32x4 v0 = *(const f32x4*)(cs + row * 128 + (((2 * ec) ^ (row & 31)) << 2));
    const f32x4 v1 = *(const f32x4*)(cs + row * 128 + (((2 * ec + 1) ^ (row & 31)) << 2));
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = g.alpha * v[r] + b8[r];
    const int64_t ci = crow * g.ldc + gn;
    if (g.pre_out) {
      uint4 p4; unsigned* pp = (unsigned*)&p4;
#pragma unroll
      for (int r = 0; r < 4; ++r) pp[r] = f2bf_pack2(v[2 * r], v[2 * r + 1]);
      *(uint4*)(g.pre_out + ci) = p4;
    }
    if (g.epi == EPI_GELU) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = gelu_tanh_fast_f(v[r]);
    }
    if (g.aux) {
      const uint4 x4 = *(const uint4*)(g.aux + ci); const unsigned* xp = (const unsigned*)&x4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float xa = unpack_lo(xp[r]), xb = unpack_hi(xp[r]);
        if (g.epi == EPI_MUL_GELU_GRAD) { v[2 * r] *= gelu_tanh_grad_fast_f(xa); v[2 * r + 1] *= gelu_tanh_grad_fast_f(xb); }
        else { v[2 * r] += xa; v[2 * r + 1] += xb; }
      }
    }
    if (g.out_f32) {
      float4* cp = (float4*)((float*)g.C + ci);
      if (g.accumulate) { const float4 o0 = cp[0], o1 = cp[1];
        v[0] += o0.x; v[1] += o0.y; v[2] += o0.z; v[3] += o0.w; v[4] += o1.x; v[5] += o1.y; v[6] += o1.z; v[7] += o1.w; }
      cp[0] = make_float4(v[0], v[1], v[2], v[3]); cp[1] = make_float4(v[4], v[5], v[6], v[7]);
    } else {
      uint4* cp = (uint4*)((bf16_t*)g.C + ci);
      if (g.accumulate) { const uint4 o4 = *cp; const unsigned* op = (const unsigned*)&o4;
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[2 * r] += unpack_lo(op[r]); v[2 * r + 1] += unpack_hi(op[r]); } }
      uint4 o4; unsigned* op = (unsigned*)&o4;
#pragma unroll
      for (int r = 0; r < 4; ++r) op[r] = f2bf_pack2(v[2 * r], v[2 * r + 1]);
      *cp = o4;
    }
  }
}


// =================================================================================================================
// High-occupancy variant for short K: ONE 32-KiB LDS buffer, direct register epilogue, 4 workgroups per CU.  No
// intra-workgroup overlap (stage -> barrier -> MFMA -> barrier), instead four workgroups per CU interleave: one
// block's first-tile latency and store tail hide under the others' K loops.
// =================================================================================================================
__global__ __launch_bounds__(256, 4) void gemm_nt_occ_kernel(NtArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [A 16 KiB | B 16 KiB]
  const int bid = blockIdx.x;
  const int xcd = bid & 7, idx = bid >> 3;
  const int tm = (idx / g.tiles_n) * 8 + xcd, tn = idx % g.tiles_n;
  if (tm >= g.tiles_m) return;
  const int64_t m0 = (int64_t)tm * 128;
  const int n0 = tn * 128;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  const int fr = lane & 15, fq = lane >> 4;
  const int sr = lane >> 3, scp = lane & 7;
  const int sc = (scp ^ sr) * 8;
  const bf16_t* ga[4]; const bf16_t* gb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = (w * 4 + i) * 8 + sr;
    int64_t am = m0 + row; if (am > g.M - 1) am = g.M - 1;
    int bn = n0 + row; if (bn > g.N - 1) bn = g.N - 1;
    ga[i] = g.A + am * g.lda + sc;
    gb[i] = g.Bt + (int64_t)bn * g.ldb + sc;
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nt = g.K / 64;
  const int a_off = (wm + fr) * 128, b_off = (wn + fr) * 128;
  const int x0 = ((fq) ^ (fr & 7)) * 16, x1 = ((4 + fq) ^ (fr & 7)) * 16;
  char* const sa0 = smem + (w * 4) * 1024; char* const sb0 = sa0 + 16384;
  for (int t = 0; t < nt; ++t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) GLDS16(ga[i] + t * 64, sa0 + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) GLDS16(gb[i] + t * 64, sb0 + i * 1024);
    __syncthreads();
    const char* sa = smem; const char* sb = smem + 16384;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int xo = ks ? x1 : x0;
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(sa + a_off + i * 2048 + xo);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(sb + b_off + j * 2048 + xo);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = MFMA16(bfr[j], af[i], acc[i][j]);
    }
    if (t + 1 < nt) __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t gm = m0 + wm + i * 16 + fr;
    int64_t crow = gm;
    if (g.crow_group > 0) crow = gm + (gm / g.crow_group + 1) * (int64_t)g.crow_skip;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gn = n0 + wn + j * 16 + fq * 4;
      if (gm >= g.M || gn >= g.N) continue;
      float vv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) vv[r] = g.alpha * acc[i][j][r];
      if (g.bias) { const float4 b4 = *(const float4*)(g.bias + gn); vv[0] += b4.x; vv[1] += b4.y; vv[2] += b4.z; vv[3] += b4.w; }
      const int64_t ci = crow * g.ldc + gn;
      if (g.pre_out) { u16x4 p4;
#pragma unroll
        for (int r = 0; r < 4; ++r) p4[r] = f2bf(vv[r]);
        *(u16x4*)(g.pre_out + ci) = p4; }
      if (g.epi == EPI_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) vv[r] = gelu_tanh_fast_f(vv[r]);
      }
      if (g.aux) {
        const u16x4 x4 = *(const u16x4*)(g.aux + ci);
        if (g.epi == EPI_MUL_GELU_GRAD) {
#pragma unroll
          for (int r = 0; r < 4; ++r) vv[r] *= gelu_tanh_grad_fast_f(bf2f(x4[r]));
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) vv[r] += bf2f(x4[r]);
        }
      }
      if (g.out_f32) {
        float4* cp = (float4*)((float*)g.C + ci);
        if (g.accumulate) { const float4 o = *cp; vv[0] += o.x; vv[1] += o.y; vv[2] += o.z; vv[3] += o.w; }
        *cp = make_float4(vv[0], vv[1], vv[2], vv[3]);
      } else {
        u16x4* cp = (u16x4*)((bf16_t*)g.C + ci);
        if (g.accumulate) { const u16x4 o = *cp;
#pragma unroll
          for (int r = 0; r < 4; ++r) vv[r] += bf2f(o[r]); }
        u16x4 o4;
#pragma unroll
        for (int r = 0; r < 4; ++r) o4[r] = f2bf(vv[r]);
        *cp = o4;
      }
    }
  }
}


// =================================================================================================================
// 256x256x64, 8 waves, 8-phase schedule (cdna_hip_programming.md "The 256^2 8-phase template"), written for this
// library's operand layout (A [M][K], Bt [N][K], XOR-swizzled 128-B LDS rows, swapped-operand MFMA).
//   waves 2(M) x 4(N), 128x64 per wave; a K-tile is four half-tiles of 16 KiB, split by NEED, not by position:
//     A-h0 / A-h1 = the first / second 64 rows of each wave-row's 128;  B-h0 / B-h1 = the first / second 32 columns of each
//     wave-column's 64.  Phase p of K-tile t works on one quadrant of the wave's output:
//       P0: read B-q0 (4 ds_read_b128, retired before the barrier), A-q0 (8);  16 MFMA  (rows 0-63,   cols 0-31)
//       P1: read B-q1 (4);                                                      16 MFMA  (rows 0-63,   cols 32-63)
//       P2: read A-q1 (8);                                                      16 MFMA  (rows 64-127, cols 32-63)
//       P3: --                                                                  16 MFMA  (rows 64-127, cols 0-31)
//     LDS: A in a 3-slot ring, B double-buffered (160 KiB / 144 KiB).  Everything issued during K-tile t is for K-tile t+2:
//       P0: A-h0(t+2), A-h1(t+2) into the ring slot last read in K-tile t-1;  P1: B-h0(t+2) (its slot's last reads retired before
//       P0's barrier);  P3: B-h1(t+2) (last read in P1).  The A panel is the operand streamed from HBM: it gets a full two
//       K-tiles of lead; the weights come from L2.
//   One counted wait per K-tile, in P3 before its first barrier: vmcnt(#LDS-DMA of one K-tile) leaves K-tile t+2 in flight and
//   retires all of K-tile t+1, which is first read one phase later.  Two raw s_barrier per phase; the wave-row-1 waves run
//   one barrier behind, so on every SIMD one wave is in its MFMA section while the other is in its read/stage section.
// =================================================================================================================
// 8 consecutive columns of one output row: alpha/bias, optional pre-activation copy, GELU, residual / GELU-gradient, store.
template <bool AUXPRE = false, bool REMAP = true>
__device__ __forceinline__ void nt_store8(const NtArgs& g, int64_t gm, int gn, float (&v)[8], const float (&b8)[8], uint4 auxpre = uint4{0, 0, 0, 0}) {
  int64_t crow = gm;
  if (REMAP && g.crow_group > 0) crow = gm + (gm / g.crow_group + 1) * (int64_t)g.crow_skip;
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = g.alpha * v[r] + b8[r];
  const int64_t ci = crow * g.ldc + gn;
  if (g.pre_out) {
    uint4 p4; unsigned* pp = (unsigned*)&p4;
#pragma unroll
    for (int r = 0; r < 4; ++r) pp[r] = f2bf_pack2(v[2 * r], v[2 * r + 1]);
    *(uint4*)(g.pre_out + ci) = p4;
  }
  if (g.epi == EPI_GELU) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = gelu_tanh_fast_f(v[r]);
  }
  if (g.aux) {
    const uint4 x4 = AUXPRE ? auxpre : *(const uint4*)(g.aux + ci); const unsigned* xp = (const unsigned*)&x4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float xa = unpack_lo(xp[r]), xb = unpack_hi(xp[r]);
      if (g.epi == EPI_MUL_GELU_GRAD) { v[2 * r] *= gelu_tanh_grad_fast_f(xa); v[2 * r + 1] *= gelu_tanh_grad_fast_f(xb); }
      else { v[2 * r] += xa; v[2 * r + 1] += xb; }
    }
  }
  if (g.out_f32) {
    float4* cp = (float4*)((float*)g.C + ci);
    if (g.accumulate) { const float4 o0 = cp[0], o1 = cp[1];
      v[0] += o0.x; v[1] += o0.y; v[2] += o0.z; v[3] += o0.w; v[4] += o1.x; v[5] += o1.y; v[6] += o1.z; v[7] += o1.w; }
    cp[0] = make_float4(v[0], v[1], v[2], v[3]); cp[1] = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    uint4* cp = (uint4*)((bf16_t*)g.C + ci);
    if (g.accumulate) { const uint4 o4 = *cp; const unsigned* op = (const unsigned*)&o4;
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[2 * r] += unpack_lo(op[r]); v[2 * r + 1] += unpack_hi(op[r]); } }
    uint4 o4; unsigned* op = (unsigned*)&o4;
#pragma unroll
    for (int r = 0; r < 4; ++r) op[r] = f2bf_pack2(v[2 * r], v[2 * r + 1]);
    if (g.nt_store) { typedef __attribute__((ext_vector_type(4))) unsigned u32x4; __builtin_nontemporal_store(u32x4{o4.x, o4.y, o4.z, o4.w}, (u32x4*)cp); }
    else *cp = o4;
  }
}

// epilogue math of one 8-column group with a residual / pre-activation operand, result packed as bf16 (persistent kernel: the
// store happens later, see there)
__device__ __forceinline__ uint4 nt_compute8_aux(const NtArgs& g, float (&v)[8], const float (&b8)[8], const uint4& x4) {
  const unsigned* xp = (const unsigned*)&x4;
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = g.alpha * v[r] + b8[r];
  if (g.epi == EPI_GELU) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = gelu_tanh_fast_f(v[r]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float xa = unpack_lo(xp[r]), xb = unpack_hi(xp[r]);
    if (g.epi == EPI_MUL_GELU_GRAD) { v[2 * r] *= gelu_tanh_grad_fast_f(xa); v[2 * r + 1] *= gelu_tanh_grad_fast_f(xb); }
    else { v[2 * r] += xa; v[2 * r + 1] += xb; }
  }
  uint4 o4; unsigned* op = (unsigned*)&o4;
#pragma unroll
  for (int r = 0; r < 4; ++r) op[r] = f2bf_pack2(v[2 * r], v[2 * r + 1]);
  return o4;
}

__device__ __forceinline__ void nt_store4(const NtArgs& g, int64_t gm, int gn, const f32x4& a) {
  if (gm >= g.M || gn >= g.N) return;
  int64_t crow = gm;
  if (g.crow_group > 0) crow = gm + (gm / g.crow_group + 1) * (int64_t)g.crow_skip;
  float vv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) vv[r] = g.alpha * a[r];
  if (g.bias) { const float4 b4 = *(const float4*)(g.bias + gn); vv[0] += b4.x; vv[1] += b4.y; vv[2] += b4.z; vv[3] += b4.w; }
  const int64_t ci = crow * g.ldc + gn;
  if (g.pre_out) { u16x4 p4;
#pragma unroll
    for (int r = 0; r < 4; ++r) p4[r] = f2bf(vv[r]);
    *(u16x4*)(g.pre_out + ci) = p4; }
  if (g.epi == EPI_GELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) vv[r] = gelu_tanh_fast_f(vv[r]);
  }
  if (g.aux) {
    const u16x4 x4 = *(const u16x4*)(g.aux + ci);
    if (g.epi == EPI_MUL_GELU_GRAD) {
#pragma unroll
      for (int r = 0; r < 4; ++r) vv[r] *= gelu_tanh_grad_fast_f(bf2f(x4[r]));
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) vv[r] += bf2f(x4[r]);
    }
  }
  if (g.out_f32) {
    float4* cp = (float4*)((float*)g.C + ci);
    if (g.accumulate) { const float4 o = *cp; vv[0] += o.x; vv[1] += o.y; vv[2] += o.z; vv[3] += o.w; }
    *cp = make_float4(vv[0], vv[1], vv[2], vv[3]);
  } else {
    u16x4* cp = (u16x4*)((bf16_t*)g.C + ci);
    if (g.accumulate) { const u16x4 o = *cp;
#pragma unroll
      for (int r = 0; r < 4; ++r) vv[r] += bf2f(o[r]); }
    u16x4 o4;
#pragma unroll
    for (int r = 0; r < 4; ++r) o4[r] = f2bf(vv[r]);
    *cp = o4;
  }
}

// diagnostic library only (tools/ablate_gemm.py, -DSPA3D_ABLATION_BUILD -DSPA3D_ABL_NT=1: csrc/ablate.inc): the persistent kernel without its output stores = loop-only time
#if SPA3D_ABL_NT
#define NT_ABLATE_STORES && !g.ablate
#else
#define NT_ABLATE_STORES
#endif
#define NT8P_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define NT8P_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define NT8P_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")

// WMT x WNT = 16x16 output tiles per wave; waves 2(M) x 4(N); tile = (32 WMT) x (64 WNT): <8,4> = 256x256, <4,6> = 128x384
// (N = 384 in ONE tile: the A panel is read from HBM exactly once).  64 KiB per LDS buffer in both.
// <4,2> = 128x128 on 80 KiB of LDS and <= 128 registers: TWO workgroups per CU, so one's epilogue (the store drain that is purely additive at
// K = 384) can run under the other's K-loop -- the round-3 measurement of that hypothesis (SPA3D_NT_8P=42)
template <int WMT, int WNT, bool COARSE = false, bool EMB = false>  // EMB: the one-pass input-embedding operands of NtArgs are live
__global__ __launch_bounds__(512, (WMT * WNT <= 8 ? 4 : 2)) void gemm_nt8p_kernel(NtArgs g) {
  constexpr int BM = 32 * WMT, BN = 64 * WNT;
  constexpr int NA = WMT / 4, NB = WNT / 2;   // LDS-DMA per thread per A / B half-tile
  constexpr int HM = WMT / 2, HN = WNT / 2;   // tiles per quadrant side
  constexpr int ASLOT = BM * 128, BBUF = BN * 128, BOFF = 3 * ASLOT;  // A ring of 3, then two B buffers
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = blockIdx.x;
  const int xcd = bid & 7, idx = bid >> 3;
  const int tm = (idx / g.tiles_n) * 8 + xcd, tn = idx % g.tiles_n;
  if (tm >= g.tiles_m) return;
  const int64_t m0 = (int64_t)tm * BM;
  const int n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int sr = lane >> 3, scp = lane & 7;
  const int sc = (scp ^ sr) * 8;
  // staging: 8-row group gi = 8i + w of a half-tile.  A-h: wave-row block gi / WMT, group gi % WMT; B-h: wave-column block gi / WNT
  const bf16_t* pa[2][NA]; const bf16_t* pb[2][NB];
  const bf16_t* pa2[2][NA];  // second A source (columns K1 .. K), same rows
  int la[2][NA], lb[2][NB];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int gi = i * 8 + w, ra = (gi / WMT) * (WMT * 16) + h * (WMT * 8) + (gi % WMT) * 8;
      int64_t am = m0 + ra + sr; if (am > g.M - 1) am = g.M - 1;
      if (EMB && g.arow_idx) am = g.arow_idx[am];  // gathered input rows (the rows of a tile are the same for every K-tile: one lookup per lane)
      pa[h][i] = g.A + am * g.lda + sc; la[h][i] = ra * 128;
      pa2[h][i] = (EMB && g.A2) ? g.A2 + am * g.lda2 + sc : nullptr;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int gi = i * 8 + w, rb = (gi / WNT) * (WNT * 16) + h * (WNT * 8) + (gi % WNT) * 8;
      int bn = n0 + rb + sr; if (bn > g.N - 1) bn = g.N - 1;
      pb[h][i] = g.Bt + (int64_t)bn * g.ldb + sc; lb[h][i] = BOFF + rb * 128;
    }
  }
  const bool a_once = g.tiles_n == 1;  // the A panel is read by this workgroup only: stream it past L2
  const int k1t = (EMB && g.A2) ? g.K1 / 64 : 0x7fffffff;  // K-tiles [0, k1t) from A, the rest from A2
  auto stageA = [&](int kt, int slot) {  // both halves of K-tile kt
    char* base = smem + slot * ASLOT;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const bf16_t* src = (!EMB || kt < k1t) ? pa[h][i] + kt * 64 : pa2[h][i] + (kt - k1t) * 64;
        if (a_once) GLDS16_NT(src, base + la[h][i]); else GLDS16(src, base + la[h][i]);
      }
  };
  auto stageB = [&](int h, int kt) {
    char* base = smem + (kt & 1) * BBUF;
#pragma unroll
    for (int i = 0; i < NB; ++i) GLDS16(pb[h][i] + kt * 64, base + lb[h][i]);
  };
  f32x4 acc[WMT][WNT];
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nt = g.K / 64;
  constexpr int NKT = 2 * NA + 2 * NB;  // LDS-DMA per thread per K-tile
  stageA(0, 0); stageB(0, 0); stageB(1, 0);
  if (nt > 1) { stageA(1, 1); stageB(0, 1); stageB(1, 1); NT8P_WAIT_VM(NKT); }
  else NT8P_WAIT_VM(0);
  NT8P_BAR();                 // K-tile 0 is visible to every wave
  if (wr == 1) NT8P_BAR();    // the stagger: wave-row 1 runs one barrier behind
  const int a_off = (wr * WMT * 16 + fr) * 128, b_off = BOFF + (wc * WNT * 16 + fr) * 128;
  const int x0 = ((fq) ^ (fr & 7)) * 16, x1 = ((4 + fq) ^ (fr & 7)) * 16;
  bf16x8 aq[HM][2], bq0[HN][2], bq1[HN][2];
#define NT8P_MFMA(AI, BJ, BQ)                                                                                                         \
  __builtin_amdgcn_s_setprio(1);                                                                                                      \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < HM; ++i) _Pragma("unroll") for (int j = 0; j < HN; ++j) \
      acc[(AI) + i][(BJ) + j] = MFMA16(BQ[j][ks], aq[i][ks], acc[(AI) + i][(BJ) + j]);       \
  __builtin_amdgcn_s_setprio(0);
  int aslot = 0, aslot2 = 2;  // ring slots of K-tiles t and t+2
  for (int t = 0; t < nt; ++t) {
    const char* sa = smem + aslot * ASLOT;
    const char* sb = smem + (t & 1) * BBUF;
    if constexpr (COARSE) {
      // two phases per K-tile (2 HM HN x 2 MFMAs per barrier pair): one wave-row's MFMA section covers the other's reads and their
      // LDS latency.  Every read is retired BEFORE its phase's first barrier, so a slot may be restaged one phase later.
      // ---------------- PA: B-q0, B-q1 (retired first), A-q0 | stage A(t+2) | quadrants (0,0) (0,1)
#pragma unroll
      for (int j = 0; j < HN; ++j) { bq0[j][0] = *(const bf16x8*)(sb + b_off + j * 2048 + x0); bq0[j][1] = *(const bf16x8*)(sb + b_off + j * 2048 + x1); }
#pragma unroll
      for (int j = 0; j < HN; ++j) { bq1[j][0] = *(const bf16x8*)(sb + b_off + (HN + j) * 2048 + x0); bq1[j][1] = *(const bf16x8*)(sb + b_off + (HN + j) * 2048 + x1); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < HM; ++i) { aq[i][0] = *(const bf16x8*)(sa + a_off + i * 2048 + x0); aq[i][1] = *(const bf16x8*)(sa + a_off + i * 2048 + x1); }
      if (t + 2 < nt) stageA(t + 2, aslot2);
      NT8P_WAIT_LGKM(2 * HM);
      NT8P_BAR();
      NT8P_WAIT_LGKM(0);
      NT8P_MFMA(0, 0, bq0)
      NT8P_MFMA(0, HN, bq1)
      NT8P_BAR();
      // ---------------- PB: A-q1 (retired before the barrier) | stage B-h0(t+2), B-h1(t+2) | wait K-tile t+1 | quadrants (1,1) (1,0)
#pragma unroll
      for (int i = 0; i < HM; ++i) { aq[i][0] = *(const bf16x8*)(sa + a_off + (HM + i) * 2048 + x0); aq[i][1] = *(const bf16x8*)(sa + a_off + (HM + i) * 2048 + x1); }
      if (t + 2 < nt) { stageB(0, t + 2); stageB(1, t + 2); NT8P_WAIT_VM(NKT); }
      else NT8P_WAIT_VM(0);
      NT8P_WAIT_LGKM(0);
      NT8P_BAR();
      NT8P_MFMA(HM, HN, bq1)
      NT8P_MFMA(HM, 0, bq0)
      NT8P_BAR();
    } else {
    // ---------------- P0
#pragma unroll
    for (int j = 0; j < HN; ++j) { bq0[j][0] = *(const bf16x8*)(sb + b_off + j * 2048 + x0); bq0[j][1] = *(const bf16x8*)(sb + b_off + j * 2048 + x1); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < HM; ++i) { aq[i][0] = *(const bf16x8*)(sa + a_off + i * 2048 + x0); aq[i][1] = *(const bf16x8*)(sa + a_off + i * 2048 + x1); }
    if (t + 2 < nt) stageA(t + 2, aslot2);
    NT8P_WAIT_LGKM(2 * HM);  // the B-q0 reads have returned: B-h0 may be restaged one phase from now
    NT8P_BAR();
    NT8P_WAIT_LGKM(0);
    NT8P_MFMA(0, 0, bq0)
    NT8P_BAR();
    // ---------------- P1
#pragma unroll
    for (int j = 0; j < HN; ++j) { bq1[j][0] = *(const bf16x8*)(sb + b_off + (HN + j) * 2048 + x0); bq1[j][1] = *(const bf16x8*)(sb + b_off + (HN + j) * 2048 + x1); }
    if (t + 2 < nt) stageB(0, t + 2);
    NT8P_BAR();
    NT8P_WAIT_LGKM(0);
    NT8P_MFMA(0, HN, bq1)
    NT8P_BAR();
    // ---------------- P2
#pragma unroll
    for (int i = 0; i < HM; ++i) { aq[i][0] = *(const bf16x8*)(sa + a_off + (HM + i) * 2048 + x0); aq[i][1] = *(const bf16x8*)(sa + a_off + (HM + i) * 2048 + x1); }
    NT8P_BAR();
    NT8P_WAIT_LGKM(0);
    NT8P_MFMA(HM, HN, bq1)
    NT8P_BAR();
    // ---------------- P3
    if (t + 2 < nt) { stageB(1, t + 2); NT8P_WAIT_VM(NKT); }  // K-tile t+1 has landed (this wave's part); t+2 stays in flight
    else NT8P_WAIT_VM(0);
    NT8P_BAR();
    NT8P_MFMA(HM, 0, bq0)
    NT8P_BAR();
    }
    aslot = aslot == 2 ? 0 : aslot + 1; aslot2 = aslot2 == 2 ? 0 : aslot2 + 1;
  }
#undef NT8P_MFMA
  if (wr == 0) NT8P_BAR();  // pairs with wave-row 1's extra barrier
  // ---- epilogue through wave-private LDS (the K-loop's buffers are dead: every wave has passed the final barrier).
  // Region per wave: [8 WMT rows][4 WNT chunks of 16 B] f32 (16 KiB / 12 KiB), chunk index swizzled with the row so that both the
  // accumulator writes and the row-wise reads spread over the banks.  Two passes (upper / lower half of the wave's rows); read
  // back as 8 columns per lane, rows contiguous: every store instruction writes whole 128-B / 192-B row segments.
  constexpr int RB = WNT * 64;           // row bytes
  constexpr int CPR = WNT * 2;           // 8-column groups per row
  constexpr int EPI_STRIDE = ((3 * BM + 2 * BN) * 128 / 8) & ~1023;  // an eighth of the kernel's LDS per wave (>= its 8 WMT x 64 WNT byte region)
  static_assert(EPI_STRIDE >= 8 * WMT * WNT * 64, "epilogue region");
  char* reg = smem + w * EPI_STRIDE;
  auto swz = [](int chunk, int row) { return WNT == 4 ? (chunk ^ (row & 15)) : WNT == 2 ? (chunk ^ (row & 7)) : ((chunk & ~7) | ((chunk & 7) ^ ((row >> 1) & 7))); };
  constexpr int NIT = WMT * WNT / 4;
  // residual / pre-activation operand: ALL of the tile's loads go out before the first store (vmcnt retires in order: a load issued
  // after the first half's stores would be waited for together with their drain); 2 NIT x 4 VGPRs, the K-loop's fragments are dead
  uint4 auxv[EMB ? 1 : 2][EMB ? 1 : NIT];  // (the embedding form has no aux operand: host)
  if constexpr (!EMB) if (g.aux) {
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int id = it * 64 + lane, row = id / CPR, c8 = id - row * CPR;
        const int64_t gm = m0 + wr * (WMT * 16) + half * (WMT * 8) + row;
        const int gn = n0 + wc * (WNT * 16) + c8 * 8;
        auxv[half][it] = uint4{0, 0, 0, 0};
        if (gm < g.M && gn < g.N) {
          int64_t crow = gm;
          if (g.crow_group > 0) crow = gm + (gm / g.crow_group + 1) * (int64_t)g.crow_skip;
          auxv[half][it] = *(const uint4*)(g.aux + crow * g.ldc + gn);
        }
      }
  }
  // embedding form: output row (scatter map) and the rank-1 operand's value per item, loaded before the first store like the aux operand
  constexpr bool emb = EMB;
  int crow_i[EMB ? 2 : 1][EMB ? NIT : 1]; float r1v[EMB ? 2 : 1][EMB ? NIT : 1];
  if constexpr (EMB) {
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int id = it * 64 + lane, row = id / CPR;
        const int64_t gm = m0 + wr * (WMT * 16) + half * (WMT * 8) + row;
        crow_i[half][it] = -1; r1v[half][it] = 0.f;
        if (gm < g.M) {
          crow_i[half][it] = g.crow_idx ? g.crow_idx[gm] : (int)gm;
          if (g.r1_x) r1v[half][it] = bf2f(g.r1_x[g.arow_idx ? g.arow_idx[gm] : gm]);
        }
      }
  }
  // bias likewise before the first store; a lane's column group is the same in every iteration when 64 % CPR == 0
  constexpr bool CONSTC = (64 % CPR) == 0;
  constexpr int NBV = CONSTC ? 1 : NIT;
  float bv[NBV][8];
#pragma unroll
  for (int it = 0; it < NBV; ++it) {
    const int id = it * 64 + lane, c8 = id % CPR;
    const int gn = n0 + wc * (WNT * 16) + c8 * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) bv[it][r] = 0.f;
    if (g.bias && gn < g.N) { const float4 b0 = *(const float4*)(g.bias + gn), b1 = *(const float4*)(g.bias + gn + 4);
      bv[it][0] = b0.x; bv[it][1] = b0.y; bv[it][2] = b0.z; bv[it][3] = b0.w; bv[it][4] = b1.x; bv[it][5] = b1.y; bv[it][6] = b1.z; bv[it][7] = b1.w; }
  }
  float wv[EMB ? NBV : 1][8];  // rank-1 column weights
  if constexpr (EMB)
#pragma unroll
  for (int it = 0; it < NBV; ++it) {
    const int id = it * 64 + lane, c8 = id % CPR;
    const int gn = n0 + wc * (WNT * 16) + c8 * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) wv[it][r] = 0.f;
    if (g.r1_x && gn < g.N) { const float4 b0 = *(const float4*)(g.r1_w + gn), b1 = *(const float4*)(g.r1_w + gn + 4);
      wv[it][0] = b0.x; wv[it][1] = b0.y; wv[it][2] = b0.z; wv[it][3] = b0.w; wv[it][4] = b1.x; wv[it][5] = b1.y; wv[it][6] = b1.z; wv[it][7] = b1.w; }
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int i = 0; i < HM; ++i)
#pragma unroll
      for (int j = 0; j < WNT; ++j) *(f32x4*)(reg + (i * 16 + fr) * RB + (swz(j * 4 + fq, i * 16 + fr) << 4)) = acc[half * HM + i][j];
    __builtin_amdgcn_wave_barrier();  // same wave, LDS is in order: only the compiler must keep the order
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int id = it * 64 + lane, row = id / CPR, c8 = id - row * CPR;
      const f32x4 v0 = *(const f32x4*)(reg + row * RB + (swz(2 * c8, row) << 4));
      const f32x4 v1 = *(const f32x4*)(reg + row * RB + (swz(2 * c8 + 1, row) << 4));
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      const int64_t gm = m0 + wr * (WMT * 16) + half * (WMT * 8) + row;
      const int gn = n0 + wc * (WNT * 16) + c8 * 8;
      if constexpr (EMB) {  // y = acc + bias + x_row * w_col (all f32), rounded once, into the mapped row
        if (crow_i[half][it] >= 0 && gn < g.N) {
          const float xr = r1v[half][it];
          uint4 o4; unsigned* op = (unsigned*)&o4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float y0 = (g.alpha * v[2 * r] + bv[CONSTC ? 0 : it][2 * r]) + xr * wv[CONSTC ? 0 : it][2 * r];
            const float y1 = (g.alpha * v[2 * r + 1] + bv[CONSTC ? 0 : it][2 * r + 1]) + xr * wv[CONSTC ? 0 : it][2 * r + 1];
            op[r] = f2bf_pack2(y0, y1);
          }
          *(uint4*)((bf16_t*)g.C + (int64_t)crow_i[half][it] * g.ldc + gn) = o4;
        }
      } else if (gm < g.M && gn < g.N) nt_store8<true>(g, gm, gn, v, bv[CONSTC ? 0 : it], auxv[half][it]);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int WMT, int WNT, bool EMB = false>
static void launch_nt8p(spa3d_ctx* c, const NtArgs& g) {
  NtArgs g2 = g; g2.tiles_m = (int)((g.M + 32 * WMT - 1) / (32 * WMT)); g2.tiles_n = g.N / (64 * WNT);
  const int64_t b2 = (int64_t)((g2.tiles_m + 7) / 8) * 8 * g2.tiles_n;
  static bool attr = false;
  constexpr int LDS = (3 * 32 * WMT + 2 * 64 * WNT) * 128;  // 160 KiB <8,4>, 144 KiB <4,6>
  if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_nt8p_kernel<WMT, WNT, false, EMB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr = true; }
  gemm_nt8p_kernel<WMT, WNT, false, EMB><<<(unsigned)b2, 512, LDS, c->stream>>>(g2);
}

// =================================================================================================================
// Persistent form of the 256x256 8-phase kernel: one workgroup per CU walks its XCD's tile list (stride 32).  What it removes
// from every tile: the dispatch, the prologue's load latency and the store drain.  After a tile's K-loop the first two K-tiles of
// the NEXT tile are issued (A ring slots 0 and 1, both B buffers); the epilogue then runs out of ring slot 2 (4 KiB per wave,
// one 16-row accumulator row per pass), and the next K-loop starts behind a COUNTED wait: vmcnt retires in issue order and counts
// stores too (MI355X_MICROARCH.md "s_waitcnt vmcnt(N)"), so "all but the second K-tile's loads and this epilogue's stores" means
// the first K-tile has landed while the stores are still draining.  Edge tiles (rows past M) skip stores, so they drain to 0.
// =================================================================================================================
template <int WMT, int WNT, bool COARSE, bool AUX>
__global__ __launch_bounds__(512, 2) void gemm_nt8pp_kernel(NtArgs g) {
  constexpr int BM = 32 * WMT, BN = 64 * WNT, NA = WMT / 4, NB = WNT / 2, HM = WMT / 2, HN = WNT / 2;  // <8,4>: 256x256, <4,6>: 128x384
  constexpr int ASLOT = BM * 128, BBUF = BN * 128, BOFF = 3 * ASLOT, NKT = 2 * NA + 2 * NB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int xcd = blockIdx.x & 7, cu_slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int per_xcd = ((g.tiles_m + 7) / 8) * g.tiles_n;  // tile list of one XCD (entries with tm >= tiles_m are skipped)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int sr = lane >> 3, scp = lane & 7;
  const int sc = (scp ^ sr) * 8;
  const int nt = g.K / 64;  // >= 2 (host)
  auto next_valid = [&](int idx) { while (idx < per_xcd && (idx / g.tiles_n) * 8 + xcd >= g.tiles_m) idx += nslot; return idx; };
  // addresses = per-tile uniform base (SGPRs) + 32-bit lane offsets (rows are clamped per tile on the M edge only): four VGPRs per
  // operand instead of eight 64-bit pointers, and the K-tile advance is a scalar add
  // One lane-offset VGPR per operand: (row-in-group * ld + swizzled column) * 2.  The 8-row group of each LDS-DMA is wave-uniform and
  // goes into the scalar base; on the M edge the GROUP is clamped (M % 8 == 0 on the host: a group is valid or invalid as a whole, and
  // an invalid one only feeds accumulator rows that are never stored).
  int rga[1][2][NA], rgb[1][2][NB];  // row groups (uniform)
#pragma unroll
  for (int v = 0; v < 1; ++v) {
    const int vw = w;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < NA; ++i) { const int gi = i * 8 + vw; rga[v][h][i] = (gi / WMT) * (WMT * 16) + h * (WMT * 8) + (gi % WMT) * 8; }
#pragma unroll
      for (int i = 0; i < NB; ++i) { const int gi = i * 8 + vw; rgb[v][h][i] = (gi / WNT) * (WNT * 16) + h * (WNT * 8) + (gi % WNT) * 8; }
    }
  }
  const unsigned oal = (unsigned)(sr * g.lda + sc) * 2u, obl = (unsigned)(sr * g.ldb + sc) * 2u;
  const int64_t lda2 = g.lda * 2, ldb2 = g.ldb * 2;
  const char* baseA = nullptr; const char* baseB = nullptr;
  int maxgrp = 0;
  auto set_tile = [&](int64_t m0, int n0) {
    baseA = (const char*)(g.A + m0 * g.lda); baseB = (const char*)(g.Bt + (int64_t)n0 * g.ldb);
    const int64_t mg = g.M - 8 - m0; maxgrp = mg > BM - 1 ? BM - 1 : (mg < 0 ? 0 : (int)mg);
  };
  auto stageA = [&](int kt, int slot) {
    char* base = smem + slot * ASLOT;
    const char* src = baseA + kt * 128;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < NA; ++i) { const int rg = rga[0][h][i] < maxgrp ? rga[0][h][i] : maxgrp; glds16_s(src + rg * lda2, oal, base + rga[0][h][i] * 128); }
  };
  auto stageB = [&](int h, int kt) {
    char* base = smem + (kt & 1) * BBUF + BOFF;
    const char* src = baseB + kt * 128;
#pragma unroll
    for (int i = 0; i < NB; ++i) glds16_s(src + rgb[0][h][i] * ldb2, obl, base + rgb[0][h][i] * 128);
  };
  auto prologue = [&]() { stageA(0, 0); stageB(0, 0); stageB(1, 0); stageA(1, 1); stageB(0, 1); stageB(1, 1); };

  int idx = next_valid(cu_slot);
  if (idx >= per_xcd) return;
  int64_t m0 = (int64_t)((idx / g.tiles_n) * 8 + xcd) * BM; int n0 = (idx % g.tiles_n) * BN;
  set_tile(m0, n0);
  prologue();
  NT8P_WAIT_VM(NKT);
  const int a_off = (wr * WMT * 16 + fr) * 128, b_off = BOFF + (wc * WNT * 16 + fr) * 128;
  const int x0 = ((fq) ^ (fr & 7)) * 16, x1 = ((4 + fq) ^ (fr & 7)) * 16;
  const int nst_code = (g.pre_out || g.out_f32) ? 1 : 0;  // 32 or 16 stores per wave per tile

  while (true) {
    NT8P_BAR();                 // K-tile 0 is visible to every wave; every wave has left the previous epilogue
    if (wr == 1) NT8P_BAR();    // the stagger
    f32x4 acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
      for (int j = 0; j < WNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 aq[HM][2], bq0[HN][2], bq1[HN][2];
#define NT8P_MFMA(AI, BJ, BQ)                                                                                                         \
  __builtin_amdgcn_s_setprio(1);                                                                                                      \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < HM; ++i) _Pragma("unroll") for (int j = 0; j < HN; ++j) \
      acc[(AI) + i][(BJ) + j] = MFMA16(BQ[j][ks], aq[i][ks], acc[(AI) + i][(BJ) + j]);       \
  __builtin_amdgcn_s_setprio(0);
    int aslot = 0, aslot2 = 2;
    for (int t = 0; t < nt; ++t) {
      const char* sa = smem + aslot * ASLOT;
      const char* sb = smem + (t & 1) * BBUF;
      if constexpr (COARSE) {
        // two phases per K-tile, 32 MFMAs (512 cycles) per barrier pair: one wave-row's MFMA section covers the other's 16 reads and
        // their LDS latency.  Every read is retired BEFORE its phase's first barrier, so a slot may be restaged one phase later.
        // ---------------- PA: B-q0, B-q1 (retired first), A-q0 | stage A(t+2) | quadrants (0,0) (0,1)
#pragma unroll
        for (int j = 0; j < HN; ++j) { bq0[j][0] = *(const bf16x8*)(sb + b_off + j * 2048 + x0); bq0[j][1] = *(const bf16x8*)(sb + b_off + j * 2048 + x1); }
#pragma unroll
        for (int j = 0; j < HN; ++j) { bq1[j][0] = *(const bf16x8*)(sb + b_off + (HN + j) * 2048 + x0); bq1[j][1] = *(const bf16x8*)(sb + b_off + (HN + j) * 2048 + x1); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < HM; ++i) { aq[i][0] = *(const bf16x8*)(sa + a_off + i * 2048 + x0); aq[i][1] = *(const bf16x8*)(sa + a_off + i * 2048 + x1); }
        if (t + 2 < nt) stageA(t + 2, aslot2);   // slot of K-tile t-1: its A-q1 reads were retired before the previous phase's barrier
        NT8P_WAIT_LGKM(2 * HM);                  // the B reads have returned: both B halves may be restaged next phase
        NT8P_BAR();
        NT8P_WAIT_LGKM(0);
        NT8P_MFMA(0, 0, bq0)
        NT8P_MFMA(0, HN, bq1)
        NT8P_BAR();
        // ---------------- PB: A-q1 (retired before the barrier) | stage B-h0(t+2), B-h1(t+2) | wait K-tile t+1 | quadrants (1,1) (1,0)
#pragma unroll
        for (int i = 0; i < HM; ++i) { aq[i][0] = *(const bf16x8*)(sa + a_off + (HM + i) * 2048 + x0); aq[i][1] = *(const bf16x8*)(sa + a_off + (HM + i) * 2048 + x1); }
        if (t + 2 < nt) { stageB(0, t + 2); stageB(1, t + 2); NT8P_WAIT_VM(NKT); }
        else NT8P_WAIT_VM(0);
        NT8P_WAIT_LGKM(0);
        NT8P_BAR();
        NT8P_MFMA(HM, HN, bq1)
        NT8P_MFMA(HM, 0, bq0)
        NT8P_BAR();
      } else {
#pragma unroll
      for (int j = 0; j < HN; ++j) { bq0[j][0] = *(const bf16x8*)(sb + b_off + j * 2048 + x0); bq0[j][1] = *(const bf16x8*)(sb + b_off + j * 2048 + x1); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < HM; ++i) { aq[i][0] = *(const bf16x8*)(sa + a_off + i * 2048 + x0); aq[i][1] = *(const bf16x8*)(sa + a_off + i * 2048 + x1); }
      if (t + 2 < nt) stageA(t + 2, aslot2);
      NT8P_WAIT_LGKM(2 * HM);
      NT8P_BAR();
      NT8P_WAIT_LGKM(0);
      NT8P_MFMA(0, 0, bq0)
      NT8P_BAR();
#pragma unroll
      for (int j = 0; j < HN; ++j) { bq1[j][0] = *(const bf16x8*)(sb + b_off + (HN + j) * 2048 + x0); bq1[j][1] = *(const bf16x8*)(sb + b_off + (HN + j) * 2048 + x1); }
      if (t + 2 < nt) stageB(0, t + 2);
      NT8P_BAR();
      NT8P_WAIT_LGKM(0);
      NT8P_MFMA(0, HN, bq1)
      NT8P_BAR();
#pragma unroll
      for (int i = 0; i < HM; ++i) { aq[i][0] = *(const bf16x8*)(sa + a_off + (HM + i) * 2048 + x0); aq[i][1] = *(const bf16x8*)(sa + a_off + (HM + i) * 2048 + x1); }
      NT8P_BAR();
      NT8P_WAIT_LGKM(0);
      NT8P_MFMA(HM, HN, bq1)
      NT8P_BAR();
      if (t + 2 < nt) { stageB(1, t + 2); NT8P_WAIT_VM(NKT); }
      else NT8P_WAIT_VM(0);
      NT8P_BAR();
      NT8P_MFMA(HM, 0, bq0)
      NT8P_BAR();
      }
      aslot = aslot == 2 ? 0 : aslot + 1; aslot2 = aslot2 == 2 ? 0 : aslot2 + 1;
    }
#undef NT8P_MFMA
    if (wr == 0) NT8P_BAR();  // pairs with wave-row 1's extra barrier: every LDS read of this tile is done

    // ---- residual / pre-activation operand of this tile first (its data is needed first), then the next tile's two K-tiles
    const int64_t cm0 = m0; const int cn0 = n0;
    const bool interior = cm0 + BM <= g.M;
    // ---- epilogue geometry.  A pass = one 16-row accumulator row-tile i x one column split cs (JW column tiles, <= 64 columns, so the
    // wave-private f32 image is always [16 rows][16 chunk slots of 16 B] = 4 KiB, chunk ^= row: conflict-free both ways).  Read back as
    // 8-column groups: item id = 64 it + lane -> row id / NG, group id % NG, valid while id < 16 NG (<4,6>: the second iteration is half full).
    constexpr int CS = WNT > 4 ? 2 : 1, JW = WNT / CS, NG = 2 * JW, NP = WMT * CS;
    int prow[2], pg8[2]; bool pval[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int id = it * 64 + lane; pval[it] = id < 16 * NG;
      const int idc = pval[it] ? id : 0; prow[it] = idc / NG; pg8[it] = idc - prow[it] * NG;
    }
    int rbase = wr * (WMT * 16);
    asm volatile("" : "+v"(rbase));  // opaque per tile: otherwise the per-row 64-bit addresses are hoisted out of the tile loop and spilled
    auto gn_of = [&](int cs, int it) { return cn0 + wc * (WNT * 16) + cs * (JW * 16) + pg8[it] * 8; };
    // The epilogue's loads (bias here, the aux operand below) are issued and consumed BEFORE the next tile's LDS-DMA: vmcnt retires in
    // order and the compiler does not see the inline-asm LDS-DMA, so a load used after them would be waited for with a count that also
    // covers the prologue's latency.
    float b8[CS][2][8];
#pragma unroll
    for (int cs = 0; cs < CS; ++cs)
#pragma unroll
      for (int it = 0; it < 2; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) b8[cs][it][r] = 0.f;
        if (g.bias) { const int gn = gn_of(cs, it); const float4 b0 = *(const float4*)(g.bias + gn), b1 = *(const float4*)(g.bias + gn + 4);
          b8[cs][it][0] = b0.x; b8[cs][it][1] = b0.y; b8[cs][it][2] = b0.z; b8[cs][it][3] = b0.w;
          b8[cs][it][4] = b1.x; b8[cs][it][5] = b1.y; b8[cs][it][6] = b1.z; b8[cs][it][7] = b1.w; }
#pragma unroll
        for (int r = 0; r < 8; ++r) asm volatile("" ::"v"(b8[cs][it][r]));
      }
    // staging region: ring slot 2 (and, for the 16-KiB slots of <4,6>, the 16 KiB past the B buffers for waves 4-7)
    char* reg = (ASLOT >= 32768 || w < 4) ? smem + 2 * ASLOT + w * 4096 : smem + 3 * ASLOT + 2 * BBUF + (w - 4) * 4096;
    auto stage_pass = [&](int i, int cs) {
#pragma unroll
      for (int jj = 0; jj < JW; ++jj) *(f32x4*)(reg + fr * 256 + (((jj * 4 + fq) ^ fr) << 4)) = acc[i][cs * JW + jj];
      __builtin_amdgcn_wave_barrier();  // same wave, LDS is in order: only the compiler must keep the order
    };
    auto read_item = [&](int it, float (&v)[8]) {
      const int row = prow[it];
      const f32x4 v0 = *(const f32x4*)(reg + row * 256 + (((2 * pg8[it]) ^ row) << 4));
      const f32x4 v1 = *(const f32x4*)(reg + row * 256 + (((2 * pg8[it] + 1) ^ row) << 4));
      v[0] = v0[0]; v[1] = v0[1]; v[2] = v0[2]; v[3] = v0[3]; v[4] = v1[0]; v[5] = v1[1]; v[6] = v1[2]; v[7] = v1[3];
    };
    constexpr bool has_aux = AUX;              // residual / pre-activation operand present: bf16 output, no pre_out (host)
    uint4 held[AUX ? NP : 1][2];               // aux path: finished rows wait here, in the registers their accumulators vacated
    if constexpr (has_aux) {
      // All loads and all math first (aux double-buffered, one pass ahead), stores last: vmcnt retires in order, so a load issued
      // behind a store would wait for the store's drain, and one issued behind the prologue's LDS-DMA for its latency.
      uint4 ax[2][2];
      auto load_ax = [&](int p, uint4 (&dst)[2]) {
        const int i = p / CS, cs = p % CS;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          int64_t gm = cm0 + (rbase + i * 16 + prow[it]);
          if (gm > g.M - 1) gm = g.M - 1;                                  // unconditional load (rows past M re-read the last row, result
          dst[it] = *(const uint4*)(g.aux + gm * g.ldc + gn_of(cs, it));    // unused): straight-line code lets the compiler emit counted waits
        }
      };
      load_ax(0, ax[0]);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if (p + 1 < NP) load_ax(p + 1, ax[(p + 1) & 1]);
        stage_pass(p / CS, p % CS);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          float v[8]; read_item(it, v);
          held[p][it] = nt_compute8_aux(g, v, b8[p % CS][it], ax[p & 1][it]);
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    idx = next_valid(idx + nslot);
    const bool more = idx < per_xcd;
    if (more) {
      m0 = (int64_t)((idx / g.tiles_n) * 8 + xcd) * BM; n0 = (idx % g.tiles_n) * BN;
      set_tile(m0, n0);
      prologue();
    }
    if constexpr (has_aux) {
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int64_t gm = cm0 + (rbase + (p / CS) * 16 + prow[it]);
          if (pval[it] && gm < g.M NT_ABLATE_STORES) {
            u32x4* cp = (u32x4*)((bf16_t*)g.C + gm * g.ldc + gn_of(p % CS, it));
            const u32x4 o = u32x4{held[p][it].x, held[p][it].y, held[p][it].z, held[p][it].w};
            if (g.nt_store) __builtin_nontemporal_store(o, cp); else *cp = o;
          }
        }
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        stage_pass(p / CS, p % CS);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          float v[8]; read_item(it, v);
          const int64_t gm = cm0 + (rbase + (p / CS) * 16 + prow[it]);
          if (pval[it] && gm < g.M NT_ABLATE_STORES) nt_store8<true, false>(g, gm, gn_of(p % CS, it), v, b8[p % CS][it]);
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (!more) break;
    // the next tile's first K-tile has landed when only its second K-tile and this epilogue's stores can still be outstanding
    if (!interior) NT8P_WAIT_VM(0);
    else if (nst_code) NT8P_WAIT_VM(NKT + 4 * NP);   // two outputs (or f32): 4 NP 16-byte stores per lane and tile
    else NT8P_WAIT_VM(NKT + 2 * NP);
  }
}

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

bool gemm_nt_bf16(spa3d_ctx* c, const GemmDesc& d) {
  // needs B as [N][K] with K contiguous: either the explicit transposed copy or B itself when sBk == 1
  const bf16_t* Bt = nullptr; int64_t ldb = 0;
  if (d.Bt) { Bt = (const bf16_t*)d.Bt; ldb = d.ldBt; }
  else if (d.sBk == 1) { Bt = (const bf16_t*)d.B; ldb = d.sBn; }
  else return false;
  if (d.sAk != 1 || d.nb1 != 1 || d.nb2 != 1 || d.atomic) return false;
  if (d.K % 64 || d.N % 8 || d.M < 1 || d.K < 64) return false;
  if (d.sAm % 8 || ldb % 8 || d.sCm % 8 || !aligned16(d.A) || !aligned16(Bt) || !aligned16(d.C)) return false;
  if (d.aux && (!aligned16(d.aux) || d.out_f32)) return false;
  if (d.bias && !aligned16(d.bias)) return false;
  if (d.pre_out && (!aligned16(d.pre_out) || d.out_f32)) return false;
  if ((int64_t)d.M * d.N < 128 * 128) return false;  // tiny problems: the generic kernel has less tail waste
  const bool emb = d.A2 || d.arow_idx || d.crow_idx || d.r1_x;
  if (emb) {  // one-pass input embedding: only the non-persistent 128 x 384 8-phase kernel carries these operands.  Checked BEFORE the dry-run return: the sizing pass
              // must refuse exactly what the real run refuses, or the caller's fallback allocates buffers the workspace was never sized for
    if (d.N != 384 || d.out_f32 || d.accumulate || d.aux || d.pre_out || d.epi != EPI_NONE || d.crow_group) return false;
    if (d.A2 && (d.K1 % 64 || d.K1 <= 0 || d.K1 >= d.K || d.sA2m % 8 || (!c->dry && !aligned16(d.A2)))) return false;
    if (d.r1_x && !d.r1_w) return false;
  }
  if (c->dry) return true;
  NtArgs g;
  g.A = (const bf16_t*)d.A; g.Bt = Bt; g.C = d.C; g.M = d.M; g.N = d.N; g.K = d.K; g.lda = d.sAm; g.ldb = ldb; g.ldc = d.sCm;
  g.bias = d.bias; g.aux = (const bf16_t*)d.aux; g.pre_out = (bf16_t*)d.pre_out; g.epi = d.epi; g.out_f32 = d.out_f32; g.accumulate = d.accumulate; g.alpha = d.alpha;
  g.tiles_m = (int)((d.M + 127) / 128); g.tiles_n = (d.N + 127) / 128;
  g.crow_group = d.crow_group; g.crow_skip = d.crow_skip;
  g.A2 = (const bf16_t*)d.A2; g.lda2 = d.sA2m; g.K1 = d.K1; g.arow_idx = d.arow_idx; g.crow_idx = d.crow_idx; g.r1_x = (const bf16_t*)d.r1_x; g.r1_w = d.r1_w;
#if SPA3D_ABL_NT
  { const char* e = getenv("SPA3D_ABLATE"); g.ablate = e ? atoi(e) : 0; }
#endif
  g.nt_store = (!d.out_f32 && (double)d.M * d.N * 2.0 >= 512e6 && c->nt_stream) ? 1 : 0;
  const int64_t blocks = (int64_t)((g.tiles_m + 7) / 8) * 8 * g.tiles_n;
  if (blocks > 0x7fffffffLL) return false;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); attr_set = true; }
  // algorithmic bytes: A, B, C once, plus the residual / pre-activation operand read and the second (pre-activation) output
  ProfScope ps(c, PROF_GEMM_NT, 2.0 * (double)d.M * d.N * d.K,
               ((double)d.M * d.K + (double)d.K * d.N + (double)d.M * d.N * (d.out_f32 ? 2.0 : 1.0) * (d.accumulate ? 2.0 : 1.0) + (double)d.M * d.N * ((d.aux ? 1.0 : 0.0) + (d.pre_out ? 1.0 : 0.0))) * 2.0);
  ps.tag(d.M, d.N, d.K, d.epi | (d.aux ? 4 : 0) | (d.pre_out ? 8 : 0) | (d.out_f32 ? 16 : 0) | (d.accumulate ? 32 : 0) | (d.crow_group ? 64 : 0) | (d.sAm != d.K ? 128 : 0));
  const int KT = d.K / 64;
  if (emb) { g.nt_store = 0; launch_nt8p<4, 6, true>(c, g); SPA_LAUNCH_CHECK(c); return true; }
  // 8-phase kernels: 256x256 when 256 | N, 128x384 when 384 | N (see the kernel header for the measurements); nt_8p == 2 (tests): any M
  if (c->nt_8p && (d.N % 256 == 0 || d.N % 384 == 0) && (d.M >= 256 * 64 || c->nt_8p == 2)) {
    // persistent forms (nt_8pp; accumulate would add loads to the counted wait).  nt_8pp == 1 (tests): the 256x256 one only
    const bool pers_ok = c->nt_8pp && d.K >= 128 && !d.accumulate && d.crow_group == 0 && d.M % 8 == 0 && (!d.aux || (!d.pre_out && !d.out_f32));
    const bool use384 = pers_ok && c->nt_8pp == 5 && !d.pre_out && !d.out_f32 && d.N % 384 == 0 && d.N % 256 != 0;
    if (pers_ok && d.N % 256 == 0) {
      NtArgs g2 = g; g2.tiles_m = (int)((g.M + 255) / 256); g2.tiles_n = g.N / 256;
      static bool attrp = false;
      if (!attrp) {
        (void)hipFuncSetAttribute((const void*)gemm_nt8pp_kernel<8, 4, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        (void)hipFuncSetAttribute((const void*)gemm_nt8pp_kernel<8, 4, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        attrp = true;
      }
      if (d.aux) gemm_nt8pp_kernel<8, 4, true, true><<<256, 512, 163840, c->stream>>>(g2);
      else gemm_nt8pp_kernel<8, 4, true, false><<<256, 512, 163840, c->stream>>>(g2);
    } else if (use384) {  // persistent 128x384: slower than the non-persistent kernel on a PLAIN epilogue at K = 768 (759 vs 840 TF/s), faster on every N = 384 shape of the step, whose epilogues mostly carry a residual (out-projection +19 %, MLP-out +12 %, dX shapes +1.5 %)
      NtArgs g2 = g; g2.tiles_m = (int)((g.M + 127) / 128); g2.tiles_n = g.N / 384;
      static bool attrq = false;
      if (!attrq) {
        (void)hipFuncSetAttribute((const void*)gemm_nt8pp_kernel<4, 6, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        (void)hipFuncSetAttribute((const void*)gemm_nt8pp_kernel<4, 6, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        attrq = true;
      }
      if (d.aux) gemm_nt8pp_kernel<4, 6, false, true><<<256, 512, 163840, c->stream>>>(g2);
      else gemm_nt8pp_kernel<4, 6, false, false><<<256, 512, 163840, c->stream>>>(g2);
    } else if (d.N % 256 == 0) launch_nt8p<8, 4>(c, g);
    else launch_nt8p<4, 6>(c, g);
    SPA_LAUNCH_CHECK(c);
    return true;
  }
  if (c->nt_occ && KT <= 8) {  // short K: single LDS buffer, 4 workgroups/CU (+12 % at K = 384: 594 -> 667 TF/s)
    gemm_nt_occ_kernel<<<(unsigned)blocks, 256, 32768, c->stream>>>(g);
    SPA_LAUNCH_CHECK(c);
    return true;
  }
  gemm_nt_kernel<<<(unsigned)blocks, 256, 65536, c->stream>>>(g);
  SPA_LAUNCH_CHECK(c);
  return true;
}

// =================================================================================================================
// TN: C[Ki][N] += sum_m A[m][Ki] * B[m][N]
// =================================================================================================================

__device__ __forceinline__ uint2 ds_read_tr16_b64(const void* p) {
  uint2 v;
  const unsigned a = (unsigned)(uintptr_t)p;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(a) : "memory");
  return v;
}
// byte offset of 16-B chunk `ch` of row `row` in a [rows][128 x bf16] image (T10 layout (b))
__device__ __forceinline__ int tr_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TnArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A 64 rows x 256 B | B 64 rows x 256 B]
  // XCD-aware id: blocks b, b+8, .. share an XCD (and its L2); all output tiles of one M-split read the same rows of
  // A and B, so they are given to ONE XCD, adjacent in dispatch order -> the rows come from HBM once, not once per XCD
  const int xcd = blockIdx.x & 7; int j = blockIdx.x >> 3;
  const int ntile = g.tiles_i * g.tiles_n;
  const int sp = (j / ntile) * 8 + xcd; j %= ntile;
  if (sp >= g.splits) return;
  const int tn = j % g.tiles_n, ti = j / g.tiles_n;
  const int i0 = ti * 128, n0 = tn * 128;
  const int64_t mbeg = (int64_t)sp * g.rows_per_split;
  int64_t mend = mbeg + g.rows_per_split; if (mend > g.M) mend = g.M;
  const int nt = (int)((mend - mbeg + 63) / 64);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wi = (w >> 1) * 64, wn = (w & 1) * 64;

  // ---- staging: one wave-instruction = 4 rows x 256 B; wave w fills rows 16w..16w+15 (4 instructions) of each operand
  const int sr = lane >> 4, scp = lane & 15;
  int acol[4], bcol[4];  // source column (elements) per instruction
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = w * 16 + q * 4 + sr;  // row inside the 64-row tile
    const int ch = scp ^ (((row & 3) << 2) | ((row >> 2) & 3));
    int ca = i0 + ch * 8; if (ca > g.Ki - 8) ca = g.Ki - 8;   // Ki % 8 == 0; clamped columns are never stored
    int cb = n0 + ch * 8; if (cb > g.N - 8) cb = g.N - 8;
    acol[q] = ca; bcol[q] = cb;
  }
  auto stage = [&](int buf, int t) {
    char* sa = smem + buf * 32768 + w * 4096;
    char* sb = sa + 16384;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t m = mbeg + (int64_t)t * 64 + w * 16 + q * 4 + sr;
      const bool ok = m < mend;
      int64_t mb = ok ? m : mend - 1;
      if (g.brow_group > 0) mb = mb + (mb / g.brow_group + 1) * (int64_t)g.brow_skip;
      const bf16_t* pa = ok ? g.A + m * g.lda + acol[q] : g.zero;  // rows past the end contribute exactly 0
      const bf16_t* pb = g.B + mb * g.ldb + bcol[q];
      GLDS16(pa, sa + q * 1024);
      GLDS16(pb, sb + q * 1024);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposed-read addressing for the 32x32x16 operands: 16-lane group gq = lane>>4 covers operand rows
  // (output index) 16*(gq&1)..+15 and k = 8*(gq>>1)..+7; lane 4q+p of the group supplies row k0+q, chunk c0+(p>>1), +8*(p&1)
  const int gq = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
  const int kq = 8 * (gq >> 1) + lq;       // row inside a 16-row k-step (first read; second read is +4)
  const int cbase = 2 * (gq & 1) + (lp >> 1);  // chunk inside the 32-wide sub-tile
  if (nt > 0) stage(0, 0);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) stage(cur ^ 1, t + 1);
    const char* sa = smem + cur * 32768;
    const char* sb = sa + 16384;
    // fragments of k-step ks+1 are requested before the MFMAs of k-step ks issue (two register sets, static indices)
    uint2 fa[2][4], fb[2][4];  // [set][i*2 + {lo,hi}]
    auto load_frags = [&](int set, int ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int cha = (wi + i * 32) / 8 + cbase, chb = (wn + i * 32) / 8 + cbase;
        fa[set][i * 2] = ds_read_tr16_b64(sa + tr_off(ks * 16 + kq, cha) + 8 * (lp & 1));
        fa[set][i * 2 + 1] = ds_read_tr16_b64(sa + tr_off(ks * 16 + kq + 4, cha) + 8 * (lp & 1));
        fb[set][i * 2] = ds_read_tr16_b64(sb + tr_off(ks * 16 + kq, chb) + 8 * (lp & 1));
        fb[set][i * 2 + 1] = ds_read_tr16_b64(sb + tr_off(ks * 16 + kq + 4, chb) + 8 * (lp & 1));
      }
    };
    auto mma = [&](int set) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint4 au = make_uint4(fa[set][i * 2].x, fa[set][i * 2].y, fa[set][i * 2 + 1].x, fa[set][i * 2 + 1].y);
        const bf16x8 af = __builtin_bit_cast(bf16x8, au);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const uint4 bu = make_uint4(fb[set][j * 2].x, fb[set][j * 2].y, fb[set][j * 2 + 1].x, fb[set][j * 2 + 1].y);
          acc[i][j] = MFMA32(af, __builtin_bit_cast(bf16x8, bu), acc[i][j]);
        }
      }
    };
    load_frags(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);  // MFMAs stay below the wait (cdna_hip_programming.md rule 18)
    load_frags(1, 1);
    mma(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    load_frags(0, 2);
    mma(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    load_frags(1, 3);
    mma(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    mma(1);
    __syncthreads();
  }
  if (nt == 0) return;
  // ---- epilogue: 32x32 C/D map col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5); one register = two 128-B row segments
  const int col = lane & 31, rb = 4 * (lane >> 5);
  const DetCfg dc = det_load();   // deterministic-gradient switch (common.hpp), read once
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gn = n0 + wn + j * 32 + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gi = i0 + wi + i * 32 + (r & 3) + 8 * (r >> 2) + rb;
        if (gi < g.Ki && gn < g.N) { float* pr = g.C + (int64_t)gi * g.ldc + gn; if (dc.shadow) grad_add(dc, pr, acc[i][j][r]); else atomicAdd(pr, acc[i][j][r]); }
      }
    }
}

// =================================================================================================================
// TN, 8-phase form.  Output tile (64 WIT) x (128 WNT) with WIT + 2 WNT = 8: <4,2> = 256x256, <2,3> = 128x384, <6,1> = 384x128
// (the 384-wide weight dimensions of the model fit without padding).  8 waves as 2 (i) x 4 (n), 32x32x16 MFMA.
//   The reduction runs over "quarters" of 16 rows of m.  One quarter in LDS = four [16 rows][128 columns] sub-images (A's, then
//   B's) in the transposed-read layout above = 16 KiB = 2 LDS-DMA per thread.  Phase q:
//       12 (11 / 14) ds_read_b64_tr_b16 pairs of quarter q | 2 LDS-DMA of quarter q+8 | s_waitcnt vmcnt(14) (quarter q+1 landed)
//       s_barrier | lgkmcnt(0) | 8 (6) MFMA | s_barrier
//   Ring of 10 quarters = 160 KiB: quarter q+8 takes the slot of quarter q-2, two phases after its last read; both operands are
//   streamed from HBM and get an 8-phase lead.  Wave-row 1 runs one barrier behind wave-row 0 (one wave of each per SIMD), so
//   one reads/stages while the other issues MFMAs.  Hazard rules as in the NT kernel (cdna_hip_programming.md, 8-phase template).
// =================================================================================================================
// QP = quarters per phase.  QP = 2: 16 MFMAs (512 cycles) between barrier pairs instead of 8, so one wave-row's MFMA section covers
// the other's 24 transposed reads and their LDS latency (with QP = 1 the read/stage section is the longer one: ~46 % MFMA-busy);
// the lead shrinks to 6 quarters (the slot of quarter q+6 was last read two phases ago).
template <int WIT, int WNT, int QP>
__global__ __launch_bounds__(512, 2) void gemm_tn8p_kernel(TnArgs g) {
  constexpr int TI = 64 * WIT, TNN = 128 * WNT, NSA = TI / 128;
  static_assert(TI / 128 + TNN / 128 == 4, "four sub-images per quarter");
  constexpr int R = 10, D = QP == 1 ? 8 : 6, QB = 16384;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int xcd = blockIdx.x & 7; int jb = blockIdx.x >> 3;
  const int ntile = g.tiles_i * g.tiles_n;
  const int sp = (jb / ntile) * 8 + xcd; jb %= ntile;
  if (sp >= g.splits) return;
  const int tn = jb % g.tiles_n, ti = jb / g.tiles_n;
  const int i0 = ti * TI, n0 = tn * TNN;
  const int64_t mbeg = (int64_t)sp * g.rows_per_split;
  int64_t mend = mbeg + g.rows_per_split; if (mend > g.M) mend = g.M;
  const int nq = (int)((mend - mbeg + 15) / 16);
  if (nq <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;

  // ---- staging: LDS-DMA ii = 8 i2 + w of a quarter -> sub-image ii >> 2, rows 4 (ii & 3) .. +3; lane -> (row sr, physical chunk scp)
  const int sr = lane >> 4, scp = lane & 15;
  // The steady state is a pointer increment per LDS-DMA (the read/stage section has to fit under the other wave-row's 8 MFMAs);
  // the row-bound test exists only in the tail branch, and the B-row remap (row -> row + (row / G + 1) * S) is carried incrementally.
  // Named scalars, not arrays: an array of pointers selected against g.zero is demoted to scratch (vmcnt(0) in the loop).
  const bf16_t *spA, *spB; int64_t stepA, stepB, skipA = 0, skipB = 0; int ldsA, ldsB, r16A, r16B, boffA = 0, boffB = 0;
  auto setup = [&](int i2, const bf16_t*& sp, int64_t& step, int64_t& skip, int& ldso, int& r16o, int& boff) {
    const int ii = i2 * 8 + w, si = ii >> 2, rg = ii & 3;
    const int r16 = rg * 4 + sr;
    const int ch = scp ^ (((r16 & 3) << 2) | ((r16 >> 2) & 3));
    r16o = r16; ldso = si * 4096 + rg * 1024;
    const int64_t m = mbeg + r16;
    if (si < NSA) {
      int ca = i0 + si * 128 + ch * 8; if (ca > g.Ki - 8) ca = g.Ki - 8;
      sp = g.A + m * g.lda + ca; step = 16 * g.lda;
    } else {
      int cb = n0 + (si - NSA) * 128 + ch * 8; if (cb > g.N - 8) cb = g.N - 8;
      int64_t row = m;
      if (g.brow_group > 0) { const int64_t grp = m / g.brow_group; boff = (int)(m - grp * g.brow_group); row = m + (grp + 1) * (int64_t)g.brow_skip;
        skip = (int64_t)g.brow_skip * g.ldb; }
      sp = g.B + row * g.ldb + cb; step = 16 * g.ldb;
    }
  };
  setup(0, spA, stepA, skipA, ldsA, r16A, boffA);
  setup(1, spB, stepB, skipB, ldsB, r16B, boffB);
  const int nq_full = (int)((mend - mbeg) / 16);  // quarters whose 16 rows all exist
  const bool remap = g.brow_group > 0;
  const int G = g.brow_group;
  int q_issue = 0, slot_issue = 0;
  auto stage = [&]() {  // quarters are issued strictly in order
    char* base = smem + slot_issue * QB;
    if (q_issue < nq_full) {
      GLDS16(spA, base + ldsA);
      GLDS16(spB, base + ldsB);
    } else {  // rows past the end contribute exactly 0 (both operands read the zero page)
      const int64_t mq = mbeg + (int64_t)q_issue * 16;
      const bf16_t* pa = (mq + r16A < mend) ? spA : g.zero;
      const bf16_t* pb = (mq + r16B < mend) ? spB : g.zero;
      GLDS16(pa, base + ldsA);
      GLDS16(pb, base + ldsB);
    }
    spA += stepA; spB += stepB;
    if (remap) {
      boffA += 16; if (boffA >= G) { boffA -= G; spA += skipA; }
      boffB += 16; if (boffB >= G) { boffB -= G; spB += skipB; }
    }
    ++q_issue; slot_issue = slot_issue == R - 1 ? 0 : slot_issue + 1;
  };

  f32x16 acc[WIT][WNT];
#pragma unroll
  for (int i = 0; i < WIT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposed-read addressing (as in gemm_tn_kernel): 16-lane group gq covers operand rows 16*(gq&1)..+15 and k = 8*(gq>>1)..+7
  const int gq = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
  const int kq = 8 * (gq >> 1) + lq;
  const int cbase = 2 * (gq & 1) + (lp >> 1);
  int offa[WIT][2], offb[WNT][2];  // byte offsets inside a quarter
#pragma unroll
  for (int i = 0; i < WIT; ++i) {
    const int it = wr * WIT + i, si = it >> 2, ch = (it & 3) * 4 + cbase;
    offa[i][0] = si * 4096 + tr_off(kq, ch) + 8 * (lp & 1); offa[i][1] = si * 4096 + tr_off(kq + 4, ch) + 8 * (lp & 1);
  }
#pragma unroll
  for (int j = 0; j < WNT; ++j) {
    const int jt = wc * WNT + j, si = NSA + (jt >> 2), ch = (jt & 3) * 4 + cbase;
    offb[j][0] = si * 4096 + tr_off(kq, ch) + 8 * (lp & 1); offb[j][1] = si * 4096 + tr_off(kq + 4, ch) + 8 * (lp & 1);
  }

  // bias gradient on the side: wave-row 0 of the ti == 0 workgroups already holds every B fragment (lane: column lane & 31,
  // 8 of the quarter's 16 rows); v_dot2c_f32_bf16 against (1, 1) adds two rows per instruction, f32 accumulate
  const bool do_cs = g.colsum != nullptr && ti == 0 && wr == 0;
  float cs[WNT];
#pragma unroll
  for (int j = 0; j < WNT; ++j) cs[j] = 0.f;
  const unsigned ones2 = ONES2_16;
  const int nqp = (nq + QP - 1) / QP * QP;  // padded to whole phases: the extra quarter lies past mend and stages the zero page
  const int npro = nqp < D ? nqp : D;
  for (int q = 0; q < npro; ++q) stage();
  if (nqp >= D) NT8P_WAIT_VM(2 * (D - QP)); else NT8P_WAIT_VM(0);
  NT8P_BAR();                 // the first phase's quarters are visible to every wave
  if (wr == 1) NT8P_BAR();    // the stagger
  int slot = 0;
  for (int q = 0; q < nqp; q += QP) {
    const char* sq = smem + slot * QB;
    uint2 fb[QP][WNT][2], fa[QP][WIT][2];
#pragma unroll
    for (int u = 0; u < QP; ++u) {
#pragma unroll
#if (SPA3D_ABL_TN & 1)  // diagnostic build only (tools/ablate_gemm_tn.py): half of the transposed reads (WRONG results: timing only)
      for (int j = 0; j < WNT; ++j) { fb[u][j][0] = ds_read_tr16_b64(sq + u * QB + offb[j][0]); fb[u][j][1] = fb[u][j][0]; }
#pragma unroll
      for (int i = 0; i < WIT; ++i) { fa[u][i][0] = ds_read_tr16_b64(sq + u * QB + offa[i][0]); fa[u][i][1] = fa[u][i][0]; }
#else
      for (int j = 0; j < WNT; ++j) { fb[u][j][0] = ds_read_tr16_b64(sq + u * QB + offb[j][0]); fb[u][j][1] = ds_read_tr16_b64(sq + u * QB + offb[j][1]); }
#pragma unroll
      for (int i = 0; i < WIT; ++i) { fa[u][i][0] = ds_read_tr16_b64(sq + u * QB + offa[i][0]); fa[u][i][1] = ds_read_tr16_b64(sq + u * QB + offa[i][1]); }
#endif
    }
    if (q + D < nqp) {   // the next phase's quarters have landed (this wave's part); D - QP quarters stay in flight
#pragma unroll
      for (int u = 0; u < QP; ++u) stage();
      NT8P_WAIT_VM(2 * (D - QP));
    } else NT8P_WAIT_VM(0);
    NT8P_BAR();
    NT8P_WAIT_LGKM(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int u = 0; u < QP; ++u)
#pragma unroll
      for (int i = 0; i < WIT; ++i) {
        const bf16x8 af = __builtin_bit_cast(bf16x8, make_uint4(fa[u][i][0].x, fa[u][i][0].y, fa[u][i][1].x, fa[u][i][1].y));
#pragma unroll
        for (int j = 0; j < WNT; ++j)
#if (SPA3D_ABL_TN & 2)  // no MFMAs (timing only)
          asm volatile("" ::"v"(af), "v"(fb[u][j][0]), "v"(fb[u][j][1]));
#else
          acc[i][j] = MFMA32(af, __builtin_bit_cast(bf16x8, make_uint4(fb[u][j][0].x, fb[u][j][0].y, fb[u][j][1].x, fb[u][j][1].y)), acc[i][j]);
#endif
      }
    if (do_cs) {
#pragma unroll
      for (int u = 0; u < QP; ++u)
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
          asm(DOT2C_F32_16 " %0, %1, %2" : "+v"(cs[j]) : "v"(fb[u][j][0].x), "v"(ones2));
          asm(DOT2C_F32_16 " %0, %1, %2" : "+v"(cs[j]) : "v"(fb[u][j][0].y), "v"(ones2));
          asm(DOT2C_F32_16 " %0, %1, %2" : "+v"(cs[j]) : "v"(fb[u][j][1].x), "v"(ones2));
          asm(DOT2C_F32_16 " %0, %1, %2" : "+v"(cs[j]) : "v"(fb[u][j][1].y), "v"(ones2));
        }
    }
    __builtin_amdgcn_s_setprio(0);
    NT8P_BAR();
    slot += QP; if (slot >= R) slot -= R;
  }
  if (wr == 0) NT8P_BAR();
  if (do_cs) {
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      const float t = cs[j] + __shfl_xor(cs[j], 32, 64);
      const int gn = n0 + wc * (WNT * 32) + j * 32 + (lane & 31);
      if (lane < 32 && gn < g.N) grad_add(g.colsum + gn, t);
    }
  }
  // ---- epilogue: 32x32 C/D map col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5); one register = two 128-B row segments
  const int col = lane & 31, rb = 4 * (lane >> 5);
  const DetCfg dc = det_load();   // deterministic-gradient switch (common.hpp), read once
#pragma unroll
  for (int i = 0; i < WIT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      const int gn0 = n0 + wc * (WNT * 32) + j * 32;   // wave-uniform: a 32-column group never straddles a segment (seg_n % 32 == 0)
      float* cb = g.C; int cn = gn0 + col;
      if (g.seg_n > 0) { const int sg = gn0 / g.seg_n; if (sg > 0) { cb = g.Cseg[sg > 1]; cn -= sg * g.seg_n; } }
      const int gn = gn0 + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gi = i0 + wr * (WIT * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + rb;
        if (gi < g.Ki && gn < g.N) { float* pr = cb + (int64_t)gi * g.ldc + cn; if (dc.shadow) grad_add(dc, pr, acc[i][j][r]); else atomicAdd(pr, acc[i][j][r]); }
      }
    }
}

template <int WIT, int WNT, int QP>
static void launch_tn8p_q(spa3d_ctx* c, TnArgs g, int rounds) {
  constexpr int TI = 64 * WIT, TNN = 128 * WNT;
  g.tiles_i = (g.Ki + TI - 1) / TI; g.tiles_n = (g.N + TNN - 1) / TNN;
  const int64_t tiles = (int64_t)g.tiles_i * g.tiles_n;
  // M-splits: one workgroup per CU, equal-length splits.  Pick the count that minimises a makespan model:
  //   rounds on the fullest XCD x (M / s) rows x (time per row at ~4 TF/s per CU)  +  s x (Ki x N x 4 B of f32 atomics at ~1.3 TB/s)
  int64_t splits;
  if (rounds > 0) splits = std::max<int64_t>(8, (256 * (int64_t)rounds / tiles) / 8 * 8);
  else {
    const double t_row = 2.0 * TI * TNN / 4.0e12, t_atom = (double)g.Ki * g.N * 4.0 / 1.3e12;
    const int64_t smax = std::max<int64_t>(1, std::min<int64_t>(g.M / 4096, 2048));
    double best = 1e30; splits = 1;
    for (int64_t sc = 1; sc <= smax; ++sc) {
      const int64_t rps_c = ((g.M + sc - 1) / sc + 63) / 64 * 64;
      const int64_t per_xcd = tiles * ((sc + 7) / 8);  // all tiles of a split run on one XCD (32 CUs), splits are dealt round-robin
      const double t = (double)((per_xcd + 31) / 32) * (double)rps_c * t_row + (double)sc * t_atom;
      if (t < best * 0.999) { best = t; splits = sc; }
    }
  }
  splits = std::min<int64_t>(splits, std::max<int64_t>(1, g.M / 4096));
  int64_t rps = ((g.M + splits - 1) / splits + 63) / 64 * 64;
  splits = (g.M + rps - 1) / rps;
  g.splits = (int)splits; g.rows_per_split = rps;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_tn8p_kernel<WIT, WNT, QP>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840); attr = true; }
  gemm_tn8p_kernel<WIT, WNT, QP><<<(unsigned)(tiles * ((splits + 7) / 8 * 8)), 512, 163840, c->stream>>>(g);
}

template <int WIT, int WNT>
static void launch_tn8p(spa3d_ctx* c, const TnArgs& g, int rounds) {
  launch_tn8p_q<WIT, WNT, 2>(c, g, rounds);  // two quarters per phase (+7-10 % over one: NOTEBOOK.md, "How the GEMMs got ...", item 5)
}

bool gemm_tn_bf16(spa3d_ctx* c, const GemmDesc& d) {
  // A[m'=i][k'=m] = X[m][i]: sAm == 1, sAk == lda ; B[k'=m][n]: sBn == 1, sBk == ldb ; f32 accumulate
  if (d.sAm != 1 || d.sBn != 1 || !d.out_f32 || !d.accumulate || d.nb1 != 1 || d.nb2 != 1) return false;
  if (d.epi != EPI_NONE || d.aux || d.bias || d.alpha != 1.f || !d.zero_page) return false;
  const int Ki = (int)d.M, N = d.N; const int64_t M = d.K;
  if (Ki % 8 || N % 8 || Ki < 8 || N < 8 || M < 256) return false;
  if (d.sAk % 8 || d.sBk % 8 || !aligned16(d.A) || !aligned16(d.B)) return false;
  if (c->dry) return true;
  TnArgs g;
  g.A = (const bf16_t*)d.A; g.B = (const bf16_t*)d.B; g.C = (float*)d.C; g.zero = (const bf16_t*)d.zero_page;
  g.M = M; g.Ki = Ki; g.N = N; g.lda = d.sAk; g.ldb = d.sBk; g.ldc = d.sCm;
  g.tiles_i = (Ki + 127) / 128; g.tiles_n = (N + 127) / 128;
  g.brow_group = d.brow_group; g.brow_skip = d.brow_skip; g.colsum = nullptr;
  g.seg_n = d.seg_n; g.Cseg[0] = (float*)d.C_seg[0]; g.Cseg[1] = (float*)d.C_seg[1];
  if (d.seg_n > 0 && (d.seg_n % 32 || N % d.seg_n || N / d.seg_n > 3 || d.colsum_out)) return false;
  c->tn_colsum_fused = false;
  if (c->tn_big && (M >= 65536 || c->tn_big == 2) && ((Ki % 384 == 0 && N % 256 == 0) || (Ki % 256 == 0 && N % 384 == 0))) {  // large register tile (gemm_tnb.hip)
    ProfScope ps(c, PROF_GEMM_TN, 2.0 * (double)M * Ki * N, ((double)M * Ki + (double)M * N) * 2.0);
    ps.tag(M, N, Ki, 1 << 20);
    g.colsum = d.colsum_out;
    if (gemm_tnb(c, g)) { c->tn_colsum_fused = d.colsum_out != nullptr; SPA_LAUNCH_CHECK(c); return true; }
    g.colsum = nullptr; ps.on = false;
  }
  if (c->tn_8p && (M >= 65536 || c->tn_8p == 2) && (g.brow_group == 0 || g.brow_group >= 16)) {
    // tile shape with the least padding: 384 x 128 / 128 x 384 when one dimension is an odd multiple of 384, else 256 x 256
    auto waste = [&](int TI, int TNN) { return (double)((Ki + TI - 1) / TI * TI) * ((N + TNN - 1) / TNN * TNN) / ((double)Ki * N); };
    const double w0 = waste(256, 256), w1 = waste(128, 384), w2 = waste(384, 128);
    const double wb = std::min(w0, std::min(w1, w2));
    if (wb <= 1.25 || c->tn_8p == 2) {
      ProfScope ps(c, PROF_GEMM_TN, 2.0 * (double)M * Ki * N, ((double)M * Ki + (double)M * N) * 2.0);
      ps.tag(M, N, Ki, 0);
      const int rounds = 0;  // M-split count from the makespan model
      g.colsum = d.colsum_out; c->tn_colsum_fused = d.colsum_out != nullptr;
      if (w0 <= wb * 1.0001) launch_tn8p<4, 2>(c, g, rounds);
      else if (w1 <= wb * 1.0001) launch_tn8p<2, 3>(c, g, rounds);
      else launch_tn8p<6, 1>(c, g, rounds);
      SPA_LAUNCH_CHECK(c);
      return true;
    }
  }
  if (d.seg_n > 0) return false;  // segmented outputs exist in the 8-phase kernels only: the caller falls back to one GEMM per segment
  const int64_t tiles = (int64_t)g.tiles_i * g.tiles_n;
  // enough workgroups to fill 256 CUs several times over, but >= 4096 reduction rows per split so the
  // f32 atomic traffic (4 B per output element per split) stays a few % of the tile's MFMA time
  // (small M -- the latent stacks' dW at M = 1 408: with one split a 144-tile problem is 144 workgroups walking 22 k-steps one memory latency at a time;
  // there the split goes down to 256 rows as long as the grid stays under ~768 workgroups, whose atomics are a few MB)
  const int64_t by_rows = std::max<int64_t>((M + 4095) / 4096, std::min<int64_t>(M / 256, (768 + tiles - 1) / tiles));
  int64_t splits = std::max<int64_t>(1, std::min<int64_t>((4096 + tiles - 1) / tiles, by_rows));
  int64_t rps = ((M + splits - 1) / splits + 63) / 64 * 64;
  splits = (M + rps - 1) / rps;
  g.splits = (int)splits; g.rows_per_split = rps;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); attr_set = true; }
  ProfScope ps(c, PROF_GEMM_TN, 2.0 * (double)M * Ki * N, ((double)M * Ki + (double)M * N) * 2.0 + (double)Ki * N * 4.0 * splits);
  ps.tag(M, N, Ki, splits);
  gemm_tn_kernel<<<(unsigned)(tiles * ((splits + 7) / 8 * 8)), 256, 65536, c->stream>>>(g);
  SPA_LAUNCH_CHECK(c);
  return true;
}
SPA_DET_UPLOAD_DEF(det_upload_gemm_fast)
}  // namespace SPA_NS
