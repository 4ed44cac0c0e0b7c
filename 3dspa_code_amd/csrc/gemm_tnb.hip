// gemm_tnb.hip -- dW = X^T . dY (TN) with a large register tile (gfx950; round 5).
//
// The 8-wave dW kernels of gemm_fast.hip spend ~75 % of their time on the operand stream into LDS (tools/ablate_gemm_tn.py, round 4: without any MFMA
// the 128 x 384 form still takes 3.0 of 4.1 ms): every 128 x 384 output tile streams 256 + 768 B per m-row through LDS-DMA, 0.33 KiB per MFMA.  Here ONE
// wave per SIMD holds a 96 x 256 (or 256 x 96) accumulator tile = 3 x 8 MFMA tiles of 32 x 32 = 384 accumulator registers, and the four waves of a
// workgroup cover 384 x 256 (or 256 x 384) of dW: 0.21 KiB of LDS-DMA and 0.46 KiB of transposed fragment reads per MFMA.
//
// Registers.  384 accumulators do not fit one register file (256 VGPR + 256 AGPR per lane at one wave per SIMD) and hipcc picks ONE file for all MFMA
// results of a kernel (AGPR at this occupancy): the round-4 version spilled 112-176 registers.  The MFMAs are therefore inline assembly with the
// accumulator file chosen per tile: 16 tiles "+a" (256 AGPRs, all of them), 8 tiles "+v" (128 VGPRs), which leaves 128 VGPRs for the fragments
// (3 x 4 x 2 sets + 8 x 4), the read / staging addresses and temporaries.  0 spilled registers.
//
// Layout.  The reduction runs over "quarters" of 16 m-rows.  A quarter in LDS = five [16 rows][128 columns] sub-images (A's 3 (2), then B's 2 (3)) = 20 KiB.
// A sub-image is four 1-KiB pieces (rows 4g .. 4g+3), one LDS-DMA instruction each (wave w stages piece w of every sub-image: 4 rows x 256 B of global
// memory), and INSIDE a piece the order is [32-column tile t][row r][64 B]: the 4 rows x 64 B that one 32-lane half of ds_read_b64_tr_b16 takes are 256
// contiguous bytes (conflict-free without a swizzle) and every tile / half / sub-image of a fragment read is an INSTRUCTION OFFSET from one lane address
// (the first version used the XOR-swizzled 256-B rows of gemm_fast.hip: 14 lane constants and a v_add per read, and the kernel is issue-bound -- 22 adds
// per quarter were 11 % of its time, tools/ablate_tnb.py mask 16).
// Ring of 8 quarters = 160 KiB.  Phase = 2 quarters = 48 MFMAs per wave between two barriers; phase p issues the LDS-DMA of quarters 2p+6, 2p+7 into the slots
// of quarters 2p-2, 2p-1 (last read in phase p-1) and ends with vmcnt(15) (this wave's pieces of quarters <= 2p+4 landed) + s_barrier.
//
// Schedule.  One wave per SIMD: nothing hides what the wave's own in-order stream does not, so every MFMA is followed by at most one or two other
// instructions (an MFMA holds vector issue for 8 of its 32 cycles).  Quarter q, triple j = the three MFMAs of wide tile j:
//     T0: the narrow operand's fragments of quarter q+1 (second register set), one tile behind each MFMA
//     Tj (j = 1..7): wide fragment j-1 of quarter q+1 replaces fragment j-1 behind MFMAs 0 and 1 (its last use was triple j-1);
//                    behind MFMA 2: one LDS-DMA piece (j = 1..5), the next quarter's read addresses (j = 6), wide fragment 7 (j = 7)
// Reads and LDS-DMA are untracked inline asm; LDS operations return in order, so two counted waits per quarter are exact: lgkmcnt(8) ahead of T0 (all but
// wide fragments 4..7 of the previous quarter's reads have returned) and lgkmcnt(12) ahead of T4 (this quarter has issued 12 reads by then).
//
// Rows.  The kernel takes whole phases only: M % 32 rows of the reduction go to a small scalar tail kernel, so no lane ever needs a bounds test or a
// zero page and both operands are addressed as (wave-uniform 64-bit base in SGPRs) + (32-bit lane offset) + (instruction offset).
#include <algorithm>

#include "tn_args.hpp"

namespace SPA_NS {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef mfma16x8 bf16x8;

#define TNB_R 8
#define TNB_QB 20480
#define TNB_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define TNB_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define TNB_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")
// LDS-DMA: 64 lanes x 16 B from (SGPR base + lane offset + IMM) to LDS at M0 + IMM + 16 lane (the instruction offset moves BOTH addresses: the caller
// subtracts it from the M0 value)
#define TNB_GLDS(voff, sbase, m0v, IMM) \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(voff), "s"(sbase), "s"(m0v), "n"(IMM) : "memory", "m0")
// transposed fragment read at (lane address) + IMM.  A function, not a macro: clang does not capture a name that a generic lambda uses only as an asm operand
template <int IMM> __device__ __forceinline__ void tnb_rd(uint2& dst, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM) : "memory");
}

constexpr int TNB_ABL = SPA3D_ABL_TNB;  // csrc/ablate.inc: 0 in libspa3d_hip.so
template <int N_> struct TnIC { static constexpr int v = N_; };
// compile-time loop (a `#pragma unroll` loop this large falls under LLVM's pragma-unroll size threshold and stays a loop: accumulators in scratch)
template <int I, int N, typename F> __device__ __forceinline__ void tn_for(F&& f) { if constexpr (I < N) { f(TnIC<I>{}); tn_for<I + 1, N>(f); } }

// WB: true  = tile 384 (i: columns of A) x 256 (n: columns of B); wave w owns i in [96w, 96w+96) x all n: narrow operand A (3 tiles), wide operand B (8)
//     false = tile 256 (i) x 384 (n);                              wave w owns all i x n in [96w, 96w+96): narrow operand B,           wide operand A
// CS: also accumulate the column sums of B (the bias gradient);  REMAP: B row m sits at m + (m / brow_group + 1) * brow_skip
template <bool WB, bool CS, bool REMAP>
__global__ __launch_bounds__(256, 1) void gemm_tnb_kernel(TnArgs g) {
  constexpr int R = TNB_R, QB = TNB_QB;
  constexpr int NSA = WB ? 3 : 2;                           // A's sub-images of a quarter (B's follow)
  constexpr int NARROW0 = WB ? 0 : 2, WIDE0 = WB ? 3 : 0;   // first sub-image of the narrow / wide operand
  constexpr int TI = WB ? 384 : 256, TNN = WB ? 256 : 384;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // XCD-aware id (as gemm_tn8p_kernel): all output tiles of one M-split read the same rows, so they are given to ONE XCD
  const int xcd = blockIdx.x & 7; int jb = blockIdx.x >> 3;
  const int ntile = g.tiles_i * g.tiles_n;
  const int sp = (jb / ntile) * 8 + xcd; jb %= ntile;
  if (sp >= g.splits) return;
  const int tn = jb % g.tiles_n, ti = jb / g.tiles_n;
  const int i0 = ti * TI, n0 = tn * TNN;
  const int64_t mbeg = (int64_t)sp * g.rows_per_split;
  int64_t mend = mbeg + g.rows_per_split; if (mend > g.M) mend = g.M;
  const int nq = (int)((mend - mbeg) >> 4);  // whole quarters, an even count (host: M % 32 == 0, rows_per_split % 64 == 0)
  if (nq <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)smem;

  // ---- staging: piece (sub-image s, row group w); lane -> (tile st, row sr of the group, 16-B chunk sc of the tile's 64-B row): LDS-DMA writes lane-linear,
  // so this IS the [tile][row][64 B] order inside the piece
  const int st = lane >> 4, sr = (lane >> 2) & 3, sc = lane & 3, r16 = 4 * w + sr;
  unsigned voffA = (unsigned)(((int64_t)r16 * g.lda + i0 + 32 * st + 8 * sc) * 2);
  unsigned voffB = (unsigned)(((int64_t)r16 * g.ldb + n0 + 32 * st + 8 * sc) * 2);
  const char* sA = (const char*)(g.A + mbeg * g.lda);
  const char* sB;
  int boff = 0; unsigned skipB = 0; int G = 0;
  if constexpr (REMAP) {
    G = g.brow_group;
    const int64_t grp0 = mbeg / G;
    sB = (const char*)(g.B + (mbeg + (grp0 + 1) * (int64_t)g.brow_skip) * g.ldb);
    const int64_t m = mbeg + r16, grp = m / G;
    boff = (int)(m - grp * G);
    skipB = (unsigned)((int64_t)g.brow_skip * g.ldb * 2);
    voffB += (unsigned)(grp - grp0) * skipB;
  } else sB = (const char*)(g.B + mbeg * g.ldb);
  const unsigned stepA = (unsigned)(32 * g.lda), stepB = (unsigned)(32 * g.ldb);  // 16 rows in bytes
  int si = 0;                                  // ring slot the next staged quarter goes to
  unsigned m0s = lds0 + (unsigned)w * 1024u;   // its LDS address for this wave's row group
  bool in_loop = false;                        // (diagnostic builds: the prologue always stages)
  auto stage_piece = [&](auto s_) {
    constexpr int s = decltype(s_)::v;
    constexpr int IMM = 256 * (s < NSA ? s : s - NSA);   // the sub-image's first column in bytes
    const unsigned m0v = m0s + (unsigned)(s * 4096 - IMM);
    const unsigned vo = s < NSA ? voffA : voffB; const char* sb = s < NSA ? sA : sB;  // locals: clang does not capture names used only as asm operands
    if (!(TNB_ABL & 1) || !in_loop) TNB_GLDS(vo, sb, m0v, IMM);
  };
  auto stage_advance = [&]() {
    sA += stepA; sB += stepB;
    si = si == R - 1 ? 0 : si + 1; m0s = lds0 + (unsigned)w * 1024u + (unsigned)si * QB;
    if constexpr (REMAP) { boff += 16; const bool wrap = boff >= G; boff = wrap ? boff - G : boff; voffB = wrap ? voffB + skipB : voffB; }
  };

  // ---- transposed-read addressing: 16-lane group gq covers operand index 16 (gq & 1) .. +15 and k = 8 (gq >> 1) .. +7; lane 4 lq + lp of the group supplies row
  // k0 + lq, chunk 2 (gq & 1) + (lp >> 1), + 8 (lp & 1) bytes.  Half h of a fragment = rows + 4 h = the next piece.  Byte offset inside a quarter of
  // (sub-image s, tile t, half h) = 4096 s + 1024 (2 (gq >> 1) + h) + 256 t + 64 lq + 16 chunk + 8 (lp & 1): one lane constant + an instruction offset
  const int gq = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
  const unsigned lc = (unsigned)(2048 * (gq >> 1) + 64 * lq + 16 * (2 * (gq & 1) + (lp >> 1)) + 8 * (lp & 1));
  unsigned lcn[3];  // the wave's narrow tiles 3w .. 3w+2 straddle the sub-images: their sub-image / tile offsets ride in the lane constant
#pragma unroll
  for (int i = 0; i < 3; ++i) { const int it = 3 * w + i; lcn[i] = lc + (unsigned)((NARROW0 + (it >> 2)) * 4096 + (it & 3) * 256); }
  unsigned cur[2][4];  // read addresses of the quarter being prefetched: [q & 1][wide | narrow tile 0..2] = its slot + the constants above

  f32x16 accA[16], accV[8];  // tile t = 3 j + i: t < 16 in AGPRs, the rest in VGPRs
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) accA[t][r] = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) accV[t][r] = 0.f;
  uint2 fn[2][3][2], fw[8][2];
  const bool do_cs = CS && g.colsum != nullptr && ti == 0;
  float cs[3] = {0.f, 0.f, 0.f};
  const unsigned ones2 = ONES2_16;

  auto mma = [&](auto t_, const uint2& a0, const uint2& a1, const uint2& b0, const uint2& b1) {
    constexpr int t = decltype(t_)::v;
    const bf16x8 av = __builtin_bit_cast(bf16x8, make_uint4(a0.x, a0.y, a1.x, a1.y));
    const bf16x8 bv = __builtin_bit_cast(bf16x8, make_uint4(b0.x, b0.y, b1.x, b1.y));
    if constexpr (TNB_ABL & 2) asm volatile("" ::"v"(av), "v"(bv));
    else if constexpr (t < 16) { f32x16& acc = accA[t]; asm volatile(MFMA32_ASM " %0, %1, %2, %0" : "+a"(acc) : "v"(av), "v"(bv)); }
    else { f32x16& acc = accV[t - 16]; asm volatile(MFMA32_ASM " %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv)); }
  };
  auto dot4 = [&](float& s, const uint2& x0, const uint2& x1) {  // s += the 8 values of a fragment (v_dot2c against (1, 1), f32 accumulate)
    asm volatile(DOT2C_F32_16 " %0, %1, %2" : "+v"(s) : "v"(x0.x), "v"(ones2));
    asm volatile(DOT2C_F32_16 " %0, %1, %2" : "+v"(s) : "v"(x0.y), "v"(ones2));
    asm volatile(DOT2C_F32_16 " %0, %1, %2" : "+v"(s) : "v"(x1.x), "v"(ones2));
    asm volatile(DOT2C_F32_16 " %0, %1, %2" : "+v"(s) : "v"(x1.y), "v"(ones2));
  };
  auto rd_wide = [&](auto j_, auto h_, unsigned addr) {   // half h of wide fragment j from the quarter at `addr`
    constexpr int j = decltype(j_)::v, h = decltype(h_)::v;
    tnb_rd<(WIDE0 + (j >> 2)) * 4096 + h * 1024 + (j & 3) * 256>(fw[j][h], addr);
  };

  // ---- prologue: quarters 0 .. 5 in flight, 0 .. 2 landed (phase 0 reads the fragments of quarters 0, 1 and 2)
  const int npro = nq < 6 ? nq : 6;
  for (int q = 0; q < npro; ++q) { tn_for<0, 5>([&](auto s_) { stage_piece(s_); }); stage_advance(); }
  if (nq >= 6) TNB_WAIT_VM(15); else TNB_WAIT_VM(0);
  TNB_BAR();
  {
    const unsigned a0 = lds0 + lc;
    tn_for<0, 3>([&](auto i_) { constexpr int i = decltype(i_)::v; const unsigned an = lds0 + lcn[i]; tnb_rd<0>(fn[0][i][0], an); tnb_rd<1024>(fn[0][i][1], an); });
    tn_for<0, 8>([&](auto j_) { rd_wide(j_, TnIC<0>{}, a0); rd_wide(j_, TnIC<1>{}, a0); });
  }
  int sr_slot = 1;  // ring slot of quarter q + 1 (the fragment prefetch target)
  cur[0][0] = lds0 + QB + lc;
#pragma unroll
  for (int i = 0; i < 3; ++i) cur[0][1 + i] = lds0 + QB + lcn[i];

  auto quarter = [&](auto u_, auto more_) {
    constexpr int u = decltype(u_)::v; constexpr bool MORE = decltype(more_)::v != 0;
    constexpr bool RD = !(TNB_ABL & 4);
    tn_for<0, 8>([&](auto j_) {
      constexpr int j = decltype(j_)::v;
      if constexpr (j == 0) TNB_WAIT_LGKM(8); else if constexpr (j == 4) TNB_WAIT_LGKM(12);
      tn_for<0, 3>([&](auto i_) {
        constexpr int i = decltype(i_)::v;
        if constexpr (WB) mma(TnIC<3 * j + i>{}, fn[u][i][0], fn[u][i][1], fw[j][0], fw[j][1]);
        else mma(TnIC<3 * j + i>{}, fw[j][0], fw[j][1], fn[u][i][0], fn[u][i][1]);
        // ---- the gap behind MFMA i of triple j
        if constexpr (RD) {
          if constexpr (j == 0) { tnb_rd<0>(fn[u ^ 1][i][0], cur[u][1 + i]); tnb_rd<1024>(fn[u ^ 1][i][1], cur[u][1 + i]); }
          else if constexpr (i < 2) rd_wide(TnIC<(j >= 1 ? j - 1 : 0)>{}, i_, cur[u][0]);
          else if constexpr (j == 7) {
            if constexpr (CS && WB) { if (do_cs && w == 3) dot4(cs[1], fw[7][0], fw[7][1]); }   // tile 7's column sums BEFORE its registers are refilled
            rd_wide(TnIC<7>{}, TnIC<0>{}, cur[u][0]); rd_wide(TnIC<7>{}, TnIC<1>{}, cur[u][0]);
          }
        }
        if constexpr (MORE && i == 2 && j >= 1 && j <= 5) stage_piece(TnIC<(j >= 1 && j <= 5 ? j - 1 : 0)>{});
        if constexpr (i == 2 && j == 6) {  // the next quarter prefetches from the next slot
          sr_slot = sr_slot == R - 1 ? 0 : sr_slot + 1;
          const unsigned sq = lds0 + (unsigned)sr_slot * QB;
          cur[u ^ 1][0] = sq + lc;
#pragma unroll
          for (int k = 0; k < 3; ++k) cur[u ^ 1][1 + k] = sq + lcn[k];
        }
      });
      if constexpr (CS) {
        if constexpr (WB) { if constexpr (j < 7 || (TNB_ABL & 4)) { if (do_cs && w == (j >> 1)) dot4(cs[j & 1], fw[j][0], fw[j][1]); } }   // every wave holds all of B: wave w sums tiles 2w, 2w+1 (tile 7: above)
        else if constexpr (j < 3) { if (do_cs) dot4(cs[j], fn[u][j][0], fn[u][j][1]); }                // the wave's own three B tiles, one per MFMA triple
      }
    });
    if constexpr (MORE) stage_advance();
  };

  const int P = nq >> 1, Pm = nq >= 8 ? (nq - 6) >> 1 : 0;  // phases; those that still stage two quarters (2p + 7 <= nq - 1)
  in_loop = true;
  for (int p = 0; p < Pm; ++p) {
    quarter(TnIC<0>{}, TnIC<1>{}); quarter(TnIC<1>{}, TnIC<1>{});
    if constexpr (!(TNB_ABL & 8)) {
      TNB_WAIT_VM(15);  // this wave's pieces of quarters <= 2p + 4 have landed (three quarters = 15 pieces may still be in flight)
      TNB_BAR();        // ... and everyone's; every wave has left the two slots the next phase refills
    }
  }
  for (int p = Pm; p < P; ++p) {
    quarter(TnIC<0>{}, TnIC<0>{}); quarter(TnIC<1>{}, TnIC<0>{});
    if constexpr (!(TNB_ABL & 8)) { TNB_WAIT_VM(0); TNB_BAR(); }
  }
  TNB_WAIT_VM(0);
  TNB_WAIT_LGKM(0);  // the last quarter's prefetch reads (unused) must have returned before their registers are reused
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // MFMA results -> VALU reads: the compiler cannot see the producers

  if constexpr (CS) {
    if (do_cs) {
#pragma unroll
      for (int k = 0; k < (WB ? 2 : 3); ++k) {
        const float t = cs[k] + __shfl_xor(cs[k], 32, 64);
        const int gn = n0 + (WB ? 32 * (2 * w + k) : 96 * w + 32 * k) + (lane & 31);
        if (lane < 32) grad_add(g.colsum + gn, t);
      }
    }
  }
  // ---- epilogue: 32x32 C/D map col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); one register = two 128-B row segments (the full-rate atomic shape)
  const int col = lane & 31, rb = 4 * (lane >> 5);
  const DetCfg dc = det_load();   // deterministic-gradient switch (common.hpp), read once: nullptr = float atomics
  tn_for<0, 24>([&](auto t_) {
    constexpr int t = decltype(t_)::v; constexpr int j = t / 3, i = t % 3;
    const int it = WB ? 3 * w + i : j, jt = WB ? j : 3 * w + i;   // the tile's place in the workgroup tile
    const int gn0 = n0 + 32 * jt;                                  // wave-uniform: a 32-column group never straddles a segment (seg_n % 32 == 0)
    float* cb = g.C; int cn = gn0 + col;
    if (g.seg_n > 0) { const int sg = gn0 / g.seg_n; if (sg > 0) { cb = g.Cseg[sg > 1]; cn -= sg * g.seg_n; } }
    float* p0 = cb + (int64_t)(i0 + 32 * it + rb) * g.ldc + cn;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v;
      if constexpr (t < 16) v = accA[t][r]; else v = accV[t - 16][r];
      float* pr = p0 + (int64_t)((r & 3) + 8 * (r >> 2)) * g.ldc;
      if (dc.shadow) grad_add(dc, pr, v); else atomicAdd(pr, v);
    }
  });
}

// the M % 32 rows the large-tile kernel leaves: one thread per output element
__global__ __launch_bounds__(256) void gemm_tn_tail_kernel(TnArgs g, int64_t m0, int nrows) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)g.Ki * g.N) return;
  const int n = (int)(idx % g.N), i = (int)(idx / g.N);
  float s = 0.f, sb = 0.f;
  for (int r = 0; r < nrows; ++r) {
    const int64_t m = m0 + r;
    int64_t mb = m; if (g.brow_group > 0) mb = m + (m / g.brow_group + 1) * (int64_t)g.brow_skip;
    const float b = bf2f(g.B[mb * g.ldb + n]);
    s += bf2f(g.A[m * g.lda + i]) * b; sb += b;
  }
  float* cb = g.C; int cn = n;
  if (g.seg_n > 0) { const int sg = n / g.seg_n; if (sg > 0) { cb = g.Cseg[sg > 1]; cn -= sg * g.seg_n; } }
  grad_add(cb + (int64_t)i * g.ldc + cn, s);
  if (g.colsum && i == 0) grad_add(g.colsum + n, sb);
}

template <bool WB, bool CS, bool REMAP>
static void launch_tnb_k(spa3d_ctx* c, const TnArgs& g, unsigned blocks) {
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_tnb_kernel<WB, CS, REMAP>, hipFuncAttributeMaxDynamicSharedMemorySize, TNB_R * TNB_QB); attr = true; }
  gemm_tnb_kernel<WB, CS, REMAP><<<blocks, 256, TNB_R * TNB_QB, c->stream>>>(g);
}
template <bool WB>
static void launch_tnb(spa3d_ctx* c, const TnArgs& g, unsigned blocks) {
  const bool cs = g.colsum != nullptr, rm = g.brow_group > 0;
  if (cs) { if (rm) launch_tnb_k<WB, true, true>(c, g, blocks); else launch_tnb_k<WB, true, false>(c, g, blocks); }
  else { if (rm) launch_tnb_k<WB, false, true>(c, g, blocks); else launch_tnb_k<WB, false, false>(c, g, blocks); }
}

bool gemm_tnb(spa3d_ctx* c, TnArgs g) {
  const bool wb = g.Ki % 384 == 0 && g.N % 256 == 0, wa = g.Ki % 256 == 0 && g.N % 384 == 0;
  if (!wb && !wa) return false;
  if (g.M < 256 || g.lda % 8 || g.ldb % 8) return false;
  if (g.seg_n > 0 && g.seg_n % 32) return false;
  if (g.lda > (1 << 24) || g.ldb > (1 << 24)) return false;                        // 32-bit lane offsets
  if (g.brow_group > 0) {                                                           // the skips a lane accumulates over a split stay in 32 bits
    if (g.brow_group < 16 || (g.M / g.brow_group + 2) * (int64_t)g.brow_skip * g.ldb * 2 >= (int64_t(1) << 31)) return false;
  }
  const int64_t Mmain = g.M / 32 * 32;
  const int TI = wb ? 384 : 256, TNN = wb ? 256 : 384;
  TnArgs gm = g; gm.M = Mmain;
  gm.tiles_i = g.Ki / TI; gm.tiles_n = g.N / TNN;
  const int64_t tiles = (int64_t)gm.tiles_i * gm.tiles_n;
  // M-splits from the makespan model of the 8-wave kernels: rounds on the fullest XCD x rows x time per row + splits x atomic bytes at ~1.3 TB/s
  const double t_row = 2.0 * TI * TNN / 5.0e12, t_atom = (double)g.Ki * g.N * 4.0 / 1.3e12;
  const int64_t smax = std::max<int64_t>(1, std::min<int64_t>(Mmain / 2048, 2048));
  double best = 1e30; int64_t splits = 1;
  for (int64_t sc = 1; sc <= smax; ++sc) {
    const int64_t rps_c = ((Mmain + sc - 1) / sc + 63) / 64 * 64;
    const int64_t per_xcd = tiles * ((sc + 7) / 8);
    const double t = (double)((per_xcd + 31) / 32) * (double)rps_c * t_row + (double)sc * t_atom;
    if (t < best * 0.999) { best = t; splits = sc; }
  }
  const int64_t rps = ((Mmain + splits - 1) / splits + 63) / 64 * 64;
  splits = (Mmain + rps - 1) / rps;
  gm.splits = (int)splits; gm.rows_per_split = rps;
  const unsigned blocks = (unsigned)(tiles * ((splits + 7) / 8 * 8));
  if (wb) launch_tnb<true>(c, gm, blocks); else launch_tnb<false>(c, gm, blocks);
  if (Mmain < g.M) {
    const int64_t el = (int64_t)g.Ki * g.N;
    gemm_tn_tail_kernel<<<(unsigned)((el + 255) / 256), 256, 0, c->stream>>>(g, Mmain, (int)(g.M - Mmain));
  }
  return true;
}

SPA_DET_UPLOAD_DEF(det_upload_gemm_tnb)
}  // namespace SPA_NS
