// gemm_ntb.hip -- C[M, N] = A[M, K] . W (+ bias) (NT: both operands K-contiguous) with a large register tile (gfx950; round 5).
//
// The forward / dX GEMMs whose output is 384 wide (dX of MLP-in and of q|k|v, K = 1536 / 2304) ran on the 8-wave 128 x 384 tile: 64 KiB of LDS-DMA per
// 64-deep k-step for 192 MFMAs (0.33 KiB per MFMA), three quarters of it the weight panel that every tile streams again from L2.  This is the dW kernel's
// recipe (gemm_tnb.hip) turned to NT: ONE wave per SIMD holds 24 MFMA tiles of 32 x 32 = 384 accumulator registers (16 tiles pinned in AGPRs, 8 in VGPRs
// by inline-asm MFMAs), the four waves (2 x 2) cover a workgroup tile of 256 x 384 (wave 128 x 192; "WM, WN = 4, 6") or 384 x 256 (wave 192 x 128; "6, 4"):
// 0.21 KiB of LDS-DMA and 10 fragment reads of 1 KiB per 24 MFMAs.
//
// Operands.  MFMA A operand = W fragment (output i = n), B operand = activation rows (output j = m): a lane ends up with 4 consecutive n of one row m.
//   * activations: row-major [M][K], streamed from HBM by LDS-DMA in k32 "phases": a phase image is [64 WM rows][64 B], one instruction = 16 rows x 64 B
//     (measured: 5.8-6.0 TB/s against 6.1-6.3 for 128-B row pieces, tools/experiments/probe_dma_pieces.hip).  16-B chunk c of row r sits at chunk
//     c ^ ((r >> 2) & 3): the four 16-lane groups of ds_read_b128 ({0-3,12-15,20-27}, ...) then touch every bank once.
//   * W: PRE-PACKED per (n-tile of the workgroup, phase) as 2 x 2 WN fragments of 1 KiB in consumption order (ntb_pack_kernel), so a fragment is one
//     LDS-DMA instruction and one conflict-free ds_read_b128 at (lane address + instruction offset); identical for every m-tile, L2-resident.
// Ring of 4 phases x 40 KiB = 160 KiB (4 WM KiB of rows + 4 WN KiB of W fragments per phase).  Phase p (48 MFMAs per wave between two barriers) issues the
// LDS-DMA of phase p + 3 into the slot phase p - 1 has left and ends with vmcnt(10) (this wave's pieces of phases <= p + 2 landed) + s_barrier; fragment
// reads run one k16 step ahead of the MFMAs (across the barrier: phase p + 1 was guaranteed by the barrier before).
//
// Schedule of a k16 step (24 MFMAs, tile t = 4 j + i: j = the wide operand's fragment (6), i = the narrow operand's (4)):
//     gap (0, i): narrow fragment i of the next step (second register set)
//     gap (j, 0), j >= 1: wide fragment j - 1 of the next step replaces fragment j - 1 (last used by MFMA (j - 1, 3)); gap (5, 3): wide fragment 5
//     gap (j, 2), j = 1 .. 5: one LDS-DMA piece (the wave stages WM row pieces + WN W pieces per phase = 5 per step)
// Two counted waits per step (LDS returns in order): lgkmcnt(3) ahead of MFMA (0, 0), lgkmcnt(6) ahead of MFMA (3, 0).
//
// Persistent: a workgroup walks its tile list (XCD-major: the n-tiles of an m-tile go to one XCD) with the phase pipeline running ACROSS tiles -- the
// next tile's first three phases are in flight during the epilogue, which converts the accumulators, transposes them through the one free ring slot
// (wave-private, 32 rows x 64 columns at a time) and stores whole 128-B row pieces.  The first k-step of a tile uses srcC = 0 (no accumulator clears).
#include <algorithm>

#include "common.hpp"

namespace SPA_NS {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float ntb_f32x4;
typedef mfma16x8 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned ntb_u32x4;

#define NTB_PH 40960
#define NTB_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define NTB_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define NTB_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")
// LDS-DMA: 64 lanes x 16 B from (SGPR base + lane offset + IMM) to LDS at M0 + IMM + 16 lane (the instruction offset moves both addresses)
#define NTB_GLDS(voff, sbase, m0v, IMM) \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(voff), "s"(sbase), "s"(m0v), "n"(IMM) : "memory", "m0")
template <int IMM> __device__ __forceinline__ void ntb_rd(uint4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM) : "memory");
}

constexpr int NTB_ABL = SPA3D_ABL_NTB;  // csrc/ablate.inc: 0 in libspa3d_hip.so
template <int N_> struct NtIC { static constexpr int v = N_; };
template <int I, int N, typename F> __device__ __forceinline__ void nt_for(F&& f) { if constexpr (I < N) { f(NtIC<I>{}); nt_for<I + 1, N>(f); } }

struct NtbArgs {
  const bf16_t* A; int64_t lda;   // [M][K], K contiguous
  const char* wpk;                // ntb_pack_kernel's stream
  bf16_t* C; int64_t ldc;
  const float* bias;              // [N] or nullptr
  int64_t M; int N, K;
  int tiles_n, U;                 // n-tiles of the workgroup tile, phases (K / 32)
  int64_t ntiles;
};

// ---- weight stream: [n-tile tn][phase p][step s][fragment t < 2 WN][lane][8]: lane (r = lane & 31, hh = lane >> 5), element j  ->
// W[k = 32 p + 16 s + 8 hh + j][n = 64 WN tn + 32 t + r];  W element (k, n) is read from w[k * sk + n * sn]
template <typename S>
__global__ void ntb_pack_kernel(const S* __restrict__ w, int64_t sk, int64_t sn, int U, int WN, int64_t total, bf16_t* __restrict__ out) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= total) return;
  const int lane = (int)(id & 63); int64_t f = id >> 6;
  const int t = (int)(f % (2 * WN)); f /= 2 * WN;
  const int s = (int)(f & 1); f >>= 1;
  const int p = (int)(f % U); const int tn = (int)(f / U);
  const int n = 64 * WN * tn + 32 * t + (lane & 31), k0 = 32 * p + 16 * s + 8 * (lane >> 5);
  bf16_t v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = f2bf(ld<S>(w + (int64_t)(k0 + j) * sk + (int64_t)n * sn));
  uint4 o;
  o.x = (unsigned)v[0] | ((unsigned)v[1] << 16); o.y = (unsigned)v[2] | ((unsigned)v[3] << 16);
  o.z = (unsigned)v[4] | ((unsigned)v[5] << 16); o.w = (unsigned)v[6] | ((unsigned)v[7] << 16);
  *(uint4*)(out + id * 8) = o;
}

template <int WM, int WN, bool BIAS>
__global__ __launch_bounds__(256, 1) void gemm_ntb_kernel(NtbArgs g) {
  constexpr int TM = 64 * WM, TNN = 64 * WN;
  constexpr int AB = 4096 * WM;       // bytes of the activation image of a phase (the W fragments follow)
  constexpr int WSTEP = 2048 * WN;    // W bytes of one k16 step
  constexpr bool NA = WM < WN;        // narrow operand (4 fragments, two register sets) = activation rows, wide (6, rolling) = W; else the other way round
  static_assert((WM == 4 && WN == 6) || (WM == 6 && WN == 4), "24 MFMA tiles per wave as 4 x 6 or 6 x 4");
  static_assert(4 * WM + 4 * WN == 40, "a phase slot is 40 KiB");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wr = w >> 1, wc = w & 1;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;

  // ---- tile list: round i of this workgroup = tile ((8 i + xcd) per + jx): the 32 workgroups of an XCD take consecutive tiles (n fastest)
  const int per = (int)gridDim.x >> 3, xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  int nmine = 0;
  {
    const int rem = (int)g.ntiles - jx;
    if (rem > 0) { const int q = (rem + per - 1) / per; if (q > xcd) nmine = (q - xcd + 7) >> 3; }
  }
  if (nmine == 0) return;
  auto tile_of = [&](int i, int& tm, int& tn) {
    const int id = (i * 8 + xcd) * per + jx;   // < ntiles + 8 per < 2^31 (host)
    tm = id / g.tiles_n; tn = id - tm * g.tiles_n;
  };

  // ---- staging cursor: the (tile, phase) whose pieces are issued next.  Row pieces: piece q of the wave = rows 16 (WM w + q) .. + 15, lane -> row + (lane >> 2),
  // physical chunk lane & 3 = logical chunk (lane & 3) ^ ((lane >> 4) & 3).  Rows past M read row M - 1 (never stored).
  unsigned voffA[WM];
  const unsigned voffW = (unsigned)(1024 * WN * w + 16 * lane), voffW2 = voffW + 4096u;
  const char* sA = nullptr; const char* sW = nullptr;
  int ci = 0, cp = 0, si = 0;   // cursor round, cursor phase, ring slot the cursor's phase goes to
  auto cursor_tile = [&]() {
    int tm, tn; tile_of(ci < nmine ? ci : nmine - 1, tm, tn);
    const int64_t row0 = (int64_t)tm * TM;
    const int lim = (int)std::min<int64_t>(TM - 1, g.M - 1 - row0);
    const int lc = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
    for (int q = 0; q < WM; ++q) {
      int r = 16 * (WM * w + q) + (lane >> 2); r = r < lim ? r : lim;
      voffA[q] = (unsigned)(((int64_t)r * g.lda + 8 * lc) * 2);
    }
    sA = (const char*)(g.A + row0 * g.lda);
    sW = g.wpk + (int64_t)tn * g.U * (4096 * WN);
  };
  bool in_loop = false;
  auto stage_piece = [&](auto q_) {
    constexpr int q = decltype(q_)::v;
    if ((NTB_ABL & 1) && in_loop) return;
    const unsigned sb = lds0 + (unsigned)si * NTB_PH;
    if constexpr (q < WM) {
      const unsigned m0v = sb + (unsigned)(1024 * (WM * w + q));
      const unsigned vo = voffA[q]; const char* sbp = sA;
      NTB_GLDS(vo, sbp, m0v, 0);
    } else {
      constexpr int j = q - WM; constexpr int IMM = 1024 * (j & 3);
      const unsigned m0v = sb + (unsigned)(AB + 1024 * (WN * w + j) - IMM);
      const unsigned vo = j < 4 ? voffW : voffW2; const char* sbp = sW;
      NTB_GLDS(vo, sbp, m0v, IMM);
    }
  };
  auto stage_advance = [&]() {
    sA += 64; sW += 4096 * WN; si = (si + 1) & 3;
    if (++cp == g.U) { cp = 0; ++ci; cursor_tile(); }
  };

  // ---- fragment read addresses.  Rows image: lane (r = lane & 31, hh = lane >> 5) of m-tile mt reads row 32 (WM wr + mt) + r, chunk (2 s + hh) ^ ((r >> 2) & 3):
  // one lane constant, XOR 32 for step 1, + 2048 mt as the instruction offset.  W fragments: AB + 1024 (WN wc + nt) + 16 lane, + WSTEP for step 1.
  const int fr = lane & 31, fh = lane >> 5;
  const unsigned laA = (unsigned)((32 * WM * wr + fr) * 64 + ((fh ^ ((fr >> 2) & 3)) * 16));
  const unsigned laW = (unsigned)(AB + 1024 * WN * wc + 16 * lane);
  unsigned cA1, cW, nA, nW;   // this phase's slot (step 1: cA1 = base + (laA ^ 32), cW = base + laW), the next phase's slot (step 0)
  int rs = 0;                 // ring slot of the phase being computed

  f32x16 accA[16], accV[8];   // tile t = 4 j + i: t < 16 in AGPRs, the rest in VGPRs
  uint4 fn[2][4], fw[6];

  auto mma = [&](auto t_, auto first_, const uint4& wf, const uint4& af) {
    constexpr int t = decltype(t_)::v; constexpr bool FIRST = decltype(first_)::v != 0;
    const bf16x8 av = __builtin_bit_cast(bf16x8, wf);   // MFMA A operand: the W fragment (i = n)
    const bf16x8 bv = __builtin_bit_cast(bf16x8, af);   // MFMA B operand: activation rows (j = m)
    if constexpr (NTB_ABL & 2) asm volatile("" ::"v"(av), "v"(bv));
    else if constexpr (t < 16) {
      f32x16& acc = accA[t];
      if constexpr (FIRST) asm volatile(MFMA32_ASM " %0, %1, %2, 0" : "=a"(acc) : "v"(av), "v"(bv));
      else asm volatile(MFMA32_ASM " %0, %1, %2, %0" : "+a"(acc) : "v"(av), "v"(bv));
    } else {
      f32x16& acc = accV[t - 16];
      if constexpr (FIRST) asm volatile(MFMA32_ASM " %0, %1, %2, 0" : "=v"(acc) : "v"(av), "v"(bv));
      else asm volatile(MFMA32_ASM " %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv));
    }
  };
  // fragment i of the narrow / j of the wide operand of step `sp` of the slot whose addresses are (aA, aW): aA already carries the step's XOR, aW does not
  auto rd_narrow = [&](auto i_, auto sp_, uint4& dst, unsigned aA, unsigned aW) {
    constexpr int i = decltype(i_)::v, sp = decltype(sp_)::v;
    if constexpr (NA) ntb_rd<2048 * i>(dst, aA); else ntb_rd<sp * WSTEP + 1024 * i>(dst, aW);
  };
  auto rd_wide = [&](auto j_, auto sp_, uint4& dst, unsigned aA, unsigned aW) {
    constexpr int j = decltype(j_)::v, sp = decltype(sp_)::v;
    if constexpr (NA) ntb_rd<sp * WSTEP + 1024 * j>(dst, aW); else ntb_rd<2048 * j>(dst, aA);
  };

  // one k16 step: MFMAs on (fn[sp], fw), prefetch of the next step's fragments into (fn[sp ^ 1], fw): step 1 of this slot (sp = 0) or step 0 of the next slot
  auto step = [&](auto sp_, auto first_) {
    constexpr int sp = decltype(sp_)::v;
    constexpr bool RD = !(NTB_ABL & 4);
    nt_for<0, 6>([&](auto j_) {
      constexpr int j = decltype(j_)::v;
      if constexpr (!(NTB_ABL & 8)) { if constexpr (j == 0) NTB_WAIT_LGKM(3); else if constexpr (j == 3) NTB_WAIT_LGKM(6); }
      nt_for<0, 4>([&](auto i_) {
        constexpr int i = decltype(i_)::v;
        if constexpr (NA) mma(NtIC<4 * j + i>{}, first_, fw[j], fn[sp][i]); else mma(NtIC<4 * j + i>{}, first_, fn[sp][i], fw[j]);
        // ---- the gap behind MFMA (j, i)
        const unsigned aA = sp == 0 ? cA1 : nA, aW = sp == 0 ? cW : nW;
        if constexpr (RD) {
          if constexpr (j == 0) rd_narrow(i_, NtIC<sp ^ 1>{}, fn[sp ^ 1][i], aA, aW);
          else if constexpr (i == 0) rd_wide(NtIC<(j >= 1 ? j - 1 : 0)>{}, NtIC<sp ^ 1>{}, fw[j >= 1 ? j - 1 : 0], aA, aW);
          else if constexpr (j == 5 && i == 3) rd_wide(NtIC<5>{}, NtIC<sp ^ 1>{}, fw[5], aA, aW);
        }
        if constexpr (i == 2 && j >= 1) stage_piece(NtIC<5 * sp + (j >= 1 ? j - 1 : 0)>{});
        if constexpr (sp == 0 && i == 1 && j == 2) {   // the next phase's slot: read addresses of its step 0
          const unsigned sq = lds0 + (unsigned)((rs + 1) & 3) * NTB_PH;
          nA = sq + laA; nW = sq + laW;
        }
        if constexpr (sp == 1 && i == 1 && j == 5) { cA1 = nA ^ 32u; cW = nW; rs = (rs + 1) & 3; }   // (the reads of this step are all issued by gap (5, 0) except wide 5, which uses nA / nW)
      });
    });
  };

  // ---- prologue: phases 0, 1, 2 of the first tile(s) in flight, 0 and 1 landed; fragments of step 0
  cursor_tile();
  for (int p = 0; p < 3; ++p) { nt_for<0, 10>([&](auto q_) { stage_piece(q_); }); stage_advance(); }
  NTB_WAIT_VM(10);
  NTB_BAR();
  {
    const unsigned a0 = lds0 + laA, w0 = lds0 + laW;
    nt_for<0, 4>([&](auto i_) { constexpr int i = decltype(i_)::v; rd_narrow(i_, NtIC<0>{}, fn[0][i], a0, w0); });
    nt_for<0, 6>([&](auto j_) { constexpr int j = decltype(j_)::v; rd_wide(j_, NtIC<0>{}, fw[j], a0, w0); });
    cA1 = (lds0 + laA) ^ 32u; cW = lds0 + laW; nA = cA1; nW = cW;
  }
  in_loop = true;

  const int er = lane & 31, ehh = lane >> 5;
  for (int ti = 0; ti < nmine; ++ti) {
    int tm, tn; tile_of(ti, tm, tn);
    const int64_t row0 = (int64_t)tm * TM;
    // phase 0 is peeled: its first k-step writes the accumulators (srcC = 0), so the accumulator values flow straight from here into the loop
    step(NtIC<0>{}, NtIC<1>{}); step(NtIC<1>{}, NtIC<0>{});
    stage_advance();
    if constexpr (!(NTB_ABL & 8)) {
      // this wave's pieces of phases <= p + 2 have landed (the 10 of phase p + 3 may be in flight).  Behind an epilogue its 48 stores (buffer stores: always
      // issued, rows past M dropped by the range check) may still be draining: they are younger than every piece this barrier stands for
      if (ti > 0) NTB_WAIT_VM(58); else NTB_WAIT_VM(10);
      NTB_BAR();
    }
    // do-while (U >= 2, host): with a skippable loop the epilogue's accumulators become phis of (peeled phase, loop) and hipcc spills 23 of the 24 tiles at the loop exit
    { int p = 1; do {
      step(NtIC<0>{}, NtIC<0>{}); step(NtIC<1>{}, NtIC<0>{});
      stage_advance();
      if constexpr (!(NTB_ABL & 8)) { NTB_WAIT_VM(10); NTB_BAR(); }
    } while (++p < g.U); }
    // ---- epilogue.  32x32 C/D map: j (= m) = lane & 31, i (= n) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5): register group gq = r >> 2 holds n = 8 gq + 4 hh .. + 3
    NTB_WAIT_LGKM(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results -> VALU reads: the compiler cannot see the producers
    char* stg = smem + ((rs + 3) & 3) * NTB_PH + w * 10240;   // the slot of the phase just finished: nothing is in flight to it until the next phase issues
    const int ncol0 = tn * TNN + 32 * WN * wc;
    // the tile's rows of C as a raw buffer: offsets past the last valid row fall outside num_records and the store is dropped -- no branch, and the
    // store count the next phase's vmcnt allowance relies on is exact
    const int vrows = (int)std::min<int64_t>(TM, g.M - row0);
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((void*)(g.C + row0 * g.ldc), 0, (int)(vrows * g.ldc * 2), 0x00020000);
    const int cbase = (int)(((32 * WM * wr + (lane >> 3)) * g.ldc + ncol0 + 8 * (lane & 7)) * 2);
    // 12 rounds of (m-tile mt, n-tile pair np) = 32 rows x 64 columns through a 4.5-KiB buffer (two buffers, alternating), software-pipelined: the staged
    // rows of round r are read back, round r + 1 is converted and written while those reads are in flight, then round r is stored
    constexpr int NR = WM * (WN / 2);
    auto conv_write = [&](auto r_) {
      constexpr int r = decltype(r_)::v, mt = r / (WN / 2), np = r % (WN / 2);
      char* buf = stg + (r & 1) * 4608;
      nt_for<0, 2>([&](auto q_) {
        constexpr int q = decltype(q_)::v, nt = 2 * np + q;
        constexpr int t = NA ? 4 * nt + mt : 4 * mt + nt;
        nt_for<0, 4>([&](auto gq_) {   // compile-time indices only: a loop LLVM declines to unroll would index the accumulator arrays dynamically (= scratch)
          constexpr int gq = decltype(gq_)::v;
          float v0, v1, v2, v3;
          if constexpr (t < 16) { v0 = accA[t][4 * gq]; v1 = accA[t][4 * gq + 1]; v2 = accA[t][4 * gq + 2]; v3 = accA[t][4 * gq + 3]; }
          else { v0 = accV[t - 16][4 * gq]; v1 = accV[t - 16][4 * gq + 1]; v2 = accV[t - 16][4 * gq + 2]; v3 = accV[t - 16][4 * gq + 3]; }
          if constexpr (BIAS) {
            const ntb_f32x4 b4 = *(const ntb_f32x4*)(g.bias + ncol0 + 32 * nt + 8 * gq + 4 * ehh);
            v0 += b4[0]; v1 += b4[1]; v2 += b4[2]; v3 += b4[3];
          }
          uint2 pk;
          pk.x = f2bf_pack2(v0, v1); pk.y = f2bf_pack2(v2, v3);
          *(uint2*)(buf + er * 144 + (32 * q + 8 * gq + 4 * (ehh ^ ((er >> 3) & 1))) * 2) = pk;   // rows r and r + 8 share banks at this stride: their 8-B halves of a 16-B chunk are swapped
        });
      });
    };
    conv_write(NtIC<0>{});
    nt_for<0, NR>([&](auto r_) {
      constexpr int r = decltype(r_)::v, mt = r / (WN / 2), np = r % (WN / 2);
      const char* buf = stg + (r & 1) * 4608;
      ntb_u32x4 rv[4];
      __builtin_amdgcn_sched_barrier(0);
      nt_for<0, 4>([&](auto it_) {
        constexpr int it = decltype(it_)::v;
        const ntb_u32x4 t4 = *(const ntb_u32x4*)(buf + (8 * it + (lane >> 3)) * 144 + (lane & 7) * 16);
        rv[it] = (it & 1) ? ntb_u32x4{t4[2], t4[3], t4[0], t4[1]} : t4;   // rows 8 .. 15 and 24 .. 31 of the round were written with the halves swapped
      });
      __builtin_amdgcn_sched_barrier(0);   // (the fences keep one round's accumulator copies live at a time: left free, hipcc hoists all 256 AGPR reads and spills)
      if constexpr (r + 1 < NR) conv_write(NtIC<(r + 1 < NR ? r + 1 : 0)>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(NTB_ABL & 16)) {
        nt_for<0, 4>([&](auto it_) {
          constexpr int it = decltype(it_)::v;
          __builtin_amdgcn_raw_buffer_store_b128(rv[it], crs, cbase + (int)((32 * mt + 8 * it) * g.ldc * 2) + 128 * np, 0, 0);
        });
      }
    });
    NTB_BAR();   // every wave has left the staging slot before the next phase's LDS-DMA lands in it
    {  // the next tile's step-0 fragments again: the copies the last step prefetched are not kept across the epilogue (40 registers the staging needs)
      const unsigned a0 = cA1 ^ 32u, w0 = cW;
      nt_for<0, 4>([&](auto i_) { constexpr int i = decltype(i_)::v; rd_narrow(i_, NtIC<0>{}, fn[0][i], a0, w0); });
      nt_for<0, 6>([&](auto j_) { constexpr int j = decltype(j_)::v; rd_wide(j_, NtIC<0>{}, fw[j], a0, w0); });
    }
  }
  NTB_WAIT_VM(0);    // the cursor ran three phases past the end (re-reading the last tile): nothing may be in flight into LDS when the workgroup ends
  NTB_WAIT_LGKM(0);
}

// ---- host
int ntb_wn(int N) { return N % 384 == 0 ? 6 : (N % 256 == 0 ? 4 : 0); }
bool gemm_ntb_ok(int K, int N) { return ntb_wn(N) != 0 && K % 32 == 0 && K >= 64; }   // K >= 64: the kernel's phase loop is a do-while behind the peeled first phase
int64_t gemm_ntb_pack_elems(int K, int N) { return (int64_t)K * N; }

template <typename S> void gemm_ntb_pack(spa3d_ctx* c, const S* w, int64_t sk, int64_t sn, int K, int N, bf16_t* wpk) {
  if (c->dry) return;
  const int WN = ntb_wn(N);
  const int64_t total = (int64_t)K * N / 8;
  ntb_pack_kernel<S><<<(unsigned)((total + 255) / 256), 256, 0, c->stream>>>(w, sk, sn, K / 32, WN, total, wpk);
  SPA_LAUNCH_CHECK(c);
}
template void gemm_ntb_pack<float>(spa3d_ctx*, const float*, int64_t, int64_t, int, int, bf16_t*);
template void gemm_ntb_pack<bf16_t>(spa3d_ctx*, const bf16_t*, int64_t, int64_t, int, int, bf16_t*);

template <int WM, int WN, bool BIAS>
static void launch_ntb(spa3d_ctx* c, NtbArgs g) {
  constexpr int TM = 64 * WM, TNN = 64 * WN;
  g.tiles_n = g.N / TNN; g.U = g.K / 32;
  g.ntiles = ((g.M + TM - 1) / TM) * g.tiles_n;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_ntb_kernel<WM, WN, BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * NTB_PH); attr = true; }
  gemm_ntb_kernel<WM, WN, BIAS><<<256, 256, 4 * NTB_PH, c->stream>>>(g);
}

// C[M, N] (16-bit) = A[M, K] . W (+ bias) with W as gemm_ntb_pack's stream.  False when the shape / layout is not this kernel's.
bool gemm_ntb(spa3d_ctx* c, const bf16_t* A, int64_t lda, const bf16_t* wpk, const float* bias, bf16_t* C, int64_t ldc, int64_t M, int N, int K) {
  if (!wpk || !gemm_ntb_ok(K, N) || M < 1) return false;
  if (lda % 8 || ldc % 8 || lda > (1 << 20) || ldc > (1 << 20) || (((uintptr_t)A | (uintptr_t)C | (uintptr_t)wpk) & 15) || (bias && ((uintptr_t)bias & 15))) return false;
  if (((M + 255) / 256) * (int64_t)(N / 256 + 1) > (int64_t(1) << 30)) return false;   // the kernel's tile ids are 32-bit
  if (c->dry) return true;
  NtbArgs g{};
  g.A = A; g.lda = lda; g.wpk = (const char*)wpk; g.C = C; g.ldc = ldc; g.bias = bias; g.M = M; g.N = N; g.K = K;
  ProfScope ps(c, PROF_GEMM_NT, 2.0 * (double)M * N * K, ((double)M * K + (double)K * N + (double)M * N) * 2.0);
  ps.tag(M, N, K, (1 << 21) | (bias ? 1 : 0));
  if (ntb_wn(N) == 6) { if (bias) launch_ntb<4, 6, true>(c, g); else launch_ntb<4, 6, false>(c, g); }
  else { if (bias) launch_ntb<6, 4, true>(c, g); else launch_ntb<6, 4, false>(c, g); }
  SPA_LAUNCH_CHECK(c);
  return true;
}

}  // namespace SPA_NS
