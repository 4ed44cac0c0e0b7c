// gemm_rs.hip -- "row-stationary" NT GEMM for the K = 384 projections of the d = 384 blocks (QKV, attention.py:154-173; the cross-attention
// K/V projection of tracks_to_latents; dX = dY . W^T products whose contraction is 384 wide):   C[M, N] = A[M, 384] . W[384, N] (+ bias).
//
// These GEMMs move 768 B of A and 2 N B of C per row for 768 N FLOP: at N = 2304 the tiled 256x256 kernel spends MFMA time + the HBM time of
// its 128-KB store burst per tile, in series (NOTEBOOK.md, "What does not overlap").  Here the A rows never touch LDS and the stores never burst:
//   * persistent workgroup = 4 waves = one wave per SIMD; a wave owns 64 rows of a 256-row tile and holds them as 2 x 24 MFMA B fragments
//     (192 registers) for the whole tile -- A is read from HBM exactly once, as rows;
//   * W comes as a pre-packed stream of 1-KiB MFMA A fragments in consumption order (N / 64 segments of 48 KiB, L2-resident, identical for
//     every tile) through a 3-slot LDS ring by LDS-DMA (the ring / counted-vmcnt / one-barrier-per-phase skeleton of mlp_fused.hip);
//     every fragment read from LDS feeds TWO MFMAs (the wave's two 32-row blocks): half the LDS bytes and half the LDS-DMA bytes per MFMA
//     of the 128-row fused MLP kernel;
//   * phase c = the 96 MFMAs (32x32x16) of output columns [64c, 64c + 64) into one of two ping-pong accumulator sets, while the OTHER set --
//     columns of phase c-1 -- is converted, staged through the wave's own 12 KiB of the ring slot this phase refills, and stored as whole
//     128-byte row pieces: 8 stores of 1 KiB per wave and phase, spread over the MFMAs of the next phase.  No epilogue burst exists.
#include <cstdlib>

#include "common.hpp"

namespace SPA_NS {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef mfma16x8 bf16x8;
#if SPA_F16
typedef __attribute__((ext_vector_type(2))) _Float16 rs_h16x2;
#else
typedef __attribute__((ext_vector_type(2))) __bf16 rs_h16x2;
#endif

constexpr int RS_K = 384;
constexpr int RS_SEG = 48 * 1024;   // one phase's weights: 48 fragments of 1 KiB = 64 output columns x 384 k
constexpr int RS_RING = 3 * RS_SEG;
constexpr int RS_MAXN = 2304;       // bias rows in LDS: 144 KiB ring + 9 KiB
constexpr int RS_LDS = RS_RING + RS_MAXN * 4;

// ---- weight stream.  Segment c (output columns 64c .. 64c+63), fragment f = t*24 + s (1 KiB = 64 lanes x 8 elements), lane (r = lane & 31, hh = lane >> 5),
// element j  ->  W[k = 16s + 8hh + j][n = 64c + 32t + r];  W element (k, n) is read from w[k * sk + n * sn] (sk = N, sn = 1: a [K][N] matrix; sk = 1, sn = ld: the
// transposed view of an [N][K'] matrix, i.e. the dX = dY . W^T products)
template <typename S>
__global__ void rs_pack_kernel(const S* __restrict__ w, int64_t sk, int64_t sn, int nseg, bf16_t* __restrict__ out) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= nseg * 48 * 64) return;
  const int lane = id & 63, frag = (id >> 6) % 48, seg = id / (64 * 48);
  const int r = lane & 31, hh = lane >> 5, t = frag / 24, s = frag % 24, n = 64 * seg + 32 * t + r;
  bf16_t v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = f2bf(ld<S>(w + (int64_t)(16 * s + 8 * hh + j) * sk + (int64_t)n * sn));
  u32x4 p;
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = (unsigned)v[2 * j] | ((unsigned)v[2 * j + 1] << 16);
  *(u32x4*)(out + (int64_t)id * 8) = p;
}

struct RsArgs {
  const bf16_t* A; int64_t lda; bf16_t* C; int64_t ldc; const char* wpk; const float* bias;
  const bf16_t* aux; int64_t ldaux;  // AUX kernels: C = (A . W + bias) o gelu'(aux), aux in C's layout (the MLP backward's dh = (dy . W_out^T) o gelu'(hpre))
  int64_t M; int N, nseg, tiles, nt_store;
  unsigned long long* dbg;  // diagnostic builds with mask 32 only: per-wave cycle sums (s_memtime), else unused
};
// diagnostic builds only (tools/ablate_gemm_rs.py compiles a SEPARATE library per mask with -DSPA3D_ABLATION_BUILD -DSPA3D_ABL_RS=mask, never the product: csrc/ablate.inc): compile-time mask, 1 stores wrapped into a 1-MiB window,
// 2 (AUX) no gelu' arithmetic, 64 (AUX) no aux loads, 4 no LDS-DMA, 8 no MFMAs, 16 no stores and no staging, 32 s_memtime stamps (per-wave sums to RsArgs::dbg), 128 no counted waits / 256 no barriers (WRONG results: timing only)
constexpr int RS_ABL = SPA3D_ABL_RS;  // csrc/ablate.inc: 0 in libspa3d_hip.so
__device__ __forceinline__ unsigned long long rs_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

__device__ __forceinline__ unsigned rs_lane() {  // recomputed where used (mlp_fused.hip lane_now): lane constants must not be hoisted out of the tile loop and spilled
  unsigned l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(l));
  return l;
}
__device__ __forceinline__ unsigned rs_pack2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, rs_h16x2)); }
template <int OFF> __device__ __forceinline__ void rs_glds(const void* base_uniform, unsigned off, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(off), "s"(base_uniform), "s"(lds_dst), "n"(OFF) : "memory", "m0");
}
#define RS_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define RS_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
template <int N_> struct RsIC { static constexpr int v = N_; };
// compile-time loop (a `#pragma unroll` loop this large falls under LLVM's pragma-unroll size threshold, stays a loop, and the accumulator arrays go to scratch)
template <int I, int N, typename F> __device__ __forceinline__ void rs_for(F&& f) { if constexpr (I < N) { f(RsIC<I>{}); rs_for<I + 1, N>(f); } }

template <bool AUX>
__global__ __launch_bounds__(256, 1) void gemm_rs_kernel(RsArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [3][48 KiB] ring | bias f32[N]
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* sbias = (float*)(smem + RS_RING);
  for (int i = tid; i < g.N; i += 256) sbias[i] = g.bias ? g.bias[i] : 0.f;
  __syncthreads();
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  int seg = 0, slot = 0;  // segment the next phase consumes and the ring slot it sits in
  unsigned long long tsum[4] = {0, 0, 0, 0};  // mask 32: barrier, phase body, end wait, tile head (A rows)

  // piece i (0..11) of this wave's twelve 1-KiB pieces of segment sg into slot sl (four consecutive pieces share a base / M0 value through the immediate offset)
  auto dma1 = [&](int sg, int sl, auto i_, unsigned lane16) {
    constexpr int i = decltype(i_)::v;
    if constexpr (RS_ABL & 4) return;
    const int p0 = w * 12 + (i & ~3);
    rs_glds<(i & 3) * 1024>(g.wpk + (int64_t)sg * RS_SEG + p0 * 1024, lane16, lds0 + (unsigned)(sl * RS_SEG + p0 * 1024));
  };
  {  // prologue: segments 0 and 1 (nseg >= 2)
    const unsigned l16 = rs_lane() * 16u;
    dma1(0, 0, RsIC<0>(), l16); dma1(0, 0, RsIC<1>(), l16); dma1(0, 0, RsIC<2>(), l16); dma1(0, 0, RsIC<3>(), l16);
    dma1(0, 0, RsIC<4>(), l16); dma1(0, 0, RsIC<5>(), l16); dma1(0, 0, RsIC<6>(), l16); dma1(0, 0, RsIC<7>(), l16);
    dma1(0, 0, RsIC<8>(), l16); dma1(0, 0, RsIC<9>(), l16); dma1(0, 0, RsIC<10>(), l16); dma1(0, 0, RsIC<11>(), l16);
    dma1(1, 1, RsIC<0>(), l16); dma1(1, 1, RsIC<1>(), l16); dma1(1, 1, RsIC<2>(), l16); dma1(1, 1, RsIC<3>(), l16);
    dma1(1, 1, RsIC<4>(), l16); dma1(1, 1, RsIC<5>(), l16); dma1(1, 1, RsIC<6>(), l16); dma1(1, 1, RsIC<7>(), l16);
    dma1(1, 1, RsIC<8>(), l16); dma1(1, 1, RsIC<9>(), l16); dma1(1, 1, RsIC<10>(), l16); dma1(1, 1, RsIC<11>(), l16);
  }
  RS_WAIT_VM(12);  // segment 0 has landed (this wave's pieces; the first phase's barrier covers the others')

  bf16x8 nb[2][24];   // this wave's 64 A rows: row block rb, k-step s: lane (r, hh) holds A[row 32 rb + r][16 s + 8 hh .. + 7]
  f32x16 S[2][2][2];  // [ping-pong][row block][32-column half]: register 4q+e = column 8q + 4hh + e of the half, of the lane's row
  int64_t st_row0 = 0; int st_col = 0; bool st_on = false, st_edge = false;  // the chunk waiting in the other accumulator set: its rows, columns, and whether stores are masked
  // AUX: aux values as the read-back steps want them (step j = rows 16 (j & 3) + (ln >> 2), columns 32 (j >> 2) + 8 (ln & 3) .. + 7), two chunks in flight: the phase that
  // computes chunk c stores chunk c-1 with ax[(c-1) & 1] and reloads that set for chunk c+1.  The TWO-phase lead is for the compiler's waits: it counts the aux loads but
  // cannot see the LDS-DMA instructions between them, so its s_waitcnt vmcnt(<= 7) ahead of a use also retires every LDS-DMA issued since -- with a one-phase lead that
  // was everything issued ~6 MFMA groups earlier (6.4 ms at N = 1536, 4.3 ms with the loads removed); now its window is half a phase older.
  u32x4 ax[2][8];

  auto load_a = [&](int tile) {  // k-step-major: the first phase's MFMA group i needs fragments 2i, 2i+1 of both row blocks -- they arrive in that order
    const unsigned ln = rs_lane(); const int r = ln & 31, hh = ln >> 5;
    const bf16_t* ap[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      int64_t row = (int64_t)tile * 256 + w * 64 + rb * 32 + r; if (row > g.M - 1) row = g.M - 1;
      ap[rb] = g.A + row * g.lda + hh * 8;
    }
#pragma unroll
    for (int s = 0; s < 24; ++s) { nb[0][s] = *(const bf16x8*)(ap[0] + 16 * s); nb[1][s] = *(const bf16x8*)(ap[1] + 16 * s); }
  };

  // One phase on ring slot `slot`.  MF: the 96 MFMAs of output columns [64 cc, 64 cc + 64) into S[BUF] (bias as the first MFMA's C operand).  The chunk in S[BUF ^ 1] (if
  // st_on) leaves meanwhile through the wave's own 12 KiB of the slot this phase refills (image [64 rows][128 B + 16] = 9 KiB): packed to 16 bit and written in groups 0-3,
  // read back as whole 128-byte row pieces and stored, 8 rows per instruction, interleaved with the LDS-DMA of segment seg + 2 (schedules below).  (nx_row0, nx_col): AUX only,
  // the rows / first column of the chunk the NEXT phase computes (its aux values are requested two phases ahead of their use).
  auto phase = [&](auto buf_, auto mf_, int cc, int64_t nx_row0, int nx_col) {
    constexpr int BUF = decltype(buf_)::v; constexpr bool MF = decltype(mf_)::v;
    unsigned long long t0 = 0, t1 = 0;
    if constexpr (RS_ABL & 32) t0 = rs_stamp();
    if constexpr (!(RS_ABL & 256)) RS_BAR();  // segment `seg` is visible to every wave; every wave has left the slot this phase refills
    if constexpr (RS_ABL & 32) { t1 = rs_stamp(); tsum[0] += t1 - t0; }
    const unsigned ln = rs_lane(); const int hh = ln >> 5;
    const unsigned l16 = ln * 16u;
    const char* sb = smem + slot * RS_SEG + l16;
    const int sg2 = seg + 2 >= g.nseg ? seg + 2 - g.nseg : seg + 2, sl2 = slot == 0 ? 2 : slot - 1;
    char* stg = smem + sl2 * RS_SEG + w * 12288;
    bf16x8 fa[3][2];
    if constexpr (MF) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { fa[0][i] = *(const bf16x8*)(sb + i * 1024); fa[1][i] = *(const bf16x8*)(sb + (2 + i) * 1024); }
    }
    u32x4 sv;
    auto stage_wr = [&](int rb, int t) {  // S[BUF ^ 1][rb][t] -> rows 32 rb + r, bytes 64 t + 16 q + 8 hh
      char* wp = stg + (rb * 32 + (ln & 31)) * 144 + 64 * t + 8 * hh;
      const f32x16& X = S[BUF ^ 1][rb][t];
#pragma unroll
      for (int q = 0; q < 4; ++q) *(u32x2*)(wp + 16 * q) = u32x2{rs_pack2(X[4 * q], X[4 * q + 1]), rs_pack2(X[4 * q + 2], X[4 * q + 3])};
    };
    auto stage_rd = [&](int k) { sv = *(const u32x4*)(stg + (8 * k + (ln >> 3)) * 144 + (ln & 7) * 16); };
    // this lane's store address of row group 0 (row st_row0 + (ln >> 3), 16-byte piece ln & 7 of the chunk's 128 B); row group k is 8 rows further
    char* cp0 = (char*)(g.C + (st_row0 + (ln >> 3)) * g.ldc + st_col + (ln & 7) * 8);
    const int64_t cstep = g.ldc * 16;  // bytes per 8 rows
    auto stage_st = [&](int k) {
      u32x4* dp = (u32x4*)(cp0 + k * cstep);
      if constexpr (RS_ABL & 1) dp = (u32x4*)((char*)g.C + ((uintptr_t)((char*)dp - (char*)g.C) & 0xffff0));
      if (st_edge) {  // uniform: only a tile that runs past M masks its stores
        if (st_row0 + 8 * k + (int64_t)(ln >> 3) < g.M) { if (g.nt_store) __builtin_nontemporal_store(sv, dp); else *dp = sv; }
      } else { if (g.nt_store) __builtin_nontemporal_store(sv, dp); else *dp = sv; }
    };
    // AUX: the waiting chunk goes through the image as f32, one 32-column half at a time ([64 rows][32 x 4 B + 16]: the same 9 KiB), so that the product with
    // gelu'(aux) is formed in f32 and rounded once, as the tiled kernel's epilogue does.  Read-back step j: rows 16 (j & 3) + (ln >> 2), 8 columns per lane:
    // 16 rows x 64 B per store instruction; ax[j] is reloaded for the chunk being computed right behind its use.
    f32x4 v0, v1;
    auto stage_wr32 = [&](int rb, int t) {
      char* wp = stg + (rb * 32 + (ln & 31)) * 144 + 16 * hh;
      const f32x16& X = S[BUF ^ 1][rb][t];
#pragma unroll
      for (int q = 0; q < 4; ++q) *(f32x4*)(wp + 32 * q) = f32x4{X[4 * q], X[4 * q + 1], X[4 * q + 2], X[4 * q + 3]};
    };
    auto stage_rd32 = [&](int j) { const char* rp = stg + (16 * (j & 3) + (ln >> 2)) * 144 + (ln & 3) * 32; v0 = *(const f32x4*)rp; v1 = *(const f32x4*)(rp + 16); };
    char* cpa = (char*)(g.C + (st_row0 + (ln >> 2)) * g.ldc + st_col + (ln & 3) * 8);
    float eo[8];  // one step's eight products; the step is spread over three MFMA groups (3 + 3 + 2 values) so that its ~130 VALU instructions ride behind
                  // twelve MFMAs instead of four -- as one block per group the kernel measured 6.4 ms (tiled: 6.7), the MFMA pipe idle behind the VALU issue
    auto ep_val = [&](int j, int e) {
      const unsigned aw = ax[BUF ^ 1][j][e >> 1];
      const float x = (e & 1) ? unpack_hi(aw) : unpack_lo(aw);
      eo[e] = (e < 4 ? v0[e & 3] : v1[e & 3]) * ((RS_ABL & 2) ? x : gelu_tanh_grad_fast_f(x));
    };
    auto ep_store = [&](int j) {
      const u32x4 ov = u32x4{rs_pack2(eo[0], eo[1]), rs_pack2(eo[2], eo[3]), rs_pack2(eo[4], eo[5]), rs_pack2(eo[6], eo[7])};
      u32x4* dp = (u32x4*)(cpa + (j & 3) * (2 * cstep) + (j >> 2) * 64);
      if constexpr (RS_ABL & 1) dp = (u32x4*)((char*)g.C + ((uintptr_t)((char*)dp - (char*)g.C) & 0xffff0));
      if (st_edge) {
        if (st_row0 + 16 * (j & 3) + (int64_t)(ln >> 2) < g.M) { if (g.nt_store) __builtin_nontemporal_store(ov, dp); else *dp = ov; }
      } else { if (g.nt_store) __builtin_nontemporal_store(ov, dp); else *dp = ov; }
    };
    auto aux_ld = [&](int j) {  // of the chunk computed in the NEXT phase (columns nx_col of the tile at nx_row0), into the set this phase has just consumed
      if constexpr (RS_ABL & 64) return;
      int64_t row = nx_row0 + 16 * (j & 3) + (ln >> 2); if (row > g.M - 1) row = g.M - 1;
      ax[BUF ^ 1][j] = *(const u32x4*)(g.aux + row * g.ldaux + nx_col + 32 * (j >> 2) + 8 * (ln & 3));
    };
    // the chunk's bias in the accumulator layout (register 4q+e = column 8q + 4hh + e of half t): it enters as the C operand of each chain's first MFMA
    f32x16 bz[2];
    auto load_bias = [&](int t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b = *(const f32x4*)(sbias + 64 * cc + 32 * t + 8 * q + 4 * hh);
        bz[t][4 * q] = b[0]; bz[t][4 * q + 1] = b[1]; bz[t][4 * q + 2] = b[2]; bz[t][4 * q + 3] = b[3];
      }
    };
    if constexpr (MF) load_bias(0);
    if constexpr (AUX) { if (st_on && !(RS_ABL & 16)) { stage_wr32(0, 0); stage_wr32(1, 0); } }
    rs_for<0, 24>([&](auto gq_) {
      constexpr int gq = decltype(gq_)::v;
      if constexpr (MF) {
        if (gq + 2 < 24) {
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[(gq + 2) % 3][i] = *(const bf16x8*)(sb + ((gq + 2) * 2 + i) * 1024);
        }
        if (gq == 8) load_bias(1);
      }
      if constexpr (AUX) {
        // step j = groups 3j .. 3j+2: its 16 rows of the f32 image are read at 3j, its store and the reload of ax[j] (for the chunk being computed) follow at 3j+3.
        // Half 0 of the image was written ahead of group 0, half 1 is written at groups 10 / 11 (behind step 3's read).  LDS-DMA: pieces 9-11 (behind the
        // image) early; pieces 2m, 2m+1 of the image's 9 KiB are free once step 4+m has read its rows (2304 bytes per step).
        if (st_on && !(RS_ABL & 16)) {
          if (gq % 3 == 0 && gq > 0) ep_store(gq / 3 - 1);
          if (gq == 10) stage_wr32(0, 1);
          if (gq == 11) stage_wr32(1, 1);
          if (gq % 3 == 0) stage_rd32(gq / 3);
        }
        if constexpr (MF) {
          if (gq % 3 == 0 && gq > 0) aux_ld(gq / 3 - 1);
          if (gq == 1) dma1(sg2, sl2, RsIC<9>(), l16);
          if (gq == 2) dma1(sg2, sl2, RsIC<10>(), l16);
          if (gq == 4) dma1(sg2, sl2, RsIC<11>(), l16);
          if (gq == 13) dma1(sg2, sl2, RsIC<0>(), l16); if (gq == 14) dma1(sg2, sl2, RsIC<1>(), l16); if (gq == 16) dma1(sg2, sl2, RsIC<2>(), l16);
          if (gq == 17) dma1(sg2, sl2, RsIC<3>(), l16); if (gq == 19) dma1(sg2, sl2, RsIC<4>(), l16); if (gq == 20) dma1(sg2, sl2, RsIC<5>(), l16);
          if (gq == 22) dma1(sg2, sl2, RsIC<6>(), l16); if (gq == 23) { dma1(sg2, sl2, RsIC<7>(), l16); dma1(sg2, sl2, RsIC<8>(), l16); }
        }
      } else {  // one vector-memory instruction per group, stores and LDS-DMA alternating (8 + 12 in 24 groups; as two blocks -- stores in groups 6-13, LDS-DMA in 14-22 -- the
         // kernel measured 6.20 ms against 5.75 at N = 2304; staggering the four waves behind the barrier with s_sleep: slower).  Row group k is read at group 4 + 2k and
         // stored at 6 + 2k; piece j of the image's 9 KiB is free once row groups <= j have been read (rows 8k .. 8k+7 end at byte 1152 (k+1))
        if (st_on && !(RS_ABL & 16)) {
          if (gq == 0) stage_wr(0, 0);
          if (gq == 1) stage_wr(0, 1);
          if (gq == 2) stage_wr(1, 0);
          if (gq == 3) stage_wr(1, 1);
          if (gq >= 6 && gq <= 20 && !(gq & 1)) stage_st((gq - 6) >> 1);   // (the store first: it reads the register the next row group is read into)
          if (gq >= 4 && gq <= 18 && !(gq & 1)) stage_rd((gq - 4) >> 1);
        }
        if constexpr (MF) {
          if (gq == 1) dma1(sg2, sl2, RsIC<9>(), l16);
          if (gq == 3) dma1(sg2, sl2, RsIC<10>(), l16);
          if (gq == 5) dma1(sg2, sl2, RsIC<11>(), l16);
          if (gq == 7) dma1(sg2, sl2, RsIC<0>(), l16); if (gq == 9) dma1(sg2, sl2, RsIC<1>(), l16); if (gq == 11) dma1(sg2, sl2, RsIC<2>(), l16);
          if (gq == 13) dma1(sg2, sl2, RsIC<3>(), l16); if (gq == 15) dma1(sg2, sl2, RsIC<4>(), l16); if (gq == 17) dma1(sg2, sl2, RsIC<5>(), l16);
          if (gq == 19) dma1(sg2, sl2, RsIC<6>(), l16); if (gq == 21) dma1(sg2, sl2, RsIC<7>(), l16); if (gq == 23) dma1(sg2, sl2, RsIC<8>(), l16);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (MF) {
        constexpr int t = gq / 12, s0 = 2 * (gq % 12);
        if constexpr (RS_ABL & 8) { asm volatile("" ::"v"(fa[gq % 3][0]), "v"(fa[gq % 3][1])); }
        else {
          S[BUF][0][t] = MFMA32(fa[gq % 3][0], nb[0][s0], s0 == 0 ? bz[t] : S[BUF][0][t]);
          S[BUF][1][t] = MFMA32(fa[gq % 3][0], nb[1][s0], s0 == 0 ? bz[t] : S[BUF][1][t]);
          S[BUF][0][t] = MFMA32(fa[gq % 3][1], nb[0][s0 + 1], S[BUF][0][t]);
          S[BUF][1][t] = MFMA32(fa[gq % 3][1], nb[1][s0 + 1], S[BUF][1][t]);
        }
      }
      if constexpr (AUX) {
        if (st_on && !(RS_ABL & 16)) {
          constexpr int j = gq / 3, part = gq % 3;
          if (part == 0) { ep_val(j, 0); ep_val(j, 1); ep_val(j, 2); }
          if (part == 1) { ep_val(j, 3); ep_val(j, 4); ep_val(j, 5); }
          if (part == 2) { ep_val(j, 6); ep_val(j, 7); }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (AUX) {  // step 7's store and reload
      if (st_on && !(RS_ABL & 16)) ep_store(7);
      if constexpr (MF) aux_ld(7);
    }
    if constexpr (RS_ABL & 32) { t0 = rs_stamp(); tsum[1] += t0 - t1; }
    if constexpr (MF) {
      // counted wait: this wave's pieces of segment seg + 1 (issued one phase earlier) have landed; this phase's 12 LDS-DMA and the (up to) 8 stores issued
      // before the last 9 of them may stay in flight (vmcnt retires in order).  Masked or absent stores are not counted on.
      if constexpr (!(RS_ABL & 128)) {
        if constexpr (AUX) { if (st_on && !st_edge && !(RS_ABL & 16)) RS_WAIT_VM(28); else RS_WAIT_VM(20); }   // + this phase's 8 aux loads
        else { if (st_on && !st_edge && !(RS_ABL & 16)) RS_WAIT_VM(20); else RS_WAIT_VM(12); }
      }
      if constexpr (RS_ABL & 32) tsum[2] += rs_stamp() - t0;
      seg = seg + 1 == g.nseg ? 0 : seg + 1; slot = slot == 2 ? 0 : slot + 1;
    }
  };

  for (int tile = blockIdx.x; tile < g.tiles; tile += gridDim.x) {
    const int64_t row0 = (int64_t)tile * 256 + w * 64;
    const bool edge = (int64_t)tile * 256 + 256 > g.M;
    unsigned long long th = 0;
    if constexpr (RS_ABL & 32) th = rs_stamp();
    load_a(tile);  // (prefetching the next tile's rows behind the last phase was measured: the 48 loads -- 32-byte pieces of 32 rows each -- cost that phase what they cost here)
    if constexpr (RS_ABL & 32) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tsum[3] += rs_stamp() - th; }
    const int ntile = tile + (int)gridDim.x < g.tiles ? tile + (int)gridDim.x : tile;
    const int64_t nrow0 = (int64_t)ntile * 256 + w * 64;
    if constexpr (AUX) {
      if (tile == (int)blockIdx.x) {  // chunk 0's aux values (set 0): every later set is loaded two phases ahead of its use
        const unsigned ln = rs_lane();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          int64_t row = row0 + 16 * (j & 3) + (ln >> 2); if (row > g.M - 1) row = g.M - 1;
          ax[0][j] = *(const u32x4*)(g.aux + row * g.ldaux + 32 * (j >> 2) + 8 * (ln & 3));
        }
      }
    }
    // The tile's first two phases are PEELED: the A rows were requested just above and arrive while phase 0 runs (the compiler's waits for them count down
    // through its MFMA groups).  Inside one loop those waits are executed by every phase that shares the code -- s_waitcnt vmcnt(9) .. vmcnt(3) in the middle of
    // every other phase, each retiring nearly every LDS-DMA and store in flight -- or are hoisted as one vmcnt(0) ahead of the loop, which exposes the whole load.
    phase(RsIC<0>(), RsIC<1>(), 0, row0, 64);                                                // computes chunk 0, stores the previous tile's last chunk (set 1), reloads set 1 for chunk 1
    st_row0 = row0; st_col = 0; st_on = true; st_edge = edge;
    phase(RsIC<1>(), RsIC<1>(), 1, row0, 128);                                               // computes 1, stores 0 (set 0), reloads set 0 for chunk 2
    st_col = 64;
    for (int c = 2; c < g.nseg; c += 2) {
      phase(RsIC<0>(), RsIC<1>(), c, row0, 64 * (c + 1));                                   // computes chunk c, stores chunk c-1 (set 1), reloads set 1 for chunk c+1
      st_col = 64 * c;
      const bool last = c + 2 >= g.nseg;
      phase(RsIC<1>(), RsIC<1>(), c + 1, last ? nrow0 : row0, last ? 0 : 64 * (c + 2));     // computes c+1, stores c (set 0), reloads set 0 for chunk c+2 / the next tile's 0
      st_col = 64 * (c + 1);
    }
  }
  if (st_on) phase(RsIC<0>(), RsIC<0>(), 0, 0, 0);  // drain: the last chunk (in S[1]) leaves; no MFMAs, no LDS-DMA
  RS_WAIT_VM(0);
  if constexpr (RS_ABL & 32) {
    if (g.dbg && rs_lane() == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) g.dbg[((int64_t)blockIdx.x * 4 + w) * 4 + k] = tsum[k];
    }
  }
}

// host: pack W (element (k, n) at w[k * sk + n * sn]) into the stream; S = float (model) or bf16_t (op test)
template <typename S> void gemm_rs_pack(spa3d_ctx* c, const S* w, int64_t sk, int64_t sn, int N, bf16_t* wpk) {
  if (c->dry) return;
  const int nseg = N / 64, n = nseg * 48 * 64;
  rs_pack_kernel<S><<<(n + 255) / 256, 256, 0, c->stream>>>(w, sk, sn, nseg, wpk);
  SPA_LAUNCH_CHECK(c);
}
template void gemm_rs_pack<float>(spa3d_ctx*, const float*, int64_t, int64_t, int, bf16_t*);
template void gemm_rs_pack<bf16_t>(spa3d_ctx*, const bf16_t*, int64_t, int64_t, int, bf16_t*);

bool gemm_rs_ok(int K, int N) { return K == RS_K && N >= 256 && N <= RS_MAXN && N % 128 == 0; }
int64_t gemm_rs_pack_elems(int N) { return (int64_t)(N / 64) * 48 * 512; }

// C[M, N] = A[M, 384] . W (+ bias) with W as the packed stream.  Returns false when the shape / layout is not this kernel's.
bool gemm_rs(spa3d_ctx* c, const bf16_t* A, int64_t lda, const bf16_t* wpk, const float* bias, bf16_t* C, int64_t ldc, int64_t M, int N, const bf16_t* gelu_pre,
             int64_t ldpre) {
  if (!wpk || !gemm_rs_ok(RS_K, N) || M < 1 || lda % 8 || ldc % 8 || (((uintptr_t)A | (uintptr_t)C) & 15)) return false;
  if (gelu_pre && (ldpre % 8 || ((uintptr_t)gelu_pre & 15))) return false;
  if (c->dry) return true;
  RsArgs g{};
  g.A = A; g.lda = lda; g.C = C; g.ldc = ldc; g.wpk = (const char*)wpk; g.bias = bias; g.M = M; g.N = N; g.nseg = N / 64;
  g.aux = gelu_pre; g.ldaux = ldpre;
  g.tiles = (int)((M + 255) / 256);
  g.nt_store = (c->nt_stream && (double)M * N * 2.0 >= 512.0 * 1024 * 1024) ? 1 : 0;
#ifdef SPA3D_RS_PLAIN_ST
  g.nt_store = 0;
#endif
  g.dbg = nullptr;
  if (RS_ABL & 32) { const char* e = getenv("SPA3D_RS_DBG"); if (e) g.dbg = (unsigned long long*)strtoull(e, nullptr, 0); }
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_rs_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_rs_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS);
    attr = true;
  }
  ProfScope ps(c, PROF_GEMM_NT, 2.0 * (double)M * N * RS_K, ((double)M * (RS_K + N * (gelu_pre ? 2.0 : 1.0)) + (double)RS_K * N) * 2.0);
  ps.tag(M, N, RS_K, gelu_pre ? 512 + 6 : 512);
  const int grid = g.tiles < 256 ? g.tiles : 256;
  if (gelu_pre) gemm_rs_kernel<true><<<grid, 256, RS_LDS, c->stream>>>(g);
  else gemm_rs_kernel<false><<<grid, 256, RS_LDS, c->stream>>>(g);
  SPA_LAUNCH_CHECK(c);
  return true;
}

}  // namespace SPA_NS
