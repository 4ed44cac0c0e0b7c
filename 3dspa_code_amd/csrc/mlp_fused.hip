// mlp_fused.hip -- sequence-resident MLP forward of ImprovedTransformerBlock (/root/reference/attention.py:103-108) for the track
// encoder's widths (d = 384, mlp = 1536):   y = a + W_out^T gelu(W_in^T na + b_in) + b_out,   h / hpre kept for the backward.
//
// One launch replaces the MLP-in GEMM (dual output h, hpre) and the MLP-out GEMM (residual): `h` is never read back from HBM.
// Flash-attention-shaped ("sequence-resident block kernels"): a persistent workgroup = 4 waves = one wave per SIMD with the
// whole 512-register file, each wave owns 32 token rows of a 128-row tile:
//   * na rows of the wave: 24 MFMA B fragments (k = model dim) held in registers for the whole tile;
//   * the weights never come from HBM per tile: a pre-packed stream (2.36 MB, L2 / Infinity-Cache resident, identical for every tile)
//     in consumption order, one 1-KiB MFMA A fragment per LDS-DMA instruction, through a 3 x 48 KiB LDS ring shared by the 4 waves;
//   * S^T[hidden 64][rows 32] = W_in^T-chunk . na^T accumulates with the TOKEN ROW ON THE LANE, so gelu(S) packs straight into the B
//     operand of the next product Y^T[out 384][rows 32] += W_out^T-chunk . P (k order permuted, cdna_hip_programming.md section 3
//     "An accumulator tile as the next MFMA's operand": the packed W_out fragments carry the same permutation);
//   * Y^T lives in 192 accumulator registers per lane for the whole tile (zero-initialised; b_out and the residual a are added in the epilogue);
//   * h / hpre leave during the GEMM2 phases through the wave's own 12 KiB of the ring slot that phase refills, as whole 128-byte row pieces; y through the same scratch.
// Phase = 48 MFMAs (32x32x16) on one 48-KiB ring slot: X_c = GEMM1 of hidden chunk c (ping-pong accumulators) with all eight gelu quads of chunk c-1 riding along,
// Y_c = GEMM2 of chunk c-1 with that chunk's stores.  One barrier per phase; the LDS-DMA of segment p+2 is issued in phase p and waited for with a COUNTED vmcnt at the end
// of phase p+1 (never 0 in the tile), so the h / hpre stores of a chunk have two phases to drain.  Measurements, ablations and what bounds it: NOTEBOOK.md, "Round 4".
#include <cstdlib>

#include "common.hpp"

namespace SPA_NS {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef mfma16x8 bf16x8;

constexpr int MF_D = 384, MF_H = 1536;
constexpr int MF_SEG = 48 * 1024;  // one phase's weights: 48 MFMA A fragments of 1 KiB
constexpr int MF_NSEG = 48;        // segments per tile: Wi0, (Wi_c, Wo_{c-1}) for c = 1..23, Wo23
constexpr int MF_RING = 3 * MF_SEG;
constexpr int MF_LDS = MF_RING + MF_H * 4 + MF_D * 4;

// ---- weight stream.  Segment sigma, fragment f (1 KiB = 64 lanes x 8 elements), lane (r = lane & 31, hh = lane >> 5), element j:
//   W_in  segment of chunk c: f = t*24 + s   -> W_in [k = 16s + 8hh + j][hidden = 64c + 32t + r]
//   W_out segment of chunk c: f = o*4 + s    -> W_out[hidden = 64c + 16s + 8(j>>2) + 4hh + (j&3)][out = 32o + r]
template <typename S>
__global__ void mlp_pack_kernel(const S* __restrict__ w_in, const S* __restrict__ w_out, bf16_t* __restrict__ out) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= MF_NSEG * 48 * 64) return;
  const int lane = id & 63, frag = (id >> 6) % 48, seg = id / (64 * 48);
  const int r = lane & 31, hh = lane >> 5;
  bool is_in; int c;
  if (seg == 0) { is_in = true; c = 0; }
  else if (seg == MF_NSEG - 1) { is_in = false; c = 23; }
  else if (seg & 1) { is_in = true; c = (seg + 1) >> 1; }
  else { is_in = false; c = (seg >> 1) - 1; }
  bf16_t v[8];
  if (is_in) {
    const int t = frag / 24, s = frag % 24, hid = 64 * c + 32 * t + r;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = f2bf(ld<S>(w_in + (int64_t)(16 * s + 8 * hh + j) * MF_H + hid));
  } else {
    const int o = frag >> 2, s = frag & 3, oc = 32 * o + r;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = f2bf(ld<S>(w_out + (int64_t)(64 * c + 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3)) * MF_D + oc));
  }
  u32x4 p;
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = (unsigned)v[2 * j] | ((unsigned)v[2 * j + 1] << 16);
  *(u32x4*)(out + (int64_t)id * 8) = p;
}

struct MlpFusedArgs {
  const bf16_t* na; const bf16_t* a; bf16_t* y; bf16_t* h; bf16_t* hpre; const char* wpk; const float* b_in; const float* b_out;
  int64_t M; int tiles; int nt_store;
  unsigned long long* dbg;  // diagnostic builds with mask 32 only: per-wave cycle sums (s_memtime), else unused
};
// diagnostic builds only (tools/ablate_mlp_fused.py compiles a SEPARATE library per mask with -DSPA3D_ABLATION_BUILD -DSPA3D_ABL_MF=mask, never the product: csrc/ablate.inc): compile-time mask,
// 1 h / hpre stores wrapped into a 1-MiB window (no HBM write stream), 2 no gelu arithmetic, 4 no LDS-DMA, 8 no MFMAs, 16 no y stores,
// 32 s_memtime stamps at the phase seams (per-wave sums to MlpFusedArgs::dbg; shares, not run time), 64 plain instead of non-temporal stores,
// 128 no counted wait at the phase ends (WRONG results: timing only), 256 no phase barriers (WRONG results: timing only)
constexpr int MF_ABL = SPA3D_ABL_MF;  // csrc/ablate.inc: 0 in libspa3d_hip.so
#ifndef SPA3D_MF_FDEPTH
#define SPA3D_MF_FDEPTH 1
#endif
constexpr int MF_FDEPTH = SPA3D_MF_FDEPTH;
#ifndef SPA3D_MF_NAPF
#define SPA3D_MF_NAPF 1
#endif
constexpr bool MF_NAPF = SPA3D_MF_NAPF;  // next tile's na rows prefetched behind the last GEMM1 phase

// LDS-DMA of one 1-KiB piece: wave-uniform 64-bit base in SGPRs + 32-bit lane offset; the instruction's immediate offset moves BOTH the global
// address and the LDS address (LDS address = M0 + offset + 16 * lane), so four consecutive pieces share one base / M0 value.
template <int OFF> __device__ __forceinline__ void mf_glds(const void* base_uniform, unsigned off, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(off), "s"(base_uniform), "s"(lds_dst), "n"(OFF) : "memory", "m0");
}
#define MF_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define MF_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
template <int N> struct IC { static constexpr int v = N; };

// The lane id recomputed where it is used (two VALU instructions) and made opaque: lane-constant addresses computed once at kernel entry are
// hoisted, spilled, and reloaded inside the tile loop behind s_waitcnt vmcnt(0), which would drain the LDS-DMA ring and the output stores
// every phase (cdna_hip_programming.md, 4-wave persistent structure, "Pitfalls").
__device__ __forceinline__ unsigned lane_now() {
  unsigned l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(l));
  return l;
}
__device__ __forceinline__ unsigned long long mf_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
typedef __attribute__((ext_vector_type(2))) float f32x2;
#if SPA_F16
typedef __attribute__((ext_vector_type(2))) _Float16 h16x2;
#else
typedef __attribute__((ext_vector_type(2))) __bf16 h16x2;
#endif
// two f32 -> one packed dword, round-to-nearest-even (one v_cvt_pk instruction)
__device__ __forceinline__ unsigned pack2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, h16x2)); }

__global__ __launch_bounds__(256, 1) void mlp_fused_fwd_kernel(MlpFusedArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [3][48 KiB] ring | b_in f32[1536] | b_out f32[384]
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* sbin = (float*)(smem + MF_RING);
  float* sbout = sbin + MF_H;
  for (int i = tid; i < MF_H / 4; i += 256) ((float4*)sbin)[i] = ((const float4*)g.b_in)[i];
  for (int i = tid; i < MF_D / 4; i += 256) ((float4*)sbout)[i] = ((const float4*)g.b_out)[i];
  __syncthreads();
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // mask 32: barrier, phase body X, phase body Y, end wait, tile head, exposed gelu, epilogue, stores
  int seg = 0, slot = 0;  // segment the next phase consumes and the ring slot it sits in (slot == seg % 3: 48 % 3 == 0)

  // piece i (0..11, compile-time) of this wave's twelve 1-KiB pieces of segment sg into slot sl
  auto dma1 = [&](int sg, int sl, auto i_, unsigned lane16) {
    constexpr int i = decltype(i_)::v;
    if constexpr (MF_ABL & 4) return;
    const int p0 = w * 12 + (i & ~3);
    mf_glds<(i & 3) * 1024>(g.wpk + (int64_t)sg * MF_SEG + p0 * 1024, lane16, lds0 + (unsigned)(sl * MF_SEG + p0 * 1024));
  };
  {  // prologue: segments 0 and 1
    const unsigned l16 = lane_now() * 16u;
    dma1(0, 0, IC<0>(), l16); dma1(0, 0, IC<1>(), l16); dma1(0, 0, IC<2>(), l16); dma1(0, 0, IC<3>(), l16);
    dma1(0, 0, IC<4>(), l16); dma1(0, 0, IC<5>(), l16); dma1(0, 0, IC<6>(), l16); dma1(0, 0, IC<7>(), l16);
    dma1(0, 0, IC<8>(), l16); dma1(0, 0, IC<9>(), l16); dma1(0, 0, IC<10>(), l16); dma1(0, 0, IC<11>(), l16);
    dma1(1, 1, IC<0>(), l16); dma1(1, 1, IC<1>(), l16); dma1(1, 1, IC<2>(), l16); dma1(1, 1, IC<3>(), l16);
    dma1(1, 1, IC<4>(), l16); dma1(1, 1, IC<5>(), l16); dma1(1, 1, IC<6>(), l16); dma1(1, 1, IC<7>(), l16);
    dma1(1, 1, IC<8>(), l16); dma1(1, 1, IC<9>(), l16); dma1(1, 1, IC<10>(), l16); dma1(1, 1, IC<11>(), l16);
  }
  MF_WAIT_VM(12);  // segment 0 has landed (this wave's pieces; the barrier of the first phase covers the others')

  // this wave's na rows as 24 B fragments: lane (r, hh) holds na[row r][16 s + 8 hh .. + 7].  Loaded for the first tile here and for every
  // later tile behind the current tile's last GEMM1 phase (two phases of MFMAs cover the latency).
  bf16x8 nb[24];
  auto load_na = [&](int tile) {
    const unsigned ln = lane_now(); const int r = ln & 31, hh = ln >> 5;
    int64_t row = (int64_t)tile * 128 + w * 32 + r; if (row > g.M - 1) row = g.M - 1;
    const bf16_t* nap = g.na + row * MF_D + hh * 8;
#pragma unroll
    for (int s = 0; s < 24; ++s) nb[s] = *(const bf16x8*)(nap + 16 * s);
  };
  if constexpr (MF_NAPF) load_na(blockIdx.x);

  for (int tile = blockIdx.x; tile < g.tiles; tile += gridDim.x) {
    const int64_t row0 = (int64_t)tile * 128 + w * 32;  // uniform
    const bool edge = (int64_t)tile * 128 + 128 > g.M;  // uniform: stores of some lanes are masked off, so waits are not counted against them
    unsigned long long th = 0;
    if constexpr (MF_ABL & 32) th = mf_stamp();
    if constexpr (!MF_NAPF) load_na(tile);
    f32x16 Yacc[12];  // Y^T: register 4q+e of out tile o is column 32 o + 8 q + 4 hh + e of the lane's row
#pragma unroll
    for (int o = 0; o < 12; ++o)
#pragma unroll
      for (int e = 0; e < 16; ++e) Yacc[o][e] = 0.f;
    if constexpr (MF_ABL & 32) { asm volatile("" ::"v"(Yacc[0]), "v"(Yacc[11])); tsum[4] += mf_stamp() - th; }
    f32x16 Sa[2], Sb[2];  // GEMM1 accumulators of two consecutive hidden chunks: one accumulates while gelu reads the other
    bf16x8 P[4];
    unsigned pn[4][4], hn[4][4];  // packed gelu / pre-activation of the chunk in flight: [k-step s][dword], dwords 0,1 = quad 2s, 2,3 = quad 2s+1

    // gelu of one register quad: quad qd = 4 t + q of the chunk's S (S = its tile t) -> k-step s = qd >> 1, half = qd & 1
    auto gelu_quad = [&](const f32x16 (&S2)[2], auto qd_) {
      constexpr int qd = decltype(qd_)::v, q = qd & 3, s = qd >> 1, half = qd & 1;
      const f32x16& S = S2[qd >> 2];
      float v0 = S[4 * q], v1 = S[4 * q + 1], v2 = S[4 * q + 2], v3 = S[4 * q + 3];
      hn[s][2 * half] = pack2(v0, v1); hn[s][2 * half + 1] = pack2(v2, v3);
      if constexpr (!(MF_ABL & 2)) { v0 = gelu_tanh_fast_f(v0); v1 = gelu_tanh_fast_f(v1); v2 = gelu_tanh_fast_f(v2); v3 = gelu_tanh_fast_f(v3); }
      pn[s][2 * half] = pack2(v0, v1); pn[s][2 * half + 1] = pack2(v2, v3);
    };
    // one MFMA phase on ring slot `slot`: kind 0 = X (GEMM1 of chunk cc into Sn), 1 = Y (GEMM2 with P into Yacc).  GQ0 >= 0: gelu quads
    // GQ0 .. GQ0 + 3 of S ride along (the first one ahead of the MFMAs, under the phase's first LDS reads).
    // ST = 1 (every Y phase): h / hpre of hidden chunk gs (P and hn) leave in groups 0-4 through this wave's OWN 12 KiB of the slot the phase
    // refills (a wave's twelve LDS-DMA pieces are its own region, so the region is scratch between the barrier that frees the slot and the
    // wave's first LDS-DMA into it): two [32 rows][128 B + 16] images, written as the lanes hold them and read back as whole 128-byte row
    // pieces -- 8 rows per store instruction instead of 32 partial lines.  The phase's LDS-DMA follows in groups 6-11, two pieces per group.
    auto phase = [&](auto kind_, auto gq0_, auto st_, int cc, f32x16 (&Sn)[2], const f32x16 (&S)[2], int gs) {
      constexpr int KIND = decltype(kind_)::v, GQ0 = decltype(gq0_)::v, ST = decltype(st_)::v;
      unsigned long long t0 = 0, t1 = 0;
      if constexpr (MF_ABL & 32) t0 = mf_stamp();
      if constexpr (!(MF_ABL & 256)) MF_BAR();  // segment `seg` is visible to every wave; every wave has left the slot this phase refills
      if constexpr (MF_ABL & 32) { t1 = mf_stamp(); tsum[0] += t1 - t0; }
      const unsigned ln = lane_now(); const int hh = ln >> 5;
      const unsigned l16 = ln * 16u;
      const char* sb = smem + slot * MF_SEG + l16;
      const int sg2 = seg + 2 >= MF_NSEG ? seg + 2 - MF_NSEG : seg + 2, sl2 = slot == 0 ? 2 : slot - 1;
      char* stg = smem + sl2 * MF_SEG + w * 12288;
      u32x4 sv[2];
      auto stage_rd = [&](int arr, int k0) {  // rows 8k + (lane >> 3), 16-byte piece lane & 7; two row groups per call
#pragma unroll
        for (int k = 0; k < 2; ++k) sv[k] = *(const u32x4*)(stg + arr * 4608 + (8 * (k0 + k) + (ln >> 3)) * 144 + (ln & 7) * 16);
      };
      auto stage_st = [&](bf16_t* dst, int k0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int64_t rw = row0 + 8 * (k0 + k) + (ln >> 3);
          if (rw < g.M) {
            const int64_t e2 = rw * MF_H + 64 * gs + (ln & 7) * 8;
            u32x4* dp = (u32x4*)(dst + ((MF_ABL & 1) ? (e2 & 0x7fff8) : e2));
            if (g.nt_store) __builtin_nontemporal_store(sv[k], dp); else *dp = sv[k];
          }
        }
      };
      bf16x8 fa[3][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[0][i] = *(const bf16x8*)(sb + i * 1024);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[1][i] = *(const bf16x8*)(sb + (4 + i) * 1024);
      if constexpr (KIND == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 b = *(const f32x4*)(sbin + 64 * cc + 32 * t + 8 * q + 4 * hh);
            Sn[t][4 * q] = b[0]; Sn[t][4 * q + 1] = b[1]; Sn[t][4 * q + 2] = b[2]; Sn[t][4 * q + 3] = b[3];
          }
      }
#pragma unroll
      for (int gq = 0; gq < 12; ++gq) {
        if (gq + 2 < 12) {
#pragma unroll
          for (int i = 0; i < 4; ++i) fa[(gq + 2) % 3][i] = *(const bf16x8*)(sb + ((gq + 2) * 4 + i) * 1024);
        }
        if constexpr (ST) {
          if (gq == 0) {  // lane (r, hh): dwords 0,1 of k-step s = hidden 16 s + 4 hh .. + 3, dwords 2,3 = hidden 16 s + 8 + 4 hh .. + 3
            char* wp = stg + (ln & 31) * 144 + hh * 8;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const u32x4 pv = __builtin_bit_cast(u32x4, P[s]);
              *(u32x2*)(wp + 32 * s) = u32x2{pv[0], pv[1]}; *(u32x2*)(wp + 32 * s + 16) = u32x2{pv[2], pv[3]};
              *(u32x2*)(wp + 4608 + 32 * s) = u32x2{hn[s][0], hn[s][1]}; *(u32x2*)(wp + 4608 + 32 * s + 16) = u32x2{hn[s][2], hn[s][3]};
            }
          }
          if (gq == 1) stage_rd(0, 0);
          if (gq == 2) { stage_st(g.h, 0); stage_rd(0, 2); }
          if (gq == 3) { stage_st(g.h, 2); stage_rd(1, 0); }
          if (gq == 4) { stage_st(g.hpre, 0); stage_rd(1, 2); }
          if (gq == 5) stage_st(g.hpre, 2);
          // LDS-DMA into the scratch's slot only after its last read has been consumed (group 5's stores)
          if (gq == 6) { dma1(sg2, sl2, IC<0>(), l16); dma1(sg2, sl2, IC<1>(), l16); }
          if (gq == 7) { dma1(sg2, sl2, IC<2>(), l16); dma1(sg2, sl2, IC<3>(), l16); }
          if (gq == 8) { dma1(sg2, sl2, IC<4>(), l16); dma1(sg2, sl2, IC<5>(), l16); }
          if (gq == 9) { dma1(sg2, sl2, IC<6>(), l16); dma1(sg2, sl2, IC<7>(), l16); }
          if (gq == 10) { dma1(sg2, sl2, IC<8>(), l16); dma1(sg2, sl2, IC<9>(), l16); }
          if (gq == 11) { dma1(sg2, sl2, IC<10>(), l16); dma1(sg2, sl2, IC<11>(), l16); }
        } else {
        // one LDS-DMA piece per group (switch over the compile-time group index)
        if (gq == 0) dma1(sg2, sl2, IC<0>(), l16); if (gq == 1) dma1(sg2, sl2, IC<1>(), l16); if (gq == 2) dma1(sg2, sl2, IC<2>(), l16);
        if (gq == 3) dma1(sg2, sl2, IC<3>(), l16); if (gq == 4) dma1(sg2, sl2, IC<4>(), l16); if (gq == 5) dma1(sg2, sl2, IC<5>(), l16);
        if (gq == 6) dma1(sg2, sl2, IC<6>(), l16); if (gq == 7) dma1(sg2, sl2, IC<7>(), l16); if (gq == 8) dma1(sg2, sl2, IC<8>(), l16);
        if (gq == 9) dma1(sg2, sl2, IC<9>(), l16); if (gq == 10) dma1(sg2, sl2, IC<10>(), l16); if (gq == 11) dma1(sg2, sl2, IC<11>(), l16);
        }
        if constexpr (GQ0 >= 0) { if (gq == 0) gelu_quad(S, IC<0>()); }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KIND == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if constexpr (MF_ABL & 8) asm volatile("" ::"v"(fa[gq % 3][i])); else Sn[gq / 6] = MFMA32(fa[gq % 3][i], nb[(gq % 6) * 4 + i], Sn[gq / 6]);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if constexpr (MF_ABL & 8) asm volatile("" ::"v"(fa[gq % 3][i])); else Yacc[gq] = MFMA32(fa[gq % 3][i], P[i], Yacc[gq]);
          }
        }
        if constexpr (GQ0 >= 0) {  // all eight quads of the previous chunk ride in the X phase (the Y phases carry the stores)
          if (gq == 1) gelu_quad(S, IC<1>());
          if (gq == 3) gelu_quad(S, IC<2>());
          if (gq == 4) gelu_quad(S, IC<3>());
          if (gq == 6) gelu_quad(S, IC<4>());
          if (gq == 7) gelu_quad(S, IC<5>());
          if (gq == 9) gelu_quad(S, IC<6>());
          if (gq == 10) gelu_quad(S, IC<7>());
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (MF_ABL & 32) { tsum[1 + KIND] += mf_stamp() - t1; }
    };
    // counted wait at a phase's end: this wave's pieces of segment seg + 1 (issued one phase earlier) have landed; everything younger --
    // this phase's 12 LDS-DMA and, in a Y phase, the 8 h / hpre stores issued before them -- may stay in flight (vmcnt retires in order)
    auto end_phase = [&](auto nw_) {
      constexpr int NW = decltype(nw_)::v;
      unsigned long long t0 = 0;
      if constexpr (MF_ABL & 32) t0 = mf_stamp();
      if constexpr (!(MF_ABL & 128)) { if (edge) MF_WAIT_VM(12); else MF_WAIT_VM(NW); }
      if constexpr (MF_ABL & 32) tsum[3] += mf_stamp() - t0;
      seg = seg + 1 == MF_NSEG ? 0 : seg + 1; slot = slot == 2 ? 0 : slot + 1;
    };
    auto take_p = [&]() {
#pragma unroll
      for (int s = 0; s < 4; ++s) P[s] = __builtin_bit_cast(bf16x8, u32x4{pn[s][0], pn[s][1], pn[s][2], pn[s][3]});
    };

    phase(IC<0>(), IC<-1>(), IC<0>(), 0, Sa, Sb, 0); end_phase(IC<12>());                       // X_0 -> Sa
    for (int c = 1; c < 23; c += 2) {
      phase(IC<0>(), IC<0>(), IC<0>(), c, Sb, Sa, 0); end_phase(IC<12>());                      // X_c   : GEMM1(c) -> Sb   || gelu(c-1) from Sa
      take_p();                                                                                 // P = gelu(chunk c-1)
      phase(IC<1>(), IC<-1>(), IC<1>(), 0, Sa, Sa, c - 1); end_phase(IC<20>());                  // Y_c   : GEMM2(c-1) || stores(c-1)
      phase(IC<0>(), IC<0>(), IC<0>(), c + 1, Sa, Sb, 0); end_phase(IC<12>());                  // X_c+1 : GEMM1(c+1) -> Sa || gelu(c) from Sb
      take_p();
      phase(IC<1>(), IC<-1>(), IC<1>(), 0, Sa, Sa, c); end_phase(IC<20>());                      // Y_c+1
    }
    phase(IC<0>(), IC<0>(), IC<0>(), 23, Sb, Sa, 0); end_phase(IC<12>());                       // X_23 -> Sb || gelu(22); nb is dead behind it
    if constexpr (MF_NAPF) {
      const int nt = tile + (int)gridDim.x;
      load_na(nt < g.tiles ? nt : tile);                                                        // the next tile's na rows, under Y_23 / Y_24
    }
    take_p();
    phase(IC<1>(), IC<-1>(), IC<1>(), 0, Sa, Sa, 22); end_phase(IC<20>());                       // Y_23
    if constexpr (MF_ABL & 32) th = mf_stamp();
    gelu_quad(Sb, IC<0>()); gelu_quad(Sb, IC<1>()); gelu_quad(Sb, IC<2>()); gelu_quad(Sb, IC<3>());   // gelu(23), exposed once per tile
    gelu_quad(Sb, IC<4>()); gelu_quad(Sb, IC<5>()); gelu_quad(Sb, IC<6>()); gelu_quad(Sb, IC<7>());
    take_p();
    if constexpr (MF_ABL & 32) { asm volatile("" ::"v"(P[3])); tsum[5] += mf_stamp() - th; }
    phase(IC<1>(), IC<-1>(), IC<1>(), 0, Sa, Sa, 23); end_phase(IC<20>());                       // Y_24 : GEMM2(23) || stores(23); LDS-DMA = next tile's segment 1

    // ---- epilogue: y = (Y + b_out) + a, rounded once.  Six rounds of 64 columns: the f32 accumulators of two out tiles go through this
    // wave's 12 KiB of the slot Y_24 just read (free until the next X0's LDS-DMA) and come back row-major, so the residual loads and the
    // y stores are whole 128-byte row pieces.  Residual loads run one round ahead.
    if constexpr (MF_ABL & 32) th = mf_stamp();
    MF_BAR();
    {
      char* stg = smem + (slot == 0 ? 2 : slot - 1) * MF_SEG + w * 12288;
      const unsigned lane = lane_now(); const int r = lane & 31, hh = lane >> 5;
      const int pc = lane & 15, rr0 = lane >> 4;  // read-back: rows rr0 + 4 k, 16-byte piece pc (4 columns)
      u32x2 av[2][8];
      auto load_a = [&](int rd, u32x2 (&dst)[8]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          int64_t rw = row0 + rr0 + 4 * k; if (rw > g.M - 1) rw = g.M - 1;
          dst[k] = *(const u32x2*)(g.a + rw * MF_D + rd * 64 + pc * 4);
        }
      };
      load_a(0, av[0]);
#pragma unroll
      for (int rd = 0; rd < 6; ++rd) {
        if (rd + 1 < 6) load_a(rd + 1, av[(rd + 1) & 1]);
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x16& Y = Yacc[rd * 2 + o];
            *(f32x4*)(stg + r * 272 + (o * 32 + 8 * q + 4 * hh) * 4) = f32x4{Y[4 * q], Y[4 * q + 1], Y[4 * q + 2], Y[4 * q + 3]};
          }
        __builtin_amdgcn_wave_barrier();
        const f32x4 bo = *(const f32x4*)(sbout + rd * 64 + pc * 4);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int rr = rr0 + 4 * k;
          const f32x4 v = *(const f32x4*)(stg + rr * 272 + pc * 16);
          const u32x2 ax = av[rd & 1][k];
          const u32x2 o2 = u32x2{pack2((v[0] + bo[0]) + unpack_lo(ax[0]), (v[1] + bo[1]) + unpack_hi(ax[0])),
                                 pack2((v[2] + bo[2]) + unpack_lo(ax[1]), (v[3] + bo[3]) + unpack_hi(ax[1]))};
          if constexpr (MF_ABL & 16) { asm volatile("" ::"v"(o2)); }
          else if (row0 + rr < g.M) {
            u32x2* yp = (u32x2*)(g.y + (row0 + rr) * MF_D + rd * 64 + pc * 4);
            if (g.nt_store) __builtin_nontemporal_store(o2, yp); else *yp = o2;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if constexpr (MF_ABL & 32) tsum[6] += mf_stamp() - th;
  }
  MF_WAIT_VM(0);
  if constexpr (MF_ABL & 32) {
    if (g.dbg && lane_now() == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) g.dbg[((int64_t)blockIdx.x * 4 + w) * 8 + k] = tsum[k];
    }
  }
}

// host: pack the two kernels of an MLP into the fused kernel's weight stream (2 x 589 824 elements); S = float (model) or bf16_t (op test)
template <typename S> void mlp_fused_pack(spa3d_ctx* c, const S* w_in, const S* w_out, bf16_t* wpk) {
  if (c->dry) return;
  const int n = MF_NSEG * 48 * 64;
  mlp_pack_kernel<S><<<(n + 255) / 256, 256, 0, c->stream>>>(w_in, w_out, wpk);
  SPA_LAUNCH_CHECK(c);
}
template void mlp_fused_pack<float>(spa3d_ctx*, const float*, const float*, bf16_t*);
template void mlp_fused_pack<bf16_t>(spa3d_ctx*, const bf16_t*, const bf16_t*, bf16_t*);

int64_t mlp_fused_pack_elems() { return (int64_t)MF_NSEG * 48 * 512; }

// y[M,384] = a + MLP(na);  h, hpre [M,1536] for the backward.  Returns false when the shape is not the fused kernel's.
bool mlp_fused_fwd(spa3d_ctx* c, const bf16_t* na, const bf16_t* a, bf16_t* y, bf16_t* h, bf16_t* hpre, int64_t M, int d, int mlp,
                   const bf16_t* wpk, const float* b_in, const float* b_out) {
  if (d != MF_D || mlp != MF_H || M < 1 || !wpk || !b_in || !b_out) return false;
  if (c->dry) return true;
  // 16-byte vector loads / stores and LDS-DMA on every operand: a misaligned pointer from a C-ABI caller is refused, not faulted on (gemm_rs / gemm_nt_bf16 do the same)
  for (const void* p : {(const void*)na, (const void*)a, (const void*)y, (const void*)h, (const void*)hpre, (const void*)wpk})
    if (((uintptr_t)p) & 15) return false;
  MlpFusedArgs g{};
  g.na = na; g.a = a; g.y = y; g.h = h; g.hpre = hpre; g.wpk = (const char*)wpk; g.b_in = b_in; g.b_out = b_out; g.M = M;
  g.tiles = (int)((M + 127) / 128);
  g.dbg = nullptr;
  if (MF_ABL & 32) { const char* e = getenv("SPA3D_MF_DBG"); if (e) g.dbg = (unsigned long long*)strtoull(e, nullptr, 0); }
  g.nt_store = (c->nt_stream && (double)M * MF_H * 2.0 >= 512.0 * 1024 * 1024) ? 1 : 0;
  if (MF_ABL & 64) g.nt_store = 0;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)mlp_fused_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MF_LDS); attr = true; }
  ProfScope ps(c, PROF_GEMM_NT, 2.0 * 2.0 * (double)M * MF_D * MF_H, ((double)M * (3.0 * MF_D + 2.0 * MF_H) + 2.0 * MF_D * MF_H) * 2.0);
  ps.tag(M, MF_D, MF_H, 256);
  const int grid = g.tiles < 256 ? g.tiles : 256;
  mlp_fused_fwd_kernel<<<grid, 256, MF_LDS, c->stream>>>(g);
  SPA_LAUNCH_CHECK(c);
  return true;
}

}  // namespace SPA_NS
