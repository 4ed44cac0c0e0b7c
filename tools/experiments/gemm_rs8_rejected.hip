// REJECTED (round 4), kept for the record; not built.  The eight-wave form of csrc/gemm_rs.hip's row-stationary K = 384 GEMM: two waves per SIMD, 32 rows per wave
// (184 VGPRs, 0 spills, correct on the first run).  Hypothesis: with one wave per SIMD the store time, the LDS-DMA time and the MFMA time of a phase ADD (ablation of
// the four-wave kernel at N = 2304, M = 3.07 M: stores alone 2.7 ms, + LDS-DMA 4.0 ms, + MFMA 5.8 ms) because a wave sitting in the issue of a store cannot issue MFMAs;
// a second wave per SIMD should fill those gaps.  Measured (tools/ablate_gemm_rs.py, VARIANT=-DSPA3D_RS_EIGHT=1, same box, same call):
//     N = 2304:  5.50 ms (986 TF/s)  against 5.56 ms for the four-wave kernel;  without MFMAs 3.69 ms (four waves: 3.6-4.0), stores into an L2-resident window 4.62, no LDS-DMA 4.57
//     N =  768:  1.97 ms (920 TF/s)  against 2.10 ms
// i.e. the sum stays a sum with two instruction streams per SIMD: it is not issue blocking.  What the second wave costs -- every fragment read from LDS feeds one MFMA
// instead of two: 506 KiB of LDS traffic per phase and CU = 3.95 k cycles at 128 B/clk against 3.07 k cycles of MFMA time -- eats what it gains.  The s_memtime stamps of
// the four-wave kernel also say the chip runs this kernel at ~1.86 GHz (230 k cycles per tile x 46.8 tiles in 5.8 ms), below the 2.07 GHz of the MFMA-only loops.
// To build it again: paste the kernel into csrc/gemm_rs.hip ahead of the host section and launch it with 512 threads for plain (no aux) problems.

// ---- eight-wave form: TWO waves per SIMD, 32 rows per wave (256 registers each).  Same tile (256 rows), same ring, same weight stream.  What it buys: while one wave of a
// SIMD sits in the issue of a store or an LDS-DMA piece that the memory pipeline is not taking yet, the other issues MFMAs -- with one wave per SIMD the store time, the
// LDS-DMA time and the MFMA time of a phase ADD (ablation: 2.7 + 1.4 + 1.8 ms of 5.8 at N = 2304).  What it costs: every fragment read from LDS feeds ONE MFMA again.
__global__ __launch_bounds__(512, 1) void gemm_rs8_kernel(RsArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [3][48 KiB] ring | bias f32[N]
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7
  float* sbias = (float*)(smem + RS_RING);
  for (int i = tid; i < g.N; i += 512) sbias[i] = g.bias ? g.bias[i] : 0.f;
  __syncthreads();
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  int seg = 0, slot = 0;
  auto dma1 = [&](int sg, int sl, auto i_, unsigned lane16) {  // piece i (0..5) of this wave's six 1-KiB pieces
    constexpr int i = decltype(i_)::v;
    if constexpr (RS_ABL & 4) return;
    const int p0 = w * 6 + (i & ~3);
    rs_glds<(i & 3) * 1024>(g.wpk + (int64_t)sg * RS_SEG + p0 * 1024, lane16, lds0 + (unsigned)(sl * RS_SEG + p0 * 1024));
  };
  {
    const unsigned l16 = rs_lane() * 16u;
    dma1(0, 0, RsIC<0>(), l16); dma1(0, 0, RsIC<1>(), l16); dma1(0, 0, RsIC<2>(), l16); dma1(0, 0, RsIC<3>(), l16); dma1(0, 0, RsIC<4>(), l16); dma1(0, 0, RsIC<5>(), l16);
    dma1(1, 1, RsIC<0>(), l16); dma1(1, 1, RsIC<1>(), l16); dma1(1, 1, RsIC<2>(), l16); dma1(1, 1, RsIC<3>(), l16); dma1(1, 1, RsIC<4>(), l16); dma1(1, 1, RsIC<5>(), l16);
  }
  RS_WAIT_VM(6);

  bf16x8 nb[24];    // this wave's 32 A rows
  f32x16 S[2][2];   // [ping-pong][32-column half]
  int64_t st_row0 = 0; int st_col = 0; bool st_on = false, st_edge = false;

  auto load_a = [&](int tile) {
    const unsigned ln = rs_lane(); const int r = ln & 31, hh = ln >> 5;
    int64_t row = (int64_t)tile * 256 + w * 32 + r; if (row > g.M - 1) row = g.M - 1;
    const bf16_t* ap = g.A + row * g.lda + hh * 8;
#pragma unroll
    for (int s = 0; s < 24; ++s) nb[s] = *(const bf16x8*)(ap + 16 * s);
  };

  // phase: 48 MFMAs in 24 groups of 2; the waiting chunk leaves through the wave's own 6 KiB of the slot being refilled (image [32 rows][128 B + 16] = 4.5 KiB): written at
  // groups 0 / 1, row group k (8 rows) read at 3 + 4k and stored at 5 + 4k; LDS-DMA: piece 5 (behind the image) at 2, pieces 0..4 once the row groups they overlap are read
  auto phase = [&](auto buf_, auto mf_, int cc) {
    constexpr int BUF = decltype(buf_)::v; constexpr bool MF = decltype(mf_)::v;
    if constexpr (!(RS_ABL & 256)) RS_BAR();
    const unsigned ln = rs_lane(); const int hh = ln >> 5;
    const unsigned l16 = ln * 16u;
    const char* sb = smem + slot * RS_SEG + l16;
    const int sg2 = seg + 2 >= g.nseg ? seg + 2 - g.nseg : seg + 2, sl2 = slot == 0 ? 2 : slot - 1;
    char* stg = smem + sl2 * RS_SEG + w * 6144;
    bf16x8 fa[3][2];
    if constexpr (MF) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { fa[0][i] = *(const bf16x8*)(sb + i * 1024); fa[1][i] = *(const bf16x8*)(sb + (2 + i) * 1024); }
    }
    u32x4 sv;
    auto stage_wr = [&](int t) {
      char* wp = stg + (ln & 31) * 144 + 64 * t + 8 * hh;
      const f32x16& X = S[BUF ^ 1][t];
#pragma unroll
      for (int q = 0; q < 4; ++q) *(u32x2*)(wp + 16 * q) = u32x2{rs_pack2(X[4 * q], X[4 * q + 1]), rs_pack2(X[4 * q + 2], X[4 * q + 3])};
    };
    auto stage_rd = [&](int k) { sv = *(const u32x4*)(stg + (8 * k + (ln >> 3)) * 144 + (ln & 7) * 16); };
    char* cp0 = (char*)(g.C + (st_row0 + (ln >> 3)) * g.ldc + st_col + (ln & 7) * 8);
    const int64_t cstep = g.ldc * 16;
    auto stage_st = [&](int k) {
      u32x4* dp = (u32x4*)(cp0 + k * cstep);
      if constexpr (RS_ABL & 1) dp = (u32x4*)((char*)g.C + ((uintptr_t)((char*)dp - (char*)g.C) & 0xffff0));
      if (st_edge) {
        if (st_row0 + 8 * k + (int64_t)(ln >> 3) < g.M) { if (g.nt_store) __builtin_nontemporal_store(sv, dp); else *dp = sv; }
      } else { if (g.nt_store) __builtin_nontemporal_store(sv, dp); else *dp = sv; }
    };
    f32x16 bz;
    auto load_bias = [&](int t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b = *(const f32x4*)(sbias + 64 * cc + 32 * t + 8 * q + 4 * hh);
        bz[4 * q] = b[0]; bz[4 * q + 1] = b[1]; bz[4 * q + 2] = b[2]; bz[4 * q + 3] = b[3];
      }
    };
    if constexpr (MF) load_bias(0);
    rs_for<0, 24>([&](auto gq_) {
      constexpr int gq = decltype(gq_)::v;
      if constexpr (MF) {
        if (gq + 2 < 24) {
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[(gq + 2) % 3][i] = *(const bf16x8*)(sb + ((gq + 2) * 2 + i) * 1024);
        }
        if (gq == 11) load_bias(1);
      }
      if (st_on && !(RS_ABL & 16)) {
        if (gq == 0) stage_wr(0);
        if (gq == 1) stage_wr(1);
        if (gq >= 5 && gq <= 17 && (gq - 5) % 4 == 0) stage_st((gq - 5) / 4);
        if (gq >= 3 && gq <= 15 && (gq - 3) % 4 == 0) stage_rd((gq - 3) / 4);
      }
      if constexpr (MF) {
        if (gq == 2) dma1(sg2, sl2, RsIC<5>(), l16);
        if (gq == 6) dma1(sg2, sl2, RsIC<0>(), l16);    // bytes 0 .. 1023: row group 0 (read at 3)
        if (gq == 10) dma1(sg2, sl2, RsIC<1>(), l16);   // .. 2047: row groups 0, 1 (7)
        if (gq == 14) dma1(sg2, sl2, RsIC<2>(), l16);   // .. 3071: 1, 2 (11)
        if (gq == 18) dma1(sg2, sl2, RsIC<3>(), l16);   // .. 4095: 2, 3 (15)
        if (gq == 20) dma1(sg2, sl2, RsIC<4>(), l16);   // .. 5119: 3
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (MF) {
        constexpr int t = gq / 12, s0 = 2 * (gq % 12);
        if constexpr (RS_ABL & 8) { asm volatile("" ::"v"(fa[gq % 3][0]), "v"(fa[gq % 3][1])); }
        else {
          if constexpr (s0 == 0) S[BUF][t] = MFMA32(fa[gq % 3][0], nb[s0], bz); else S[BUF][t] = MFMA32(fa[gq % 3][0], nb[s0], S[BUF][t]);
          S[BUF][t] = MFMA32(fa[gq % 3][1], nb[s0 + 1], S[BUF][t]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (MF) {
      if constexpr (!(RS_ABL & 128)) { if (st_on && !st_edge && !(RS_ABL & 16)) RS_WAIT_VM(10); else RS_WAIT_VM(6); }  // this phase's 6 LDS-DMA (+ 4 stores) may stay in flight
      seg = seg + 1 == g.nseg ? 0 : seg + 1; slot = slot == 2 ? 0 : slot + 1;
    }
  };

  for (int tile = blockIdx.x; tile < g.tiles; tile += gridDim.x) {
    const int64_t row0 = (int64_t)tile * 256 + w * 32;
    const bool edge = (int64_t)tile * 256 + 256 > g.M;
    load_a(tile);
    phase(RsIC<0>(), RsIC<1>(), 0);   // peeled (see gemm_rs_kernel): the A rows arrive under phase 0
    st_row0 = row0; st_col = 0; st_on = true; st_edge = edge;
    phase(RsIC<1>(), RsIC<1>(), 1);
    st_col = 64;
    for (int c = 2; c < g.nseg; c += 2) {
      phase(RsIC<0>(), RsIC<1>(), c);
      st_col = 64 * c;
      phase(RsIC<1>(), RsIC<1>(), c + 1);
      st_col = 64 * (c + 1);
    }
  }
  if (st_on) phase(RsIC<0>(), RsIC<0>(), 0);
  RS_WAIT_VM(0);
}

