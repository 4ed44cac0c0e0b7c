// REJECTED EXPERIMENT (round 2), kept for the record; not built, not part of libspa3d_hip.so.
// Paced-store persistent NT kernel: measured 8.42 ms against 6.85 ms for the shipped 256x256 persistent kernel on the QKV shape
// (M = 3.06 M, N = 2304, K = 384) and 20-25 % slower on every other short-K shape (profiles/r02_gemm_paced_rejected.log) -- the same
// as the plain 128x256 kernel with unpaced stores.  Why pacing cannot work on this chip: vmcnt retires in issue order and counts
// stores, so every LDS-DMA load issued after a store chunk retires behind that chunk's acknowledgement; with a ring two K-tiles
// deep the first wait that depends on a chunk comes two K-tile periods (about 2 us) after it was issued, less than a store's
// round trip under load.  Hiding the stores would need the next D K-tiles in flight BEFORE the stores, i.e. a ring deeper than LDS holds.
// It slots into gemm_fast.hip before aligned16() together with the dispatch branch quoted at the end of this file.
// =================================================================================================================
// Paced-store persistent kernel for short K (K = 384 ... 768): 128x256 tile (64 accumulator registers per lane), so a finished tile's
// packed 16-bit rows (32 registers, 64 with a second output) can wait in registers BESIDE the next tile's accumulators, and its stores
// are issued during the next tile's first four K-tiles, one 16-row pass per K-tile, each right AFTER that K-tile's LDS-DMA.
// vmcnt retires in issue order and counts stores: a wait for the loads of K-tile t+1 never covers stores issued after them, so every
// chunk has two K-tile periods to drain under MFMA work -- the 256x256 kernel's stores all sit between the next tile's K-tiles 1 and 2
// and are waited for in full during K-tile 0 (35 % of a K = 384 tile's time).  Edge tiles (rows past M) store at once and drain.
// Same two-phase-per-K-tile schedule, LDS layout and epilogue math as gemm_nt8pp_kernel<., ., true, .>.  Host: K >= 384 (the four
// chunks need four K-tiles with a successor), 256 | N, bf16 outputs, no accumulate / row remap.
// =================================================================================================================
template <bool AUX>
__global__ __launch_bounds__(512, 2) void gemm_nt8pq_kernel(NtArgs g) {
  constexpr int WMT = 4, WNT = 4;
  constexpr int BM = 32 * WMT, BN = 64 * WNT, NA = WMT / 4, NB = WNT / 2, HM = WMT / 2, HN = WNT / 2;
  constexpr int ASLOT = BM * 128, BBUF = BN * 128, BOFF = 3 * ASLOT, NKT = 2 * NA + 2 * NB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int xcd = blockIdx.x & 7, cu_slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int per_xcd = ((g.tiles_m + 7) / 8) * g.tiles_n;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int sr = lane >> 3, scp = lane & 7;
  const int sc = (scp ^ sr) * 8;
  const int nt = g.K / 64;  // >= 6 (host)
  auto next_valid = [&](int idx) { while (idx < per_xcd && (idx / g.tiles_n) * 8 + xcd >= g.tiles_m) idx += nslot; return idx; };
  int rga[2][NA], rgb[2][NB];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < NA; ++i) { const int gi = i * 8 + w; rga[h][i] = (gi / WMT) * (WMT * 16) + h * (WMT * 8) + (gi % WMT) * 8; }
#pragma unroll
    for (int i = 0; i < NB; ++i) { const int gi = i * 8 + w; rgb[h][i] = (gi / WNT) * (WNT * 16) + h * (WNT * 8) + (gi % WNT) * 8; }
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
      for (int i = 0; i < NA; ++i) { const int rg = rga[h][i] < maxgrp ? rga[h][i] : maxgrp; glds16_s(src + rg * lda2, oal, base + rga[h][i] * 128); }
  };
  auto stageB = [&](int h, int kt) {
    char* base = smem + (kt & 1) * BBUF + BOFF;
    const char* src = baseB + kt * 128;
#pragma unroll
    for (int i = 0; i < NB; ++i) glds16_s(src + rgb[h][i] * ldb2, obl, base + rgb[h][i] * 128);
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
  const bool dual = g.pre_out != nullptr;
  // epilogue geometry: a pass = one 16-row accumulator row-tile of the wave's 64 columns: [16 rows][16 chunks of 16 B] f32 in a
  // wave-private 4-KiB image (chunk ^= row), read back as 8-column groups: item id = 64 it + lane -> row id / 8, group id % 8
  constexpr int NP = WMT;
  int prow[2], pg8[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) { const int id = it * 64 + lane; prow[it] = id / 8; pg8[it] = id - prow[it] * 8; }
  char* reg = w < 4 ? smem + 2 * ASLOT + w * 4096 : smem + 3 * ASLOT + 2 * BBUF + (w - 4) * 4096;
  // the previous tile's finished rows and where they go
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  uint4 hC[NP][2], hP[NP][2];
  bool have_held = false; int64_t pm0 = 0; int pn0 = 0;
  auto issue_pass = [&](auto pc) {
    constexpr int p = decltype(pc)::value;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int64_t ci = (pm0 + wr * (WMT * 16) + p * 16 + prow[it]) * g.ldc + (pn0 + wc * (WNT * 16) + pg8[it] * 8);
#ifdef SPA3D_ABLATE
      if (g.ablate & 1) continue;
#endif
      u32x4* cp = (u32x4*)((bf16_t*)g.C + ci);
      const u32x4 o = u32x4{hC[p][it].x, hC[p][it].y, hC[p][it].z, hC[p][it].w};
      if (g.nt_store) __builtin_nontemporal_store(o, cp); else *cp = o;
      if (dual) {
        u32x4* pp = (u32x4*)(g.pre_out + ci);
        const u32x4 q = u32x4{hP[p][it].x, hP[p][it].y, hP[p][it].z, hP[p][it].w};
        if (g.nt_store) __builtin_nontemporal_store(q, pp); else *pp = q;
      }
    }
  };

  while (true) {
    NT8P_BAR();                 // K-tile 0 is visible to every wave; every wave has left the previous epilogue
    if (wr == 1) NT8P_BAR();    // the stagger
    f32x4 acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
      for (int j = 0; j < WNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 aq[HM][2], bq0[HN][2], bq1[HN][2];
#define NT8Q_MFMA(AI, BJ, BQ)                                                                                                         \
  __builtin_amdgcn_s_setprio(1);                                                                                                      \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < HM; ++i) _Pragma("unroll") for (int j = 0; j < HN; ++j) \
      acc[(AI) + i][(BJ) + j] = MFMA16(BQ[j][ks], aq[i][ks], acc[(AI) + i][(BJ) + j]);       \
  __builtin_amdgcn_s_setprio(0);
    int aslot = 0, aslot2 = 2;
    for (int t = 0; t < nt; ++t) {
      const char* sa = smem + aslot * ASLOT;
      const char* sb = smem + (t & 1) * BBUF;
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
      NT8Q_MFMA(0, 0, bq0)
      NT8Q_MFMA(0, HN, bq1)
      NT8P_BAR();
      // ---------------- PB: A-q1 | stage B(t+2), then one pass of the previous tile's stores | wait K-tile t+1 | quadrants (1,1) (1,0)
#pragma unroll
      for (int i = 0; i < HM; ++i) { aq[i][0] = *(const bf16x8*)(sa + a_off + (HM + i) * 2048 + x0); aq[i][1] = *(const bf16x8*)(sa + a_off + (HM + i) * 2048 + x1); }
      if (t + 2 < nt) {
        stageB(0, t + 2); stageB(1, t + 2);
        if (have_held) {
          // younger than K-tile t+1's loads: pass t-1's stores, K-tile t+2's loads, pass t's stores (passes exist for t = 0..3)
          if (t == 0) { issue_pass(std::integral_constant<int, 0>{}); if (dual) NT8P_WAIT_VM(NKT + 4); else NT8P_WAIT_VM(NKT + 2); }
          else if (t == 1) { issue_pass(std::integral_constant<int, 1>{}); if (dual) NT8P_WAIT_VM(NKT + 8); else NT8P_WAIT_VM(NKT + 4); }
          else if (t == 2) { issue_pass(std::integral_constant<int, 2>{}); if (dual) NT8P_WAIT_VM(NKT + 8); else NT8P_WAIT_VM(NKT + 4); }
          else if (t == 3) { issue_pass(std::integral_constant<int, 3>{}); if (dual) NT8P_WAIT_VM(NKT + 8); else NT8P_WAIT_VM(NKT + 4); }
          else if (t == 4) { if (dual) NT8P_WAIT_VM(NKT + 4); else NT8P_WAIT_VM(NKT + 2); }
          else NT8P_WAIT_VM(NKT);
        } else NT8P_WAIT_VM(NKT);
      } else NT8P_WAIT_VM(0);
      NT8P_WAIT_LGKM(0);
      NT8P_BAR();
      NT8Q_MFMA(HM, HN, bq1)
      NT8Q_MFMA(HM, 0, bq0)
      NT8P_BAR();
      aslot = aslot == 2 ? 0 : aslot + 1; aslot2 = aslot2 == 2 ? 0 : aslot2 + 1;
    }
#undef NT8Q_MFMA
    if (wr == 0) NT8P_BAR();  // pairs with wave-row 1's extra barrier: every LDS read of this tile is done

    const int64_t cm0 = m0; const int cn0 = n0;
    const bool interior = cm0 + BM <= g.M;
    int rbase = wr * (WMT * 16);
    asm volatile("" : "+v"(rbase));
    auto gn_of = [&](int it) { return cn0 + wc * (WNT * 16) + pg8[it] * 8; };
    float b8[2][8];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r) b8[it][r] = 0.f;
      if (g.bias) { const int gn = gn_of(it); const float4 b0 = *(const float4*)(g.bias + gn), b1 = *(const float4*)(g.bias + gn + 4);
        b8[it][0] = b0.x; b8[it][1] = b0.y; b8[it][2] = b0.z; b8[it][3] = b0.w; b8[it][4] = b1.x; b8[it][5] = b1.y; b8[it][6] = b1.z; b8[it][7] = b1.w; }
#pragma unroll
      for (int r = 0; r < 8; ++r) asm volatile("" ::"v"(b8[it][r]));
    }
    auto stage_pass = [&](int i) {
#pragma unroll
      for (int jj = 0; jj < WNT; ++jj) *(f32x4*)(reg + fr * 256 + (((jj * 4 + fq) ^ fr) << 4)) = acc[i][jj];
      __builtin_amdgcn_wave_barrier();
    };
    auto read_item = [&](int it, float (&v)[8]) {
      const int row = prow[it];
      const f32x4 v0 = *(const f32x4*)(reg + row * 256 + (((2 * pg8[it]) ^ row) << 4));
      const f32x4 v1 = *(const f32x4*)(reg + row * 256 + (((2 * pg8[it] + 1) ^ row) << 4));
      v[0] = v0[0]; v[1] = v0[1]; v[2] = v0[2]; v[3] = v0[3]; v[4] = v1[0]; v[5] = v1[1]; v[6] = v1[2]; v[7] = v1[3];
    };
    auto pack8 = [](const float (&v)[8]) {
      uint4 o4; unsigned* op = (unsigned*)&o4;
#pragma unroll
      for (int r = 0; r < 4; ++r) op[r] = (unsigned)f2bf(v[2 * r]) | ((unsigned)f2bf(v[2 * r + 1]) << 16);
      return o4;
    };
    // all loads and all math of the epilogue first (aux double-buffered one pass ahead); the results wait in hC / hP
    uint4 ax[2][2];
    auto load_ax = [&](int p, uint4 (&dst)[2]) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        int64_t gm = cm0 + (rbase + p * 16 + prow[it]);
        if (gm > g.M - 1) gm = g.M - 1;
        dst[it] = *(const uint4*)(g.aux + gm * g.ldc + gn_of(it));
      }
    };
    if constexpr (AUX) load_ax(0, ax[0]);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if constexpr (AUX) { if (p + 1 < NP) load_ax(p + 1, ax[(p + 1) & 1]); }
      stage_pass(p);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        float v[8]; read_item(it, v);
        if constexpr (AUX) hC[p][it] = nt_compute8_aux(g, v, b8[it], ax[p & 1][it]);
        else {
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = g.alpha * v[r] + b8[it][r];
          if (dual) hP[p][it] = pack8(v);
          if (g.epi == EPI_GELU) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = gelu_tanh_fast_f(v[r]);
          }
          hC[p][it] = pack8(v);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    pm0 = cm0; pn0 = cn0; have_held = interior;
    idx = next_valid(idx + nslot);
    const bool more = idx < per_xcd;
    if (more) {
      m0 = (int64_t)((idx / g.tiles_n) * 8 + xcd) * BM; n0 = (idx % g.tiles_n) * BN;
      set_tile(m0, n0);
      prologue();
    }
    if (!interior) {  // edge tile: predicated stores now, drained before the next K-loop
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int64_t gm = cm0 + (rbase + p * 16 + prow[it]);
          if (gm < g.M NT_ABLATE_STORES) {
            const int64_t ci = gm * g.ldc + gn_of(it);
            *(uint4*)((bf16_t*)g.C + ci) = hC[p][it];
            if (dual) *(uint4*)(g.pre_out + ci) = hP[p][it];
          }
        }
    }
    if (!more) {
      if (have_held) { issue_pass(std::integral_constant<int, 0>{}); issue_pass(std::integral_constant<int, 1>{});
                       issue_pass(std::integral_constant<int, 2>{}); issue_pass(std::integral_constant<int, 3>{}); }
      break;
    }
    if (!interior) NT8P_WAIT_VM(0);
    else NT8P_WAIT_VM(NKT);   // only the next tile's second K-tile may be outstanding: its first has landed (no stores were issued)
  }
}


/* dispatch branch (gemm_nt_bf16):
    if (pers_ok && c->nt_paced && d.N % 256 == 0 && d.K >= 384 && d.K <= c->nt_paced_kmax && !d.out_f32) {  // paced-store 128x256 (see the kernel)
      NtArgs g2 = g; g2.tiles_m = (int)((g.M + 127) / 128); g2.tiles_n = g.N / 256;
      static bool attrq = false;
      if (!attrq) {
        (void)hipFuncSetAttribute((const void*)gemm_nt8pq_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)gemm_nt8pq_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        attrq = true;
      }
      if (d.aux) gemm_nt8pq_kernel<true><<<256, 512, 131072, c->stream>>>(g2);
      else gemm_nt8pq_kernel<false><<<256, 512, 131072, c->stream>>>(g2);
    } else
*/
