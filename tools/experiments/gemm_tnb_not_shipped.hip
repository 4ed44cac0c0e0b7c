// NOT SHIPPED (round 4), kept for the next round; not built.  The dW (TN) kernel with a large register tile: one wave per SIMD, workgroup tile 384 x 256 (all rows of a
// d = 384 weight gradient), wave tile 96 x 256 = 384 accumulator registers.  Motivation and expected gain: DESIGN.md "what comes next", item 4 (the 8-wave kernel spends
// ~75 % of its time on its LDS-DMA operand stream; this geometry moves 0.21 KiB per MFMA instead of 0.33).  State when it was put aside:
//   * CORRECT: tools/race_screen.py, TN M = 70000, N = 768, Ki = 384: max relative error 5.2e-7 over 40 runs (the 8-wave kernel: 6e-7), incl. the fused column sums' code path;
//   * SLOW: 10.2 ms at M = 3.07 M, N = 1536 against 4.1 ms for gemm_tn8p_kernel<2,3,2>, because it does not fit its registers: 256 VGPR + 256 AGPR, 112 spilled registers in the
//     plain instance (176 with the column sums), 16 scratch accesses per phase -- each a vector-memory operation inside the counted-vmcnt scheme.  A first restructuring
//     (branch-free zero-page select, barrier at the phase end) fell off the allocator's cliff: 441 spilled registers.  Budget as designed: 384 accumulators + 24 (A fragments,
//     two sets) + 32 (B fragments, rolling) + 2 lane bases + 4 pointers + temporaries; what the fused MLP forward needed at the same budget was one feature at a time with the
//     spill count checked after each.
// It belongs in csrc/gemm_fast.hip behind gemm_tn8p_kernel (it uses TnArgs, tr_off, ds_read_tr16_b64, GLDS16, the NT8P_* macros) with this dispatch in gemm_tn_bf16:
//     if (Ki == 384 && N % 256 == 0 && g.brow_group == 0 && g.lda % 8 == 0 && g.ldb % 8 == 0) { launch_tnb(c, g); SPA_LAUNCH_CHECK(c); return true; }

// =================================================================================================================
// TN, large register tile (round 4).  The 8-wave kernel above spends ~75 % of its time on its operand stream (tools/ablate_gemm_tn.py: without any MFMA
// the 128 x 384 form still takes 3.0 of 4.1 ms): every output tile streams its own 256 + 768 B per m-row through LDS-DMA, 12-18 tiles per launch.  Here ONE wave
// per SIMD holds a 96 x 256 accumulator tile (3 x 8 MFMA tiles = 384 registers), four waves cover ALL 384 rows of the weight gradient x 256 columns: 768 + 512 B per
// m-row and tile, 0.21 KiB of LDS-DMA and 0.46 KiB of transposed fragment reads per MFMA instead of 0.33 / 0.83.  Ki == 384, 256 | N, no row remap.
//   Quarter (16 m-rows) = five [16][128] sub-images (A columns 0-127, 128-255, 256-383; B columns n0 .. +127, +128 .. +255) in the transposed-read layout = 20 KiB,
//   5 LDS-DMA per wave (sub-image i2, rows 4w .. 4w+3).  Ring of 7 quarters; phase = 2 quarters = 48 MFMAs per wave between barriers; the LDS-DMA of quarters q+5, q+6
//   is issued in phase q/2 (the slots of the previous phase's quarters).  No second wave per SIMD to stagger against, so the fragment reads are pipelined by hand:
//   while the MFMAs of quarter q issue (column block j = 0..7: three MFMAs each), the A fragments of quarter q+1 are read into the second A set and B fragment j of
//   quarter q+1 replaces B fragment j right behind its last MFMA.  The reads are untracked (asm); every fragment has 20 younger LDS operations at its first use,
//   so one s_waitcnt lgkmcnt(15) (the counter's maximum; LDS operations return in order) ahead of each MFMA triple is enough.
// =================================================================================================================
template <bool CS>  // CS: also accumulate the column sums of B (the bias gradient)
__global__ __launch_bounds__(256, 1) void gemm_tnb_kernel(TnArgs g) {
  constexpr int R = 7, QB = 20480, D = 5;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int xcd = blockIdx.x & 7; int jb = blockIdx.x >> 3;
  const int ntile = g.tiles_n;
  const int sp = (jb / ntile) * 8 + xcd; jb %= ntile;
  if (sp >= g.splits) return;
  const int n0 = jb * 256;
  const int64_t mbeg = (int64_t)sp * g.rows_per_split;
  int64_t mend = mbeg + g.rows_per_split; if (mend > g.M) mend = g.M;
  const int nq = (int)((mend - mbeg + 15) / 16);
  if (nq <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- staging.  Named scalars (an array of pointers selected against g.zero is demoted to scratch: see the kernel above)
  const int sr = lane >> 4, scp = lane & 15, r16 = 4 * w + sr;
  const int chs = scp ^ (((r16 & 3) << 2) | ((r16 >> 2) & 3));
  const bf16_t* pa0 = g.A + (mbeg + r16) * g.lda + chs * 8;      // sub-images 1, 2 / 4: + 128, + 256 / + 128 columns (added at the call: five live pointers cost 10 registers)
  const bf16_t* pb0 = g.B + (mbeg + r16) * g.ldb + n0 + chs * 8;
  const int64_t stepA = 16 * g.lda, stepB = 16 * g.ldb;
  const int ldsw = w * 1024;
  int q_issue = 0, slot_issue = 0;
  bool row_ok = mbeg + r16 < mend;  // this lane's row of quarter q_issue exists
  auto stage1 = [&](int i2) {  // one LDS-DMA of quarter q_issue (i2 is a compile-time constant at every call)
    char* base = smem + slot_issue * QB + i2 * 4096 + ldsw;
    const bf16_t* p = i2 < 3 ? pa0 + 128 * i2 : pb0 + 128 * (i2 - 3);
    p = row_ok ? p : g.zero;  // rows past the end contribute exactly 0 (a select, not a branch: branches here split every live range of the loop)
    GLDS16(p, base);
  };
  auto stage_next = [&]() { pa0 += stepA; pb0 += stepB; ++q_issue; row_ok = mbeg + (int64_t)q_issue * 16 + r16 < mend; slot_issue = slot_issue == R - 1 ? 0 : slot_issue + 1; };

  f32x16 acc[3][8];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposed-read addressing (as above): 16-lane group gq covers operand rows 16 (gq & 1) .. +15 and k = 8 (gq >> 1) .. +7
  const int gq = lane >> 4, lq = (lane & 15) >> 2, lp = lane & 3;
  const int kq = 8 * (gq >> 1) + lq;
  const int cbase = 2 * (gq & 1) + (lp >> 1);
  // byte offset of tile t's fragment inside its sub-image = tr_off(row, 4 t + cbase) = 256 row + 16 ((4 t + cbase) ^ swz(row)) = (256 row | 16 (cbase ^ swz(row))) ^ 64 t:
  // one lane constant per row (kq, kq + 4) and one v_xor per read instead of 22 offset registers; the sub-image offset rides in the instruction
  const int swz0 = ((kq & 3) << 2) | ((kq >> 2) & 3), swz1 = (((kq + 4) & 3) << 2) | (((kq + 4) >> 2) & 3);
  const int lb0 = (256 * kq + 16 * (cbase ^ swz0)) + 8 * (lp & 1), lb1 = (256 * (kq + 4) + 16 * (cbase ^ swz1)) + 8 * (lp & 1);
  const int ita = 3 * w;  // this wave's first A tile (0, 3, 6, 9): tile it sits in sub-image it >> 2 at 64 (it & 3)
  const bool do_cs = CS && w == 0;  // every wave holds all B fragments: wave 0 adds them up (v_dot2c against (1, 1))
  float cs[CS ? 8 : 1];
#pragma unroll
  for (int j = 0; j < (CS ? 8 : 1); ++j) cs[j] = 0.f;
  const unsigned ones2 = ONES2_16;

  const int nqp = (nq + 1) / 2 * 2;  // whole phases; a padding quarter lies past mend and stages the zero page
  const int npro = nqp < D ? nqp : D;
  for (int q = 0; q < npro; ++q) { stage1(0); stage1(1); stage1(2); stage1(3); stage1(4); stage_next(); }
  // quarters 0 .. 2 must be visible to every wave before the first phase: it reads 0 and 1 for its MFMAs and 2 for the fragment prefetch
  if (nqp >= D) NT8P_WAIT_VM(10); else NT8P_WAIT_VM(0);
  NT8P_BAR();
  auto rd_a = [&](const char* sq, int i, int h) {  // A tile ita + i (the wave's tiles can straddle two sub-images: runtime sub-image offset)
    const int it = ita + i;
    return ds_read_tr16_b64(sq + (it >> 2) * 4096 + ((h ? lb1 : lb0) ^ (64 * (it & 3))));
  };
  auto rd_b = [&](const char* sq, int j, int h) { return ds_read_tr16_b64(sq + (3 + (j >> 2)) * 4096 + ((h ? lb1 : lb0) ^ (64 * (j & 3)))); };
  uint2 fa[2][3][2], fb[8][2];
  {
    const char* sq = smem;  // quarter 0 sits in slot 0
#pragma unroll
    for (int i = 0; i < 3; ++i) { fa[0][i][0] = rd_a(sq, i, 0); fa[0][i][1] = rd_a(sq, i, 1); }
#pragma unroll
    for (int j = 0; j < 8; ++j) { fb[j][0] = rd_b(sq, j, 0); fb[j][1] = rd_b(sq, j, 1); }
    NT8P_WAIT_LGKM(0);
  }
  int slot = 0;
  for (int q = 0; q < nqp; q += 2) {
    const bool more = q + D < nqp;  // this phase still issues LDS-DMA (quarters q + 5, q + 6)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int sn = slot + u + 1; if (sn >= R) sn -= R;
      const char* sqn = smem + sn * QB;  // the next quarter's slot (fragment prefetch)
      // A fragments of the next quarter into the other set
#pragma unroll
      for (int i = 0; i < 3; ++i) { fa[u ^ 1][i][0] = rd_a(sqn, i, 0); fa[u ^ 1][i][1] = rd_a(sqn, i, 1); }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (more && j >= 1 && j <= 5) stage1(j - 1);
        NT8P_WAIT_LGKM(15);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 bf = __builtin_bit_cast(bf16x8, make_uint4(fb[j][0].x, fb[j][0].y, fb[j][1].x, fb[j][1].y));
#pragma unroll
        for (int i = 0; i < 3; ++i)
          acc[i][j] = MFMA32(__builtin_bit_cast(bf16x8, make_uint4(fa[u][i][0].x, fa[u][i][0].y, fa[u][i][1].x, fa[u][i][1].y)), bf, acc[i][j]);
        if constexpr (CS) if (do_cs) {
          asm(DOT2C_F32_16 " %0, %1, %2" : "+v"(cs[j]) : "v"(fb[j][0].x), "v"(ones2));
          asm(DOT2C_F32_16 " %0, %1, %2" : "+v"(cs[j]) : "v"(fb[j][0].y), "v"(ones2));
          asm(DOT2C_F32_16 " %0, %1, %2" : "+v"(cs[j]) : "v"(fb[j][1].x), "v"(ones2));
          asm(DOT2C_F32_16 " %0, %1, %2" : "+v"(cs[j]) : "v"(fb[j][1].y), "v"(ones2));
        }
        __builtin_amdgcn_sched_barrier(0);
        fb[j][0] = rd_b(sqn, j, 0); fb[j][1] = rd_b(sqn, j, 1);  // B fragment j of the next quarter
      }
      if (more) stage_next();
    }
    // quarter q + 4 (the next phase's prefetch target) has landed when only this phase's ten LDS-DMA can still be in flight
    if (more) NT8P_WAIT_VM(10); else NT8P_WAIT_VM(0);
    NT8P_BAR();  // quarters up to q + 4 are visible; every wave has left the two slots the next phase refills
    slot += 2; if (slot >= R) slot -= R;
  }
  NT8P_WAIT_LGKM(0);
  if constexpr (CS) if (do_cs) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float t = cs[j] + __shfl_xor(cs[j], 32, 64);
      if (lane < 32) atomicAdd(g.colsum + n0 + j * 32 + lane, t);
    }
  }
  // ---- epilogue: 32x32 C/D map col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const int col = lane & 31, rb = 4 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int gn0 = n0 + j * 32;
      float* cb = g.C; int cn = gn0 + col;
      if (g.seg_n > 0) { const int sg = gn0 / g.seg_n; if (sg > 0) { cb = g.Cseg[sg > 1]; cn -= sg * g.seg_n; } }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gi = 96 * w + i * 32 + (r & 3) + 8 * (r >> 2) + rb;
        atomicAdd(cb + (int64_t)gi * g.ldc + cn, acc[i][j][r]);
      }
    }
}

static void launch_tnb(spa3d_ctx* c, TnArgs g) {
  g.tiles_i = 1; g.tiles_n = g.N / 256;
  const int64_t tiles = g.tiles_n;
  // M-splits from the same makespan model as the 8-wave kernels (rounds on the fullest XCD x rows x time per row + splits x atomic bytes)
  const double t_row = 2.0 * 384 * 256 / 4.5e12, t_atom = (double)g.Ki * g.N * 4.0 / 1.3e12;
  const int64_t smax = std::max<int64_t>(1, std::min<int64_t>(g.M / 4096, 2048));
  double best = 1e30; int64_t splits = 1;
  for (int64_t sc = 1; sc <= smax; ++sc) {
    const int64_t rps_c = ((g.M + sc - 1) / sc + 63) / 64 * 64;
    const int64_t per_xcd = tiles * ((sc + 7) / 8);
    const double t = (double)((per_xcd + 31) / 32) * (double)rps_c * t_row + (double)sc * t_atom;
    if (t < best * 0.999) { best = t; splits = sc; }
  }
  int64_t rps = ((g.M + splits - 1) / splits + 63) / 64 * 64;
  splits = (g.M + rps - 1) / rps;
  g.splits = (int)splits; g.rows_per_split = rps;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_tnb_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 7 * 20480);
    (void)hipFuncSetAttribute((const void*)gemm_tnb_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 7 * 20480);
    attr = true;
  }
  if (g.colsum) gemm_tnb_kernel<true><<<(unsigned)(tiles * ((splits + 7) / 8 * 8)), 256, 7 * 20480, c->stream>>>(g);
  else gemm_tnb_kernel<false><<<(unsigned)(tiles * ((splits + 7) / 8 * 8)), 256, 7 * 20480, c->stream>>>(g);
}

