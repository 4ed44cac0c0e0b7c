// Evicted from 3dspa_code_amd/csrc/gemm_fast.hip in round 4 (all were opt-in, correct, and rejected by measurement in round 3):
//  * LayerNorm folded into the epilogue of the 128 x 384 8-phase kernel (SPA3D_LN_FOLD=1): outputs bit-identical to the stand-alone kernel, net zero in the
//    step (LayerNorm class -10.5 ms/step, NT GEMM class +10..12): profiles/r03_ln_fold.log, DESIGN.md "Round 3".  The block below sat in
//    gemm_nt8p_kernel<4, 6>'s epilogue behind `if (g.ln_out)`; NtArgs / GemmDesc carried ln_out, ln_stats, ln_scale.
//  * the 8-phase schedule at 128 x 128 tiles with two workgroups per CU (gemm_nt8p_kernel<4, 2>, SPA3D_NT_8P=42) and the 128 x 256 single-workgroup tile
//    (<4, 4>, =44): 602 / 640 TF/s against 788 TF/s at the QKV shape: profiles/r03_gemm_two_workgroups_per_cu.log.  Those were instantiations of the
//    shipped template (launch_nt8p<4, 2>, <4, 4>), no separate source.
//  * SPA3D_NT_PREF384 (persistent 128 x 384 kernel also for K = 384 shapes with 384 | N: NT class 934 -> 991 ms/step) and SPA3D_NT_COARSE384 (two-phase
//    K-tiles in that kernel: no change): profiles/r03_in_step_toggles.log.  Dispatch conditions only.
// Not built.
  if constexpr (WMT == 4 && WNT == 6) {
    // ---- LayerNorm folded into this epilogue (the tile holds whole 384-wide rows; a row's columns sit in the four waves of a wave-row).
    // Per half: (A) final values (bias, residual) -> C, and their 16-bit-rounded copies back into the wave's region; (B) lanes 0..31 sum
    // their row's 96 values, partial (sum, sum of squares) into the wave's slack behind its region; workgroup barrier; (C) every lane
    // combines the four partials of its row and writes LN(row) * scale; wave-column 0 writes (mean, rstd).  Host guarantees: N == 384,
    // no GELU / f32 output / accumulate / row remap.  The scale vector is staged in LDS before the first store (a load issued behind the
    // stores would wait for their drain: vmcnt retires in order).
    if (g.ln_out) {
      static_assert(EPI_STRIDE >= 12288 + 512 + 384, "slack for the LayerNorm partials and the scale slice");
      float2* mypart = (float2*)(reg + 12288);          // [2 halves][32 rows]
      float* mysc = (float*)(reg + 12288 + 512);        // this wave's 96 scale values
      for (int t = lane; t < 96; t += 64) mysc[t] = g.ln_scale[wc * 96 + t];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int i = 0; i < HM; ++i)
#pragma unroll
          for (int j = 0; j < WNT; ++j) *(f32x4*)(reg + (i * 16 + fr) * RB + (swz(j * 4 + fq, i * 16 + fr) << 4)) = acc[half * HM + i][j];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {  // (A)
          const int id = it * 64 + lane, row = id / CPR, c8 = id - row * CPR;
          f32x4* p0 = (f32x4*)(reg + row * RB + (swz(2 * c8, row) << 4)); f32x4* p1 = (f32x4*)(reg + row * RB + (swz(2 * c8 + 1, row) << 4));
          const f32x4 v0 = *p0, v1 = *p1;
          float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          const int64_t gm = m0 + wr * (WMT * 16) + half * (WMT * 8) + row;
          const int gn = wc * (WNT * 16) + c8 * 8;
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = g.alpha * v[r] + bv[it][r];
          if (g.aux) {
            const unsigned* xp = (const unsigned*)&auxv[half][it];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[2 * r] += unpack_lo(xp[r]); v[2 * r + 1] += unpack_hi(xp[r]); }
          }
          uint4 o4; unsigned* op = (unsigned*)&o4;
#pragma unroll
          for (int r = 0; r < 4; ++r) op[r] = (unsigned)f2bf(v[2 * r]) | ((unsigned)f2bf(v[2 * r + 1]) << 16);
          if (gm < g.M) {
            uint4* cp = (uint4*)((bf16_t*)g.C + gm * g.ldc + gn);
            if (g.nt_store) { typedef __attribute__((ext_vector_type(4))) unsigned u32x4; __builtin_nontemporal_store(u32x4{o4.x, o4.y, o4.z, o4.w}, (u32x4*)cp); }
            else *cp = o4;
          }
          *p0 = f32x4{unpack_lo(op[0]), unpack_hi(op[0]), unpack_lo(op[1]), unpack_hi(op[1])};   // what the stand-alone LayerNorm would read back
          *p1 = f32x4{unpack_lo(op[2]), unpack_hi(op[2]), unpack_lo(op[3]), unpack_hi(op[3])};
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 32) {  // (B) row `lane` of this half: 24 chunks of 4 columns
          float s_ = 0.f, ss_ = 0.f;
#pragma unroll
          for (int ch = 0; ch < 4 * WNT; ++ch) {
            const f32x4 t = *(const f32x4*)(reg + lane * RB + (swz(ch, lane) << 4));
#pragma unroll
            for (int r = 0; r < 4; ++r) { s_ += t[r]; ss_ += t[r] * t[r]; }
          }
          mypart[half * 32 + lane] = float2{s_, ss_};
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {  // (C)
          const int id = it * 64 + lane, row = id / CPR, c8 = id - row * CPR;
          float s_ = 0.f, ss_ = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float2 pj = ((const float2*)(smem + (wr * 4 + j) * EPI_STRIDE + 12288))[half * 32 + row];
            s_ += pj.x; ss_ += pj.y;
          }
          const float mu = s_ * (1.f / 384.f);
          const float var = fmaxf(ss_ * (1.f / 384.f) - mu * mu, 0.f);
          const float rs = rsqrtf(var + 1e-6f);
          const f32x4 v0 = *(const f32x4*)(reg + row * RB + (swz(2 * c8, row) << 4));
          const f32x4 v1 = *(const f32x4*)(reg + row * RB + (swz(2 * c8 + 1, row) << 4));
          const f32x4 s0 = *(const f32x4*)(mysc + c8 * 8), s1 = *(const f32x4*)(mysc + c8 * 8 + 4);
          const int64_t gm = m0 + wr * (WMT * 16) + half * (WMT * 8) + row;
          const int gn = wc * (WNT * 16) + c8 * 8;
          uint4 o4; unsigned* op = (unsigned*)&o4;
          op[0] = (unsigned)f2bf((v0[0] - mu) * rs * s0[0]) | ((unsigned)f2bf((v0[1] - mu) * rs * s0[1]) << 16);
          op[1] = (unsigned)f2bf((v0[2] - mu) * rs * s0[2]) | ((unsigned)f2bf((v0[3] - mu) * rs * s0[3]) << 16);
          op[2] = (unsigned)f2bf((v1[0] - mu) * rs * s1[0]) | ((unsigned)f2bf((v1[1] - mu) * rs * s1[1]) << 16);
          op[3] = (unsigned)f2bf((v1[2] - mu) * rs * s1[2]) | ((unsigned)f2bf((v1[3] - mu) * rs * s1[3]) << 16);
          if (gm < g.M) {
            uint4* yp = (uint4*)(g.ln_out + gm * 384 + gn);
            if (g.nt_store) { typedef __attribute__((ext_vector_type(4))) unsigned u32x4; __builtin_nontemporal_store(u32x4{o4.x, o4.y, o4.z, o4.w}, (u32x4*)yp); }
            else *yp = o4;
            if (wc == 0 && c8 == 0) *(float2*)(g.ln_stats + gm * 2) = float2{mu, rs};
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      return;
    }
  }
