// Evicted from 3dspa_code_amd/csrc/attention_fused.hip in round 4 (was opt-in as SPA3D_ATTN_BWD_MODE=4 / 5): the single-orientation fused attention
// backward of round 3 -- correct (it passed every op test and the in-model ragged test) and measured slower than the shipped two-role kernel
// (11.2 vs 10.7 ms at the track encoder shape; 253.6 vs 236.3 ms/step in the step).  Measurements: profiles/r03_attn_bwd_experiments.log,
// profiles/r03_attn_bwd1_first_version_pmc.csv, DESIGN.md "Round 3".  Not built; needs the surrounding file (helpers, AttnBwdArgs) to compile.
// =================================================================================================================
// Single-orientation form (round 3; S <= 160, the default there).  The two-role kernel above recomputes S, P, dP, dS in BOTH orientations so that no
// sum crosses a wave: per problem that is twice the score arithmetic (exp, mask, scale: ~13 VALU operations per element and orientation) and
// twice the S / dP MFMAs, and its key-tile role (3 tiles on 4 waves) is the critical path.  Here every wave owns KEY tiles only:
//   phase 1  S, dP with the key on the lane; P and dS pack straight into the B operands of dV^T += dO^T P and dK^T += Xq^T dS (as role (b));
//            the 16-bit dS tile ALSO goes to an LDS image dS[key][query] -- the one hand-off of the problem;
//   barrier
//   phase 2  every wave owns QUERY tiles: dQ^T[d][q] = sum_keys Xk^T[d][key] dS^T[key][q], both operands by ds_read_b64_tr_b16 from the Xk and
//            dS images, then the RMSNorm backward and the store (through wave tiles in the dead dO image).
// The q / k images hold the NORMALISED rows WITHOUT the learned scale, xq = 16-bit(q r_q), xk = 16-bit(k r_k) (r = rsqrt(mean x^2 + eps), kept per row
// in LDS): the scales enter in fp32 -- S = alpha sum_d xq[d] (xk g_k g_q)[d] with the wave's own key rows re-normalised from global memory as B
// operands (one 16-bit rounding, as q^ k^ have), dK^ = alpha g_q o (Xq^T dS), dQ^ = alpha g_k o (Xk^T dS^T) -- and the RMSNorm backward
// dX = r (g o dY - x^ mean(g o dY o x^)) reads x^ from the images in the accumulator layout: no raw q / k row is read a second time
// (the two-role kernel's 1.27x read traffic).  No V image: a wave's own v rows are its B operands, loaded from global memory in fragment shape.
// LDS at S_pad = 160: three images + dS[160][160] = 152 KB + 6 KB of row / column constants.
// Row constants ride in the accumulators' initial value (cdna_hip_programming.md App. B "Attention backward"): S' = Xq Kb^T - (m + l) / alpha and
// dP' = dO V^T - delta leave the MFMA chains ready, so without a key mask a score element costs fma + exp2 + mul + two half conversions.
// MASK: the key mask's finfo.min semantics (fully-masked rows attend uniformly) need the reference's operation order on the logits.
// =================================================================================================================
#ifndef SPA3D_ABL1
#define SPA3D_ABL1 0
#endif
// structure switches of the single-orientation kernel (measured alternatives; tools/ablate_attn.py builds the variants):
#ifndef SPA3D_B1_PREFETCH   // 1: the next problem's q / k rows are requested at the start of phase 2 (software pipeline over problems)
#define SPA3D_B1_PREFETCH 0
#endif
#ifndef SPA3D_B1_PIPE       // 1: hand-sequenced software pipeline of the key tile's query-tile pairs (no-mask 8-wave form)
#define SPA3D_B1_PIPE 0
#endif
#ifndef SPA3D_B1_FLUSH      // scale-gradient partials: 0 per-phase DPP reduction + LDS atomics, 1 DPP reduction only (WRONG sums: timing experiment)
#define SPA3D_B1_FLUSH 0
#endif
constexpr int DSROW = 352;  // bytes per dS-image row (160 queries x 2 B + 32): odd multiple of 32 mod 256 -> the transposed reads of phase 2 are conflict-free
constexpr int bwd1_tile_bytes(int KT) { return img_bytes(KT * 16) > 8 * WTILE ? img_bytes(KT * 16) : 8 * WTILE; }  // dO image / phase-2 wave tiles
constexpr int bwd1_small_bytes(int KT) { return 6 * KT * 16 * 4 + 7 * DH * 4; }
constexpr int bwd1_lds_bytes(int KT) { return 2 * img_bytes(KT * 16) + bwd1_tile_bytes(KT) + KT * 16 * DSROW + bwd1_small_bytes(KT); }

// rows into an LDS image as x^ = 16-bit(x r) (no learned scale); r per row into rinv[]
template <int NP, int RPP>
__device__ __forceinline__ void rows_store_xhat(RawRows<NP>& r, int S_pad, char* lds, float* rinv, int tid = threadIdx.x) {
  const int part = tid & 3, r0 = tid >> 2;
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int row = r0 + RPP * ps;
    float f[24]; float ss = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) { f[c * 8 + j] = bf2f(r.x[ps][c][j]); ss += f[c * 8 + j] * f[c * 8 + j]; }
    ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64);
    const float rr = rsqrtf(ss / DH + 1e-6f);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) r.x[ps][c][j] = f2bf(f[c * 8 + j] * rr);
    if (row < S_pad) {
      u16x8* d = (u16x8*)(lds + row_off(row)) + part;
      d[0] = r.x[ps][0]; d[4] = r.x[ps][1]; d[8] = r.x[ps][2];
      if (part == 0) rinv[row] = rr;
    }
  }
}

// RMSNorm backward of one 16-row tile held as (row fr, d = 16dt + 4fq + r): acc = dY (gradient w.r.t. x^ g, fp32), xrow = the row's x^ in the image
// (accumulator layout: 8 bytes per dt), rr its r, sc = g.  Emits dX through `emit(dt, u16x4)`, adds the scale-gradient partials dY o x^.
template <typename F>
__device__ __forceinline__ void rms_bwd_tile(int lane, f32x4 (&acc)[6], const char* xrow, float rr, const float* sc_, bool valid, float (&ds_acc)[6][4], F&& emit) {
  const int fq = lane >> 4;
  float gx = 0.f;
#pragma unroll
  for (int dt = 0; dt < 6; ++dt) {
    const u16x4 xv = *(const u16x4*)(xrow + (16 * dt + 4 * fq) * 2);
    const f32x4 sc = *(const f32x4*)(sc_ + dt * 16 + fq * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) gx += acc[dt][r] * sc[r] * bf2f(xv[r]);
  }
  gx += __shfl_xor(gx, 16, 64); gx += __shfl_xor(gx, 32, 64);
  gx /= DH;
#pragma unroll
  for (int dt = 0; dt < 6; ++dt) {
    const u16x4 xv = *(const u16x4*)(xrow + (16 * dt + 4 * fq) * 2);  // read again rather than held: 24 registers, and the LDS pipe is idle here
    const f32x4 sc = *(const f32x4*)(sc_ + dt * 16 + fq * 4);
    u16x4 o4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float x = bf2f(xv[r]);
      o4[r] = f2bf(rr * (acc[dt][r] * sc[r] - x * gx));
      if (valid) ds_acc[dt][r] += acc[dt][r] * x;
    }
    emit(dt, o4);
  }
}

// A phase's scale-gradient partials (lane (fr, fq): d = 16dt + 4fq + r, summed over the wave's tiles) -> LDS sums[96]: reduce over the 16 lanes that
// hold the same d, one LDS atomic per d and wave.  Keeps the 24 accumulators out of the staging phase's live set (with both q and k sets live
// across problems the compiler spilled the staging loads' addresses and serialised them behind vmcnt(0) reloads).
// sum over the 16 lanes of a DPP row (lanes 16g .. 16g+15), result in the row's lane 15: an inclusive scan by row_shr 1, 2, 4, 8 with zeros
// shifted in -- four VALU instructions (as __shfl_xor this is four ds_bpermute_b32 through the LDS pipe, and their latency)
__device__ __forceinline__ float row16_sum_lane15(float x) {
#define SPA_DPP_ADD(ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, true))
  SPA_DPP_ADD(0x111); SPA_DPP_ADD(0x112); SPA_DPP_ADD(0x114); SPA_DPP_ADD(0x118);
#undef SPA_DPP_ADD
  return x;
}
__device__ __forceinline__ void flush_ds_acc(float (&acc)[6][4], float* sums) {
  const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int dt = 0; dt < 6; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a = row16_sum_lane15(acc[dt][r]);
#if SPA3D_B1_FLUSH == 1
      if (fr == 15 && a == 12345.678f) sums[dt * 16 + fq * 4 + r] = a;
#else
      if (fr == 15) atomicAdd(sums + dt * 16 + fq * 4 + r, a);  // ds_add_f32
#endif
      acc[dt][r] = 0.f;
    }
}

struct Bwd1Lds {  // row / column constants of the single-orientation kernel
  float *kbias, *mrow, *lrow, *ndrow, *rq, *rk;  // [S_pad] each: key bias; (m, l) or -(m+l)/alpha; -delta; r of the q / k rows
  float *sred;                                   // [2][96] scale-gradient staging
  float *gq, *gk, *gqk, *agq, *agk;              // [96] each: g_q, g_k, g_q g_k, alpha g_q, alpha g_k
};

// `lane`: an OPAQUE copy of the lane index made by the caller per phase (opaque_tid): every lane-constant address below is then recomputed per
// tile instead of being hoisted out of the problem loop, spilled, and reloaded behind s_waitcnt vmcnt(0) -- which would also wait for the next
// problem's prefetched rows
template <int KT, bool MASK, bool SPLIT_TR = false>  // SPLIT_TR: the dO^T and xq^T fragments of a query-tile pair are read one after the other (24 fewer live registers: the 12-wave form)
__device__ __forceinline__ void bwd1_key_tile(int lane, const char* Qs, const char* dOs, const char* krow /* x^k image row of this lane's key */, char* dSrow,
                                              const Bwd1Lds& L, const mfma16x8 (&kb)[3], const mfma16x8 (&vb)[3], float kbv, float rrk, bool valid,
                                              bf16_t* gk_row, bf16_t* gv_row /* this lane's dk / dv row (d = 4fq), or nullptr past the end */,
                                              float (&ds_acc)[6][4]) {
  const int fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
  const float alpha = 0.10206207261596575f, c2 = 0.10206207261596575f * 1.4426950408889634f;  // 1/sqrt(96), times log2(e)
  const bool keep = kbv == 0.f;
  const float kb2 = kbv * 1.4426950408889634f;  // no mask: 0 (real key) or -inf (padding key: P = 0 exactly)
  f32x4 dva[6], dka[6];
#pragma unroll
  for (int dt = 0; dt < 6; ++dt) { dva[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dka[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#if SPA3D_B1_PIPE
  if constexpr (!MASK && !SPLIT_TR) {
    // Software pipeline over query-tile pairs, LDS reads sequenced by hand: while pair s2's transposed fragments (group T, 24 reads) are in
    // flight (T1: dO^T, T2: xq^T), the scores of pair s2+1 are computed from fragment reads issued before T1 (group A: first query tile) and behind
    // T2 (group B: second); every group is retired by a counted wait <= 15 (lgkmcnt is a 4-bit counter; LDS operations return in order).
    const char* qa = Qs + row_off(fr) + fq * 16;   // row-major fragment base of this lane (query-tile 0); tile qt: + qt * ROW16
    const char* da = dOs + row_off(fr) + fq * 16;
    const char* ma = (const char*)(L.mrow + fq * 4); const char* na = (const char*)(L.ndrow + fq * 4);  // + qt * 64
    struct Half { uint4 st0, dp0, q[3], d[3]; };
    auto issue_half = [&](Half& hh, int qt) {
      const char* q_ = qa + qt * ROW16; const char* d_ = da + qt * ROW16;
      hh.st0 = lds_b128_o<0>(ma + qt * 64); hh.dp0 = lds_b128_o<0>(na + qt * 64);
      hh.q[0] = lds_b128_o<0>(q_); hh.q[1] = lds_b128_o<64>(q_); hh.q[2] = lds_b128_o<128>(q_);
      hh.d[0] = lds_b128_o<0>(d_); hh.d[1] = lds_b128_o<64>(d_); hh.d[2] = lds_b128_o<128>(d_);
    };
    auto score_half = [&](const Half& hh, int hf, u16x8& tp_, u16x8& tds, char* dsp) {
      f32x4 st = __builtin_bit_cast(f32x4, hh.st0), dpt = __builtin_bit_cast(f32x4, hh.dp0);
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        st = MFMA16(__builtin_bit_cast(mfma16x8, hh.q[s]), kb[s], st);
        dpt = MFMA16(__builtin_bit_cast(mfma16x8, hh.d[s]), vb[s], dpt);
      }
      u16x4 d4_;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(st[r], c2, kb2));
        tp_[hf * 4 + r] = f2bf(p);
        d4_[r] = f2bf(p * dpt[r]);
        tds[hf * 4 + r] = d4_[r];
      }
      *(u16x4*)dsp = d4_;
    };
    u16x8 tpc, tdc;
    {
      Half h0, h1;
      issue_half(h0, 0); issue_half(h1, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      score_half(h0, 0, tpc, tdc, dSrow + fq * 8);
      score_half(h1, 1, tpc, tdc, dSrow + 32 + fq * 8);
    }
#pragma unroll 1
    for (int s2 = 0; s2 < KT / 2; ++s2) {
      const mfma16x8 pb = __builtin_bit_cast(mfma16x8, tpc), dsb = __builtin_bit_cast(mfma16x8, tdc);
      const bool more = s2 + 1 < KT / 2;
      const int qn = more ? 2 * s2 + 2 : 0;  // (last trip: harmless re-read of pair 0, results unused)
      Half ha, hb;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the dS store of the previous scores is out of the counter
      issue_half(ha, qn);                                                     // group A: 8 reads
      const int roff = 2 * s2 * ROW16 + row_off(4 * fq + tq) + tp * 8;
      const char* ob = dOs + roff; const char* qb_ = Qs + roff;
      uint2 olo[6], ohi[6];                                                   // group T1: 12 transposed reads (dO^T)
      static_for<0, 6>([&](auto dtc) { constexpr int dt = decltype(dtc)::value; olo[dt] = lds_tr16_b64_o<dt * 32>(ob); ohi[dt] = lds_tr16_b64_o<ROW16 + dt * 32>(ob); });
      u16x8 tpn, tdn;
      asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");                     // A landed (lgkmcnt is a 4-bit counter: groups are sized to waits <= 15)
      __builtin_amdgcn_sched_barrier(0);
      if (more) score_half(ha, 0, tpn, tdn, dSrow + (qn * 16 + fq * 4) * 2);
      __builtin_amdgcn_sched_barrier(0);
      uint2 qlo[6], qhi[6];                                                   // group T2: 12 transposed reads (xq^T)
      static_for<0, 6>([&](auto dtc) { constexpr int dt = decltype(dtc)::value; qlo[dt] = lds_tr16_b64_o<dt * 32>(qb_); qhi[dt] = lds_tr16_b64_o<ROW16 + dt * 32>(qb_); });
      asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");                     // T1 (and the dS store) landed
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) dva[dt] = MFMA16(__builtin_bit_cast(mfma16x8, make_uint4(olo[dt].x, olo[dt].y, ohi[dt].x, ohi[dt].y)), pb, dva[dt]);
      __builtin_amdgcn_sched_barrier(0);
      issue_half(hb, qn + 1);                                                 // group B: 8 reads
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                      // T2 landed
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) dka[dt] = MFMA16(__builtin_bit_cast(mfma16x8, make_uint4(qlo[dt].x, qlo[dt].y, qhi[dt].x, qhi[dt].y)), dsb, dka[dt]);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (more) score_half(hb, 1, tpn, tdn, dSrow + ((qn + 1) * 16 + fq * 4) * 2);
      tpc = tpn; tdc = tdn;
    }
  } else
#endif
#pragma unroll 1
  for (int s2 = 0; s2 < KT / 2; ++s2) {
    u16x8 tp_, tds;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int qt = 2 * s2 + hf;
      f32x4 st, dpt = *(const f32x4*)(L.ndrow + qt * 16 + fq * 4);                // -delta
      if constexpr (MASK) st = f32x4{0.f, 0.f, 0.f, 0.f};
      else st = *(const f32x4*)(L.mrow + qt * 16 + fq * 4);                      // -(m + l) / alpha
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const mfma16x8 qf = *(const mfma16x8*)(Qs + qt * ROW16 + row_off(fr) + (s * 32 + fq * 8) * 2);
        const mfma16x8 df = *(const mfma16x8*)(dOs + qt * ROW16 + row_off(fr) + (s * 32 + fq * 8) * 2);
        st = MFMA16(qf, kb[s], st);     // S[q = 16qt+4fq+r][key = k0+fr] (/ alpha)
        dpt = MFMA16(df, vb[s], dpt);   // dP[q][key] - delta[q]
      }
      u16x4 d4_;
      if constexpr (MASK) {
        const f32x4 m4 = *(const f32x4*)(L.mrow + qt * 16 + fq * 4);
        const f32x4 l4 = *(const f32x4*)(L.lrow + qt * 16 + fq * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __expf((st[r] * alpha + kbv - m4[r]) - l4[r]);
          tp_[hf * 4 + r] = f2bf(p);
          d4_[r] = f2bf(keep ? p * dpt[r] : 0.f);  // where() passes no gradient to masked logits
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(st[r], c2, kb2));
          tp_[hf * 4 + r] = f2bf(p);
          d4_[r] = f2bf(p * dpt[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) tds[hf * 4 + r] = d4_[r];
      *(u16x4*)(dSrow + (qt * 16 + fq * 4) * 2) = d4_;  // dS[key = k0+fr][q = 16qt+4fq .. +3] (without alpha)
    }
    const mfma16x8 pb = __builtin_bit_cast(mfma16x8, tp_), dsb = __builtin_bit_cast(mfma16x8, tds);
    // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^T[d][key] += Xq^T[d][q] dS[q][key]
    const int roff = 2 * s2 * ROW16 + row_off(4 * fq + tq) + tp * 8;
    const char* ob = dOs + roff; const char* qb_ = Qs + roff;
    if constexpr (SPLIT_TR) {
      uint2 lo[6], hi[6];
      static_for<0, 6>([&](auto dtc) { constexpr int dt = decltype(dtc)::value; lo[dt] = lds_tr16_b64_o<dt * 32>(ob); hi[dt] = lds_tr16_b64_o<ROW16 + dt * 32>(ob); });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) dva[dt] = MFMA16(__builtin_bit_cast(mfma16x8, make_uint4(lo[dt].x, lo[dt].y, hi[dt].x, hi[dt].y)), pb, dva[dt]);
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, 6>([&](auto dtc) { constexpr int dt = decltype(dtc)::value; lo[dt] = lds_tr16_b64_o<dt * 32>(qb_); hi[dt] = lds_tr16_b64_o<ROW16 + dt * 32>(qb_); });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) dka[dt] = MFMA16(__builtin_bit_cast(mfma16x8, make_uint4(lo[dt].x, lo[dt].y, hi[dt].x, hi[dt].y)), dsb, dka[dt]);
    } else {
      uint2 olo[6], ohi[6], qlo[6], qhi[6];
      static_for<0, 6>([&](auto dtc) {
        constexpr int dt = decltype(dtc)::value;
        olo[dt] = lds_tr16_b64_o<dt * 32>(ob); ohi[dt] = lds_tr16_b64_o<ROW16 + dt * 32>(ob);
        qlo[dt] = lds_tr16_b64_o<dt * 32>(qb_); qhi[dt] = lds_tr16_b64_o<ROW16 + dt * 32>(qb_);
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) {
        const uint4 uo = make_uint4(olo[dt].x, olo[dt].y, ohi[dt].x, ohi[dt].y);
        const uint4 uq = make_uint4(qlo[dt].x, qlo[dt].y, qhi[dt].x, qhi[dt].y);
        dva[dt] = MFMA16(__builtin_bit_cast(mfma16x8, uo), pb, dva[dt]);
        dka[dt] = MFMA16(__builtin_bit_cast(mfma16x8, uq), dsb, dka[dt]);
      }
    }
  }
  // lane: key = k0 + fr, d = 16dt + 4fq + r.  dV rows are packed first (their registers die), then dK^ = alpha g_q o dka and the RMSNorm backward of k
#pragma unroll
  for (int dt = 0; dt < 6; ++dt) {
    u16x4 v4;
#pragma unroll
    for (int r = 0; r < 4; ++r) v4[r] = f2bf(dva[dt][r]);
    if (gv_row) *(u16x4*)(gv_row + dt * 16) = v4;  // 8-byte row pieces straight from the accumulator layout
  }
#pragma unroll
  for (int dt = 0; dt < 6; ++dt) {
    const f32x4 a4 = *(const f32x4*)(L.agq + dt * 16 + fq * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) dka[dt][r] *= a4[r];
  }
  rms_bwd_tile(lane, dka, krow, rrk, L.gk, valid, ds_acc, [&](int dt, u16x4 k4) { if (gk_row) *(u16x4*)(gk_row + dt * 16) = k4; });
}

// phase 2: one 16-query tile.  dQ^[q][d] = alpha g_k[d] sum_keys xk[key][d] dS[q][key]; then the RMSNorm backward of q (lane: query fr, d = 16dt+4fq+r)
template <int KT>
__device__ __forceinline__ void bwd1_query_tile(int lane, const char* Ks, const char* dSs, const char* qrow /* x^q image row of this lane's query */, int qt,
                                                const Bwd1Lds& L, float rrq, bool valid, char* wt, bf16_t* gtile, int64_t ld_, int nrows,
                                                float (&ds_acc)[6][4]) {
  const int fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
  // B operand dS^T[key][q]: group fq's block = keys 16*tile + 4fq .. +3 (image rows), queries 16qt .. +15 (image columns); element j <-> key
  // 16(2 s2 + (j>>2)) + 4fq + (j&3), the k order of the Xk^T fragments below
  const char* dbase = dSs + (4 * fq + tq) * DSROW + qt * 32 + tp * 8;
  mfma16x8 dsb[KT / 2];
  {
    uint2 lo[KT / 2], hi[KT / 2];
    static_for<0, KT / 2>([&](auto sc_) {
      constexpr int s2 = decltype(sc_)::value;
      lo[s2] = lds_tr16_b64_o<2 * s2 * 16 * DSROW>(dbase); hi[s2] = lds_tr16_b64_o<(2 * s2 + 1) * 16 * DSROW>(dbase);
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < KT / 2; ++s2) dsb[s2] = __builtin_bit_cast(mfma16x8, make_uint4(lo[s2].x, lo[s2].y, hi[s2].x, hi[s2].y));
  }
  f32x4 dqa[6];
  const char* kbase = Ks + tp * 8 + row_off(4 * fq + tq);
  static_for<0, 6>([&](auto dtc) {
    constexpr int dt = decltype(dtc)::value;
    dqa[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint2 lo[KT / 2], hi[KT / 2];
    static_for<0, KT / 2>([&](auto sc_) {
      constexpr int s2 = decltype(sc_)::value;
      lo[s2] = lds_tr16_b64_o<dt * 32 + 2 * s2 * ROW16>(kbase); hi[s2] = lds_tr16_b64_o<dt * 32 + (2 * s2 + 1) * ROW16>(kbase);
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < KT / 2; ++s2) {
      const uint4 u = make_uint4(lo[s2].x, lo[s2].y, hi[s2].x, hi[s2].y);
      dqa[dt] = MFMA16(__builtin_bit_cast(mfma16x8, u), dsb[s2], dqa[dt]);
    }
  });
#pragma unroll
  for (int dt = 0; dt < 6; ++dt) {
    const f32x4 a4 = *(const f32x4*)(L.agk + dt * 16 + fq * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) dqa[dt][r] *= a4[r];
  }
  if (wt) {  // wave-uniform
    rms_bwd_tile(lane, dqa, qrow, rrq, L.gq, valid, ds_acc, [&](int dt, u16x4 o4) { tile_put(wt, dt, o4, lane); });
    tile_flush(wt, gtile, ld_, nrows, lane);
  } else {
    rms_bwd_tile(lane, dqa, qrow, rrq, L.gq, valid, ds_acc, [&](int dt, u16x4 o4) {
      if (fr < nrows) *(u16x4*)(gtile + (int64_t)fr * ld_ + dt * 16 + fq * 4) = o4;
    });
  }
}

// NW waves per workgroup: 8 (two per SIMD, 256 registers) or 12 (three per SIMD, 168 registers: every key / query tile of S <= 160 in ONE round, and a
// third dependency chain per SIMD to hide the LDS / MFMA latencies the phases are bound by)
template <int KT, bool MASK, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void attn_bwd1_kernel(AttnBwdArgs g) {
  constexpr int NTH = NW * 64, RPP = NW * 16;
  constexpr int S_pad = KT * 16, IMG = img_bytes(S_pad);
  static_assert(S_pad * 2 + 32 <= DSROW, "dS image row");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem; char* Ks = Qs + IMG; char* dOs = Ks + IMG; char* dSs = dOs + bwd1_tile_bytes(KT);
  Bwd1Lds L;
  L.kbias = (float*)(dSs + S_pad * DSROW); L.mrow = L.kbias + S_pad; L.lrow = L.mrow + S_pad; L.ndrow = L.lrow + S_pad; L.rq = L.ndrow + S_pad;
  L.rk = L.rq + S_pad; L.sred = L.rk + S_pad; L.gq = L.sred + 2 * DH; L.gk = L.gq + DH; L.gqk = L.gk + DH; L.agq = L.gqk + DH; L.agk = L.agq + DH;
  const int E = g.H * DH;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, fr = lane & 15, fq = lane >> 4;
  
  for (int t = tid; t < DH; t += NTH) {
    const float a = g.sq[t], b = g.sk[t];
    L.gq[t] = a; L.gk[t] = b; L.gqk[t] = a * b; L.agq[t] = 0.10206207261596575f * a; L.agk[t] = 0.10206207261596575f * b;
    L.sred[t] = 0.f; L.sred[DH + t] = 0.f;
  }
  __syncthreads();
  const int64_t nseq = g.nprob / g.H;
  constexpr int NP = (S_pad + RPP - 1) / RPP;
  constexpr int abl = SPA3D_ABL1;  // diagnostic builds only (tools/ablate_attn.py compiles one object per mask); 0 in the product: everything below folds away
  // Software pipeline over this workgroup's problems: the NEXT problem's q, k, dO, O rows are requested at the start of phase 2 and land in
  // registers under it (phase 2 needs few), so the top of the loop finds them there instead of paying a memory latency per problem.
  RawRows<NP> rq, rk, rd, ro;
  int64_t row0 = 0; int S = 0, h = 0;
  auto params = [&](int64_t pi_, int64_t& row0_, int& S_, int& h_) {
    const int64_t prob = map_prob(pi_, nseq, g.H);
    const int64_t seq = prob / g.H; h_ = (int)(prob - seq * g.H);
    row0_ = g.seq_off ? (int64_t)g.seq_off[seq] : seq * g.S;
    S_ = g.seq_off ? g.seq_off[seq + 1] - (int)row0_ : g.S;
  };
  // q, k rows (48 registers) are the prefetched half: with all four matrices in flight under phase 2 the compiler spills 36 of the 96 row
  // registers right behind the loads, i.e. waits for them at once; dO, O are requested at the top of the loop and land under the q / k staging
  auto request_qk = [&](int64_t row0_, int S_, int h_) {
    if constexpr (!(abl & 2)) {
      const int t_ = opaque_tid();
      rows_load<NP, RPP>(rq, g.q + row0_ * g.ldq + h_ * DH, g.ldq, S_, t_);
      rows_load<NP, RPP>(rk, g.k + row0_ * g.ldk + h_ * DH, g.ldk, S_, t_);
    }
  };
  constexpr bool PREFETCH = SPA3D_B1_PREFETCH != 0;
  int64_t pi = blockIdx.x;
  if (pi < g.nprob) { params(pi, row0, S, h); if (PREFETCH) request_qk(row0, S, h); }
  while (pi < g.nprob) {
    const int QT = (S + 15) / 16;
    const int tid_o = opaque_tid();
    if (!PREFETCH) request_qk(row0, S, h);
    if constexpr (!(abl & 2)) {
      rows_load<NP, RPP>(rd, g.d_o + row0 * E + h * DH, E, S, tid_o);
      rows_load<NP, RPP>(ro, g.o + row0 * E + h * DH, E, S, tid_o);
    }
    __syncthreads();  // previous problem's phase-2 reads (Xq, Xk, dS) and tile traffic (dO region) are done
    if constexpr (!(abl & 2)) {
      rows_store_xhat<NP, RPP>(rq, S_pad, Qs, L.rq, tid_o);
      rows_store_xhat<NP, RPP>(rk, S_pad, Ks, L.rk, tid_o);
    }
    for (int t = tid_o; t < S_pad; t += NTH) {
      float b = 0.f, m = 0.f, ll = __builtin_inff();  // padding query: P = exp(.. - inf) = 0
      if (t >= S) b = -__builtin_inff();
      else if (MASK && g.km[row0 + t] == 0.f) b = NEG_BIG;
      if (t < S) { m = g.lse[(row0 * g.H + (int64_t)h * S + t) * 2]; ll = g.lse[(row0 * g.H + (int64_t)h * S + t) * 2 + 1]; }
      L.kbias[t] = b;
      if constexpr (MASK) { L.mrow[t] = m; L.lrow[t] = ll; }
      else L.mrow[t] = -(m + ll) * 9.797958971132712f;  // -(m + l) / alpha: the S accumulators' initial value (padding query: -inf -> P = 0)
    }
    // this wave's own key rows, k and v, as B-operand fragments straight from global memory (lane: key fr, d = 32s + 8fq + j)
    u16x8 kx[3], vx[3];
    auto load_own = [&](int kt) {
      int krow = kt * 16 + fr; if (krow > S - 1) krow = S - 1;
      const bf16_t* kp = g.k + (row0 + krow) * g.ldk + h * DH + fq * 8;
      const bf16_t* vp = g.v + (row0 + krow) * g.ldv + h * DH + fq * 8;
#pragma unroll
      for (int s = 0; s < 3; ++s) { kx[s] = *(const u16x8*)(kp + s * 32); vx[s] = *(const u16x8*)(vp + s * 32); }
    };
    if (wv < QT) load_own(wv);  // (L2-warm: the staging loads just fetched these rows)
    if constexpr (!(abl & 2)) store_do_delta<NP, RPP, true>(rd, ro, S_pad, dOs, L.ndrow, tid_o);  // -delta: the dP accumulators' initial value
    __syncthreads();
    const int64_t pn = pi + gridDim.x;
    int64_t row0n = 0; int Sn = 0, hn = 0;
    // ---------------------------------------------------------------- phase 1: key tiles -> dk, dv, dS image
    const int QT1 = (abl & 16) ? min(QT, NW) : QT;          // 16: first round of tiles only
    if constexpr (!(abl & 1)) {                            // 1: staging only
      float dsk_acc[6][4];
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) dsk_acc[i][r] = 0.f;
      for (int kt = wv; kt < QT1; kt += (NW >= KT ? KT : NW)) {  // (NW >= KT: at most one trip, and the compiler can see it)
        const int k0 = kt * 16;
        const int lo_ = opaque_tid() & 63, fr = lo_ & 15, fq = lo_ >> 4;  // shadows the kernel-scope fr / fq: recomputed per tile, not hoisted
        mfma16x8 kb[3], vb[3];
        frag_norm(kx, L.gqk, fq, kb);  // 16-bit(x r g_k g_q): with xq in the image, S = alpha sum_d xq kb
#pragma unroll
        for (int s = 0; s < 3; ++s) vb[s] = __builtin_bit_cast(mfma16x8, vx[s]);
        if (kt + NW < QT) load_own(kt + NW);  // the second tile's rows arrive under the first tile's MFMAs
        const bool st_ = !(abl & 128) && k0 + fr < S;     // 128: no dk / dv stores
        bwd1_key_tile<KT, MASK, (NW > 8)>(lo_, Qs, dOs, Ks + row_off(k0 + fr), dSs + (k0 + fr) * DSROW, L, kb, vb, L.kbias[k0 + fr], L.rk[k0 + fr], k0 + fr < S,
                                          st_ ? g.dk + (row0 + k0 + fr) * g.ldk + h * DH + fq * 4 : nullptr,
                                          st_ ? g.dv + (row0 + k0 + fr) * g.ldv + h * DH + fq * 4 : nullptr, dsk_acc);
      }
      if constexpr (!(abl & 256)) flush_ds_acc(dsk_acc, L.sred + DH);  // 256: no scale-gradient flushes
      // key tiles past the sequence end (ragged sequences): their dS rows must read as zeros in phase 2 (xk rows there are zero, but 0 x NaN = NaN)
      for (int kt = QT + wv; kt < KT; kt += NW) {
        char* rowp = dSs + (kt * 16 + fr) * DSROW + fq * 8;
#pragma unroll
        for (int qt = 0; qt < KT; ++qt) *(u16x4*)(rowp + qt * 32) = u16x4{0, 0, 0, 0};
      }
    }
    __syncthreads();
    // ---------------------------------------------------------------- phase 2: query tiles -> dq (and the held dk / dv rows)
    // next problem's rows: in flight under phase 2.  Unconditional (the last iteration re-requests its own rows, never used): a conditional
    // request makes the row registers loop-carried through phase 1 in the compiler's eyes (309 spilled VGPRs)
    params(pn < g.nprob ? pn : pi, row0n, Sn, hn); if (PREFETCH) request_qk(row0n, Sn, hn);
    if constexpr (!(abl & 1) && !(abl & 32)) {             // 32: no phase 2
      float dsq_acc[6][4];
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) dsq_acc[i][r] = 0.f;
      for (int qt = wv; qt < QT1; qt += (NW >= KT ? KT : NW)) {
        const int q0 = qt * 16;
        const int lo_ = opaque_tid() & 63, fr = lo_ & 15;
        const int wv_ = opaque_tid() >> 6;
        char* wt_ = wv_ < 8 ? dOs + wv_ * WTILE : nullptr;  // eight tiles fit the dead dO image; further waves store 8-byte pieces directly
        bwd1_query_tile<KT>(lo_, Ks, dSs, Qs + row_off(q0 + fr), qt, L, L.rq[q0 + fr], q0 + fr < S, wt_, g.dq + (row0 + q0) * g.ldq + h * DH, g.ldq,
                            S - q0, dsq_acc);
      }
      if constexpr (!(abl & 256)) flush_ds_acc(dsq_acc, L.sred);
    }
    pi = pn; row0 = row0n; S = Sn; h = hn;
  }
  __syncthreads();
  if (tid < DH) atomicAdd(g.dsq + tid, L.sred[tid]);
  else if (tid < 2 * DH) atomicAdd(g.dsk + tid - DH, L.sred[tid]);
}

// split-pass form, two images: see the section header.  NW waves per workgroup (4: two workgroups per CU at S <= 160; 8: one).
