// tools/experiments/gemm_nt_rejected.hip -- NOT part of libspa3d_hip.so (not compiled by 3dspa_code_amd/build.py).
//
// Six NT-GEMM kernel structures that were built, race-screened and MEASURED in round 1 and lost to the 8-phase kernels that the
// product ships (3dspa_code_amd/csrc/gemm_fast.hip).  Kept as a record of what was tried, with the numbers in each header, so the
// experiments are not repeated; they compiled against round 1's NtArgs / helper set (gemm_fast.hip up to commit 4d17776) and would
// need the staging helpers of that file to build again.  M = 2.47 M, K = 384, N = 2304 unless a header says otherwise, where the
// 128x128 single-buffer kernel does 667 TF/s and the shipped 8-phase kernels 690-790:
//   gemm_nt_astat_kernel   A panel resident, weights through a 4-slot ring, one wave per SIMD            452 TF/s
//   gemm_nt_persist_kernel persistent 128x128 with cross-tile prefetch (2 WG/CU)                          538 vs 594
//   gemm_nt256_kernel      plain 256x256 8-wave kernel, 2 barriers per K-tile                             +3-4 % over 128x128 at K >= 1280 only
//   gemm_nt_astat2_kernel  two 4-wave teams sharing a resident A panel                                    638 TF/s
//   gemm_nt_ring_kernel    persistent 256x128, 3-slot ring                                                500-740 TF/s
//   gemm_nt_pp_kernel      the same with ping-pong teams one barrier apart                                500-740 TF/s
// =================================================================================================================
// A-stationary NT GEMM for short K (K <= 512) and wide N: the shapes of the QKV / MLP-in projections and of the dX of
// the out / MLP-out projections (K = 384), where a 128x128 tile re-stages its whole A panel for every one of the N/128
// column tiles and the kernel is bound by LDS-DMA delivery (64 FLOP per staged byte), not by MFMA issue.
// One workgroup owns a 128-row M tile: the [128 x K] A panel is staged ONCE (K/64 swizzled 16-KiB images, 96 KiB at
// K = 384) and stays resident; the weight tiles [128 x 64] of all N tiles stream through a 4-slot LDS ring, three
// tiles in flight behind a COUNTED `s_waitcnt vmcnt` and a raw `s_barrier` per K-step (cdna_hip_programming.md
// "Pipelining across barriers"), so the stream never drains between column tiles: 128 FLOP per staged byte, one
// prologue per M tile, A read from HBM exactly once, weights always from L2.
// =================================================================================================================
template <int KT>
__global__ __launch_bounds__(256, 1) void gemm_nt_astat_kernel(NtArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [KT][A image 16 KiB] | [NB][B tile 16 KiB]
  constexpr int NB = 4;
  char* const sA = smem; char* const sB = smem + KT * 16384;
  const int tm = blockIdx.x;
  const int64_t m0 = (int64_t)tm * 128;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  const int fr = lane & 15, fq = lane >> 4;
  const int sr = lane >> 3, scp = lane & 7;
  const int sc = (scp ^ sr) * 8;
  const int nsteps = g.tiles_n * KT;

  // per-lane source rows: wave w fills 8-row groups 4w..4w+3 of every staged tile
  const bf16_t* ga[4]; int brow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (w * 4 + i) * 8 + sr;
    int64_t am = m0 + row; if (am > g.M - 1) am = g.M - 1;
    ga[i] = g.A + am * g.lda + sc;
    brow[i] = row;
  }
  auto stage_b = [&](int step) {  // weight tile of K-step `step` -> ring slot step % NB
    const int nt = step / KT, kt = step - nt * KT;
    char* sb = sB + (step % NB) * 16384 + (w * 4) * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int bn = nt * 128 + brow[i]; if (bn > g.N - 1) bn = g.N - 1;
      GLDS16(g.Bt + (int64_t)bn * g.ldb + sc + kt * 64, sb + i * 1024);
    }
  };
  // ---- prologue: A panel, then the first NB-1 weight tiles
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    char* sa = sA + kt * 16384 + (w * 4) * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) GLDS16(ga[i] + kt * 64, sa + i * 1024);
  }
#pragma unroll
  for (int p = 0; p < NB - 1; ++p) if (p < nsteps) stage_b(p);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int a_off = (wm + fr) * 128, b_off = (wn + fr) * 128;
  const int x0 = ((fq) ^ (fr & 7)) * 16, x1 = ((4 + fq) ^ (fr & 7)) * 16;

  // vmcnt counts loads, LDS-DMA and STORES in issue order.  When the epilogue is pure stores (no bias / residual /
  // accumulate loads) and the tile is interior, exactly 16 (32 with a pre-activation copy) store instructions per wave sit
  // between the weight tiles of two column tiles; counting them lets the stream run on while they drain.
  const bool pure = !g.bias && !g.aux && !g.accumulate && (m0 + 128 <= g.M) && (g.N % 128 == 0);
  const int nstore = g.pre_out ? 32 : 16;
  int kt = 0, nt = 0;
  for (int step = 0; step < nsteps; ++step) {
    // tile `step` (and the A panel before it) has landed once at most the 2 younger tiles (4 LDS-DMA each) -- plus, for the
    // three steps after an epilogue, its stores -- are outstanding; at the tail fewer tiles are in flight: wait for all.
    if (step + 2 >= nsteps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (pure && nt > 0 && kt < 3) { if (nstore == 16) asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // slot (step+3) % 4 == (step-1) % 4 was last read in step-1, which every wave has left: safe to refill now
    if (step + NB - 1 < nsteps) stage_b(step + NB - 1);
    const char* sa = sA + kt * 16384;
    const char* sb = sB + (step % NB) * 16384;
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
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (++kt == KT) {
      // ---- epilogue of column tile nt, straight from registers: lane holds C[m = wm+16i+fr][n = wn+16j+4fq .. +3]
      const int n0 = nt * 128;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t gm = m0 + wm + i * 16 + fr;
        int64_t crow = gm;
        if (g.crow_group > 0) crow = gm + (gm / g.crow_group + 1) * (int64_t)g.crow_skip;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int gn = n0 + wn + j * 16 + fq * 4;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = g.alpha * acc[i][j][r]; acc[i][j][r] = 0.f; }
          if (gm >= g.M || gn >= g.N) continue;
          if (g.bias) { const float4 b4 = *(const float4*)(g.bias + gn); v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w; }
          const int64_t ci = crow * g.ldc + gn;
          if (g.pre_out) { u16x4 p4;
#pragma unroll
            for (int r = 0; r < 4; ++r) p4[r] = f2bf(v[r]);
            *(u16x4*)(g.pre_out + ci) = p4; }
          if (g.epi == EPI_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_fast_f(v[r]);
          }
          if (g.aux) {
            const u16x4 x4 = *(const u16x4*)(g.aux + ci);
            if (g.epi == EPI_MUL_GELU_GRAD) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] *= gelu_tanh_grad_fast_f(bf2f(x4[r]));
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += bf2f(x4[r]);
            }
          }
          if (g.out_f32) {
            float4* cp = (float4*)((float*)g.C + ci);
            if (g.accumulate) { const float4 o = *cp; v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
            *cp = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            u16x4* cp = (u16x4*)((bf16_t*)g.C + ci);
            if (g.accumulate) { const u16x4 o = *cp;
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += bf2f(o[r]); }
            u16x4 o4;
#pragma unroll
            for (int r = 0; r < 4; ++r) o4[r] = f2bf(v[r]);
            *cp = o4;
          }
        }
      }
      kt = 0; ++nt;
    }
  }
}

template <int KT>
static void launch_astat(spa3d_ctx* c, const NtArgs& g) {
  const int lds = (KT + 4) * 16384;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_nt_astat_kernel<KT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
  gemm_nt_astat_kernel<KT><<<(unsigned)g.tiles_m, 256, lds, c->stream>>>(g);
}


// =================================================================================================================
// Persistent variant of gemm_nt_kernel.  Fitting T(K) of the one-tile-per-workgroup kernel on the step's shapes gives
// ~1.0 us per K-tile but ~5 us of FIXED cost per 128x128 output tile (workgroup launch, first-tile DMA latency, store
// tail) -- as much as the whole K loop at K = 384.  Here 2 workgroups per CU stay resident and walk the tile list;
// the first K-tile of the NEXT output tile is staged into the free LDS buffer under the last K-tile of the current
// one, and the epilogue goes straight from registers (no LDS, no barrier) so it overlaps that DMA.
// =================================================================================================================
__global__ __launch_bounds__(256, 2) void gemm_nt_persist_kernel(NtArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A 16 KiB | B 16 KiB]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  const int fr = lane & 15, fq = lane >> 4;
  const int sr = lane >> 3, scp = lane & 7;
  const int sc = (scp ^ sr) * 8;
  const int nvirt = ((g.tiles_m + 7) / 8) * 8 * g.tiles_n;  // virtual ids; gridDim.x % 8 == 0 keeps a workgroup on "its" XCD lane
  const int nt = g.K / 64;
  const int a_off = (wm + fr) * 128, b_off = (wn + fr) * 128;
  const int x0 = ((fq) ^ (fr & 7)) * 16, x1 = ((4 + fq) ^ (fr & 7)) * 16;

  auto decode = [&](int v, int& tm, int& tn) { const int xcd = v & 7, idx = v >> 3; tm = (idx / g.tiles_n) * 8 + xcd; tn = idx % g.tiles_n; };
  auto next_valid = [&](int v) { int tm, tn; for (; v < nvirt; v += gridDim.x) { decode(v, tm, tn); if (tm < g.tiles_m) return v; } return nvirt; };
  const bf16_t* ga[4]; const bf16_t* gb[4];
  auto setup = [&](int v, const bf16_t* (&pa)[4], const bf16_t* (&pb)[4]) {
    int tm, tn; decode(v, tm, tn);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (w * 4 + i) * 8 + sr;
      int64_t am = (int64_t)tm * 128 + row; if (am > g.M - 1) am = g.M - 1;
      int bn = tn * 128 + row; if (bn > g.N - 1) bn = g.N - 1;
      pa[i] = g.A + am * g.lda + sc; pb[i] = g.Bt + (int64_t)bn * g.ldb + sc;
    }
  };
  auto stage = [&](int buf, int kt, const bf16_t* const (&pa)[4], const bf16_t* const (&pb)[4]) {
    char* sa = smem + buf * 32768 + (w * 4) * 1024;
    char* sb = sa + 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) GLDS16(pa[i] + kt * 64, sa + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) GLDS16(pb[i] + kt * 64, sb + i * 1024);
  };

  int v = next_valid(blockIdx.x);
  if (v >= nvirt) return;
  setup(v, ga, gb);
  stage(0, 0, ga, gb);
  __syncthreads();
  int cur = 0;
  while (true) {
    int tm, tn; decode(v, tm, tn);
    const int vn = next_valid(v + gridDim.x);
    const bool has_next = vn < nvirt;
    const bf16_t* gan[4]; const bf16_t* gbn[4];
    if (has_next) setup(vn, gan, gbn);
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < nt; ++t) {
      if (t + 1 < nt) stage(cur ^ 1, t + 1, ga, gb);
      else if (has_next) stage(cur ^ 1, 0, gan, gbn);  // next output tile's first K-tile, under this tile's last MFMAs + epilogue
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
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
      if (t + 1 < nt) { __syncthreads(); cur ^= 1; }
    }
    // ---- epilogue straight from registers: lane holds C[m = wm+16i+fr][n = wn+16j+4fq .. +3]
    const int64_t m0 = (int64_t)tm * 128; const int n0 = tn * 128;
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
    if (!has_next) break;
    __syncthreads();  // next tile's first K-tile has landed (vmcnt(0)) and every wave is done reading `cur`
    cur ^= 1;
    v = vn;
#pragma unroll
    for (int i = 0; i < 4; ++i) { ga[i] = gan[i]; gb[i] = gbn[i]; }
  }
}


// =================================================================================================================
// 256x256x64 tile, 8 waves (2 x 4, 128x64 per wave, 32 accumulator tiles = 128 VGPRs), two 64-KiB LDS buffers.
// Why: with LDS-DMA staging the 128x128 kernels are bound by (bytes in flight per CU) / (DMA latency) x (FLOP per
// staged byte): 64 KiB / ~1.5 us x 64 FLOP/B ~ 700 TF/s, which is what they measure.  This tile keeps the same 64 KiB
// in flight but does 128 FLOP per staged byte, and its 64 MFMAs per wave per K-tile (x 2 waves per SIMD ~ 1 us) cover one
// DMA latency.  Same swizzle, same swapped-operand MFMA, direct register epilogue (8-byte stores).
// =================================================================================================================
__global__ __launch_bounds__(512, 2) void gemm_nt256_kernel(NtArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A 32 KiB | B 32 KiB]
  const int bid = blockIdx.x;
  const int xcd = bid & 7, idx = bid >> 3;
  const int tm = (idx / g.tiles_n) * 8 + xcd, tn = idx % g.tiles_n;  // tiles_* count 256-wide tiles here
  if (tm >= g.tiles_m) return;
  const int64_t m0 = (int64_t)tm * 256;
  const int n0 = tn * 256;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = (w >> 2) * 128, wn = (w & 3) * 64;
  const int fr = lane & 15, fq = lane >> 4;
  const int sr = lane >> 3, scp = lane & 7;
  const int sc = (scp ^ sr) * 8;
  const bf16_t* ga[4]; const bf16_t* gb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (w * 4 + i) * 8 + sr;  // 8 waves x 4 groups x 8 rows = 256 rows
    int64_t am = m0 + row; if (am > g.M - 1) am = g.M - 1;
    int bn = n0 + row; if (bn > g.N - 1) bn = g.N - 1;
    ga[i] = g.A + am * g.lda + sc;
    gb[i] = g.Bt + (int64_t)bn * g.ldb + sc;
  }
  auto stage = [&](int buf, int kt) {
    char* sa = smem + buf * 65536 + (w * 4) * 1024;
    char* sb = sa + 32768;
#pragma unroll
    for (int i = 0; i < 4; ++i) GLDS16(ga[i] + kt * 64, sa + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) GLDS16(gb[i] + kt * 64, sb + i * 1024);
  };
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nt = g.K / 64;
  stage(0, 0);
  __syncthreads();
  const int a_off = (wm + fr) * 128, b_off = (wn + fr) * 128;
  const int x0 = ((fq) ^ (fr & 7)) * 16, x1 = ((4 + fq) ^ (fr & 7)) * 16;
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) stage(cur ^ 1, t + 1);
    const char* sa = smem + cur * 65536;
    const char* sb = sa + 32768;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int xo = ks ? x1 : x0;
      bf16x8 bfr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(sb + b_off + j * 2048 + xo);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bf16x8 af = *(const bf16x8*)(sa + a_off + i * 2048 + xo);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af, acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
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
// A-stationary NT GEMM, two teams.  The per-CU LDS-DMA path, not MFMA issue, bounds the 128x128 kernels (64 FLOP per
// staged byte); the first A-stationary attempt halved the staged bytes but ran one wave per SIMD and lost.  Here a
// 512-thread workgroup owns a 128-row M tile whose [128 x K] A panel is staged once (K <= 384: 96 KiB) and two teams of
// 4 waves walk the column tiles (team t takes tiles t, t+2, ..), each streaming its own weight tiles through a private
// 2-slot ring: 128 FLOP per staged byte AND two waves per SIMD, so one team's MFMAs cover the other's DMA issue / LDS
// reads.  One s_barrier per K-step for both teams; epilogue straight from registers.
// =================================================================================================================
template <int KT>
__global__ __launch_bounds__(512, 2) void gemm_nt_astat2_kernel(NtArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [KT][A image 16 KiB] | team0 [2][16 KiB] | team1 [2][16 KiB]
  char* const sA = smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = w >> 2, tw = w & 3;
  char* const sB = smem + KT * 16384 + team * 32768;
  const int64_t m0 = (int64_t)blockIdx.x * 128;
  const int wm = (tw >> 1) * 64, wn = (tw & 1) * 64;
  const int fr = lane & 15, fq = lane >> 4;
  const int sr = lane >> 3, scp = lane & 7;
  const int sc = (scp ^ sr) * 8;
  const int my_tiles = (g.tiles_n - team + 1) / 2;           // column tiles of this team
  const int max_tiles = (g.tiles_n + 1) / 2;                 // team 0's count: the loop length for both (barriers must match)
  const int nsteps = max_tiles * KT, my_steps = my_tiles * KT;

  // A panel: 16 pieces of 1 KiB per image, 2 per wave
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int grp = w * 2 + i;
      int64_t am = m0 + grp * 8 + sr; if (am > g.M - 1) am = g.M - 1;
      GLDS16(g.A + am * g.lda + sc + kt * 64, sA + kt * 16384 + grp * 1024);
    }
  auto stage_b = [&](int step) {  // this team's weight tile of its K-step `step` -> slot step & 1
    const int ti = step / KT, kt = step - ti * KT;
    const int nt = team + 2 * ti;
    char* sb = sB + (step & 1) * 16384 + (tw * 4) * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int bn = nt * 128 + (tw * 4 + i) * 8 + sr; if (bn > g.N - 1) bn = g.N - 1;
      GLDS16(g.Bt + (int64_t)bn * g.ldb + sc + kt * 64, sb + i * 1024);
    }
  };
  if (my_steps > 0) stage_b(0);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int a_off = (wm + fr) * 128, b_off = (wn + fr) * 128;
  const int x0 = ((fq) ^ (fr & 7)) * 16, x1 = ((4 + fq) ^ (fr & 7)) * 16;

  int kt = 0, ti = 0;
  for (int step = 0; step < nsteps; ++step) {
    __syncthreads();  // tile `step` (and, first time, the A panel) landed; every wave is done with the slot refilled next
    if (step + 1 < my_steps) stage_b(step + 1);
    if (step < my_steps) {
      const char* sa = sA + kt * 16384;
      const char* sb = sB + (step & 1) * 16384;
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
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
    }
    if (++kt == KT) {
      if (step < my_steps) {
        const int n0 = (team + 2 * ti) * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t gm = m0 + wm + i * 16 + fr;
          int64_t crow = gm;
          if (g.crow_group > 0) crow = gm + (gm / g.crow_group + 1) * (int64_t)g.crow_skip;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int gn = n0 + wn + j * 16 + fq * 4;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = g.alpha * acc[i][j][r]; acc[i][j][r] = 0.f; }
            if (gm >= g.M || gn >= g.N) continue;
            if (g.bias) { const float4 b4 = *(const float4*)(g.bias + gn); v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w; }
            const int64_t ci = crow * g.ldc + gn;
            if (g.pre_out) { u16x4 p4;
#pragma unroll
              for (int r = 0; r < 4; ++r) p4[r] = f2bf(v[r]);
              *(u16x4*)(g.pre_out + ci) = p4; }
            if (g.epi == EPI_GELU) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_fast_f(v[r]);
            }
            if (g.aux) {
              const u16x4 x4 = *(const u16x4*)(g.aux + ci);
              if (g.epi == EPI_MUL_GELU_GRAD) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= gelu_tanh_grad_fast_f(bf2f(x4[r]));
              } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += bf2f(x4[r]);
              }
            }
            if (g.out_f32) {
              float4* cp = (float4*)((float*)g.C + ci);
              if (g.accumulate) { const float4 o = *cp; v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
              *cp = make_float4(v[0], v[1], v[2], v[3]);
            } else {
              u16x4* cp = (u16x4*)((bf16_t*)g.C + ci);
              if (g.accumulate) { const u16x4 o = *cp;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += bf2f(o[r]); }
              u16x4 o4;
#pragma unroll
              for (int r = 0; r < 4; ++r) o4[r] = f2bf(v[r]);
              *cp = o4;
            }
          }
        }
      }
      kt = 0; ++ti;
    }
  }
}

template <int KT>
static void launch_astat2(spa3d_ctx* c, const NtArgs& g) {
  const int lds = KT * 16384 + 65536;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)gemm_nt_astat2_kernel<KT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
  gemm_nt_astat2_kernel<KT><<<(unsigned)g.tiles_m, 512, lds, c->stream>>>(g);
}


// =================================================================================================================
// Ring kernel: persistent, 256x128x64 tile, 8 waves (4 x 2, 64x64 per wave), 3-slot LDS ring of (A 32 KiB | B 16 KiB).
// Measurements on the simpler kernels say the NT GEMMs are LDS-DMA LATENCY bound (spreading the DMA issues later in a
// K-step made them slower; deeper tiles or fewer bytes alone did not help): throughput ~ bytes in flight per CU / latency.
// This kernel keeps TWO K-tiles (96 KiB) in flight per CU behind a counted `s_waitcnt vmcnt` + one raw `s_barrier` per
// K-step, and the step sequence is flattened over all output tiles a workgroup owns (an M tile, then every N tile of
// it), so the ring never drains at a tile boundary: the epilogue (registers -> global, bias from LDS) runs with the next
// tile's operands already streaming.  One workgroup per CU, two waves per SIMD.
// =================================================================================================================
__global__ __launch_bounds__(512, 2) void gemm_nt_ring_kernel(NtArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [3][A 32 KiB | B 16 KiB] | bias f32[N<=4096]
  constexpr int SLOT = 49152;
  float* const sbias = (float*)(smem + 3 * SLOT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  const int fr = lane & 15, fq = lane >> 4;
  const int sr = lane >> 3, scp = lane & 7;
  const int sc = (scp ^ sr) * 8;
  const int KT = g.K / 64;
  const int G = gridDim.x;
  const int my_m = (g.tiles_m - (int)blockIdx.x + G - 1) / G;  // tiles_m counts 256-row tiles here
  const int nsteps = my_m * g.tiles_n * KT;
  if (g.bias) for (int i = tid; i < g.N; i += 512) sbias[i] = g.bias[i];

  // staging cursor (runs 2 steps ahead of the compute cursor)
  int s_j = 0, s_tn = 0, s_kt = 0;
  auto stage = [&](int slot) {
    const int64_t tm = (int64_t)blockIdx.x + (int64_t)s_j * G;
    char* sa = smem + slot * SLOT; char* sb = sa + 32768;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = w * 4 + i;
      int64_t am = tm * 256 + piece * 8 + sr; if (am > g.M - 1) am = g.M - 1;
      GLDS16(g.A + am * g.lda + sc + s_kt * 64, sa + piece * 1024);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = w * 2 + i;
      int bn = s_tn * 128 + piece * 8 + sr; if (bn > g.N - 1) bn = g.N - 1;
      GLDS16(g.Bt + (int64_t)bn * g.ldb + sc + s_kt * 64, sb + piece * 1024);
    }
    if (++s_kt == KT) { s_kt = 0; if (++s_tn == g.tiles_n) { s_tn = 0; ++s_j; } }
  };
  if (nsteps > 0) stage(0);
  if (nsteps > 1) stage(1);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int a_off = (wm + fr) * 128, b_off = (wn + fr) * 128;
  const int x0 = ((fq) ^ (fr & 7)) * 16, x1 = ((4 + fq) ^ (fr & 7)) * 16;
  // stores of an epilogue sit between the weight tiles in vmcnt's in-order count; when their number is known exactly
  // (interior tile, no load in the epilogue) it is added to the allowance so the ring does not wait for them
  const bool count_stores = !g.aux && !g.accumulate && (g.N % 128 == 0);
  const int nstore = g.pre_out ? 32 : 16;

  int c_j = 0, c_tn = 0, c_kt = 0, since_epi = 1000;
  for (int step = 0; step < nsteps; ++step) {
    // tile `step` has landed once only the younger tile step+1 (6 LDS-DMA per wave) [+ counted stores] is outstanding
    if (step + 1 >= nsteps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (since_epi < 2 && nstore == 16) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
    else if (since_epi < 2 && nstore == 32) asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (step + 2 < nsteps) stage((step + 2) % 3);  // slot of step-1: every wave has left it
    ++since_epi;
    const char* sa = smem + (step % 3) * SLOT;
    const char* sb = sa + 32768;
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
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (++c_kt == KT) {
      const int64_t m0 = ((int64_t)blockIdx.x + (int64_t)c_j * G) * 256;
      const int n0 = c_tn * 128;
      const bool interior = m0 + 256 <= g.M;
      since_epi = (count_stores && interior) ? 0 : 1000;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t gm = m0 + wm + i * 16 + fr;
        int64_t crow = gm;
        if (g.crow_group > 0) crow = gm + (gm / g.crow_group + 1) * (int64_t)g.crow_skip;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int gn = n0 + wn + j * 16 + fq * 4;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = g.alpha * acc[i][j][r]; acc[i][j][r] = 0.f; }
          if (gm >= g.M || gn >= g.N) continue;
          if (g.bias) { const f32x4 b4 = *(const f32x4*)(sbias + gn); v[0] += b4[0]; v[1] += b4[1]; v[2] += b4[2]; v[3] += b4[3]; }
          const int64_t ci = crow * g.ldc + gn;
          if (g.pre_out) { u16x4 p4;
#pragma unroll
            for (int r = 0; r < 4; ++r) p4[r] = f2bf(v[r]);
            *(u16x4*)(g.pre_out + ci) = p4; }
          if (g.epi == EPI_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_fast_f(v[r]);
          }
          if (g.aux) {
            const u16x4 x4 = *(const u16x4*)(g.aux + ci);
            if (g.epi == EPI_MUL_GELU_GRAD) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] *= gelu_tanh_grad_fast_f(bf2f(x4[r]));
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += bf2f(x4[r]);
            }
          }
          if (g.out_f32) {
            float4* cp = (float4*)((float*)g.C + ci);
            if (g.accumulate) { const float4 o = *cp; v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
            *cp = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            u16x4* cp = (u16x4*)((bf16_t*)g.C + ci);
            if (g.accumulate) { const u16x4 o = *cp;
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += bf2f(o[r]); }
            u16x4 o4;
#pragma unroll
            for (int r = 0; r < 4; ++r) o4[r] = f2bf(v[r]);
            *cp = o4;
          }
        }
      }
      c_kt = 0; if (++c_tn == g.tiles_n) { c_tn = 0; ++c_j; }
    }
  }
}


// =================================================================================================================
// Ping-pong kernel: the ring kernel's data path (persistent, 256x128x64 tile, 3-slot ring, counted vmcnt, steps
// flattened over output tiles) with the schedule of cdna_hip_programming.md's 8-phase template in its smallest form:
// every K-step of a wave is  [LOAD: 16 ds_read_b128 of the step's fragments + 6 LDS-DMA for the tile two steps ahead]
// s_barrier [MFMA: 32 MFMAs from registers] s_barrier,  and the two 4-wave teams (rows 0-127 / 128-255 of the tile)
// run ONE BARRIER APART, so on every SIMD one wave is in its MFMA section while its partner is in its LOAD section.
// Hazards (global barrier index b; team 0's LOAD(s) is interval (2s,2s+1), team 1's is (2s+1,2s+2)):
//   RAW  every wave confirms its pieces of tile s (vmcnt) before barrier 2s: team 0 just before it, team 1 at the end of
//        its LOAD(s-1);
//   WAR  tile s+2 reuses the slot of tile s-1, last read in interval (2s-1,2s); it is issued after barrier 2s.
// =================================================================================================================
__global__ __launch_bounds__(512, 2) void gemm_nt_pp_kernel(NtArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [3][A 32 KiB | B 16 KiB] | bias f32[N<=4096]
  constexpr int SLOT = 49152;
  float* const sbias = (float*)(smem + 3 * SLOT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = w >> 2, tw = w & 3;
  const int wm = team * 128 + (tw >> 1) * 64, wn = (tw & 1) * 64;
  const int fr = lane & 15, fq = lane >> 4;
  const int sr = lane >> 3, scp = lane & 7;
  const int sc = (scp ^ sr) * 8;
  const int KT = g.K / 64;
  const int G = gridDim.x;
  const int my_m = (g.tiles_m - (int)blockIdx.x + G - 1) / G;
  const int nsteps = my_m * g.tiles_n * KT;
  if (g.bias) for (int i = tid; i < g.N; i += 512) sbias[i] = g.bias[i];

  int s_j = 0, s_tn = 0, s_kt = 0;
  auto stage = [&](int slot) {
    const int64_t tm = (int64_t)blockIdx.x + (int64_t)s_j * G;
    char* sa = smem + slot * SLOT; char* sb = sa + 32768;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = w * 4 + i;
      int64_t am = tm * 256 + piece * 8 + sr; if (am > g.M - 1) am = g.M - 1;
      GLDS16(g.A + am * g.lda + sc + s_kt * 64, sa + piece * 1024);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = w * 2 + i;
      int bn = s_tn * 128 + piece * 8 + sr; if (bn > g.N - 1) bn = g.N - 1;
      GLDS16(g.Bt + (int64_t)bn * g.ldb + sc + s_kt * 64, sb + piece * 1024);
    }
    if (++s_kt == KT) { s_kt = 0; if (++s_tn == g.tiles_n) { s_tn = 0; ++s_j; } }
  };
  // own pieces of a tile have landed when only the ONE younger tile (6 LDS-DMA of this wave) may still be outstanding
  auto confirm = [&](bool younger_tile_issued) {
    if (younger_tile_issued) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  if (nsteps > 0) stage(0);
  if (nsteps > 1) stage(1);
  __syncthreads();  // bias visible (also drains the two staged tiles once; start-up only)
  if (team == 1) __builtin_amdgcn_s_barrier();  // the stagger: team 1 runs one barrier behind

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int a_off = (wm + fr) * 128, b_off = (wn + fr) * 128;
  const int x0 = ((fq) ^ (fr & 7)) * 16, x1 = ((4 + fq) ^ (fr & 7)) * 16;

  int c_j = 0, c_tn = 0, c_kt = 0;
  for (int step = 0; step < nsteps; ++step) {
    if (team == 0) confirm(step + 1 < nsteps);      // tile `step`: before global barrier 2*step
    __builtin_amdgcn_s_barrier();
    // ---------------- LOAD section
    const char* sa = smem + (step % 3) * SLOT;
    const char* sb = sa + 32768;
    bf16x8 af[2][4], bfr[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int xo = ks ? x1 : x0;
#pragma unroll
      for (int i = 0; i < 4; ++i) af[ks][i] = *(const bf16x8*)(sa + a_off + i * 2048 + xo);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[ks][j] = *(const bf16x8*)(sb + b_off + j * 2048 + xo);
    }
    if (step + 2 < nsteps) stage((step + 2) % 3);
    if (team == 1 && step + 1 < nsteps) confirm(step + 2 < nsteps);  // tile step+1: before global barrier 2*(step+1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // fragments are in registers: the slot may be refilled later
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- MFMA section (registers only)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    if (++c_kt == KT) {
      const int64_t m0 = ((int64_t)blockIdx.x + (int64_t)c_j * G) * 256;
      const int n0 = c_tn * 128;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t gm = m0 + wm + i * 16 + fr;
        int64_t crow = gm;
        if (g.crow_group > 0) crow = gm + (gm / g.crow_group + 1) * (int64_t)g.crow_skip;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int gn = n0 + wn + j * 16 + fq * 4;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = g.alpha * acc[i][j][r]; acc[i][j][r] = 0.f; }
          if (gm >= g.M || gn >= g.N) continue;
          if (g.bias) { const f32x4 b4 = *(const f32x4*)(sbias + gn); v[0] += b4[0]; v[1] += b4[1]; v[2] += b4[2]; v[3] += b4[3]; }
          const int64_t ci = crow * g.ldc + gn;
          if (g.pre_out) { u16x4 p4;
#pragma unroll
            for (int r = 0; r < 4; ++r) p4[r] = f2bf(v[r]);
            *(u16x4*)(g.pre_out + ci) = p4; }
          if (g.epi == EPI_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_fast_f(v[r]);
          }
          if (g.aux) {
            const u16x4 x4 = *(const u16x4*)(g.aux + ci);
            if (g.epi == EPI_MUL_GELU_GRAD) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] *= gelu_tanh_grad_fast_f(bf2f(x4[r]));
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += bf2f(x4[r]);
            }
          }
          if (g.out_f32) {
            float4* cp = (float4*)((float*)g.C + ci);
            if (g.accumulate) { const float4 o = *cp; v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
            *cp = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            u16x4* cp = (u16x4*)((bf16_t*)g.C + ci);
            if (g.accumulate) { const u16x4 o = *cp;
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += bf2f(o[r]); }
            u16x4 o4;
#pragma unroll
            for (int r = 0; r < 4; ++r) o4[r] = f2bf(v[r]);
            *cp = o4;
          }
        }
      }
      c_kt = 0; if (++c_tn == g.tiles_n) { c_tn = 0; ++c_j; }
    }
  }
  if (team == 0) __builtin_amdgcn_s_barrier();  // matches team 1's extra barrier
}

