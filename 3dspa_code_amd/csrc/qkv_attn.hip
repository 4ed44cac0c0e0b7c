// qkv_attn.hip -- QKV projection + QK-RMSNorm + softmax attention of one (sequence, head) in ONE kernel (gfx950; round 5).
//
// attention.py:154-175 is one module upstream: three DenseGeneral, two RMSNorm, dot_product_attention.  Until round 5 the track encoder ran it as a K = 384 GEMM
// that writes q | k | v (4608 B per token) and an attention kernel that reads them back (75 FLOP/B: an HBM-stream kernel at 13 % MFMA-busy).  Here a persistent
// workgroup (4 waves, one per SIMD) takes (sequence, head) problems; per problem
//   * the LayerNorm-ed rows nq[S <= 160][384] never touch LDS: every wave owns two or three 16-token tiles and holds them as MFMA B fragments in registers
//     (36 x 16-byte loads per tile, all in flight at once);
//   * the head's 384 x 288 weight slice streams through a 4-slot LDS ring as pre-packed 1-KiB MFMA A fragments in consumption order (12 k-steps of 18 KiB;
//     the stream is continuous across problems, so the next problem's first three k-steps land under this problem's attention phase);
//   * C^T = W^T . nq^T accumulates as [output column][token] tiles: a lane ends with 4 consecutive columns of ONE token, so the RMSNorm row sums are in-lane
//     plus two xor-shuffles, q | k | v leave as 8-byte pieces (they are still stored: the backward's dW GEMM and RMSNorm backward read them), the normalised
//     q^ of the wave's own tokens stay in REGISTERS as the B operands of S^T = K^ Q^T (the k index of an accumulator pair is a permutation of 32 d's; the K^ image
//     is written with its columns permuted the same way), and only K^ and V go to LDS images;
//   * the attention tile routine of attention_fused.hip (S^T in registers, softmax in-lane, V read transposed) runs on the wave's own query tiles.
// Same rounding points as the two-kernel path: q | k | v are rounded to 16 bits first and normalised from the rounded values.
#include "attn_common.hpp"

namespace SPA_NS {

constexpr int QA_KS = 12;               // k-steps of 32 over d = 384
constexpr int QA_NF = 18;               // 16-column fragments of a head's q | k | v
constexpr int QA_SLOT = QA_NF * 1024;   // one k-step of the weight stream
constexpr int QA_NSLOT = 4;
constexpr int QA_KT = 10, QA_SPAD = 160, QA_D = 384;
constexpr int64_t QA_HEAD_BYTES = (int64_t)QA_KS * QA_SLOT;
constexpr int QA_ABL = SPA3D_ABL_QKVA;  // csrc/ablate.inc: 0 in libspa3d_hip.so.  1 no attention phase, 2 no q | k | v stores, 4 no MFMAs / fragment reads in the k-loop, 8 no LDS-DMA, 16 no epilogue at all
constexpr int QA_LDS = QA_NSLOT * QA_SLOT + 2 * img_bytes(QA_SPAD) + QA_SPAD * 4 + 4 * WTILE;

// weight stream: out[h][s][f][lane][e] = W_x[k = 32 s + 8 (lane >> 4) + e][96 h + 16 (f % 6) + (lane & 15)], x = f / 6 (q, k, v); W_x is [384][E] (the Flax kernel [d, H, Dh])
template <typename S_>
__global__ void qa_pack_kernel(const S_* __restrict__ wq, const S_* __restrict__ wk, const S_* __restrict__ wv, int E, int H, bf16_t* __restrict__ out) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= H * QA_KS * QA_NF * 64) return;
  const int lane = id & 63, f = (id >> 6) % QA_NF, s = (id / (64 * QA_NF)) % QA_KS, h = id / (64 * QA_NF * QA_KS);
  const S_* w = f < 6 ? wq : (f < 12 ? wk : wv);
  const int col = 96 * h + 16 * (f % 6) + (lane & 15);
  u16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = f2bf(ld<S_>(w + (int64_t)(32 * s + 8 * (lane >> 4) + e) * E + col));
  *(u16x8*)(out + (int64_t)id * 8) = v;
}

struct QkvaArgs {
  const bf16_t* nq; int64_t ldn; const char* wpk; const float *sq, *sk, *km; const int32_t* seq_off;
  int S, H; int64_t nprob;
  bf16_t* qkv; int64_t ldq;   // [rows][3 E]: q | k | v as the projection GEMM writes them
  bf16_t* o; float* lse;      // as attn_fwd_kernel
};

template <int OFF> __device__ __forceinline__ void qa_glds(const void* base_uniform, unsigned off, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(off), "s"(base_uniform), "s"(lds_dst), "n"(OFF) : "memory", "m0");
}
#define QA_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define QA_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

__global__ __launch_bounds__(256, 1) void qkva_fwd_kernel(QkvaArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem; char* Ks = smem + QA_NSLOT * QA_SLOT; char* Vs = Ks + img_bytes(QA_SPAD);
  float* kbias = (float*)(Vs + img_bytes(QA_SPAD));
  const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* wt = (char*)(kbias + QA_SPAD) + w * WTILE;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int64_t nb = gridDim.x, nseq = g.nprob / g.H;
  const int64_t nmine = (g.nprob - (int64_t)blockIdx.x + nb - 1) / nb;   // problems blockIdx.x + i nb
  const int E = g.H * DH;
  const unsigned voff = (unsigned)lane * 16u;

  // step gs of the stream = k-step gs % 12 of problem gs / 12 into ring slot gs & 3; wave w moves pieces w, w + 4, .. (5 for waves 0, 1; 4 for waves 2, 3).
  // (The head of a step's problem is passed in: map_prob's divisions per k-step were ~400 scalar instructions.)
  auto issue = [&](int h, int s, unsigned slot) {
    if constexpr (QA_ABL & 8) return;
    const char* src = g.wpk + (int64_t)h * QA_HEAD_BYTES + s * QA_SLOT + w * 1024;
    const unsigned dst = lds0 + slot * QA_SLOT + (unsigned)w * 1024u;
    qa_glds<0>(src, voff, dst);
    qa_glds<0>(src + 4096, voff, dst + 4096);
    qa_glds<0>(src + 8192, voff, dst + 8192);
    qa_glds<0>(src + 12288, voff, dst + 12288);
    if (w < 2) qa_glds<0>(src + 16384, voff, dst + 16384);
  };
  const int nseq32 = (int)nseq, nb32 = (int)nb;
  auto prob_of = [&](int i) { return (int)map_prob((int64_t)((int)blockIdx.x + i * nb32), (int64_t)nseq32, g.H); };
  int prob_n = nmine > 0 ? prob_of(0) : 0;
  if (nmine > 0) { const int h0 = prob_n % g.H; issue(h0, 0, 0u); issue(h0, 1, 1u); issue(h0, 2, 2u); }

  for (int64_t i = 0; i < nmine; ++i) {
    const int prob = prob_n;
    prob_n = i + 1 < nmine ? prob_of((int)i + 1) : 0;
    const int seq = prob / g.H, h = prob - seq * g.H, h_n = prob_n % g.H;
    const bool more = i + 1 < nmine;
    const int64_t row0 = g.seq_off ? (int64_t)g.seq_off[seq] : (int64_t)seq * g.S;
    const int S = g.seq_off ? g.seq_off[seq + 1] - (int)row0 : g.S;
    const int QT = (S + 15) >> 4;
    const int nt = QT / 4 + (w < (QT & 3) ? 1 : 0), t0 = w * (QT / 4) + (w < (QT & 3) ? w : (QT & 3));   // this wave's token tiles t0 .. t0 + nt - 1 (nt <= 3)

    auto body = [&](auto nt_c) {
      constexpr int NT = decltype(nt_c)::value;
      // ---- the wave's rows as B fragments: lane (token 16 tile + fr, k = 32 s + 8 fq + e); tiles past nt are clamped copies whose results are dropped
      u16x8 nqf[NT][QA_KS];
      int tok[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int tile = t < nt ? t0 + t : (QT > 0 ? QT - 1 : 0);
        tok[t] = 16 * tile + fr;
        int r = tok[t] < S ? tok[t] : S - 1; if (r < 0) r = 0;
        const bf16_t* p = g.nq + (row0 + r) * g.ldn + 8 * fq;
#pragma unroll
        for (int s = 0; s < QA_KS; ++s) nqf[t][s] = *(const u16x8*)(p + 32 * s);
      }
      f32x4 acc[NT][QA_NF];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < QA_NF; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      // ---- C^T[column][token] += W^T . nq^T over 12 k-steps
      static_for<0, QA_KS>([&](auto s_) {
        constexpr int s = decltype(s_)::value;
        // this wave's pieces of this step have landed: the pieces of the next two steps (issued since) may still be in flight
        if (s + 2 < QA_KS || more) { if (w < 2) QA_WAIT_VM(10); else QA_WAIT_VM(8); } else QA_WAIT_VM(0);
        QA_BAR();   // ... and everyone's; every wave has left the slot of the previous step, which step + 3 refills.  12 steps per problem: slot = s & 3
        if constexpr (s + 3 < QA_KS) issue(h, s + 3, (unsigned)((s + 3) & 3));
        else if (more) issue(h_n, s + 3 - QA_KS, (unsigned)((s + 3) & 3));
        // the step's 18 weight fragments through 9 rolling registers: untracked reads (LDS operations return in order), one counted wait per fragment.  Left to
        // the compiler every read was followed by lgkmcnt(0) and its two MFMAs: 18 exposed LDS latencies per step, 4x the MFMA time.
        const char* sp = ring + (s & 3) * QA_SLOT + lane * 16;
        uint4 af[9];
        if constexpr (!(QA_ABL & 4)) static_for<0, 9>([&](auto j_) { constexpr int j = decltype(j_)::value; af[j] = lds_b128_o<j * 1024>(sp); });
        if constexpr (!(QA_ABL & 4)) static_for<0, QA_NF>([&](auto j_) {
          constexpr int j = decltype(j_)::value;
          constexpr int allowed = j <= 9 ? 8 : 17 - j;   // reads younger than fragment j's at this point
          asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(allowed) : "memory");
          __builtin_amdgcn_sched_barrier(0);
          const mfma16x8 a = __builtin_bit_cast(mfma16x8, af[j % 9]);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t][j] = MFMA16(a, __builtin_bit_cast(mfma16x8, nqf[t][s]), acc[t][j]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (j + 9 < QA_NF) af[j % 9] = lds_b128_o<(j + 9) * 1024>(sp);
        });
      });
      // ---- epilogue: q | k | v to memory (16-bit), q^ into B fragments, k^ and v into the LDS images.  The images' previous readers (the last problem's attention phase)
      // passed twelve barriers ago.
      mfma16x8 qb[NT][3];
#pragma unroll
      for (int t = 0; t < ((QA_ABL & 16) ? 0 : NT); ++t) {
        const bool own = t < nt, valid = own && tok[t] < S;
        bf16_t* qrow = g.qkv + (row0 + tok[t]) * g.ldq + h * DH + 4 * fq;
        u16x4 raw[QA_NF];
#pragma unroll
        for (int j = 0; j < QA_NF; ++j) {
#pragma unroll
          for (int r = 0; r < 4; ++r) raw[j][r] = f2bf(acc[t][j][r]);
          if (valid && !(QA_ABL & 2)) *(u16x4*)(qrow + (j / 6) * E + 16 * (j % 6)) = raw[j];
        }
        // RMSNorm over the 96 columns of the token (24 in this lane, the rest in lanes fq ^ 1, 2, 3), from the ROUNDED values, as the attention kernel does it
        float ssq = 0.f, ssk = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float a = bf2f(raw[j][r]), b = bf2f(raw[6 + j][r]); ssq += a * a; ssk += b * b; }
        ssq += __shfl_xor(ssq, 16, 64); ssq += __shfl_xor(ssq, 32, 64);
        ssk += __shfl_xor(ssk, 16, 64); ssk += __shfl_xor(ssk, 32, 64);
        const float rq = rsqrtf(ssq / DH + 1e-6f), rk = rsqrtf(ssk / DH + 1e-6f);
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) {   // q^: k-step s2 of S^T = accumulator tiles 2 s2, 2 s2 + 1: element e <-> d = 32 s2 + 16 (e >> 2) + 4 fq + (e & 3)
          u16x8 q8;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int j = 2 * s2 + (e >> 2), d = 16 * j + 4 * fq + (e & 3);
            q8[e] = f2bf(bf2f(raw[j][e & 3]) * rq * g.sq[d]);
          }
          qb[t][s2] = __builtin_bit_cast(mfma16x8, q8);
        }
        if (own) {
          char* krow = Ks + row_off(tok[t]); char* vrow = Vs + row_off(tok[t]);
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            u16x4 k4, v4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              k4[r] = valid ? f2bf(bf2f(raw[6 + j][r]) * rk * g.sk[16 * j + 4 * fq + r]) : (bf16_t)0;
              v4[r] = valid ? raw[12 + j][r] : (bf16_t)0;
            }
            // K^ image columns permuted like q^'s k index: d = 16 j + 4 fq + r sits at position 32 (j >> 1) + 8 fq + 4 (j & 1) + r
            *(u16x4*)(krow + (32 * (j >> 1) + 8 * fq + 4 * (j & 1)) * 2) = k4;
            *(u16x4*)(vrow + (16 * j + 4 * fq) * 2) = v4;
          }
        }
      }
      // rows of the tiles nobody owns (16 QT .. 159) and the key bias
      for (int x = tid; x < (QA_SPAD - 16 * QT) * 24; x += 256) {
        const int r = 16 * QT + x / 24, c = x % 24;
        *(uint2*)(Ks + row_off(r) + c * 8) = make_uint2(0u, 0u); *(uint2*)(Vs + row_off(r) + c * 8) = make_uint2(0u, 0u);
      }
      for (int t = tid; t < QA_SPAD; t += 256) {
        float b = 0.f;
        if (t >= S) b = -__builtin_inff();
        else if (g.km && g.km[row0 + t] == 0.f) b = NEG_BIG;
        kbias[t] = b;
      }
      __syncthreads();
      // ---- attention on the wave's own query tiles (the tile routine of attn_fwd_kernel, Q from registers)
      constexpr int KT = QA_KT;
      const float qscale = 0.10206207261596575f;  // 1/sqrt(96)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t < nt && !(QA_ABL & 1)) {
          const int q0 = tok[t] - fr;
          f32x4 sc[KT];
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 3; ++s) {
              const mfma16x8 kf = *(const mfma16x8*)(Ks + kt * ROW16 + row_off(fr) + (s * 32 + fq * 8) * 2);
              sc[kt] = MFMA16(kf, qb[t][s], sc[kt]);
            }
          }
          float m = NEG_BIG;
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            const f32x4 b4 = *(const f32x4*)(kbias + kt * 16 + fq * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) { sc[kt][r] = sc[kt][r] * qscale + b4[r]; m = fmaxf(m, sc[kt][r]); }
          }
          m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
          float l = 0.f;
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float p = __expf(sc[kt][r] - m); sc[kt][r] = p; l += p; }
          l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
          const float inv = 1.f / l;
          if (g.lse && fq == 0 && q0 + fr < S) { float* lp = g.lse + (row0 * g.H + (int64_t)h * S + q0 + fr) * 2; lp[0] = m; lp[1] = __logf(l); }
          mfma16x8 pb[KT / 2];
#pragma unroll
          for (int s2 = 0; s2 < KT / 2; ++s2) {
            u16x8 p8;
#pragma unroll
            for (int e = 0; e < 8; ++e) p8[e] = f2bf(sc[2 * s2 + (e >> 2)][e & 3] * inv);
            pb[s2] = __builtin_bit_cast(mfma16x8, p8);
          }
          const int tq = fr >> 2, tp = fr & 3;
          const char* vbase = Vs + tp * 8 + row_off(4 * fq + tq);
          static_for<0, 6>([&](auto dtc) {
            constexpr int dt = decltype(dtc)::value;
            f32x4 oacc = f32x4{0.f, 0.f, 0.f, 0.f};
            uint2 lo[KT / 2], hi[KT / 2];
            static_for<0, KT / 2>([&](auto sc_) {
              constexpr int s2 = decltype(sc_)::value;
              lo[s2] = lds_tr16_b64_o<dt * 32 + 2 * s2 * ROW16>(vbase); hi[s2] = lds_tr16_b64_o<dt * 32 + (2 * s2 + 1) * ROW16>(vbase);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s2 = 0; s2 < KT / 2; ++s2)
              oacc = MFMA16(__builtin_bit_cast(mfma16x8, make_uint4(lo[s2].x, lo[s2].y, hi[s2].x, hi[s2].y)), pb[s2], oacc);
            u16x4 o4;
#pragma unroll
            for (int r = 0; r < 4; ++r) o4[r] = f2bf(oacc[r]);
            tile_put(wt, dt, o4, lane);
          });
          tile_flush(wt, g.o + (row0 + q0) * E + h * DH, E, S - q0, lane);
        }
      }
    };
    if (nt == 3) body(std::integral_constant<int, 3>{}); else body(std::integral_constant<int, 2>{});
  }
  QA_WAIT_VM(0);
}

int64_t qkv_attn_pack_elems(int H) { return (int64_t)H * QA_KS * QA_NF * 512; }
template <typename S_> void qkv_attn_pack(spa3d_ctx* c, const S_* wq, const S_* wk, const S_* wv, int E, int H, bf16_t* wpk) {
  if (c->dry) return;
  const int n = H * QA_KS * QA_NF * 64;
  qa_pack_kernel<S_><<<(n + 255) / 256, 256, 0, c->stream>>>(wq, wk, wv, E, H, wpk);
  SPA_LAUNCH_CHECK(c);
}
template void qkv_attn_pack<float>(spa3d_ctx*, const float*, const float*, const float*, int, int, bf16_t*);
template void qkv_attn_pack<bf16_t>(spa3d_ctx*, const bf16_t*, const bf16_t*, const bf16_t*, int, int, bf16_t*);

// qkv[rows][3 E] = nq . (Wq | Wk | Wv), o / lse = attention of every (sequence, head); false = shape not covered (the caller runs the GEMM and the attention kernel)
bool qkv_attn_fwd(spa3d_ctx* c, const bf16_t* nq, int64_t ldn, const bf16_t* wpk, const float* sq, const float* sk, const float* km, int64_t nseq, int S, int H,
                  int Dh, int d, bf16_t* qkv, bf16_t* o, float* lse, const int32_t* seq_off, int64_t total_rows) {
  if (!wpk || Dh != DH || d != QA_D || S < 2 || S > QA_SPAD || H < 1 || nseq < 1 || ldn % 8) return false;
  if ((((uintptr_t)nq) | ((uintptr_t)qkv) | ((uintptr_t)o) | ((uintptr_t)wpk)) & 15) return false;
  if (nseq * H > 0x7fffffffLL) return false;
  if (c->dry) return true;
  QkvaArgs a{};
  a.nq = nq; a.ldn = ldn; a.wpk = (const char*)wpk; a.sq = sq; a.sk = sk; a.km = km; a.seq_off = seq_off; a.S = S; a.H = H; a.nprob = nseq * H;
  a.qkv = qkv; a.ldq = 3 * (int64_t)H * DH; a.o = o; a.lse = lse;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)qkva_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, QA_LDS); attr = true; }
  const double rows = total_rows > 0 ? (double)total_rows : (double)nseq * S;
  {
    ProfScope ps(c, PROF_ATTN_FWD, 2.0 * rows * QA_D * 3.0 * H * DH + 4.0 * rows * H * (rows / nseq) * Dh, rows * (QA_D + 4.0 * H * DH) * 2.0);
    ps.tag(nseq, S, H, 2);
    const unsigned grid = (unsigned)std::min<int64_t>(a.nprob, 256);
    qkva_fwd_kernel<<<grid, 256, QA_LDS, c->stream>>>(a);
  }
  SPA_LAUNCH_CHECK(c);
  return true;
}

}  // namespace SPA_NS
