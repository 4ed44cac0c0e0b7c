// attention_fused.hip -- LDS-resident short-sequence attention for 16-bit activations (gfx950): the temporal self-attention of
// the track encoder (S = T+1 = 151; 301 at BASELINE cfg#5) and the readout stack (S = 129), d_head = 96 (attention.py:166-175).
//
// One workgroup per (sequence, head).  The whole normalised K and the V of the head (S_pad x 96 x 2 B = 30 KiB each at
// S = 151, 60 KiB at S = 301) are staged into LDS once; RMSNorm of q/k (attention.py:166-167), the 1/sqrt(d) scale, the key
// mask, the softmax and both contractions are fused, so per token only q,k,v are read and o (+ a 4-byte LSE per
// head) written.  No online-softmax streaming: a full row of scores lives in registers (S <= 320).
//
// Forward, per 16-query tile (tiles are dealt round-robin to the waves), everything "transposed" so that the
// softmax reductions stay inside a lane plus two xor-shuffles and P never leaves registers:
//   S^T[key][q] = mfma(A = K^[key][:], B = Q^[q][:])          C-layout: lane (fr,fq) holds keys 16t+4fq+r of query fr
//   P^T        = exp(S^T + kbias - max) ; l = sum             in-lane over (t,r), then lanes fq via shfl_xor 16,32
//   O^T[d][q]  = mfma(A = V^T[d][keys], B = P^T[keys][q])     B operand = the S^T accumulators re-packed to 16 bit with the
//                                                             k index permuted (cdna_hip_programming.md section 3 "An accumulator
//                                                             tile as the next MFMA's operand"); A operand = V read from its
//                                                             row-major LDS image with ds_read_b64_tr_b16 using the same k order.
//   lane ends with 4 consecutive d of one query -> 8-byte stores.
//
// Memory-side structure (round 2; the kernels are HBM-bound at 75 FLOP/B):
//   * every global load a workgroup needs before its first barrier (K, V rows and the wave's own Q fragments; in the backward
//     Q, K, V, dO, O) is ISSUED before the first value is used: one memory latency per problem instead of one per matrix;
//   * blocks are mapped to problems XCD-major (map_prob): blocks b, b+8, .. share an XCD and take the 8 heads of one sequence
//     back to back, so the 128-B lines that two heads' 192-B row segments share, the key mask and the LSE rows hit that XCD's L2.
#include "attn_common.hpp"

namespace SPA_NS {

struct AttnArgs {
  const bf16_t *q, *k, *v; int64_t ldq, ldk, ldv;
  const float *sq, *sk, *km;
  int S, H; int64_t nprob;
  bf16_t* o; float* lse;  // o [nseq*S][H*96]; lse [nseq][H][S][2] = (row max, log row sum) (may be null)
  const int32_t* seq_off;  // null: dense sequences of S rows.  Else ragged (token pruning): sequence i = rows [seq_off[i], seq_off[i+1]) of q/k/v/o,
                           // at most S of them; lse of (sequence, head, t) sits at ((seq_off[i] * H + h * S_i) + t) * 2 either way
};

// S rows of [*, 96] 16-bit elements (row stride ld elements): 4 threads per row, RPP rows per pass; thread `part` of a row holds its
// 16-byte chunks part, part+4, part+8 (elements 8(part+4c)+j), so every wave-instruction moves 64 contiguous bytes of 16 rows
template <int NP> struct RawRows { u16x8 x[NP][3]; };

// `tid`: the thread index; a persistent kernel passes an OPAQUE copy made inside its problem loop (opaque_tid) so that the per-row byte offsets are
// recomputed per problem instead of being hoisted out of the loop as lane constants, spilled, and reloaded behind s_waitcnt vmcnt(0) between the
// loads they serialise (cdna_hip_programming.md App. B, 4-wave attention "Pitfalls")
__device__ __forceinline__ int opaque_tid() { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; }
template <int NP, int RPP>
__device__ __forceinline__ void rows_load(RawRows<NP>& r, const bf16_t* __restrict__ src, int64_t ld_, int S, int tid = threadIdx.x) {
  const int part = tid & 3, r0 = tid >> 2;
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int row = r0 + RPP * ps;
    if (row < S) {
      const u16x8* p = (const u16x8*)(src + (int64_t)row * ld_) + part;  // chunks part, part+4, part+8: 64 contiguous bytes per row and instruction
      r.x[ps][0] = p[0]; r.x[ps][1] = p[4]; r.x[ps][2] = p[8];
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) r.x[ps][c][j] = 0;
    }
  }
}
// into LDS rows of 192 B.  NORM: per-row RMSNorm * scale (attention.py:167) before the 16-bit round, as the unfused path stores it.
template <bool NORM, int NP, int RPP>
__device__ __forceinline__ void rows_store(RawRows<NP>& r, int S_pad, const float* __restrict__ scale, char* lds) {
  const int part = threadIdx.x & 3, r0 = threadIdx.x >> 2;
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int row = r0 + RPP * ps;
    if (NORM) {
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
        for (int j = 0; j < 8; ++j) r.x[ps][c][j] = f2bf(f[c * 8 + j] * rr * scale[(part + 4 * c) * 8 + j]);
    }
    if (row < S_pad) {
      u16x8* d = (u16x8*)(lds + row_off(row)) + part;
      d[0] = r.x[ps][0]; d[4] = r.x[ps][1]; d[8] = r.x[ps][2];
    }
  }
}

// One row's B-operand fragments (lane (row fr, d = 32s + 8fq + j)) straight from global memory, RMS-normalised over the 96 d
// (3 s x 4 fq lanes) and scaled: the same rounding points as the staged rows.
__device__ __forceinline__ void frag_norm(const u16x8 (&x)[3], const float* __restrict__ scale, int fq, mfma16x8 (&out)[3]) {
  float f[24]; float ss = 0.f;
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) { f[s * 8 + j] = bf2f(x[s][j]); ss += f[s * 8 + j] * f[s * 8 + j]; }
  ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
  const float rr = rsqrtf(ss / DH + 1e-6f);
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    u16x8 t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = f2bf(f[s * 8 + j] * rr * scale[s * 32 + fq * 8 + j]);
    out[s] = __builtin_bit_cast(mfma16x8, t);
  }
}


template <int KT, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_kernel(AttnArgs g) {
  constexpr int S_pad = KT * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem; char* Vs = smem + img_bytes(S_pad); float* kbias = (float*)(smem + 2 * img_bytes(S_pad));
  constexpr bool WIDE = fwd_wide(KT, NW);  // output rows through a wave tile (not where it would cost the second workgroup per CU)
  char* wt = (char*)(kbias + S_pad) + (threadIdx.x >> 6) * WTILE;
  unsigned seq_u, h_u; map_prob32(blockIdx.x, (unsigned)g.nprob / (unsigned)g.H, (unsigned)g.H, seq_u, h_u);
  const int64_t seq = seq_u; const int h = (int)h_u;
  const int64_t row0 = g.seq_off ? (int64_t)g.seq_off[seq] : seq * g.S;       // first row of the sequence
  const int S = g.seq_off ? g.seq_off[seq + 1] - (int)row0 : g.S, E = g.H * DH;  // its length (wave-uniform)
  const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
  const int w = tid >> 6;
  const int QT = (S + 15) / 16;

  constexpr int RPP = NW * 16, NP = (S_pad + RPP - 1) / RPP, NT = (KT + NW - 1) / NW;
  // every global load of the problem is in flight before the first use: K rows, V rows, this wave's Q fragments
  RawRows<NP> rk, rv;
  rows_load<NP, RPP>(rk, g.k + row0 * g.ldk + h * DH, g.ldk, S);
  rows_load<NP, RPP>(rv, g.v + row0 * g.ldv + h * DH, g.ldv, S);
  u16x8 qx[NT][3];
#pragma unroll
  for (int it = 0; it < NT; ++it) {
    int qrow = (w + NW * it) * 16 + fr; if (qrow > S - 1) qrow = S - 1;
    const bf16_t* qp = g.q + (row0 + qrow) * g.ldq + h * DH + fq * 8;
#pragma unroll
    for (int s = 0; s < 3; ++s) qx[it][s] = *(const u16x8*)(qp + s * 32);
  }
  rows_store<true, NP, RPP>(rk, S_pad, g.sk, Ks);
  rows_store<false, NP, RPP>(rv, S_pad, nullptr, Vs);
  for (int t = tid; t < S_pad; t += NW * 64) {
    float b = 0.f;
    if (t >= S) b = -__builtin_inff();                           // padding key: weight exactly 0
    else if (g.km && g.km[row0 + t] == 0.f) b = NEG_BIG;      // where(mask, logit, finfo.min)
    kbias[t] = b;
  }
  __syncthreads();

  const float qscale = 0.10206207261596575f;  // 1/sqrt(96)
#pragma unroll
  for (int it = 0; it < NT; ++it) {
    const int qt = w + NW * it;
    if (qt < QT) {
      const int q0 = qt * 16;
      // ---- Q fragment: B operand, lane (query fr, dh = 32s + 8fq + j); same rounding points as the unfused path:
      // 16-bit(q^ = x r scale), then the 1/sqrt(d) scale in fp32 on the logits
      mfma16x8 qb[3];
      frag_norm(qx[it], g.sq, fq, qb);
      // ---- S^T tiles
      f32x4 acc[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const mfma16x8 kf = *(const mfma16x8*)(Ks + kt * ROW16 + row_off(fr) + (s * 32 + fq * 8) * 2);
          acc[kt] = MFMA16(kf, qb[s], acc[kt]);
        }
      }
      // ---- masked softmax over keys (rows of S^T) for query fr
      float m = NEG_BIG;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const f32x4 b4 = *(const f32x4*)(kbias + kt * 16 + fq * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc[kt][r] = acc[kt][r] * qscale + b4[r]; m = fmaxf(m, acc[kt][r]); }
      }
      m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
      float l = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float p = __expf(acc[kt][r] - m); acc[kt][r] = p; l += p; }
      l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
      const float inv = 1.f / l;
      if (g.lse && fq == 0 && q0 + fr < S) { float* lp = g.lse + (row0 * g.H + (int64_t)h * S + q0 + fr) * 2; lp[0] = m; lp[1] = __logf(l); }
      // ---- P^T as B operands: k-step s2 covers key tiles 2*s2, 2*s2+1; element j <-> key 16*(2*s2 + (j>>2)) + 4*fq + (j&3)
      mfma16x8 pb[KT / 2];
#pragma unroll
      for (int s2 = 0; s2 < KT / 2; ++s2) {
        u16x8 t;
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = f2bf(acc[2 * s2 + (j >> 2)][j & 3] * inv);
        pb[s2] = __builtin_bit_cast(mfma16x8, t);
      }
      // odd KT (S = 129: nine key tiles, not ten): the last key tile enters as a half-filled k = 32 product (elements 4-7 of both operands zero)
      mfma16x8 ph;
      if constexpr (KT & 1) {
        u16x8 t8;
#pragma unroll
        for (int j = 0; j < 4; ++j) { t8[j] = f2bf(acc[KT - 1][j] * inv); t8[4 + j] = 0; }
        ph = __builtin_bit_cast(mfma16x8, t8);
      }
      // ---- O^T[d][q] = sum_keys V^T[d][key] P^T[key][q]
      const int tq = fr >> 2, tp = fr & 3;  // this lane's slot in its 16-lane transposed-read group
      const char* vbase = Vs + tp * 8 + row_off(4 * fq + tq);
      static_for<0, 6>([&](auto dtc) {
        constexpr int dt = decltype(dtc)::value;
        f32x4 oacc = f32x4{0.f, 0.f, 0.f, 0.f};
        uint2 lo[KT / 2], hi[KT / 2];
        static_for<0, KT / 2>([&](auto sc_) {
          constexpr int s2 = decltype(sc_)::value;
          lo[s2] = lds_tr16_b64_o<dt * 32 + 2 * s2 * ROW16>(vbase); hi[s2] = lds_tr16_b64_o<dt * 32 + (2 * s2 + 1) * ROW16>(vbase);
        });
        uint2 lo_last = make_uint2(0, 0);
        if constexpr (KT & 1) lo_last = lds_tr16_b64_o<dt * 32 + (KT - 1) * ROW16>(vbase);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);  // MFMAs must stay below the wait (cdna_hip_programming.md rule 18)
#pragma unroll
        for (int s2 = 0; s2 < KT / 2; ++s2) {
          const uint4 vu = make_uint4(lo[s2].x, lo[s2].y, hi[s2].x, hi[s2].y);
          oacc = MFMA16(__builtin_bit_cast(mfma16x8, vu), pb[s2], oacc);
        }
        if constexpr (KT & 1) oacc = MFMA16(__builtin_bit_cast(mfma16x8, make_uint4(lo_last.x, lo_last.y, 0u, 0u)), ph, oacc);  // upper half of k: zeros on both sides
        u16x4 o4;
#pragma unroll
        for (int r = 0; r < 4; ++r) o4[r] = f2bf(oacc[r]);
        if constexpr (WIDE) tile_put(wt, dt, o4);
        else if (q0 + fr < S) *(u16x4*)(g.o + (row0 + q0 + fr) * E + h * DH + dt * 16 + fq * 4) = o4;
      });
      if constexpr (WIDE) tile_flush(wt, g.o + (row0 + q0) * E + h * DH, E, S - q0);
    }
  }
}

template <int KT, int NW>
static void launch_fwd(spa3d_ctx* c, const AttnArgs& a) {
  const int lds = 2 * img_bytes(KT * 16) + KT * 16 * 4 + (fwd_wide(KT, NW) ? NW * WTILE : 0);
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<KT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
  attn_fwd_kernel<KT, NW><<<(unsigned)a.nprob, NW * 64, lds, c->stream>>>(a);
}


// =================================================================================================================
// cross attention (tracks_to_latents, track_autoencoder_3d.py:95-100,201: 128 latent queries against the N = 2048 track tokens of a
// sample; attention.py:92-100).  Sq <= 128 query rows, Sk keys cut into chunks of 128: one workgroup per (sequence, head, chunk) runs the
// forward tile routine of attn_fwd_kernel<8, 4> on its chunk and leaves an UNNORMALISED partial (sum_k exp(s - m_chunk) v, m_chunk,
// sum_k exp(s - m_chunk)); xattn_combine_kernel merges the chunks (the split-softmax identity) and writes o and (max, log-sum).
// Replaces a five-launch composition of strided GEMMs and row kernels that ran at 14-40 TFLOP/s (1.4 ms -> < 0.1 ms per block).
// =================================================================================================================
constexpr int XCHUNK = 128;
struct XAttnArgs {
  const bf16_t *q, *k, *v; int64_t ldq, ldk, ldv;
  const float *sq, *sk, *km;
  int Sq, Sk, H, nsplit; int64_t nprob;
  float* opart;   // [nprob][nsplit][Sq][96]
  float* mlpart;  // [nprob][nsplit][Sq][2]
};

__global__ __launch_bounds__(256, 2) void xattn_fwd_kernel(XAttnArgs g) {
  constexpr int KT = XCHUNK / 16, NW = 4, S_pad = XCHUNK, RPP = NW * 16, NP = S_pad / RPP, NT = KT / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem; char* Vs = smem + img_bytes(S_pad); float* kbias = (float*)(smem + 2 * img_bytes(S_pad));
  const unsigned pi_u = blockIdx.x, prob_u = pi_u / (unsigned)g.nsplit, seq_u = prob_u / (unsigned)g.H;   // 32-bit: nprob * nsplit < 2^31 (xattn_fwd)
  const int64_t pi = pi_u, prob = prob_u; const int sp = (int)(pi_u - prob_u * (unsigned)g.nsplit);
  const int64_t seq = seq_u; const int h = (int)(prob_u - seq_u * (unsigned)g.H);
  const int64_t qrow0 = seq * g.Sq, krow0 = seq * g.Sk + (int64_t)sp * S_pad;
  const int Sq = g.Sq, kn = min(S_pad, g.Sk - sp * S_pad), QT = (Sq + 15) / 16;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, fq = lane >> 4;
  RawRows<NP> rk, rv;
  rows_load<NP, RPP>(rk, g.k + krow0 * g.ldk + h * DH, g.ldk, kn);
  rows_load<NP, RPP>(rv, g.v + krow0 * g.ldv + h * DH, g.ldv, kn);
  u16x8 qx[NT][3];
#pragma unroll
  for (int it = 0; it < NT; ++it) {
    int qrow = (w + NW * it) * 16 + fr; if (qrow > Sq - 1) qrow = Sq - 1;
    const bf16_t* qp = g.q + (qrow0 + qrow) * g.ldq + h * DH + fq * 8;
#pragma unroll
    for (int s = 0; s < 3; ++s) qx[it][s] = *(const u16x8*)(qp + s * 32);
  }
  rows_store<true, NP, RPP>(rk, S_pad, g.sk, Ks);
  rows_store<false, NP, RPP>(rv, S_pad, nullptr, Vs);
  for (int t = tid; t < S_pad; t += NW * 64) {
    float b = 0.f;
    if (t >= kn) b = -__builtin_inff();
    else if (g.km && g.km[krow0 + t] == 0.f) b = NEG_BIG;
    kbias[t] = b;
  }
  __syncthreads();
  const float qscale = 0.10206207261596575f;  // 1/sqrt(96)
#pragma unroll
  for (int it = 0; it < NT; ++it) {
    const int qt = w + NW * it;
    if (qt < QT) {
      const int q0 = qt * 16;
      mfma16x8 qb[3];
      frag_norm(qx[it], g.sq, fq, qb);
      f32x4 acc[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const mfma16x8 kf = *(const mfma16x8*)(Ks + kt * ROW16 + row_off(fr) + (s * 32 + fq * 8) * 2);
          acc[kt] = MFMA16(kf, qb[s], acc[kt]);
        }
      }
      float m = NEG_BIG;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const f32x4 b4 = *(const f32x4*)(kbias + kt * 16 + fq * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc[kt][r] = acc[kt][r] * qscale + b4[r]; m = fmaxf(m, acc[kt][r]); }
      }
      m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
      float l = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float p = __expf(acc[kt][r] - m); acc[kt][r] = p; l += p; }
      l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
      const bool live = q0 + fr < Sq;
      if (fq == 0 && live) { float* mp = g.mlpart + (pi * Sq + q0 + fr) * 2; mp[0] = m; mp[1] = l; }
      mfma16x8 pb[KT / 2];
#pragma unroll
      for (int s2 = 0; s2 < KT / 2; ++s2) {
        u16x8 t;
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = f2bf(acc[2 * s2 + (j >> 2)][j & 3]);
        pb[s2] = __builtin_bit_cast(mfma16x8, t);
      }
      const int tq = fr >> 2, tp = fr & 3;
      const char* vbase = Vs + tp * 8 + row_off(4 * fq + tq);
      float* op = g.opart + (pi * Sq + q0 + fr) * DH + fq * 4;
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
        for (int s2 = 0; s2 < KT / 2; ++s2) {
          const uint4 vu = make_uint4(lo[s2].x, lo[s2].y, hi[s2].x, hi[s2].y);
          oacc = MFMA16(__builtin_bit_cast(mfma16x8, vu), pb[s2], oacc);
        }
        if (live) *(f32x4*)(op + dt * 16) = oacc;
      });
    }
  }
}

// one wave per (sequence, head, query): O = sum_c e^{m_c - M} O_c / L,  M = max_c m_c,  L = sum_c e^{m_c - M} l_c
__global__ __launch_bounds__(256) void xattn_combine_kernel(const float* __restrict__ opart, const float* __restrict__ mlpart, int nsplit, int Sq, int H,
                                                            int64_t nrows, bf16_t* __restrict__ o, float* __restrict__ lse) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + w;
  if (row >= nrows) return;
  const int64_t prob = row / Sq; const int qi = (int)(row - prob * Sq);
  const int64_t seq = prob / H; const int h = (int)(prob - seq * H);
  const float* ml = mlpart + (prob * nsplit * Sq + qi) * 2;  // chunk c at + c * Sq * 2
  float M = -__builtin_inff();
  for (int c = lane; c < nsplit; c += 64) M = fmaxf(M, ml[(int64_t)c * Sq * 2]);
#pragma unroll
  for (int o_ = 32; o_ > 0; o_ >>= 1) M = fmaxf(M, __shfl_xor(M, o_, 64));
  float L = 0.f;
  for (int c = lane; c < nsplit; c += 64) L += ml[(int64_t)c * Sq * 2 + 1] * __expf(ml[(int64_t)c * Sq * 2] - M);
  L = wave_sum(L);
  float a0 = 0.f, a1 = 0.f;
  const float* op = opart + (prob * nsplit * Sq + qi) * DH;
  for (int c = 0; c < nsplit; ++c) {
    const float wgt = __expf(ml[(int64_t)c * Sq * 2] - M);
    const float* oc = op + (int64_t)c * Sq * DH;
    a0 += wgt * oc[lane];
    if (lane < DH - 64) a1 += wgt * oc[64 + lane];
  }
  const float inv = 1.f / L;
  bf16_t* orow = o + (seq * Sq + qi) * (int64_t)(H * DH) + h * DH;
  orow[lane] = f2bf(a0 * inv);
  if (lane < DH - 64) orow[64 + lane] = f2bf(a1 * inv);
  if (lse && lane == 0) { float* lp = lse + ((seq * Sq) * H + (int64_t)h * Sq + qi) * 2; lp[0] = M; lp[1] = __logf(L); }
}

// backward tail of the cross attention: dq^ = sum over key chunks of the partials, then the per-head RMSNorm backward (attention.py:166)
// and the q-scale gradient.  Grid-stride over rows so that the scale gradient costs 96 atomics per workgroup, not per row.
__global__ __launch_bounds__(256) void xattn_dq_finish_kernel(const float* __restrict__ dqpart, int nsplit, int Sq, int H, int64_t nrows,
                                                              const bf16_t* __restrict__ q, int64_t ldq, const float* __restrict__ sq,
                                                              bf16_t* __restrict__ dq, float* __restrict__ dsq) {
  __shared__ float red[4][DH];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const bool hi = lane < DH - 64;
  const float sc0 = sq[lane], sc1 = hi ? sq[64 + lane] : 0.f;
  float a0 = 0.f, a1 = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < nrows; row += (int64_t)gridDim.x * 4) {
    const int64_t prob = row / Sq; const int qi = (int)(row - prob * Sq);
    const int64_t seq = prob / H; const int h = (int)(prob - seq * H);
    const float* dp = dqpart + (prob * nsplit * Sq + qi) * DH;
    float g0 = 0.f, g1 = 0.f;
    for (int c = 0; c < nsplit; ++c) { const float* dc = dp + (int64_t)c * Sq * DH; g0 += dc[lane]; if (hi) g1 += dc[64 + lane]; }
    const bf16_t* xp = q + (seq * Sq + qi) * ldq + h * DH;
    float x0 = bf2f(xp[lane]), x1 = hi ? bf2f(xp[64 + lane]) : 0.f;
    const float rr = rsqrtf(wave_sum(x0 * x0 + x1 * x1) / DH + 1e-6f);
    x0 *= rr; x1 *= rr;
    const float gx = wave_sum(g0 * sc0 * x0 + g1 * sc1 * x1) / DH;
    bf16_t* drow = dq + (seq * Sq + qi) * ldq + h * DH;
    drow[lane] = f2bf(rr * (g0 * sc0 - x0 * gx));
    if (hi) drow[64 + lane] = f2bf(rr * (g1 * sc1 - x1 * gx));
    a0 += g0 * x0; a1 += g1 * x1;
  }
  red[w][lane] = a0; if (hi) red[w][64 + lane] = a1;
  __syncthreads();
  if (tid < DH) grad_add(dsq + tid, red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
}

static bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

static bool xattn_fwd(spa3d_ctx* c, const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ldq, int64_t ldk, int64_t ldv, const float* sq,
                      const float* sk, const float* km, int64_t nseq, int Sq, int Sk, int H, bf16_t* o, float* lse, const int32_t* seq_off) {
  const int nsplit = (Sk + XCHUNK - 1) / XCHUNK;
  const int64_t nprob = nseq * H;
  if (seq_off || Sq < 1 || Sq > XCHUNK || Sk < 1 || nprob * nsplit > 0x7fffffffLL) return false;
  if (ldq % 8 || ldk % 8 || ldv % 8 || !al16(q) || !al16(k) || !al16(v) || !al16(o)) return false;
  const int64_t mk = c->ar.mark();
  XAttnArgs a; a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.sq = sq; a.sk = sk; a.km = km;
  a.Sq = Sq; a.Sk = Sk; a.H = H; a.nsplit = nsplit; a.nprob = nprob;
  a.opart = (float*)c->ar.alloc(nprob * nsplit * Sq * DH * (int64_t)sizeof(float));
  a.mlpart = (float*)c->ar.alloc(nprob * nsplit * Sq * 2 * (int64_t)sizeof(float));
  if (!c->dry) {
    ProfScope ps(c, PROF_ATTN_FWD, 4.0 * (double)nprob * Sq * Sk * DH, (double)nprob * (2.0 * Sq + 2.0 * Sk) * DH * 2.0);
    ps.tag(nseq, Sk, H, Sq);
    const int lds = 2 * img_bytes(XCHUNK) + XCHUNK * 4;
    xattn_fwd_kernel<<<(unsigned)(nprob * nsplit), 256, lds, c->stream>>>(a);
    const int64_t nrows = nprob * Sq;
    xattn_combine_kernel<<<(unsigned)((nrows + 3) / 4), 256, 0, c->stream>>>(a.opart, a.mlpart, nsplit, Sq, H, nrows, o, lse);
    SPA_LAUNCH_CHECK(c);
  }
  c->ar.release(mk);  // stream-ordered: whatever reuses the space is launched behind the two kernels
  return true;
}

constexpr int ATTN_MAX_S = 320;  // K^ and V of one head resident: 2 x 320 x 192 B = 120 KiB of the CU's 160 KiB

// returns false when the shape is outside what the fused kernels cover (the caller then composes the generic kernels)
bool attn_fused_fwd_bf16(spa3d_ctx* c, const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ldq, int64_t ldk, int64_t ldv,
                         const float* sq, const float* sk, const float* km, int64_t nseq, int Sq, int Sk, int H, int Dh, bf16_t* o,
                         float* lse, const int32_t* seq_off, int64_t total_rows) {
  if (Dh == DH && Sq != Sk) return xattn_fwd(c, q, k, v, ldq, ldk, ldv, sq, sk, km, nseq, Sq, Sk, H, o, lse, seq_off);
  if (Dh != DH || Sq != Sk || Sk < 2 || Sk > ATTN_MAX_S) return false;
  if (ldq % 8 || ldk % 8 || ldv % 8 || !al16(q) || !al16(k) || !al16(v) || !al16(o)) return false;
  if (nseq * H > 0x7fffffffLL) return false;
  if (c->dry) return true;
  AttnArgs a; a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.sq = sq; a.sk = sk; a.km = km;
  a.S = Sk; a.H = H; a.nprob = nseq * H; a.o = o; a.lse = lse; a.seq_off = seq_off;
  const double rows = total_rows > 0 ? (double)total_rows : (double)nseq * Sk;  // ragged: rows actually present
  int KT = (Sk + 15) / 16;        // key / query tiles of 16; odd counts pair up (the last pair half empty) except nine (S = 129, the readout stack: its own instance)
  if ((KT & 1) && KT != 9) KT += 1;
  ProfScope ps(c, PROF_ATTN_FWD, 4.0 * rows * H * (rows / nseq) * Dh, rows * H * Dh * 2.0 * 4.0);
  ps.tag(nseq, Sk, H, 0);
  switch (KT) {
    case 2: launch_fwd<2, 4>(c, a); break;
    case 4: launch_fwd<4, 4>(c, a); break;
    case 6: launch_fwd<6, 4>(c, a); break;
    case 8: launch_fwd<8, 4>(c, a); break;
    case 9: launch_fwd<9, 4>(c, a); break;
    case 10: launch_fwd<10, 4>(c, a); break;
    case 12: launch_fwd<12, 4>(c, a); break;
    case 14: launch_fwd<14, 8>(c, a); break;   // > 80 KiB of LDS: one workgroup per CU, so eight waves
    case 16: launch_fwd<16, 8>(c, a); break;
    case 18: launch_fwd<18, 8>(c, a); break;
    case 20: launch_fwd<20, 8>(c, a); break;
    default: return false;
  }
  SPA_LAUNCH_CHECK(c);
  return true;
}


// =================================================================================================================
// backward.  dq, dk, dv (through the per-head RMSNorms) and the two RMSNorm scale gradients.
//
// Both orientations of the score tile are recomputed so that no partial sum ever crosses a wave:
//   (a) a wave owns 16-query tiles:  S^T, dP^T over all keys (key in registers, query on the lane)
//         -> dS^T packs straight into the B operand of dQ^T[d][q] = K^^T[d][keys] dS^T[keys][q]        needs K^, V in LDS
//   (b) a wave owns 16-key tiles:    S, dP over all queries (query in registers, key on the lane)
//         -> P, dS pack straight into the B operands of dV^T[d][key] = dO^T[d][q] P[q][key] and dK^^T = Q^^T dS   needs Q^, dO in LDS
// Transposed A operands (K^^T, dO^T, Q^^T) are read from the row-major LDS images with ds_read_b64_tr_b16.
// P is rebuilt from the forward's (row max, log row sum); delta = rowsum(dO o O).
//
// Two workgroup structures over the same two tile routines:
//   attn_bwd8_kernel     (S <= 160)  all four images resident (4 x 31 KiB at S = 151); waves 0-3 run (a) while waves 4-7 run (b)
//                                    on the shared images: two waves per SIMD, compute = max(a, b); one workgroup per CU.
//   attn_bwd_split_kernel (S <= 320) two passes per problem over TWO images: pass A stages K^, V and every wave takes query
//                                    tiles (its own q^/dO fragments come straight from global memory), pass B re-uses the
//                                    same LDS for Q^, dO and every wave takes key tiles.  Half the LDS: at S = 151 two
//                                    workgroups fit a CU, so one's staging runs under the other's MFMA phase; at S = 301
//                                    (BASELINE cfg#5) it is what makes a fused backward possible at all (4 images = 240 KiB).
// =================================================================================================================
struct AttnBwdArgs {
  const bf16_t *q, *k, *v, *o, *d_o; int64_t ldq, ldk, ldv;
  const float *sq, *sk, *km, *lse;
  int S, H; int64_t nprob;
  bf16_t *dq, *dk, *dv; float *dsq, *dsk;
  const int32_t* seq_off;  // as AttnArgs::seq_off
  // cross attention (attn_bwd8_kernel<8, true>): S = query rows per sequence (<= 128); the Sk keys of a sequence are cut into nsplit chunks
  // of 128, one workgroup pass per (sequence, head, chunk); dq^ (before the RMSNorm backward) leaves as fp32 partials per chunk
  int Sk, nsplit; float* dqpart;  // [nprob][nsplit][S][96]
#if SPA3D_ABL_ATTN  // tools/ablate_attn.py builds a separate diagnostic library with this; never defined for libspa3d_hip.so
  int ablate;          // 1: no tile work, 2: no staging (garbage operands), 4: no dq/dk/dv stores
#endif
};

// (a) one 16-query tile against all keys.  qb / dob: this lane's query row as B-operand fragments (q^ normalised); mq, lq, dq_:
// the row's (max, log-sum, delta); xraw: the raw q row in the accumulator layout (d = 16dt + 4fq + r) for the RMSNorm backward.
// X (cross attention): the tile's dQ^ rows go out as fp32 partials (dqp: row 0 of the tile, 96 floats per row) -- the sum over key chunks and
// the RMSNorm backward happen in xattn_dq_finish_kernel; xraw / wt / gtile / ds_acc are unused.
// FAST (no key mask, self attention): the row constants ride in the accumulators' initial value -- mq = -(m + l) / alpha, dq_ = -delta -- so a score
// element costs fma + exp2 + mul + half a conversion instead of ~13 VALU operations, and alpha is applied once to the fp32 dQ^ sums.  Padding
// keys (kbias = -inf) still give P = 0 exactly.  With a mask the finfo.min semantics (fully-masked rows attend uniformly) need the reference's order.
template <int KT, bool X = false, bool FAST = false>
__device__ __forceinline__ void bwd_query_tile(const char* Ks, const char* Vs, const float* kbias, const float* scq,
                                               const mfma16x8 (&qb)[3], const mfma16x8 (&dob)[3], float mq, float lq, float dq_,
                                               const u16x4 (&xraw)[6], bool valid, char* wt, bf16_t* gtile, int64_t ld_, int nrows,
                                               float (&ds_acc)[6][4], float* dqp = nullptr
#if SPA3D_ABL_ATTN
                                               , int ablate = 0
#endif
                                               ) {
  const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
  const float alpha = 0.10206207261596575f;  // 1/sqrt(96)
  constexpr int NPAIR = (KT + 1) / 2;  // odd KT (nine tiles at S = 129): the last pair's second key tile does not exist -- zeros on both MFMA operands
  mfma16x8 dsb[NPAIR];
#pragma unroll
  for (int s2 = 0; s2 < NPAIR; ++s2) {
#if SPA3D_ABL_ATTN
    if (ablate & 8) {  // what a dS tile handed over through LDS would cost this role: one 16-byte read per key-tile pair, no S / dP work
      dsb[s2] = *(const mfma16x8*)(Vs + 2 * s2 * ROW16 + row_off(fr) + fq * 16);
      continue;
    }
#endif
    u16x8 t;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int kt = 2 * s2 + hf;
      if (kt >= KT) {  // compile-time after unrolling
#pragma unroll
        for (int r = 0; r < 4; ++r) t[hf * 4 + r] = 0;
        continue;
      }
      f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (FAST) { st = f32x4{mq, mq, mq, mq}; dpt = f32x4{dq_, dq_, dq_, dq_}; }
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const mfma16x8 kf = *(const mfma16x8*)(Ks + kt * ROW16 + row_off(fr) + (s * 32 + fq * 8) * 2);
        const mfma16x8 vf = *(const mfma16x8*)(Vs + kt * ROW16 + row_off(fr) + (s * 32 + fq * 8) * 2);
        st = MFMA16(kf, qb[s], st);
        dpt = MFMA16(vf, dob[s], dpt);
      }
      const f32x4 b4 = *(const f32x4*)(kbias + kt * 16 + fq * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (FAST) {
          const float p = __builtin_amdgcn_exp2f(fmaf(st[r], 0.10206207261596575f * 1.4426950408889634f, b4[r]));  // b4: 0 or -inf
          t[hf * 4 + r] = f2bf(p * dpt[r]);
        } else {
          const float p = __expf((st[r] * alpha + b4[r] - mq) - lq);
          const float ds = (b4[r] == 0.f) ? p * (dpt[r] - dq_) * alpha : 0.f;  // where() passes no gradient to masked logits
          t[hf * 4 + r] = f2bf(ds);
        }
      }
    }
    dsb[s2] = __builtin_bit_cast(mfma16x8, t);
    __builtin_amdgcn_sched_barrier(0);  // keep the unrolled key-tile pairs apart: interleaving them costs ~60 live registers
  }
  // dQ^^T[d][q] = sum_keys K^^T[d][key] dS^T[key][q]  (already times alpha)
  f32x4 dqa[6];
  const char* kbase = Ks + tp * 8 + row_off(4 * fq + tq);
  static_for<0, 6>([&](auto dtc) {
    constexpr int dt = decltype(dtc)::value;
    dqa[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint2 lo[NPAIR], hi[NPAIR];
    static_for<0, NPAIR>([&](auto sc_) {
      constexpr int s2 = decltype(sc_)::value;
      lo[s2] = lds_tr16_b64_o<dt * 32 + 2 * s2 * ROW16>(kbase);
      if constexpr (2 * s2 + 1 < KT) hi[s2] = lds_tr16_b64_o<dt * 32 + (2 * s2 + 1) * ROW16>(kbase); else hi[s2] = make_uint2(0u, 0u);
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < NPAIR; ++s2) {
      const uint4 u = make_uint4(lo[s2].x, lo[s2].y, hi[s2].x, hi[s2].y);
      dqa[dt] = MFMA16(__builtin_bit_cast(mfma16x8, u), dsb[s2], dqa[dt]);
    }
  });
  if constexpr (X) {
    if (fr < nrows) {
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) *(f32x4*)(dqp + fr * DH + dt * 16 + fq * 4) = dqa[dt];
    }
    return;
  }
  // RMSNorm backward for query fr: lane holds d = 16dt + 4fq + r
  float x[6][4]; float ss = 0.f;
#pragma unroll
  for (int dt = 0; dt < 6; ++dt) {
    const u16x4 xv = xraw[dt];
#pragma unroll
    for (int r = 0; r < 4; ++r) { x[dt][r] = bf2f(xv[r]); ss += x[dt][r] * x[dt][r]; if constexpr (FAST) dqa[dt][r] *= alpha; }
  }
  ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
  const float rr = rsqrtf(ss / DH + 1e-6f);
  float gx = 0.f;
#pragma unroll
  for (int dt = 0; dt < 6; ++dt) {
    const f32x4 sc = *(const f32x4*)(scq + dt * 16 + fq * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) { x[dt][r] *= rr; gx += dqa[dt][r] * sc[r] * x[dt][r]; }
  }
  gx += __shfl_xor(gx, 16, 64); gx += __shfl_xor(gx, 32, 64);
  gx /= DH;
#pragma unroll
  for (int dt = 0; dt < 6; ++dt) {
    const f32x4 sc = *(const f32x4*)(scq + dt * 16 + fq * 4);
    u16x4 o4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o4[r] = f2bf(rr * (dqa[dt][r] * sc[r] - x[dt][r] * gx));
      if (valid) ds_acc[dt][r] += dqa[dt][r] * x[dt][r];
    }
    tile_put(wt, dt, o4);
  }
  tile_flush(wt, gtile, ld_, nrows);  // rows past the sequence end are never written
}

// (b) one 16-key tile against all queries.  kb / vb: this lane's key row as B-operand fragments (k^ normalised, v raw); kbv: its
// key bias; xraw: the raw k row in the accumulator layout.
template <int KT, bool FAST = false>  // FAST: as bwd_query_tile -- mrow holds -(m + l) / alpha, drow holds -delta, lrow is unused
__device__ __forceinline__ void bwd_key_tile(const char* Qs, const char* dOs, const float* mrow, const float* lrow,
                                             const float* drow, const float* sck, const mfma16x8 (&kb)[3], const mfma16x8 (&vb)[3],
                                             float kbv, const u16x4 (&xraw)[6], bool valid, char* wt, bf16_t* gk, int64_t ldk_, bf16_t* gv,
                                             int64_t ldv_, int nrows, float (&ds_acc)[6][4]) {
  const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
  const float alpha = 0.10206207261596575f;
  const bool keep = kbv == 0.f;
  f32x4 dva[6], dka[6];
#pragma unroll
  for (int dt = 0; dt < 6; ++dt) { dva[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dka[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  // one pair of query tiles (2 s2, 2 s2 + 1); SECOND = false: the last pair of an odd KT (nine tiles at S = 129) holds one tile -- zeros on both MFMA operands for the other
  auto pair_step = [&](int s2, auto second_) {
    constexpr bool SECOND = decltype(second_)::value;
    u16x8 tp_, tds;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int qt = 2 * s2 + hf;
      if (!SECOND && hf == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { tp_[4 + r] = 0; tds[4 + r] = 0; }
        continue;
      }
      f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (FAST) { st = *(const f32x4*)(mrow + qt * 16 + fq * 4); dpt = *(const f32x4*)(drow + qt * 16 + fq * 4); }
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const mfma16x8 qf = *(const mfma16x8*)(Qs + qt * ROW16 + row_off(fr) + (s * 32 + fq * 8) * 2);
        const mfma16x8 df = *(const mfma16x8*)(dOs + qt * ROW16 + row_off(fr) + (s * 32 + fq * 8) * 2);
        st = MFMA16(qf, kb[s], st);     // S[q = 16qt+4fq+r][key = k0+fr]
        dpt = MFMA16(df, vb[s], dpt);   // dP[q][key]
      }
      if constexpr (FAST) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(st[r], 0.10206207261596575f * 1.4426950408889634f, kbv));  // kbv: 0 or -inf (padding key)
          tp_[hf * 4 + r] = f2bf(p);
          tds[hf * 4 + r] = f2bf(p * dpt[r]);
        }
      } else {
        const f32x4 m4 = *(const f32x4*)(mrow + qt * 16 + fq * 4);
        const f32x4 l4 = *(const f32x4*)(lrow + qt * 16 + fq * 4);
        const f32x4 d4 = *(const f32x4*)(drow + qt * 16 + fq * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __expf((st[r] * alpha + kbv - m4[r]) - l4[r]);
          tp_[hf * 4 + r] = f2bf(p);
          tds[hf * 4 + r] = f2bf(keep ? p * (dpt[r] - d4[r]) * alpha : 0.f);
        }
      }
    }
    const mfma16x8 pb = __builtin_bit_cast(mfma16x8, tp_), dsb = __builtin_bit_cast(mfma16x8, tds);
    // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^^T[d][key] += Q^^T[d][q] dS[q][key]
    uint2 olo[6], ohi[6], qlo[6], qhi[6];
    const int roff = 2 * s2 * ROW16 + row_off(4 * fq + tq) + tp * 8;
    const char* ob = dOs + roff; const char* qb_ = Qs + roff;
    static_for<0, 6>([&](auto dtc) {
      constexpr int dt = decltype(dtc)::value;
      olo[dt] = lds_tr16_b64_o<dt * 32>(ob); qlo[dt] = lds_tr16_b64_o<dt * 32>(qb_);
      if constexpr (SECOND) { ohi[dt] = lds_tr16_b64_o<ROW16 + dt * 32>(ob); qhi[dt] = lds_tr16_b64_o<ROW16 + dt * 32>(qb_); }
      else { ohi[dt] = make_uint2(0u, 0u); qhi[dt] = make_uint2(0u, 0u); }
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
  };
#pragma unroll 1
  for (int s2 = 0; s2 < KT / 2; ++s2) pair_step(s2, std::true_type{});
  if constexpr (KT & 1) pair_step(KT / 2, std::false_type{});
  // lane: key = k0 + fr, d = 16dt + 4fq + r
  float x[6][4]; float ss = 0.f;
#pragma unroll
  for (int dt = 0; dt < 6; ++dt) {
    const u16x4 xv = xraw[dt];
#pragma unroll
    for (int r = 0; r < 4; ++r) { x[dt][r] = bf2f(xv[r]); ss += x[dt][r] * x[dt][r]; if constexpr (FAST) dka[dt][r] *= alpha; }
  }
  ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
  const float rr = rsqrtf(ss / DH + 1e-6f);
  float gx = 0.f;
#pragma unroll
  for (int dt = 0; dt < 6; ++dt) {
    const f32x4 sc = *(const f32x4*)(sck + dt * 16 + fq * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) { x[dt][r] *= rr; gx += dka[dt][r] * sc[r] * x[dt][r]; }
  }
  gx += __shfl_xor(gx, 16, 64); gx += __shfl_xor(gx, 32, 64);
  gx /= DH;
#pragma unroll
  for (int dt = 0; dt < 6; ++dt) {
    const f32x4 sc = *(const f32x4*)(sck + dt * 16 + fq * 4);
    u16x4 k4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      k4[r] = f2bf(rr * (dka[dt][r] * sc[r] - x[dt][r] * gx));
      if (valid) ds_acc[dt][r] += dka[dt][r] * x[dt][r];
    }
    tile_put(wt, dt, k4);
  }
  tile_flush(wt, gk, ldk_, nrows);
#pragma unroll
  for (int dt = 0; dt < 6; ++dt) {
    u16x4 v4;
#pragma unroll
    for (int r = 0; r < 4; ++r) v4[r] = f2bf(dva[dt][r]);
    tile_put(wt, dt, v4);
  }
  tile_flush(wt, gv, ldv_, nrows);
}

// dO rows + delta = rowsum(dO o O) into LDS
template <int NP, int RPP, bool NEG = false>
__device__ __forceinline__ void store_do_delta(const RawRows<NP>& xd, const RawRows<NP>& xo, int S_pad, char* dOs, float* drow, int tid = threadIdx.x) {
  const int part = tid & 3, r0 = tid >> 2;
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int row = r0 + RPP * ps;
    float dsum = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) dsum += bf2f(xd.x[ps][c][j]) * bf2f(xo.x[ps][c][j]);
    dsum += __shfl_xor(dsum, 1, 64); dsum += __shfl_xor(dsum, 2, 64);
    if (row < S_pad) {
      u16x8* d = (u16x8*)(dOs + row_off(row)) + part;
      d[0] = xd.x[ps][0]; d[4] = xd.x[ps][1]; d[8] = xd.x[ps][2];
      if (part == 0) drow[row] = NEG ? -dsum : dsum;
    }
  }
}

// the RMSNorm scale gradients of a workgroup: lanes with equal fq hold the same d -> reduce over fr, then over waves through LDS
template <int NTHREADS>
__device__ __forceinline__ void flush_scale_grads(const AttnBwdArgs& g, float* sred, const float (&acc_q)[6][4], const float (&acc_k)[6][4],
                                                  bool has_q, bool has_k) {
  const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
  __syncthreads();
  for (int t = tid; t < 2 * DH; t += NTHREADS) sred[t] = 0.f;
  __syncthreads();
#pragma unroll
  for (int dt = 0; dt < 6; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = acc_q[dt][r], b = acc_k[dt][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
      if (fr == 0) {
        if (det_on()) {  // deterministic mode: LDS float atomics depend on arrival order too -- every wave adds straight into the fixed-point shadow
          if (has_q) grad_add(g.dsq + dt * 16 + fq * 4 + r, a);
          if (has_k) grad_add(g.dsk + dt * 16 + fq * 4 + r, b);
        } else {
          if (has_q) atomicAdd(sred + dt * 16 + fq * 4 + r, a);
          if (has_k) atomicAdd(sred + DH + dt * 16 + fq * 4 + r, b);
        }
      }
    }
  __syncthreads();
  if (det_on()) return;
  if (tid < DH) atomicAdd(g.dsq + tid, sred[tid]);
  else if (tid < 2 * DH) atomicAdd(g.dsk + tid - DH, sred[tid]);
}

// 8-wave form, four resident images: waves 0-3 take the query tiles (dQ) while waves 4-7 take the key tiles (dK, dV) of the same
// problem: two waves per SIMD to overlap LDS/exp latency with the other's MFMAs, compute = max(a, b) instead of a + b.
// X: cross attention -- a "problem" is (sequence, head, 128-key chunk): query-side rows (q, dO, O, lse: g.S per sequence) and key-side rows
// (k, v: the chunk) come from different places, role (a) emits dq^ partials instead of dq (see bwd_query_tile).
template <int KT, bool X = false, bool FAST = false>  // FAST: no key mask (g.km == nullptr), self attention -- see bwd_query_tile
__global__ __launch_bounds__(512, 2) void attn_bwd8_kernel(AttnBwdArgs g) {
  static_assert(!(X && FAST), "the cross-attention form keeps the masked arithmetic");
  constexpr int S_pad = KT * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem; char* Ks = Qs + img_bytes(S_pad); char* Vs = Ks + img_bytes(S_pad); char* dOs = Vs + img_bytes(S_pad);
  float* kbias = (float*)(dOs + img_bytes(S_pad)); float* mrow = kbias + S_pad; float* lrow = mrow + S_pad; float* drow = lrow + S_pad;
  float* sred = drow + S_pad;     // [2][96] scale-gradient staging
  float* sscale = sred + 2 * DH;  // [2][96] RMSNorm scales (LDS copies: as loop invariants in registers they cost 48 VGPRs)
  char* wt = (char*)(sscale + 2 * DH) + (threadIdx.x >> 6) * WTILE;  // wave-private output tile
  const int E = g.H * DH;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, role = wv >> 2, w = wv & 3, fr = lane & 15, fq = lane >> 4;
  float ds_acc[6][4];  // role 0: d scale_q, role 1: d scale_k
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) ds_acc[i][r] = 0.f;

  for (int t = tid; t < 2 * DH; t += 512) sscale[t] = t < DH ? g.sq[t] : g.sk[t - DH];
  __syncthreads();
  const int64_t nseq = (unsigned)g.nprob / (unsigned)g.H;
  const int64_t nwork = X ? g.nprob * g.nsplit : g.nprob;
  for (int64_t pi = blockIdx.x; pi < nwork; pi += gridDim.x) {
    // (32-bit index arithmetic, one or two divisions: the launchers refuse nprob * nsplit >= 2^31)
    unsigned seq_u, h_u, prob_u; int sp = 0;
    if constexpr (X) { prob_u = (unsigned)pi / (unsigned)g.nsplit; sp = (int)((unsigned)pi - prob_u * (unsigned)g.nsplit); seq_u = prob_u / (unsigned)g.H; h_u = prob_u - seq_u * (unsigned)g.H; }
    else { map_prob32((unsigned)pi, (unsigned)nseq, (unsigned)g.H, seq_u, h_u); prob_u = seq_u * (unsigned)g.H + h_u; }
    const int64_t prob = prob_u, seq = seq_u; const int h = (int)h_u;
    const int64_t row0 = X ? seq * g.S : g.seq_off ? (int64_t)g.seq_off[seq] : seq * g.S;   // first query-side row
    const int S = (!X && g.seq_off) ? g.seq_off[seq + 1] - (int)row0 : g.S;               // query-side rows
    const int64_t krow0 = X ? seq * g.Sk + (int64_t)sp * S_pad : row0;                   // first key-side row
    const int Sk = X ? min(S_pad, g.Sk - sp * S_pad) : S;                                 // key-side rows
    const int QT = (S + 15) / 16, QTk = (Sk + 15) / 16;
    constexpr int NP = (S_pad + 127) / 128;
    // all five matrices of the problem are requested before the first is used: one memory latency per problem
    RawRows<NP> rq, rk, rv, rd, ro;
#if SPA3D_ABL_ATTN
    if (!(g.ablate & 2)) {
#endif
    const int tid_o = opaque_tid();  // per-problem copy: the row offsets below are recomputed, not hoisted out of the loop and spilled (rows_load)
    rows_load<NP, 128>(rq, g.q + row0 * g.ldq + h * DH, g.ldq, S, tid_o);
    rows_load<NP, 128>(rk, g.k + krow0 * g.ldk + h * DH, g.ldk, Sk, tid_o);
    rows_load<NP, 128>(rv, g.v + krow0 * g.ldv + h * DH, g.ldv, Sk, tid_o);
    rows_load<NP, 128>(rd, g.d_o + row0 * E + h * DH, E, S, tid_o);
    rows_load<NP, 128>(ro, g.o + row0 * E + h * DH, E, S, tid_o);
    __syncthreads();  // previous problem's LDS reads are done
    rows_store<true, NP, 128>(rq, S_pad, sscale, Qs);
    rows_store<true, NP, 128>(rk, S_pad, sscale + DH, Ks);
    rows_store<false, NP, 128>(rv, S_pad, nullptr, Vs);
    store_do_delta<NP, 128, FAST>(rd, ro, S_pad, dOs, drow);  // FAST: -delta
#if SPA3D_ABL_ATTN
    }
#endif
    for (int t = tid; t < S_pad; t += 512) {
      float b = 0.f, m = 0.f, ll = __builtin_inff();  // padding query: P = exp(.. - inf) = 0
      if (t >= Sk) b = -__builtin_inff();
      else if (g.km && g.km[krow0 + t] == 0.f) b = NEG_BIG;
      if (t < S) { m = g.lse[(row0 * g.H + (int64_t)h * S + t) * 2]; ll = g.lse[(row0 * g.H + (int64_t)h * S + t) * 2 + 1]; }
      kbias[t] = b;
      if constexpr (FAST) mrow[t] = -(m + ll) * 9.797958971132712f;  // -(m + l) / alpha (padding query: -inf -> P = 0)
      else { mrow[t] = m; lrow[t] = ll; }
    }
    __syncthreads();
#if SPA3D_ABL_ATTN
    if (g.ablate & 1) continue;
#endif

    if (role == 0) {  // ------------------------------------------------------------------ (a) query tiles -> dq
      for (int qt = w; qt < QT; qt += 4) {
        const int q0 = qt * 16;
        // raw q row of this lane's query for the RMSNorm backward: requested NOW so the HBM/L2 latency hides under the MFMAs
        int qrow = q0 + fr; const bool valid = qrow < S; if (!valid) qrow = S - 1;
        u16x4 xraw[6];
        if constexpr (!X) {
          const bf16_t* xp = g.q + (row0 + qrow) * g.ldq + h * DH + fq * 4;
#pragma unroll
          for (int dt = 0; dt < 6; ++dt) xraw[dt] = *(const u16x4*)(xp + dt * 16);
        }
        mfma16x8 qb[3], dob[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          qb[s] = *(const mfma16x8*)(Qs + qt * ROW16 + row_off(fr) + (s * 32 + fq * 8) * 2);
          dob[s] = *(const mfma16x8*)(dOs + qt * ROW16 + row_off(fr) + (s * 32 + fq * 8) * 2);
        }
#if SPA3D_ABL_ATTN
        const int nrows_st = (g.ablate & 4) ? 0 : S - q0;
#else
        const int nrows_st = S - q0;
#endif
        bwd_query_tile<KT, X, FAST>(Ks, Vs, kbias, sscale, qb, dob, mrow[q0 + fr], lrow[q0 + fr], drow[q0 + fr], xraw, valid, wt,
                              g.dq + (row0 + q0) * g.ldq + h * DH, g.ldq, nrows_st, ds_acc,
                              X ? g.dqpart + ((pi * g.S) + q0) * DH : nullptr
#if SPA3D_ABL_ATTN
                              , g.ablate
#endif
                              );
      }
    } else {          // ------------------------------------------------------------------ (b) key tiles -> dk, dv
      for (int kt = 3 - w; kt < QTk; kt += 4) {  // real key tiles only; dealt in the reverse wave order of role (a): with 9 or 10 tiles the SIMDs whose dQ wave has three tiles get a dK/dV wave with two
        const int k0 = kt * 16;
        int krow = k0 + fr; const bool valid = krow < Sk; if (!valid) krow = Sk - 1;
        u16x4 xraw[6];  // raw k row for the RMSNorm backward, requested before the MFMA work (see part (a))
        {
          const bf16_t* xp = g.k + (krow0 + krow) * g.ldk + h * DH + fq * 4;
#pragma unroll
          for (int dt = 0; dt < 6; ++dt) xraw[dt] = *(const u16x4*)(xp + dt * 16);
        }
        mfma16x8 kb[3], vb[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          kb[s] = *(const mfma16x8*)(Ks + kt * ROW16 + row_off(fr) + (s * 32 + fq * 8) * 2);
          vb[s] = *(const mfma16x8*)(Vs + kt * ROW16 + row_off(fr) + (s * 32 + fq * 8) * 2);
        }
#if SPA3D_ABL_ATTN
        const int nrows_st = (g.ablate & 4) ? 0 : Sk - k0;
#else
        const int nrows_st = Sk - k0;
#endif
        bwd_key_tile<KT, FAST>(Qs, dOs, mrow, lrow, drow, sscale + DH, kb, vb, kbias[k0 + fr], xraw, valid, wt, g.dk + (krow0 + k0) * g.ldk + h * DH,
                               g.ldk, g.dv + (krow0 + k0) * g.ldv + h * DH, g.ldv, nrows_st, ds_acc);
      }
    }
  }
  flush_scale_grads<512>(g, sred, ds_acc, ds_acc, !X && role == 0, role == 1);
}

template <int KT, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_split_kernel(AttnBwdArgs g) {
  constexpr int S_pad = KT * 16, NTH = NW * 64, RPP = NW * 16, NP = (S_pad + RPP - 1) / RPP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* B0 = smem; char* B1 = B0 + img_bytes(S_pad);  // pass A: K^, V ; pass B: Q^, dO
  float* kbias = (float*)(B1 + img_bytes(S_pad)); float* mrow = kbias + S_pad; float* lrow = mrow + S_pad; float* drow = lrow + S_pad;
  float* sred = drow + S_pad; float* sscale = sred + 2 * DH;
  char* wtile = (char*)(sscale + 2 * DH);  // wave-private tiles: own-row staging in pass A, output rows in both passes
  const int E = g.H * DH;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, fq = lane >> 4;
  float dsq_acc[6][4], dsk_acc[6][4];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { dsq_acc[i][r] = 0.f; dsk_acc[i][r] = 0.f; }
  for (int t = tid; t < 2 * DH; t += NTH) sscale[t] = t < DH ? g.sq[t] : g.sk[t - DH];
  __syncthreads();
  const int64_t nseq = g.nprob / g.H;
  for (int64_t pi = blockIdx.x; pi < g.nprob; pi += gridDim.x) {
    unsigned seq_u, h_u; map_prob32((unsigned)pi, (unsigned)nseq, (unsigned)g.H, seq_u, h_u);
    const int64_t seq = seq_u; const int h = (int)h_u;
    const int64_t row0 = g.seq_off ? (int64_t)g.seq_off[seq] : seq * g.S;
    const int S = g.seq_off ? g.seq_off[seq + 1] - (int)row0 : g.S;
    const int QT = (S + 15) / 16;
    // ================================================================== pass A: K^, V resident; query tiles -> dq
    {
      RawRows<NP> rk, rv;
      rows_load<NP, RPP>(rk, g.k + row0 * g.ldk + h * DH, g.ldk, S);
      rows_load<NP, RPP>(rv, g.v + row0 * g.ldv + h * DH, g.ldv, S);
      __syncthreads();  // previous problem's pass-B reads are done
      rows_store<true, NP, RPP>(rk, S_pad, sscale + DH, B0);
      rows_store<false, NP, RPP>(rv, S_pad, nullptr, B1);
      for (int t = tid; t < S_pad; t += NTH) {
        float b = 0.f;
        if (t >= S) b = -__builtin_inff();
        else if (g.km && g.km[row0 + t] == 0.f) b = NEG_BIG;
        kbias[t] = b;
      }
      __syncthreads();
    }
    for (int qt = w; qt < QT; qt += NW) {
      const int q0 = qt * 16;
      int qrow = q0 + fr; const bool valid = qrow < S; if (!valid) qrow = S - 1;
      // The wave's own 16 query rows go through a wave-private 3-KiB LDS tile, one matrix at a time (q^, then dO): whole 192-B
      // rows are loaded coalesced (4 lanes per row), normalised exactly as the staged images are, and read back as B-operand
      // fragments.  (Fragment-shaped global loads + in-register normalisation cost ~390 spilled VGPRs here.)
      const int tr = lane >> 2, part = lane & 3;
      const int64_t grow = row0 + (q0 + tr < S ? q0 + tr : S - 1);
      u16x8 xq[3], xd[3], xo[3]; u16x4 xraw[6];
      {
        const u16x8* pq = (const u16x8*)(g.q + grow * g.ldq + h * DH) + part;
        const u16x8* pd = (const u16x8*)(g.d_o + grow * E + h * DH) + part;
        const u16x8* po = (const u16x8*)(g.o + grow * E + h * DH) + part;
#pragma unroll
        for (int c = 0; c < 3; ++c) { xq[c] = pq[4 * c]; xd[c] = pd[4 * c]; xo[c] = po[4 * c]; }
        const bf16_t* xp = g.q + (row0 + qrow) * g.ldq + h * DH + fq * 4;
#pragma unroll
        for (int dt = 0; dt < 6; ++dt) xraw[dt] = *(const u16x4*)(xp + dt * 16);
      }
      float mq = 0.f, lq = __builtin_inff();  // padding query: P = exp(.. - inf) = 0
      if (valid) { mq = g.lse[(row0 * g.H + (int64_t)h * S + qrow) * 2]; lq = g.lse[(row0 * g.H + (int64_t)h * S + qrow) * 2 + 1]; }
      char* wt = wtile + w * WTILE; float* wdel = (float*)(wt + 16 * WROW);
      mfma16x8 qb[3], dob[3];
      {  // q^ rows -> tile -> fragments
        float f[24]; float ss = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int j = 0; j < 8; ++j) { f[c * 8 + j] = bf2f(xq[c][j]); ss += f[c * 8 + j] * f[c * 8 + j]; }
        ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64);
        const float rr = rsqrtf(ss / DH + 1e-6f);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int j = 0; j < 8; ++j) xq[c][j] = f2bf(f[c * 8 + j] * rr * sscale[(part + 4 * c) * 8 + j]);
        u16x8* d = (u16x8*)(wt + tr * WROW) + part;
        d[0] = xq[0]; d[4] = xq[1]; d[8] = xq[2];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < 3; ++s) qb[s] = *(const mfma16x8*)(wt + fr * WROW + (s * 32 + fq * 8) * 2);
        asm volatile("" ::: "memory");
      }
      float dsum = 0.f;
      {  // dO rows (+ delta) -> the same tile -> fragments
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int j = 0; j < 8; ++j) dsum += bf2f(xd[c][j]) * bf2f(xo[c][j]);
        dsum += __shfl_xor(dsum, 1, 64); dsum += __shfl_xor(dsum, 2, 64);
        u16x8* d = (u16x8*)(wt + tr * WROW) + part;
        d[0] = xd[0]; d[4] = xd[1]; d[8] = xd[2];
        if (part == 0) wdel[tr] = dsum;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < 3; ++s) dob[s] = *(const mfma16x8*)(wt + fr * WROW + (s * 32 + fq * 8) * 2);
        dsum = wdel[fr];
        asm volatile("" ::: "memory");
      }
      bwd_query_tile<KT>(B0, B1, kbias, sscale, qb, dob, mq, lq, dsum, xraw, valid, wt, g.dq + (row0 + q0) * g.ldq + h * DH, g.ldq, S - q0,
                         dsq_acc);
    }
    // ================================================================== pass B: Q^, dO resident; key tiles -> dk, dv
    {
      RawRows<NP> rq, rd, ro;
      rows_load<NP, RPP>(rq, g.q + row0 * g.ldq + h * DH, g.ldq, S);
      rows_load<NP, RPP>(rd, g.d_o + row0 * E + h * DH, E, S);
      rows_load<NP, RPP>(ro, g.o + row0 * E + h * DH, E, S);
      __syncthreads();  // pass A's reads of K^, V are done
      rows_store<true, NP, RPP>(rq, S_pad, sscale, B0);
      store_do_delta<NP, RPP>(rd, ro, S_pad, B1, drow);
      for (int t = tid; t < S_pad; t += NTH) {
        float m = 0.f, ll = __builtin_inff();
        if (t < S) { m = g.lse[(row0 * g.H + (int64_t)h * S + t) * 2]; ll = g.lse[(row0 * g.H + (int64_t)h * S + t) * 2 + 1]; }
        mrow[t] = m; lrow[t] = ll;
      }
      __syncthreads();
    }
    for (int kt = w; kt < QT; kt += NW) {
      const int k0 = kt * 16;
      int krow = k0 + fr; const bool valid = krow < S; if (!valid) krow = S - 1;
      const bf16_t* kp = g.k + (row0 + krow) * g.ldk + h * DH;
      const bf16_t* vp = g.v + (row0 + krow) * g.ldv + h * DH + fq * 8;
      u16x8 kx[3], vx[3]; u16x4 xraw[6];
#pragma unroll
      for (int s = 0; s < 3; ++s) { kx[s] = *(const u16x8*)(kp + fq * 8 + s * 32); vx[s] = *(const u16x8*)(vp + s * 32); }
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) xraw[dt] = *(const u16x4*)(kp + fq * 4 + dt * 16);
      mfma16x8 kb[3], vb[3];
      frag_norm(kx, sscale + DH, fq, kb);
#pragma unroll
      for (int s = 0; s < 3; ++s) vb[s] = __builtin_bit_cast(mfma16x8, vx[s]);
      const float kbv = kbias[k0 + fr];  // written in pass A, untouched since
      bwd_key_tile<KT>(B0, B1, mrow, lrow, drow, sscale + DH, kb, vb, kbv, xraw, valid, wtile + w * WTILE, g.dk + (row0 + k0) * g.ldk + h * DH,
                       g.ldk, g.dv + (row0 + k0) * g.ldv + h * DH, g.ldv, S - k0, dsk_acc);
    }
  }
  flush_scale_grads<NTH>(g, sred, dsq_acc, dsk_acc, true, true);
}

template <int KT>
static void launch_bwd(spa3d_ctx* c, const AttnBwdArgs& a) {
  constexpr int S_pad = KT * 16;
  const int small = 4 * S_pad * 4 + 4 * DH * 4;
  const int lds4 = 4 * img_bytes(S_pad) + small + 8 * WTILE;
  auto lds2 = [&](int nw) { return 2 * img_bytes(S_pad) + small + nw * WTILE; };
  // mode 1: four resident images, concurrent roles (S <= 160); 2: split-pass, 4 waves, two workgroups per CU; 3: split-pass, 8 waves
  int mode = c->attn_bwd_mode;
  if (S_pad > 160) mode = 3;  // four images + wave tiles fit 160 KiB up to S = 160
  else if (mode == 0) mode = 1;
  static bool attr_set = false;
  if (!attr_set) {
    if constexpr (S_pad <= 160) {
      (void)hipFuncSetAttribute((const void*)attn_bwd8_kernel<KT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
      (void)hipFuncSetAttribute((const void*)attn_bwd_split_kernel<KT, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2(4));
    }
    (void)hipFuncSetAttribute((const void*)attn_bwd_split_kernel<KT, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2(8));
    attr_set = true;
  }
  if constexpr (S_pad <= 160) {
    if (mode == 1) {
      static bool attrf = false;
      if (!attrf) { (void)hipFuncSetAttribute((const void*)attn_bwd8_kernel<KT, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4); attrf = true; }
      if (a.km) attn_bwd8_kernel<KT><<<(unsigned)std::min<int64_t>(a.nprob, 1024), 512, lds4, c->stream>>>(a);
      else attn_bwd8_kernel<KT, false, true><<<(unsigned)std::min<int64_t>(a.nprob, 1024), 512, lds4, c->stream>>>(a);  // no key mask: cheap score arithmetic
      return;
    }
    if (mode == 2) { attn_bwd_split_kernel<KT, 4><<<(unsigned)std::min<int64_t>(a.nprob, 2048), 256, lds2(4), c->stream>>>(a); return; }
  }
  attn_bwd_split_kernel<KT, 8><<<(unsigned)std::min<int64_t>(a.nprob, 1024), 512, lds2(8), c->stream>>>(a);
}

bool attn_fused_bwd_bf16(spa3d_ctx* c, const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ldq, int64_t ldk, int64_t ldv,
                         const float* sq, const float* sk, const float* km, int64_t nseq, int Sq, int Sk, int H, int Dh, const bf16_t* o,
                         const float* lse, const bf16_t* d_o, bf16_t* dq, bf16_t* dk, bf16_t* dv, float* dsq, float* dsk,
                         const int32_t* seq_off, int64_t total_rows) {
  if (Dh == DH && Sq != Sk) {  // cross attention: chunked keys, see xattn_fwd_kernel
    const int nsplit = (Sk + XCHUNK - 1) / XCHUNK;
    const int64_t nprob = nseq * H;
    if (seq_off || !o || !lse || Sq < 1 || Sq > XCHUNK || Sk < 1 || nprob * nsplit > 0x7fffffffLL) return false;
    if (ldq % 8 || ldk % 8 || ldv % 8 || !al16(q) || !al16(k) || !al16(v) || !al16(o) || !al16(d_o) || !al16(dq) || !al16(dk) || !al16(dv) ||
        !al16(sq) || !al16(sk))
      return false;
    const int64_t mk = c->ar.mark();
    float* dqpart = (float*)c->ar.alloc(nprob * nsplit * Sq * DH * (int64_t)sizeof(float));
    if (!c->dry) {
      AttnBwdArgs a; a.q = q; a.k = k; a.v = v; a.o = o; a.d_o = d_o; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.sq = sq; a.sk = sk; a.km = km;
      a.lse = lse; a.S = Sq; a.H = H; a.nprob = nprob; a.dq = dq; a.dk = dk; a.dv = dv; a.dsq = dsq; a.dsk = dsk; a.seq_off = nullptr;
      a.Sk = Sk; a.nsplit = nsplit; a.dqpart = dqpart;
#if SPA3D_ABL_ATTN
      a.ablate = 0;
#endif
      ProfScope ps(c, PROF_ATTN_BWD, 14.0 * (double)nprob * Sq * Sk * DH, (double)nprob * (4.0 * Sq + 4.0 * Sk) * DH * 2.0);
      ps.tag(nseq, Sk, H, Sq);
      constexpr int KT = XCHUNK / 16;
      const int lds4 = 4 * img_bytes(XCHUNK) + 4 * XCHUNK * 4 + 4 * DH * 4 + 8 * WTILE;
      static bool attr_set = false;
      if (!attr_set) { (void)hipFuncSetAttribute((const void*)attn_bwd8_kernel<KT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4); attr_set = true; }
      attn_bwd8_kernel<KT, true><<<(unsigned)std::min<int64_t>(nprob * nsplit, 1024), 512, lds4, c->stream>>>(a);
      const int64_t nrows = nprob * Sq;
      xattn_dq_finish_kernel<<<(unsigned)std::min<int64_t>((nrows + 3) / 4, 512), 256, 0, c->stream>>>(dqpart, nsplit, Sq, H, nrows, q, ldq, sq, dq, dsq);
      SPA_LAUNCH_CHECK(c);
    }
    c->ar.release(mk);
    return true;
  }
  if (Dh != DH || Sq != Sk || Sk < 2 || Sk > ATTN_MAX_S || !o || !lse) return false;
  if (ldq % 8 || ldk % 8 || ldv % 8 || !al16(q) || !al16(k) || !al16(v) || !al16(o) || !al16(d_o) || !al16(dq) || !al16(dk) || !al16(dv) ||
      !al16(sq) || !al16(sk))
    return false;
  if (nseq * H > 0x7fffffffLL) return false;   // the kernels index problems in 32 bits
  if (c->dry) return true;
  AttnBwdArgs a; a.q = q; a.k = k; a.v = v; a.o = o; a.d_o = d_o; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.sq = sq; a.sk = sk; a.km = km;
  a.lse = lse; a.S = Sk; a.H = H; a.nprob = nseq * H; a.dq = dq; a.dk = dk; a.dv = dv; a.dsq = dsq; a.dsk = dsk; a.seq_off = seq_off;
  a.Sk = Sk; a.nsplit = 1; a.dqpart = nullptr;
  const double rows = total_rows > 0 ? (double)total_rows : (double)nseq * Sk;
#if SPA3D_ABL_ATTN
  { const char* e = getenv("SPA3D_ABLATE"); a.ablate = e ? atoi(e) : 0; }
#endif
  // even tile counts only: the tile routines take an odd KT (the last pair half empty, as in the forward's nine-tile instance), but at S = 129 the backward
  // measured the same with nine tiles as with ten (3.43 vs 3.44-3.6 ms: its third ROUND of tiles, not the tenth key tile, is what S = 129 pays for there)
  const int KT = ((Sk + 31) / 32) * 2;
  ProfScope ps(c, PROF_ATTN_BWD, 14.0 * rows * H * (rows / nseq) * Dh, rows * H * Dh * 2.0 * 8.0);
  ps.tag(nseq, Sk, H, 0);
  switch (KT) {
    case 2: launch_bwd<2>(c, a); break;
    case 4: launch_bwd<4>(c, a); break;
    case 6: launch_bwd<6>(c, a); break;
    case 8: launch_bwd<8>(c, a); break;
    case 10: launch_bwd<10>(c, a); break;
    case 12: launch_bwd<12>(c, a); break;
    case 14: launch_bwd<14>(c, a); break;
    case 16: launch_bwd<16>(c, a); break;
    case 18: launch_bwd<18>(c, a); break;
    case 20: launch_bwd<20>(c, a); break;
    default: return false;
  }
  SPA_LAUNCH_CHECK(c);
  return true;
}
SPA_DET_UPLOAD_DEF(det_upload_attn)
}  // namespace SPA_NS
