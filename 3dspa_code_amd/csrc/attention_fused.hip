// attention_fused.hip -- LDS-resident short-sequence attention for bf16 (gfx950): the temporal self-attention of
// the track encoder (S = T+1 = 151) and the readout stack (S = 129), d_head = 96 (attention.py:166-175).
//
// One workgroup (4 waves) per (sequence, head).  The whole normalised K and the V of the head (S_pad x 96 bf16
// = 30 KiB each) are staged into LDS once; RMSNorm of q/k (attention.py:166-167), the 1/sqrt(d) scale, the key
// mask, the softmax and both contractions are fused, so per token only q,k,v are read and o (+ a 4-byte LSE per
// head) written.  No online-softmax streaming: a full row of scores lives in registers.
//
// Forward, per 16-query tile (tiles are dealt round-robin to the 4 waves), everything "transposed" so that the
// softmax reductions stay inside a lane plus two xor-shuffles and P never leaves registers:
//   S^T[key][q] = mfma(A = K^[key][:], B = Q^[q][:])          C-layout: lane (fr,fq) holds keys 16t+4fq+r of query fr
//   P^T        = exp(S^T + kbias - max) ; l = sum             in-lane over (t,r), then lanes fq via shfl_xor 16,32
//   O^T[d][q]  = mfma(A = V^T[d][keys], B = P^T[keys][q])     B operand = the S^T accumulators re-packed to bf16 with the
//                                                             k index permuted (cdna_hip_programming.md section 3 "An accumulator
//                                                             tile as the next MFMA's operand"); A operand = V read from its
//                                                             row-major LDS image with ds_read_b64_tr_b16 using the same k order.
//   lane ends with 4 consecutive d of one query -> 8-byte stores.
#include "common.hpp"
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

#define NEG_BIG (-3.4028234663852886e38f)

struct AttnArgs {
  const bf16_t *q, *k, *v; int64_t ldq, ldk, ldv;
  const float *sq, *sk, *km;
  int S, H; int64_t nprob;
  bf16_t* o; float* lse;  // o [nseq*S][H*96]; lse [nseq][H][S][2] = (row max, log row sum) (may be null)
};

__device__ __forceinline__ uint2 lds_tr16_b64(const void* p) {
  uint2 v;
  const unsigned a = (unsigned)(uintptr_t)p;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(a) : "memory");
  return v;
}

// the same read with a compile-time byte offset in the instruction: without it every distinct address is a loop-invariant VGPR
// (the 8-wave backward spilled ~60 of them)
template <int OFF>
__device__ __forceinline__ uint2 lds_tr16_b64_o(const void* p) {
  static_assert(OFF >= 0 && OFF < 65536, "16-bit DS offset");
  uint2 v;
  const unsigned a = (unsigned)(uintptr_t)p;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
  return v;
}
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

constexpr int DH = 96, ROWB = DH * 2;  // 192-byte LDS rows

// stage S rows of [*, 96] bf16 (row stride ld elements) into LDS rows of 192 B; 4 threads per row, 24 elements each.
// NORM: per-row RMSNorm * scale (attention.py:167) before the bf16 round, as the unfused path stores it.
template <bool NORM, int NP, int RPP = 64>
__device__ __forceinline__ void stage_rows(const bf16_t* __restrict__ src, int64_t ld_, int S, int S_pad, const float* __restrict__ scale,
                                           char* lds) {
  const int part = threadIdx.x & 3, r0 = threadIdx.x >> 2;
  u16x8 x[NP][3];
  // every global load of this matrix is issued before the first use: one memory latency per matrix, not one per 64 rows
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int row = r0 + RPP * ps;
    if (row < S) {
      const u16x8* p = (const u16x8*)(src + (int64_t)row * ld_ + part * 24);
      x[ps][0] = p[0]; x[ps][1] = p[1]; x[ps][2] = p[2];
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) x[ps][c][j] = 0;
    }
  }
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int row = r0 + RPP * ps;
    if (NORM) {
      float f[24]; float ss = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) { f[c * 8 + j] = bf2f(x[ps][c][j]); ss += f[c * 8 + j] * f[c * 8 + j]; }
      ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64);
      const float r = rsqrtf(ss / DH + 1e-6f);
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) x[ps][c][j] = f2bf(f[c * 8 + j] * r * scale[part * 24 + c * 8 + j]);
    }
    if (row < S_pad) {
      u16x8* d = (u16x8*)(lds + row * ROWB + part * 48);
      d[0] = x[ps][0]; d[1] = x[ps][1]; d[2] = x[ps][2];
    }
  }
}

template <int KT>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs g) {
  constexpr int S_pad = KT * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem; char* Vs = smem + S_pad * ROWB; float* kbias = (float*)(smem + 2 * S_pad * ROWB);
  const int64_t prob = blockIdx.x;
  const int64_t seq = prob / g.H; const int h = (int)(prob - seq * g.H);
  const int S = g.S, E = g.H * DH;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, fq = lane >> 4;

  constexpr int NP = (S_pad + 63) / 64;
  stage_rows<true, NP>(g.k + seq * S * g.ldk + h * DH, g.ldk, S, S_pad, g.sk, Ks);
  stage_rows<false, NP>(g.v + seq * S * g.ldv + h * DH, g.ldv, S, S_pad, nullptr, Vs);
  for (int t = tid; t < S_pad; t += 256) {
    float b = 0.f;
    if (t >= S) b = -__builtin_inff();                           // padding key: weight exactly 0
    else if (g.km && g.km[seq * S + t] == 0.f) b = NEG_BIG;      // where(mask, logit, finfo.min)
    kbias[t] = b;
  }
  __syncthreads();

  const int QT = (S + 15) / 16;
  const float qscale = 0.10206207261596575f;  // 1/sqrt(96)
  for (int qt = w; qt < QT; qt += 4) {
    const int q0 = qt * 16;
    int qrow = q0 + fr; if (qrow > S - 1) qrow = S - 1;
    // ---- Q fragment: B operand, lane (query fr, dh = 32s + 8fq + j); RMSNorm over the 96 dh = 3 s x 4 fq lanes
    const bf16_t* qp = g.q + (seq * S + qrow) * g.ldq + h * DH + fq * 8;
    u16x8 qx[3]; float qf[24]; float ss = 0.f;
#pragma unroll
    for (int s = 0; s < 3; ++s) qx[s] = *(const u16x8*)(qp + s * 32);
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) { qf[s * 8 + j] = bf2f(qx[s][j]); ss += qf[s * 8 + j] * qf[s * 8 + j]; }
    ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
    const float rq = rsqrtf(ss / DH + 1e-6f);
    bf16x8 qb[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      u16x8 t;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // same rounding points as the unfused path: bf16(q^ = x r scale), then the 1/sqrt(d) scale in fp32 on the logits
        t[j] = f2bf(qf[s * 8 + j] * rq * g.sq[s * 32 + fq * 8 + j]);
      }
      qb[s] = __builtin_bit_cast(bf16x8, t);
    }
    // ---- S^T tiles
    f32x4 acc[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const bf16x8 kf = *(const bf16x8*)(Ks + (kt * 16 + fr) * ROWB + (s * 32 + fq * 8) * 2);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qb[s], acc[kt], 0, 0, 0);
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
    if (g.lse && fq == 0 && q0 + fr < S) { float* lp = g.lse + (prob * S + q0 + fr) * 2; lp[0] = m; lp[1] = __logf(l); }
    // ---- P^T as B operands: k-step s2 covers key tiles 2*s2, 2*s2+1; element j <-> key 16*(2*s2 + (j>>2)) + 4*fq + (j&3)
    bf16x8 pb[KT / 2];
#pragma unroll
    for (int s2 = 0; s2 < KT / 2; ++s2) {
      u16x8 t;
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = f2bf(acc[2 * s2 + (j >> 2)][j & 3] * inv);
      pb[s2] = __builtin_bit_cast(bf16x8, t);
    }
    // ---- O^T[d][q] = sum_keys V^T[d][key] P^T[key][q]
    const int tq = fr >> 2, tp = fr & 3;  // this lane's slot in its 16-lane transposed-read group
#pragma unroll
    for (int dt = 0; dt < 6; ++dt) {
      f32x4 oacc = f32x4{0.f, 0.f, 0.f, 0.f};
      uint2 lo[KT / 2], hi[KT / 2];
      const char* base = Vs + (dt * 16 + tp * 4) * 2 + (4 * fq + tq) * ROWB;
#pragma unroll
      for (int s2 = 0; s2 < KT / 2; ++s2) {
        lo[s2] = lds_tr16_b64(base + (32 * s2) * ROWB);
        hi[s2] = lds_tr16_b64(base + (32 * s2 + 16) * ROWB);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);  // MFMAs must stay below the wait (cdna_hip_programming.md rule 18)
#pragma unroll
      for (int s2 = 0; s2 < KT / 2; ++s2) {
        const uint4 vu = make_uint4(lo[s2].x, lo[s2].y, hi[s2].x, hi[s2].y);
        oacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vu), pb[s2], oacc, 0, 0, 0);
      }
      if (q0 + fr < S) {
        u16x4 o4;
#pragma unroll
        for (int r = 0; r < 4; ++r) o4[r] = f2bf(oacc[r]);
        *(u16x4*)(g.o + (seq * S + q0 + fr) * E + h * DH + dt * 16 + fq * 4) = o4;
      }
    }
  }
}

template <int KT>
static void launch_fwd(spa3d_ctx* c, const AttnArgs& a) {
  const int lds = 2 * KT * 16 * ROWB + KT * 16 * 4;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<KT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
  attn_fwd_kernel<KT><<<(unsigned)a.nprob, 256, lds, c->stream>>>(a);
}

static bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// returns false when the shape is outside what the fused kernels cover (the caller then composes the generic kernels)
bool attn_fused_fwd_bf16(spa3d_ctx* c, const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ldq, int64_t ldk, int64_t ldv,
                         const float* sq, const float* sk, const float* km, int64_t nseq, int Sq, int Sk, int H, int Dh, bf16_t* o,
                         float* lse) {
  if (Dh != DH || Sq != Sk || Sk < 2 || Sk > 192) return false;
  if (ldq % 8 || ldk % 8 || ldv % 8 || !al16(q) || !al16(k) || !al16(v) || !al16(o)) return false;
  if (nseq * H > 0x7fffffffLL) return false;
  if (c->dry) return true;
  AttnArgs a; a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.sq = sq; a.sk = sk; a.km = km;
  a.S = Sk; a.H = H; a.nprob = nseq * H; a.o = o; a.lse = lse;
  const int KT = ((Sk + 31) / 32) * 2;
  ProfScope ps(c, PROF_ATTN_FWD, 4.0 * (double)Sq * Sk * Dh * (double)a.nprob, (double)a.nprob * Sq * Dh * 2.0 * 4.0);
  switch (KT) {
    case 2: launch_fwd<2>(c, a); break;
    case 4: launch_fwd<4>(c, a); break;
    case 6: launch_fwd<6>(c, a); break;
    case 8: launch_fwd<8>(c, a); break;
    case 10: launch_fwd<10>(c, a); break;
    case 12: launch_fwd<12>(c, a); break;
    default: return false;
  }
  SPA_LAUNCH_CHECK(c);
  return true;
}


// =================================================================================================================
// backward.  dq, dk, dv (through the per-head RMSNorms) and the two RMSNorm scale gradients.
//
// Persistent workgroups (4 waves) loop over (sequence, head) problems with Q^, K^, V, dO of the head resident in LDS
// (4 x 30 KiB).  Both orientations of the score tile are recomputed so that no partial sum ever crosses a wave:
//   (a) a wave owns 16-query tiles:  S^T, dP^T over all keys (key in registers, query on the lane)
//         -> dS^T packs straight into the B operand of dQ^T[d][q] = K^^T[d][keys] dS^T[keys][q]
//   (b) a wave owns 16-key tiles:    S, dP over all queries (query in registers, key on the lane)
//         -> P, dS pack straight into the B operands of dV^T[d][key] = dO^T[d][q] P[q][key] and dK^^T = Q^^T dS
// Transposed A operands (K^^T, dO^T, Q^^T) are read from the row-major LDS images with ds_read_b64_tr_b16.
// P is rebuilt from the forward's (row max, log row sum); delta = rowsum(dO o O) is formed while staging dO.
// =================================================================================================================
struct AttnBwdArgs {
  const bf16_t *q, *k, *v, *o, *d_o; int64_t ldq, ldk, ldv;
  const float *sq, *sk, *km, *lse;
  int S, H; int64_t nprob;
  bf16_t *dq, *dk, *dv; float *dsq, *dsk;
};

template <int KT>
__global__ __launch_bounds__(256, 1) void attn_bwd_kernel(AttnBwdArgs g) {
  constexpr int S_pad = KT * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem; char* Ks = Qs + S_pad * ROWB; char* Vs = Ks + S_pad * ROWB; char* dOs = Vs + S_pad * ROWB;
  float* kbias = (float*)(dOs + S_pad * ROWB); float* mrow = kbias + S_pad; float* lrow = mrow + S_pad; float* drow = lrow + S_pad;
  float* sred = drow + S_pad;  // [2][96] scale-gradient staging
  const int S = g.S, E = g.H * DH;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int tq = fr >> 2, tp = fr & 3;
  const float alpha = 0.10206207261596575f;  // 1/sqrt(96)
  const int QT = (S + 15) / 16;
  float dsq_acc[6][4], dsk_acc[6][4];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { dsq_acc[i][r] = 0.f; dsk_acc[i][r] = 0.f; }

  for (int64_t prob = blockIdx.x; prob < g.nprob; prob += gridDim.x) {
    const int64_t seq = prob / g.H; const int h = (int)(prob - seq * g.H);
    __syncthreads();  // previous problem's LDS reads are done
    constexpr int NP = (S_pad + 63) / 64;
    stage_rows<true, NP>(g.q + seq * S * g.ldq + h * DH, g.ldq, S, S_pad, g.sq, Qs);
    stage_rows<true, NP>(g.k + seq * S * g.ldk + h * DH, g.ldk, S, S_pad, g.sk, Ks);
    stage_rows<false, NP>(g.v + seq * S * g.ldv + h * DH, g.ldv, S, S_pad, nullptr, Vs);
    {  // dO rows + delta = rowsum(dO o O); all loads first
      const int part = tid & 3, r0 = tid >> 2;
      u16x8 xd[NP][3], xo[NP][3];
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
        const int row = r0 + 64 * ps;
        if (row < S) {
          const u16x8* p = (const u16x8*)(g.d_o + (seq * S + row) * E + h * DH + part * 24);
          const u16x8* po = (const u16x8*)(g.o + (seq * S + row) * E + h * DH + part * 24);
#pragma unroll
          for (int c = 0; c < 3; ++c) { xd[ps][c] = p[c]; xo[ps][c] = po[c]; }
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) { xd[ps][c][j] = 0; xo[ps][c][j] = 0; }
        }
      }
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
        const int row = r0 + 64 * ps;
        float dsum = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int j = 0; j < 8; ++j) dsum += bf2f(xd[ps][c][j]) * bf2f(xo[ps][c][j]);
        dsum += __shfl_xor(dsum, 1, 64); dsum += __shfl_xor(dsum, 2, 64);
        if (row < S_pad) {
          u16x8* d = (u16x8*)(dOs + row * ROWB + part * 48);
          d[0] = xd[ps][0]; d[1] = xd[ps][1]; d[2] = xd[ps][2];
          if (part == 0) drow[row] = dsum;
        }
      }
    }
    for (int t = tid; t < S_pad; t += 256) {
      float b = 0.f, m = 0.f, ll = __builtin_inff();  // padding query: P = exp(.. - inf) = 0
      if (t >= S) b = -__builtin_inff();
      else {
        if (g.km && g.km[seq * S + t] == 0.f) b = NEG_BIG;
        m = g.lse[(prob * S + t) * 2]; ll = g.lse[(prob * S + t) * 2 + 1];
      }
      kbias[t] = b; mrow[t] = m; lrow[t] = ll;
    }
    __syncthreads();

    // ------------------------------------------------------------------ (a) query tiles -> dq
    for (int qt = w; qt < QT; qt += 4) {
      const int q0 = qt * 16;
      // raw q row of this lane's query for the RMSNorm backward: requested NOW so the HBM/L2 latency hides under the MFMAs
      // (the asm "memory" clobbers below pin loads where they are written)
      int qrow = q0 + fr; const bool valid = qrow < S; if (!valid) qrow = S - 1;
      u16x4 xraw[6];
      {
        const bf16_t* xp = g.q + (seq * S + qrow) * g.ldq + h * DH + fq * 4;
#pragma unroll
        for (int dt = 0; dt < 6; ++dt) xraw[dt] = *(const u16x4*)(xp + dt * 16);
      }
      bf16x8 qb[3], dob[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        qb[s] = *(const bf16x8*)(Qs + (q0 + fr) * ROWB + (s * 32 + fq * 8) * 2);
        dob[s] = *(const bf16x8*)(dOs + (q0 + fr) * ROWB + (s * 32 + fq * 8) * 2);
      }
      const float mq = mrow[q0 + fr], lq = lrow[q0 + fr], dq_ = drow[q0 + fr];
      bf16x8 dsb[KT / 2];
#pragma unroll
      for (int s2 = 0; s2 < KT / 2; ++s2) {
        u16x8 t;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int kt = 2 * s2 + hf;
          f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const bf16x8 kf = *(const bf16x8*)(Ks + (kt * 16 + fr) * ROWB + (s * 32 + fq * 8) * 2);
            const bf16x8 vf = *(const bf16x8*)(Vs + (kt * 16 + fr) * ROWB + (s * 32 + fq * 8) * 2);
            st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qb[s], st, 0, 0, 0);
            dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dob[s], dpt, 0, 0, 0);
          }
          const f32x4 b4 = *(const f32x4*)(kbias + kt * 16 + fq * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __expf((st[r] * alpha + b4[r] - mq) - lq);
            const float ds = (b4[r] == 0.f) ? p * (dpt[r] - dq_) * alpha : 0.f;  // where() passes no gradient to masked logits
            t[hf * 4 + r] = f2bf(ds);
          }
        }
        dsb[s2] = __builtin_bit_cast(bf16x8, t);
      }
      // dQ^^T[d][q] = sum_keys K^^T[d][key] dS^T[key][q]  (already times alpha)
      f32x4 dqa[6];
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) {
        dqa[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        uint2 lo[KT / 2], hi[KT / 2];
        const char* base = Ks + (dt * 16 + tp * 4) * 2 + (4 * fq + tq) * ROWB;
#pragma unroll
        for (int s2 = 0; s2 < KT / 2; ++s2) { lo[s2] = lds_tr16_b64(base + (32 * s2) * ROWB); hi[s2] = lds_tr16_b64(base + (32 * s2 + 16) * ROWB); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s2 = 0; s2 < KT / 2; ++s2) {
          const uint4 u = make_uint4(lo[s2].x, lo[s2].y, hi[s2].x, hi[s2].y);
          dqa[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, u), dsb[s2], dqa[dt], 0, 0, 0);
        }
      }
      // RMSNorm backward for query fr: lane holds d = 16dt + 4fq + r
      float x[6][4]; float ss = 0.f;
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) {
        const u16x4 xv = xraw[dt];
#pragma unroll
        for (int r = 0; r < 4; ++r) { x[dt][r] = bf2f(xv[r]); ss += x[dt][r] * x[dt][r]; }
      }
      ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
      const float rr = rsqrtf(ss / DH + 1e-6f);
      float gx = 0.f;
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) {
        const f32x4 sc = *(const f32x4*)(g.sq + dt * 16 + fq * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { x[dt][r] *= rr; gx += dqa[dt][r] * sc[r] * x[dt][r]; }
      }
      gx += __shfl_xor(gx, 16, 64); gx += __shfl_xor(gx, 32, 64);
      gx /= DH;
      if (valid) {
        bf16_t* op = g.dq + (seq * S + qrow) * g.ldq + h * DH + fq * 4;
#pragma unroll
        for (int dt = 0; dt < 6; ++dt) {
          const f32x4 sc = *(const f32x4*)(g.sq + dt * 16 + fq * 4);
          u16x4 o4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o4[r] = f2bf(rr * (dqa[dt][r] * sc[r] - x[dt][r] * gx));
            dsq_acc[dt][r] += dqa[dt][r] * x[dt][r];
          }
          *(u16x4*)(op + dt * 16) = o4;
        }
      }
    }

    // ------------------------------------------------------------------ (b) key tiles -> dk, dv
    for (int kt = w; kt < QT; kt += 4) {  // real key tiles only (S_q == S_k)
      const int k0 = kt * 16;
      int krow = k0 + fr; const bool valid = krow < S; if (!valid) krow = S - 1;
      u16x4 xraw[6];  // raw k row for the RMSNorm backward, requested before the MFMA work (see part (a))
      {
        const bf16_t* xp = g.k + (seq * S + krow) * g.ldk + h * DH + fq * 4;
#pragma unroll
        for (int dt = 0; dt < 6; ++dt) xraw[dt] = *(const u16x4*)(xp + dt * 16);
      }
      bf16x8 kb[3], vb[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        kb[s] = *(const bf16x8*)(Ks + (k0 + fr) * ROWB + (s * 32 + fq * 8) * 2);
        vb[s] = *(const bf16x8*)(Vs + (k0 + fr) * ROWB + (s * 32 + fq * 8) * 2);
      }
      const float kbv = kbias[k0 + fr];
      const bool keep = kbv == 0.f;
      f32x4 dva[6], dka[6];
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) { dva[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dka[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
      for (int s2 = 0; s2 < KT / 2; ++s2) {
        u16x8 tp_, tds;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int qt = 2 * s2 + hf;
          f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const bf16x8 qf = *(const bf16x8*)(Qs + (qt * 16 + fr) * ROWB + (s * 32 + fq * 8) * 2);
            const bf16x8 df = *(const bf16x8*)(dOs + (qt * 16 + fr) * ROWB + (s * 32 + fq * 8) * 2);
            st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kb[s], st, 0, 0, 0);     // S[q = 16qt+4fq+r][key = k0+fr]
            dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, vb[s], dpt, 0, 0, 0);   // dP[q][key]
          }
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
        const bf16x8 pb = __builtin_bit_cast(bf16x8, tp_), dsb = __builtin_bit_cast(bf16x8, tds);
        // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^^T[d][key] += Q^^T[d][q] dS[q][key]
        uint2 olo[6], ohi[6], qlo[6], qhi[6];
        const int roff = (32 * s2 + 4 * fq + tq) * ROWB + tp * 8;
#pragma unroll
        for (int dt = 0; dt < 6; ++dt) {
          olo[dt] = lds_tr16_b64(dOs + roff + dt * 32); ohi[dt] = lds_tr16_b64(dOs + roff + 16 * ROWB + dt * 32);
          qlo[dt] = lds_tr16_b64(Qs + roff + dt * 32); qhi[dt] = lds_tr16_b64(Qs + roff + 16 * ROWB + dt * 32);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < 6; ++dt) {
          const uint4 uo = make_uint4(olo[dt].x, olo[dt].y, ohi[dt].x, ohi[dt].y);
          const uint4 uq = make_uint4(qlo[dt].x, qlo[dt].y, qhi[dt].x, qhi[dt].y);
          dva[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, uo), pb, dva[dt], 0, 0, 0);
          dka[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, uq), dsb, dka[dt], 0, 0, 0);
        }
      }
      // lane: key = k0 + fr, d = 16dt + 4fq + r
      float x[6][4]; float ss = 0.f;
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) {
        const u16x4 xv = xraw[dt];
#pragma unroll
        for (int r = 0; r < 4; ++r) { x[dt][r] = bf2f(xv[r]); ss += x[dt][r] * x[dt][r]; }
      }
      ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
      const float rr = rsqrtf(ss / DH + 1e-6f);
      float gx = 0.f;
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) {
        const f32x4 sc = *(const f32x4*)(g.sk + dt * 16 + fq * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { x[dt][r] *= rr; gx += dka[dt][r] * sc[r] * x[dt][r]; }
      }
      gx += __shfl_xor(gx, 16, 64); gx += __shfl_xor(gx, 32, 64);
      gx /= DH;
      if (valid) {
        bf16_t* okp = g.dk + (seq * S + krow) * g.ldk + h * DH + fq * 4;
        bf16_t* ovp = g.dv + (seq * S + krow) * g.ldv + h * DH + fq * 4;
#pragma unroll
        for (int dt = 0; dt < 6; ++dt) {
          const f32x4 sc = *(const f32x4*)(g.sk + dt * 16 + fq * 4);
          u16x4 k4, v4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            k4[r] = f2bf(rr * (dka[dt][r] * sc[r] - x[dt][r] * gx));
            v4[r] = f2bf(dva[dt][r]);
            dsk_acc[dt][r] += dka[dt][r] * x[dt][r];
          }
          *(u16x4*)(okp + dt * 16) = k4;
          *(u16x4*)(ovp + dt * 16) = v4;
        }
      }
    }
  }
  // ---- flush the RMSNorm scale gradients: lanes with equal fq hold the same d -> reduce over fr, then over waves through LDS
  __syncthreads();
  for (int t = tid; t < 2 * DH; t += 256) sred[t] = 0.f;
  __syncthreads();
#pragma unroll
  for (int dt = 0; dt < 6; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = dsq_acc[dt][r], b = dsk_acc[dt][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
      if (fr == 0) { atomicAdd(sred + dt * 16 + fq * 4 + r, a); atomicAdd(sred + DH + dt * 16 + fq * 4 + r, b); }
    }
  __syncthreads();
  if (tid < DH) atomicAdd(g.dsq + tid, sred[tid]);
  else if (tid < 2 * DH) atomicAdd(g.dsk + tid - DH, sred[tid]);
}

// 8-wave form: the two phases only read the shared LDS images, so waves 0-3 take the query tiles (dQ) while waves 4-7 take the key
// tiles (dK, dV) of the same problem: two waves per SIMD to overlap LDS/exp latency with the other's MFMAs, compute = max(a, b)
// instead of a + b, and all eight waves stage.
template <int KT>
__global__ __launch_bounds__(512, 2) void attn_bwd8_kernel(AttnBwdArgs g) {
  constexpr int S_pad = KT * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem; char* Ks = Qs + S_pad * ROWB; char* Vs = Ks + S_pad * ROWB; char* dOs = Vs + S_pad * ROWB;
  float* kbias = (float*)(dOs + S_pad * ROWB); float* mrow = kbias + S_pad; float* lrow = mrow + S_pad; float* drow = lrow + S_pad;
  float* sred = drow + S_pad;  // [2][96] scale-gradient staging
  float* sscale = sred + 2 * DH;  // [2][96] RMSNorm scales (LDS copies: as loop invariants in registers they cost 48 VGPRs)
  const int S = g.S, E = g.H * DH;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, role = wv >> 2, w = wv & 3, fr = lane & 15, fq = lane >> 4;
  const int tq = fr >> 2, tp = fr & 3;
  const float alpha = 0.10206207261596575f;  // 1/sqrt(96)
  const int QT = (S + 15) / 16;
  float ds_acc[6][4];  // role 0: d scale_q, role 1: d scale_k
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) ds_acc[i][r] = 0.f;

  for (int t = tid; t < 2 * DH; t += 512) sscale[t] = t < DH ? g.sq[t] : g.sk[t - DH];  // visible after the first problem's barrier
  for (int64_t prob = blockIdx.x; prob < g.nprob; prob += gridDim.x) {
    const int64_t seq = prob / g.H; const int h = (int)(prob - seq * g.H);
    __syncthreads();  // previous problem's LDS reads are done
    constexpr int NP = (S_pad + 127) / 128;
    stage_rows<true, NP, 128>(g.q + seq * S * g.ldq + h * DH, g.ldq, S, S_pad, g.sq, Qs);
    stage_rows<true, NP, 128>(g.k + seq * S * g.ldk + h * DH, g.ldk, S, S_pad, g.sk, Ks);
    stage_rows<false, NP, 128>(g.v + seq * S * g.ldv + h * DH, g.ldv, S, S_pad, nullptr, Vs);
    {  // dO rows + delta = rowsum(dO o O); all loads first
      const int part = tid & 3, r0 = tid >> 2;
      u16x8 xd[NP][3], xo[NP][3];
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
        const int row = r0 + 128 * ps;
        if (row < S) {
          const u16x8* p = (const u16x8*)(g.d_o + (seq * S + row) * E + h * DH + part * 24);
          const u16x8* po = (const u16x8*)(g.o + (seq * S + row) * E + h * DH + part * 24);
#pragma unroll
          for (int c = 0; c < 3; ++c) { xd[ps][c] = p[c]; xo[ps][c] = po[c]; }
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) { xd[ps][c][j] = 0; xo[ps][c][j] = 0; }
        }
      }
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
        const int row = r0 + 128 * ps;
        float dsum = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int j = 0; j < 8; ++j) dsum += bf2f(xd[ps][c][j]) * bf2f(xo[ps][c][j]);
        dsum += __shfl_xor(dsum, 1, 64); dsum += __shfl_xor(dsum, 2, 64);
        if (row < S_pad) {
          u16x8* d = (u16x8*)(dOs + row * ROWB + part * 48);
          d[0] = xd[ps][0]; d[1] = xd[ps][1]; d[2] = xd[ps][2];
          if (part == 0) drow[row] = dsum;
        }
      }
    }
    for (int t = tid; t < S_pad; t += 512) {
      float b = 0.f, m = 0.f, ll = __builtin_inff();  // padding query: P = exp(.. - inf) = 0
      if (t >= S) b = -__builtin_inff();
      else {
        if (g.km && g.km[seq * S + t] == 0.f) b = NEG_BIG;
        m = g.lse[(prob * S + t) * 2]; ll = g.lse[(prob * S + t) * 2 + 1];
      }
      kbias[t] = b; mrow[t] = m; lrow[t] = ll;
    }
    __syncthreads();

    // ------------------------------------------------------------------ (a) query tiles -> dq
    for (int qt = w; qt < QT && role == 0; qt += 4) {
      const int q0 = qt * 16;
      // raw q row of this lane's query for the RMSNorm backward: requested NOW so the HBM/L2 latency hides under the MFMAs
      // (the asm "memory" clobbers below pin loads where they are written)
      int qrow = q0 + fr; const bool valid = qrow < S; if (!valid) qrow = S - 1;
      u16x4 xraw[6];
      {
        const bf16_t* xp = g.q + (seq * S + qrow) * g.ldq + h * DH + fq * 4;
#pragma unroll
        for (int dt = 0; dt < 6; ++dt) xraw[dt] = *(const u16x4*)(xp + dt * 16);
      }
      bf16x8 qb[3], dob[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        qb[s] = *(const bf16x8*)(Qs + (q0 + fr) * ROWB + (s * 32 + fq * 8) * 2);
        dob[s] = *(const bf16x8*)(dOs + (q0 + fr) * ROWB + (s * 32 + fq * 8) * 2);
      }
      const float mq = mrow[q0 + fr], lq = lrow[q0 + fr], dq_ = drow[q0 + fr];
      bf16x8 dsb[KT / 2];
#pragma unroll
      for (int s2 = 0; s2 < KT / 2; ++s2) {
        u16x8 t;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int kt = 2 * s2 + hf;
          f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const bf16x8 kf = *(const bf16x8*)(Ks + (kt * 16 + fr) * ROWB + (s * 32 + fq * 8) * 2);
            const bf16x8 vf = *(const bf16x8*)(Vs + (kt * 16 + fr) * ROWB + (s * 32 + fq * 8) * 2);
            st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qb[s], st, 0, 0, 0);
            dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dob[s], dpt, 0, 0, 0);
          }
          const f32x4 b4 = *(const f32x4*)(kbias + kt * 16 + fq * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __expf((st[r] * alpha + b4[r] - mq) - lq);
            const float ds = (b4[r] == 0.f) ? p * (dpt[r] - dq_) * alpha : 0.f;  // where() passes no gradient to masked logits
            t[hf * 4 + r] = f2bf(ds);
          }
        }
        dsb[s2] = __builtin_bit_cast(bf16x8, t);
        __builtin_amdgcn_sched_barrier(0);  // keep the unrolled key-tile pairs apart: interleaving them costs ~60 live registers
      }
      // dQ^^T[d][q] = sum_keys K^^T[d][key] dS^T[key][q]  (already times alpha)
      f32x4 dqa[6];
      const char* kbase = Ks + tp * 8 + (4 * fq + tq) * ROWB;
      static_for<0, 6>([&](auto dtc) {
        constexpr int dt = decltype(dtc)::value;
        dqa[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        uint2 lo[KT / 2], hi[KT / 2];
        static_for<0, KT / 2>([&](auto sc_) {
          constexpr int s2 = decltype(sc_)::value;
          lo[s2] = lds_tr16_b64_o<dt * 32 + 32 * s2 * ROWB>(kbase); hi[s2] = lds_tr16_b64_o<dt * 32 + (32 * s2 + 16) * ROWB>(kbase);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s2 = 0; s2 < KT / 2; ++s2) {
          const uint4 u = make_uint4(lo[s2].x, lo[s2].y, hi[s2].x, hi[s2].y);
          dqa[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, u), dsb[s2], dqa[dt], 0, 0, 0);
        }
      });
      // RMSNorm backward for query fr: lane holds d = 16dt + 4fq + r
      float x[6][4]; float ss = 0.f;
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) {
        const u16x4 xv = xraw[dt];
#pragma unroll
        for (int r = 0; r < 4; ++r) { x[dt][r] = bf2f(xv[r]); ss += x[dt][r] * x[dt][r]; }
      }
      ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
      const float rr = rsqrtf(ss / DH + 1e-6f);
      float gx = 0.f;
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) {
        const f32x4 sc = *(const f32x4*)(sscale + dt * 16 + fq * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { x[dt][r] *= rr; gx += dqa[dt][r] * sc[r] * x[dt][r]; }
      }
      gx += __shfl_xor(gx, 16, 64); gx += __shfl_xor(gx, 32, 64);
      gx /= DH;
      if (valid) {
        bf16_t* op = g.dq + (seq * S + qrow) * g.ldq + h * DH + fq * 4;
#pragma unroll
        for (int dt = 0; dt < 6; ++dt) {
          const f32x4 sc = *(const f32x4*)(sscale + dt * 16 + fq * 4);
          u16x4 o4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o4[r] = f2bf(rr * (dqa[dt][r] * sc[r] - x[dt][r] * gx));
            ds_acc[dt][r] += dqa[dt][r] * x[dt][r];
          }
          *(u16x4*)(op + dt * 16) = o4;
        }
      }
    }

    // ------------------------------------------------------------------ (b) key tiles -> dk, dv
    for (int kt = w; kt < QT && role == 1; kt += 4) {  // real key tiles only (S_q == S_k)
      const int k0 = kt * 16;
      int krow = k0 + fr; const bool valid = krow < S; if (!valid) krow = S - 1;
      u16x4 xraw[6];  // raw k row for the RMSNorm backward, requested before the MFMA work (see part (a))
      {
        const bf16_t* xp = g.k + (seq * S + krow) * g.ldk + h * DH + fq * 4;
#pragma unroll
        for (int dt = 0; dt < 6; ++dt) xraw[dt] = *(const u16x4*)(xp + dt * 16);
      }
      bf16x8 kb[3], vb[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        kb[s] = *(const bf16x8*)(Ks + (k0 + fr) * ROWB + (s * 32 + fq * 8) * 2);
        vb[s] = *(const bf16x8*)(Vs + (k0 + fr) * ROWB + (s * 32 + fq * 8) * 2);
      }
      const float kbv = kbias[k0 + fr];
      const bool keep = kbv == 0.f;
      f32x4 dva[6], dka[6];
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) { dva[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dka[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
      for (int s2 = 0; s2 < KT / 2; ++s2) {
        u16x8 tp_, tds;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int qt = 2 * s2 + hf;
          f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const bf16x8 qf = *(const bf16x8*)(Qs + (qt * 16 + fr) * ROWB + (s * 32 + fq * 8) * 2);
            const bf16x8 df = *(const bf16x8*)(dOs + (qt * 16 + fr) * ROWB + (s * 32 + fq * 8) * 2);
            st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kb[s], st, 0, 0, 0);     // S[q = 16qt+4fq+r][key = k0+fr]
            dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, vb[s], dpt, 0, 0, 0);   // dP[q][key]
          }
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
        const bf16x8 pb = __builtin_bit_cast(bf16x8, tp_), dsb = __builtin_bit_cast(bf16x8, tds);
        // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^^T[d][key] += Q^^T[d][q] dS[q][key]
        uint2 olo[6], ohi[6], qlo[6], qhi[6];
        const int roff = (32 * s2 + 4 * fq + tq) * ROWB + tp * 8;
        const char* ob = dOs + roff; const char* qb_ = Qs + roff;
        static_for<0, 6>([&](auto dtc) {
          constexpr int dt = decltype(dtc)::value;
          olo[dt] = lds_tr16_b64_o<dt * 32>(ob); ohi[dt] = lds_tr16_b64_o<16 * ROWB + dt * 32>(ob);
          qlo[dt] = lds_tr16_b64_o<dt * 32>(qb_); qhi[dt] = lds_tr16_b64_o<16 * ROWB + dt * 32>(qb_);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < 6; ++dt) {
          const uint4 uo = make_uint4(olo[dt].x, olo[dt].y, ohi[dt].x, ohi[dt].y);
          const uint4 uq = make_uint4(qlo[dt].x, qlo[dt].y, qhi[dt].x, qhi[dt].y);
          dva[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, uo), pb, dva[dt], 0, 0, 0);
          dka[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, uq), dsb, dka[dt], 0, 0, 0);
        }
      }
      // lane: key = k0 + fr, d = 16dt + 4fq + r
      float x[6][4]; float ss = 0.f;
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) {
        const u16x4 xv = xraw[dt];
#pragma unroll
        for (int r = 0; r < 4; ++r) { x[dt][r] = bf2f(xv[r]); ss += x[dt][r] * x[dt][r]; }
      }
      ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
      const float rr = rsqrtf(ss / DH + 1e-6f);
      float gx = 0.f;
#pragma unroll
      for (int dt = 0; dt < 6; ++dt) {
        const f32x4 sc = *(const f32x4*)(sscale + DH + dt * 16 + fq * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { x[dt][r] *= rr; gx += dka[dt][r] * sc[r] * x[dt][r]; }
      }
      gx += __shfl_xor(gx, 16, 64); gx += __shfl_xor(gx, 32, 64);
      gx /= DH;
      if (valid) {
        bf16_t* okp = g.dk + (seq * S + krow) * g.ldk + h * DH + fq * 4;
        bf16_t* ovp = g.dv + (seq * S + krow) * g.ldv + h * DH + fq * 4;
#pragma unroll
        for (int dt = 0; dt < 6; ++dt) {
          const f32x4 sc = *(const f32x4*)(sscale + DH + dt * 16 + fq * 4);
          u16x4 k4, v4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            k4[r] = f2bf(rr * (dka[dt][r] * sc[r] - x[dt][r] * gx));
            v4[r] = f2bf(dva[dt][r]);
            ds_acc[dt][r] += dka[dt][r] * x[dt][r];
          }
          *(u16x4*)(okp + dt * 16) = k4;
          *(u16x4*)(ovp + dt * 16) = v4;
        }
      }
    }
  }
  // ---- flush the RMSNorm scale gradients: lanes with equal fq hold the same d -> reduce over fr, then over waves through LDS
  __syncthreads();
  for (int t = tid; t < 2 * DH; t += 512) sred[t] = 0.f;
  __syncthreads();
#pragma unroll
  for (int dt = 0; dt < 6; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = ds_acc[dt][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) a += __shfl_xor(a, o, 64);
      if (fr == 0) atomicAdd(sred + role * DH + dt * 16 + fq * 4 + r, a);
    }
  __syncthreads();
  if (tid < DH) atomicAdd(g.dsq + tid, sred[tid]);
  else if (tid < 2 * DH) atomicAdd(g.dsk + tid - DH, sred[tid]);
}

template <int KT>
static void launch_bwd(spa3d_ctx* c, const AttnBwdArgs& a) {
  const int lds = 4 * KT * 16 * ROWB + 4 * KT * 16 * 4 + 4 * DH * 4;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<KT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)attn_bwd8_kernel<KT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
  const unsigned grid = (unsigned)std::min<int64_t>(a.nprob, 1024);
  static int w8 = -1; if (w8 < 0) { const char* e = getenv("SPA3D_ATTN_BWD_W8"); w8 = e ? atoi(e) : 1; }
  if (w8) attn_bwd8_kernel<KT><<<grid, 512, lds, c->stream>>>(a);
  else attn_bwd_kernel<KT><<<grid, 256, lds, c->stream>>>(a);
}

bool attn_fused_bwd_bf16(spa3d_ctx* c, const bf16_t* q, const bf16_t* k, const bf16_t* v, int64_t ldq, int64_t ldk, int64_t ldv,
                         const float* sq, const float* sk, const float* km, int64_t nseq, int Sq, int Sk, int H, int Dh, const bf16_t* o,
                         const float* lse, const bf16_t* d_o, bf16_t* dq, bf16_t* dk, bf16_t* dv, float* dsq, float* dsk) {
  if (Dh != DH || Sq != Sk || Sk < 2 || Sk > 192 || !o || !lse) return false;
  if (ldq % 8 || ldk % 8 || ldv % 8 || !al16(q) || !al16(k) || !al16(v) || !al16(o) || !al16(d_o) || !al16(dq) || !al16(dk) || !al16(dv) ||
      !al16(sq) || !al16(sk))
    return false;
  if (c->dry) return true;
  AttnBwdArgs a; a.q = q; a.k = k; a.v = v; a.o = o; a.d_o = d_o; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.sq = sq; a.sk = sk; a.km = km;
  a.lse = lse; a.S = Sk; a.H = H; a.nprob = nseq * H; a.dq = dq; a.dk = dk; a.dv = dv; a.dsq = dsq; a.dsk = dsk;
  const int KT = ((Sk + 31) / 32) * 2;
  ProfScope ps(c, PROF_ATTN_BWD, 14.0 * (double)Sq * Sk * Dh * (double)a.nprob, (double)a.nprob * Sq * Dh * 2.0 * 8.0);
  switch (KT) {
    case 2: launch_bwd<2>(c, a); break;
    case 4: launch_bwd<4>(c, a); break;
    case 6: launch_bwd<6>(c, a); break;
    case 8: launch_bwd<8>(c, a); break;
    case 10: launch_bwd<10>(c, a); break;
    case 12: launch_bwd<12>(c, a); break;
    default: return false;
  }
  SPA_LAUNCH_CHECK(c);
  return true;
}
