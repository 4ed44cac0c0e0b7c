// attn_common.hpp -- LDS image layout, fragment-read helpers and the block -> problem map shared by the fused attention kernels
// (attention_fused.hip) and the fused QKV-projection + attention forward (qkv_attn.hip).
#pragma once
#include <type_traits>

#include "common.hpp"

namespace SPA_NS {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

#define NEG_BIG (-3.4028234663852886e38f)

// transposed LDS read with a compile-time byte offset in the instruction: as separate addresses every one is a loop-invariant
// VGPR (the 8-wave backward spilled ~60 of them)
template <int OFF>
__device__ __forceinline__ uint2 lds_tr16_b64_o(const void* p) {
  static_assert(OFF >= 0 && OFF < 65536, "16-bit DS offset");
  uint2 v;
  const unsigned a = (unsigned)(uintptr_t)p;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
  return v;
}
// 16-byte LDS read the COMPILER DOES NOT TRACK (no s_waitcnt of its own): for hand-sequenced loops that keep several groups of reads in flight
// and retire them with counted s_waitcnt lgkmcnt(N) (LDS operations return in order)
template <int OFF>
__device__ __forceinline__ uint4 lds_b128_o(const void* p) {
  static_assert(OFF >= 0 && OFF < 65536, "16-bit DS offset");
  uint4 v;
  const unsigned a = (unsigned)(uintptr_t)p;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
  return v;
}
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

constexpr int DH = 96, ROWB = DH * 2;  // 192-byte rows
// LDS images carry 32 bytes of padding after every 4 rows: row r starts at r*192 + (r/4)*32.  With plain 192-byte rows both kinds of
// fragment read were 2-way bank conflicts (MI355X_MICROARCH.md "LDS": ds_read_b128 is served in four fixed 16-lane groups, the
// transposed ds_read_b64_tr_b16 in 32-lane halves; rows r and r+4 start on the same banks, 4*192 = 3*256); this is the smallest padding
// for which the 16 lanes of every b128 group hit 16 distinct 16-byte slots AND the 32 lanes of every transposed-read half hit 32
// distinct 8-byte slots (exhaustive search over per-row paddings).  Per-row padding (208 B) would not fit four images of S = 151.
constexpr int ROW16 = 16 * ROWB + 128;  // bytes per 16-row block
constexpr int img_bytes(int rows) { return rows * ROWB + (rows / 4) * 32; }  // rows % 4 == 0
__device__ __forceinline__ int row_off(int r) { return r * ROWB + (r >> 2) * 32; }
constexpr bool fwd_wide(int KT, int NW) { return KT <= 10 || NW == 8; }

// Block -> problem map.  The dispatcher deals blocks round-robin over the 8 XCDs (observed, speed only), so blocks b, b+8, ..
// share an L2: give the i-th block of XCD label x head (i mod H) of sequence 8*(i div H) + x.  Bijective on [0, nseq*H); sequences
// past the last multiple of 8 keep the identity order.  A different placement changes speed only.
__device__ __forceinline__ int64_t map_prob(int64_t b, int64_t nseq, int H) {
  const int64_t nfull = (nseq >> 3) << 3;
  if (b >= nfull * H) return b;
  const int64_t x = b & 7, i = b >> 3;
  return ((i / H) * 8 + x) * H + (i % H);
}
// The same map in 32 bits with ONE division, returning (sequence, head) directly (every launcher refuses nseq * H * nsplit >= 2^31).  The 64-bit form above costs
// ~400 scalar instructions per problem (four 64-bit divisions: the map's / and %, then prob / H and its remainder), all of them in front of the problem's
// first global load in a persistent kernel.
__device__ __forceinline__ void map_prob32(unsigned b, unsigned nseq, unsigned H, unsigned& seq, unsigned& h) {
  const unsigned nfull = nseq & ~7u;
  if (b >= nfull * H) { seq = b / H; h = b - seq * H; return; }
  const unsigned i = b >> 3, q = i / H;
  seq = q * 8 + (b & 7); h = i - q * H;
}

// Wave-private LDS tile (16 rows): results held as (row fr, d = 16dt + 4fq + r) -- 8 bytes per lane and dt, i.e. 32-byte pieces of 16
// rows per store instruction -- are turned into 64 contiguous bytes per row and instruction (the staging loads' shape) on the way out.
constexpr int WROW = 208;                 // tile row stride: 13 x 16 B (192-B rows would put 8 of a 16-lane group on one bank pair)
constexpr int WTILE = 16 * WROW + 64;     // + 16 floats (the split-pass backward keeps the tile's row deltas there)
__device__ __forceinline__ void tile_put(char* wt, int dt, u16x4 v, int lane = threadIdx.x & 63) {
  *(u16x4*)(wt + (lane & 15) * WROW + (16 * dt + 4 * (lane >> 4)) * 2) = v;
}
__device__ __forceinline__ void tile_flush(const char* wt, bf16_t* g0, int64_t ld_, int nrows, int lane = threadIdx.x & 63) {  // g0: row 0, column 0 of the tile in global memory
  const int tr = lane >> 2, part = lane & 3;
  asm volatile("" ::: "memory");
  if (tr < nrows) {
    const u16x8* sp = (const u16x8*)(wt + tr * WROW) + part;
    u16x8* dp = (u16x8*)(g0 + (int64_t)tr * ld_) + part;
    const u16x8 a = sp[0], b = sp[4], c = sp[8];
    dp[0] = a; dp[4] = b; dp[8] = c;
  }
  asm volatile("" ::: "memory");
}

}  // namespace SPA_NS
