// probe_tr.hip -- hardware probe (not product code): prints the lane->element map of ds_read_b64_tr_b16
// and checks the 16x16x32 / 32x32x16 bf16 MFMA operand maps with exact integer data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ unsigned short f2b(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

__global__ void probe_tr(float* out) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[32 * 64];
  for (int i = threadIdx.x; i < 32 * 64; i += 64) tile[i] = f2b((float)((i / 64) * 64 + (i % 64)) / 8.0f);  // exact in bf16? use small ints
  __syncthreads();
  int l = threadIdx.x;
  // lane 4q+p of 16-lane group g supplies row 4g+q, cols 4p..4p+3
  int g = l >> 4, q = (l & 15) >> 2, p = l & 3;
  unsigned addr = (unsigned)(size_t)(&tile[(4 * g + q) * 64 + 4 * p]);
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) {
    unsigned short h = (unsigned short)(v >> (16 * j));
    out[l * 4 + j] = __uint_as_float(((unsigned)h) << 16) * 8.0f;
  }
}
// MFMA 16x16x32: A[i][k] = (i==k%16 ? 1:0)*..., use A = delta to read B layout etc.  Generic check: compute C = A*B with
// A[i][k] = i + 0.5*(k==i), B[k][j] = small ints, compare against host.
__global__ void probe_mfma16(const float* A, const float* B, float* C) {  // A[16][32], B[32][16] row-major floats
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (__bf16)A[(l & 15) * 32 + 8 * (l >> 4) + j];
    b[j] = (__bf16)B[(8 * (l >> 4) + j) * 16 + (l & 15)];
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
__global__ void probe_mfma32(const float* A, const float* B, float* C) {  // A[32][16], B[16][32]
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (__bf16)A[(l & 31) * 16 + 8 * (l >> 5) + j];
    b[j] = (__bf16)B[(8 * (l >> 5) + j) * 32 + (l & 31)];
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
int main() {
  float* d; hipMalloc(&d, 64 * 4 * 4);
  probe_tr<<<1, 64>>>(d);
  std::vector<float> h(256);
  hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
  printf("ds_read_b64_tr_b16: lane -> 4 values as (row,col)\n");
  int ok = 1;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) { int v = (int)h[l * 4 + j]; printf(" (%d,%d)", v / 64, v % 64);
      if (v / 64 != 4 * (l >> 4) + j || v % 64 != (l & 15)) ok = 0; }
    printf("\n");
  }
  printf("TR_MAP_AS_DOCUMENTED=%d\n", ok);
  // mfma checks
  std::vector<float> A(512), B(512), C(1024), R(1024);
  for (int i = 0; i < 512; ++i) { A[i] = (float)((i * 7) % 5 - 2); B[i] = (float)((i * 3) % 7 - 3); }
  float *dA, *dB, *dC; hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 4096);
  hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
  probe_mfma16<<<1, 64>>>(dA, dB, dC); hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += A[i * 32 + k] * B[k * 16 + j]; if (s != C[i * 16 + j]) ++bad; }
  printf("MFMA16x16x32_MAP_BAD=%d\n", bad);
  probe_mfma32<<<1, 64>>>(dA, dB, dC); hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
  bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 32 + j]; if (s != C[i * 32 + j]) ++bad; }
  printf("MFMA32x32x16_MAP_BAD=%d\n", bad);
  return 0;
}
