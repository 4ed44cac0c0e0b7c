// How fast does an LDS-DMA stream of a row-major [M][K] bf16 matrix run when every wave-instruction takes 16 rows x 64 B (a k32 phase of an NT GEMM tile)
// instead of 8 rows x 128 B (k64)?  Persistent workgroups, 256 rows per tile, ring of `R` pieces-sets in flight, no compute.
//   hipcc --offload-arch=gfx950 -O3 -o probe_dma_pieces probe_dma_pieces.hip && ./probe_dma_pieces
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define GLDS(voff, sbase, m0v) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(m0v) : "memory", "m0")

// PB = bytes of a row piece (64 or 128); a phase = 256 rows x PB; 4 waves, wave w stages rows 64w .. 64w+63 = 64 * PB / 1024 instructions
template <int PB, int LEAD>
__global__ __launch_bounds__(256, 1) void stream_kernel(const char* A, int64_t M, int K2 /* row bytes */, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RPI = 1024 / PB;        // rows per instruction
  constexpr int NI = 64 / RPI;          // instructions per wave and phase
  constexpr int PHB = 256 * PB;         // phase bytes
  constexpr int R = LEAD + 1;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int lpr = PB / 16;              // lanes per row
  const unsigned voff0 = (unsigned)((lane / lpr) * K2 + (lane % lpr) * 16);
  const int U = K2 / PB;
  int issued = 0, done = 0;
  const int total = ((tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) * U;
  int ti = blockIdx.x, pi = 0;
  auto issue = [&]() {
    const char* sb = A + ((int64_t)ti * 256 + 64 * w) * K2 + (int64_t)pi * PB;
    const unsigned m0 = lds0 + (unsigned)(issued % R) * PHB + (unsigned)w * (64 * PB);
#pragma unroll
    for (int i = 0; i < NI; ++i) { const char* s2 = sb + (int64_t)i * RPI * K2; const unsigned m2 = m0 + i * 1024; GLDS(voff0, s2, m2); }
    ++issued; if (++pi == U) { pi = 0; ti += gridDim.x; }
  };
  for (int i = 0; i < LEAD && issued < total; ++i) issue();
  float acc = 0.f;
  while (done < total) {
    if (issued < total) { issue(); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LEAD * NI) : "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc += *(const float*)(smem + (done % R) * PHB + threadIdx.x * 16);
    ++done;
    __syncthreads();
  }
  if (acc == 123.456f) ((float*)A)[0] = acc;
}

template <int PB, int LEAD>
static void run(const char* A, int64_t M, int K2, const char* name) {
  const int tiles = (int)(M / 256);
  const int lds = (LEAD + 1) * 256 * PB;
  CK(hipFuncSetAttribute((const void*)stream_kernel<PB, LEAD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  stream_kernel<PB, LEAD><<<256, 256, lds>>>(A, M, K2, tiles);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) stream_kernel<PB, LEAD><<<256, 256, lds>>>(A, M, K2, tiles);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  printf("%-44s row bytes %5d  LDS %6d  %7.3f ms  %6.2f TB/s\n", name, K2, lds, ms, (double)tiles * 256 * K2 / ms / 1e9);
}

int main() {
  const int64_t M = 3065088;  // multiple of 256
  for (int K2 : {3072, 4608, 1536}) {
    char* A; CK(hipMalloc(&A, M * K2)); CK(hipMemset(A, 1, M * K2));
    run<128, 3>(A, M, K2, "8 rows x 128 B per instruction, lead 3");
    run<128, 4>(A, M, K2, "8 rows x 128 B per instruction, lead 4");
    run<64, 3>(A, M, K2, "16 rows x 64 B per instruction, lead 3");
    run<64, 6>(A, M, K2, "16 rows x 64 B per instruction, lead 6");
    run<64, 9>(A, M, K2, "16 rows x 64 B per instruction, lead 9");
    CK(hipFree(A));
  }
  return 0;
}
