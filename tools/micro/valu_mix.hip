// Microbenchmark: does the ORDER of full-rate (v_and) and half-rate (v_bcnt)
// wave64 VALU instructions matter on gfx950?  Same 8 and + 8 bcnt per block:
//   alt   : and0 bcnt0 and1 bcnt1 ...      (dependent neighbours)
//   batch : and0..and7 bcnt0..bcnt7        (dependency distance 8)
//   skew  : and0 and1 bcnt0 and2 bcnt1 ... (distance 2)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 2048;
#define AND(k) asm volatile("v_and_b32 %0, %1, %2" : "=v"(t##k) : "v"(a##k), "v"(b));
#define BCNT(k) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(c##k) : "v"(t##k));
#define DECL uint32_t a0 = s, a1 = s * 3, a2 = s * 5, a3 = s * 7, a4 = s * 9, a5 = s * 11, a6 = s * 13, a7 = s * 15; \
  uint32_t t0, t1, t2, t3, t4, t5, t6, t7; uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0; uint32_t b = s ^ 0x5555AAAAu;
#define FIN out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;

__global__ __launch_bounds__(256) void k_alt(uint32_t *out, uint32_t seed) {
  uint32_t s = seed + threadIdx.x; DECL
  for (int it = 0; it < ITERS; ++it) {
    AND(0) BCNT(0) AND(1) BCNT(1) AND(2) BCNT(2) AND(3) BCNT(3) AND(4) BCNT(4) AND(5) BCNT(5) AND(6) BCNT(6) AND(7) BCNT(7)
    AND(0) BCNT(0) AND(1) BCNT(1) AND(2) BCNT(2) AND(3) BCNT(3) AND(4) BCNT(4) AND(5) BCNT(5) AND(6) BCNT(6) AND(7) BCNT(7)
  }
  FIN
}
__global__ __launch_bounds__(256) void k_batch(uint32_t *out, uint32_t seed) {
  uint32_t s = seed + threadIdx.x; DECL
  for (int it = 0; it < ITERS; ++it) {
    AND(0) AND(1) AND(2) AND(3) AND(4) AND(5) AND(6) AND(7) BCNT(0) BCNT(1) BCNT(2) BCNT(3) BCNT(4) BCNT(5) BCNT(6) BCNT(7)
    AND(0) AND(1) AND(2) AND(3) AND(4) AND(5) AND(6) AND(7) BCNT(0) BCNT(1) BCNT(2) BCNT(3) BCNT(4) BCNT(5) BCNT(6) BCNT(7)
  }
  FIN
}
__global__ __launch_bounds__(256) void k_skew(uint32_t *out, uint32_t seed) {
  uint32_t s = seed + threadIdx.x; DECL
  AND(0)
  for (int it = 0; it < ITERS; ++it) {
    AND(1) BCNT(0) AND(2) BCNT(1) AND(3) BCNT(2) AND(4) BCNT(3) AND(5) BCNT(4) AND(6) BCNT(5) AND(7) BCNT(6) AND(0) BCNT(7)
    AND(1) BCNT(0) AND(2) BCNT(1) AND(3) BCNT(2) AND(4) BCNT(3) AND(5) BCNT(4) AND(6) BCNT(5) AND(7) BCNT(6) AND(0) BCNT(7)
  }
  FIN
}
// 4 bcnt + 5 logic (phase-1 mix) and 5 + 5 (current mix), batched
#define BITOP(k) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x28" : "=v"(t##k) : "v"(a##k), "v"(b), "v"(t0));
__global__ __launch_bounds__(256) void k_mix55(uint32_t *out, uint32_t seed) {
  uint32_t s = seed + threadIdx.x; DECL
  for (int it = 0; it < ITERS; ++it) {
    AND(0) BCNT(0) BITOP(1) BCNT(1) AND(2) BCNT(2) AND(3) BCNT(3) AND(4) BCNT(4)
    AND(0) BCNT(0) BITOP(1) BCNT(1) AND(2) BCNT(2) AND(3) BCNT(3) AND(4) BCNT(4)
    AND(0) BCNT(0) BITOP(1) BCNT(1) AND(2) BCNT(2) AND(3) BCNT(3) AND(4) BCNT(4)
    AND(0) BCNT(0) BITOP(1) BCNT(1) AND(2) BCNT(2) AND(3) BCNT(3) AND(4) BCNT(4)
  }
  FIN
}
__global__ __launch_bounds__(256) void k_mix54(uint32_t *out, uint32_t seed) {
  uint32_t s = seed + threadIdx.x; DECL
  for (int it = 0; it < ITERS; ++it) {
    AND(0) BITOP(1) BCNT(1) AND(2) BCNT(2) AND(3) BCNT(3) AND(4) BCNT(4)
    AND(0) BITOP(1) BCNT(1) AND(2) BCNT(2) AND(3) BCNT(3) AND(4) BCNT(4)
    AND(0) BITOP(1) BCNT(1) AND(2) BCNT(2) AND(3) BCNT(3) AND(4) BCNT(4)
    AND(0) BITOP(1) BCNT(1) AND(2) BCNT(2) AND(3) BCNT(3) AND(4) BCNT(4)
  }
  FIN
}
template <typename K>
int run(const char *name, K kern, uint32_t *d, int instrs_per_iter) {
  printf("%-10s", name);
  for (int blocks_per_cu : {1, 2, 4, 8}) {
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    kern<<<grid, 256>>>(d, 12345); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) kern<<<grid, 256>>>(d, 12345 + r);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double per_simd = (double)grid * 4 * ITERS * instrs_per_iter / 1024.0;
    printf("  w%d: %5.2f cyc/instr", blocks_per_cu, ms * 1e-3 * 2.4e9 / per_simd);
  }
  printf("\n");
  return 0;
}
int main() {
  uint32_t *d; CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
  if (run("alt", k_alt, d, 32)) return 1;
  if (run("batch", k_batch, d, 32)) return 1;
  if (run("skew", k_skew, d, 32)) return 1;
  if (run("mix5+5", k_mix55, d, 40)) return 1;
  if (run("mix5+4", k_mix54, d, 36)) return 1;
  return 0;
}
