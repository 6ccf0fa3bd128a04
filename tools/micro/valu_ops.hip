// Microbenchmark: issue cost of individual gfx950 VALU instructions (wave64),
// pinned with inline asm, 8 independent chains, 1/2/4/8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 2048;

#define OP8(STR) \
  asm volatile(STR : "+v"(r0) : "v"(b), "v"(c)); asm volatile(STR : "+v"(r1) : "v"(b), "v"(c)); \
  asm volatile(STR : "+v"(r2) : "v"(b), "v"(c)); asm volatile(STR : "+v"(r3) : "v"(b), "v"(c)); \
  asm volatile(STR : "+v"(r4) : "v"(b), "v"(c)); asm volatile(STR : "+v"(r5) : "v"(b), "v"(c)); \
  asm volatile(STR : "+v"(r6) : "v"(b), "v"(c)); asm volatile(STR : "+v"(r7) : "v"(b), "v"(c));

#define KERNEL(NAME, STR)                                                      \
  __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed) { \
    uint32_t r0 = seed + threadIdx.x, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7,  \
             r4 = r0 * 11, r5 = r0 * 13, r6 = r0 * 17, r7 = r0 * 19;          \
    uint32_t b = seed ^ 0x5555AAAAu, c = seed * 977;                          \
    for (int it = 0; it < ITERS; ++it) { OP8(STR) OP8(STR) OP8(STR) OP8(STR) } \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7; \
  }

KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL(k_add, "v_add_u32 %0, %0, %1")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0")
KERNEL(k_bcnt2, "v_bcnt_u32_b32 %0, %0, %2")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x28")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %1, %2, %0")
KERNEL(k_sad8, "v_sad_u8 %0, %1, %2, %0")
KERNEL(k_sad32, "v_sad_u32 %0, %1, %2, %0")
KERNEL(k_fma, "v_fma_f32 %0, %1, %2, %0")
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %1, 1, %0")
KERNEL(k_dot4, "v_dot4_u32_u8 %0, %1, %2, %0")
KERNEL(k_dot8, "v_dot8_u32_u4 %0, %1, %2, %0")

template <typename K>
int run(const char *name, K kern, uint32_t *d) {
  printf("%-12s", name);
  for (int blocks_per_cu : {1, 2, 4, 8}) {
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    kern<<<grid, 256>>>(d, 12345);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) kern<<<grid, 256>>>(d, 12345 + r);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    const double per_simd = (double)grid * 4 * ITERS * 32 / 1024.0;
    printf("  w%d: %5.2f cyc", blocks_per_cu, ms * 1e-3 * 2.4e9 / per_simd);
  }
  printf("   (cycles per wave64 instr per SIMD at 2.4 GHz)\n");
  return 0;
}

int main() {
  uint32_t *d; CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
#define RUN(K) if (run(#K, K, d)) return 1;
  RUN(k_and) RUN(k_xor) RUN(k_add) RUN(k_bcnt) RUN(k_bcnt2) RUN(k_bitop3) RUN(k_and_or)
  RUN(k_perm) RUN(k_lshl_add) RUN(k_dot4) RUN(k_dot8)
  return 0;
}
