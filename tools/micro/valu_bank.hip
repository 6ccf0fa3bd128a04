// Microbenchmark: VGPR bank conflicts on gfx950.  Same instruction stream with
// source registers in the same bank (reg % 4 equal) or in different banks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 2048;
#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63"

// 8 x (and + bcnt); sources of the AND: (v32+4a, v36+4b) same bank 0 / (v32, v37) different banks
#define BODY_SAME \
  "v_and_b32 v48, v32, v36\n v_bcnt_u32_b32 v56, v48, v56\n" \
  "v_and_b32 v49, v40, v44\n v_bcnt_u32_b32 v57, v49, v57\n" \
  "v_and_b32 v50, v33, v37\n v_bcnt_u32_b32 v58, v50, v58\n" \
  "v_and_b32 v51, v41, v45\n v_bcnt_u32_b32 v59, v51, v59\n" \
  "v_and_b32 v52, v34, v38\n v_bcnt_u32_b32 v60, v52, v60\n" \
  "v_and_b32 v53, v42, v46\n v_bcnt_u32_b32 v61, v53, v61\n" \
  "v_and_b32 v54, v35, v39\n v_bcnt_u32_b32 v62, v54, v62\n" \
  "v_and_b32 v55, v43, v47\n v_bcnt_u32_b32 v63, v55, v63\n"
#define BODY_DIFF \
  "v_and_b32 v48, v32, v37\n v_bcnt_u32_b32 v57, v48, v57\n" \
  "v_and_b32 v49, v40, v45\n v_bcnt_u32_b32 v58, v49, v58\n" \
  "v_and_b32 v50, v33, v38\n v_bcnt_u32_b32 v59, v50, v59\n" \
  "v_and_b32 v51, v41, v46\n v_bcnt_u32_b32 v60, v51, v60\n" \
  "v_and_b32 v52, v34, v39\n v_bcnt_u32_b32 v61, v52, v61\n" \
  "v_and_b32 v53, v42, v47\n v_bcnt_u32_b32 v62, v53, v62\n" \
  "v_and_b32 v54, v35, v36\n v_bcnt_u32_b32 v63, v54, v63\n" \
  "v_and_b32 v55, v43, v44\n v_bcnt_u32_b32 v56, v55, v56\n"
// bcnt where src0 and the accumulator share a bank (v48 & v56: both bank 0) is BODY_SAME's
// bcnt pattern; BODY_DIFF's bcnt uses v48 (bank 0) with v57 (bank 1).
// AND only:
#define AND_SAME \
  "v_and_b32 v48, v32, v36\n v_and_b32 v49, v40, v44\n v_and_b32 v50, v33, v37\n v_and_b32 v51, v41, v45\n" \
  "v_and_b32 v52, v34, v38\n v_and_b32 v53, v42, v46\n v_and_b32 v54, v35, v39\n v_and_b32 v55, v43, v47\n"
#define AND_DIFF \
  "v_and_b32 v48, v32, v37\n v_and_b32 v49, v40, v45\n v_and_b32 v50, v33, v38\n v_and_b32 v51, v41, v46\n" \
  "v_and_b32 v52, v34, v39\n v_and_b32 v53, v42, v47\n v_and_b32 v54, v35, v36\n v_and_b32 v55, v43, v44\n"
#define BCNT_SAME \
  "v_bcnt_u32_b32 v56, v48, v56\n v_bcnt_u32_b32 v57, v49, v57\n v_bcnt_u32_b32 v58, v50, v58\n v_bcnt_u32_b32 v59, v51, v59\n" \
  "v_bcnt_u32_b32 v60, v52, v60\n v_bcnt_u32_b32 v61, v53, v61\n v_bcnt_u32_b32 v62, v54, v62\n v_bcnt_u32_b32 v63, v55, v63\n"
#define BCNT_DIFF \
  "v_bcnt_u32_b32 v57, v48, v57\n v_bcnt_u32_b32 v58, v49, v58\n v_bcnt_u32_b32 v59, v50, v59\n v_bcnt_u32_b32 v60, v51, v60\n" \
  "v_bcnt_u32_b32 v61, v52, v61\n v_bcnt_u32_b32 v62, v53, v62\n v_bcnt_u32_b32 v63, v54, v63\n v_bcnt_u32_b32 v56, v55, v56\n"
// bitop3 with 3 sources: all same bank / all different
#define BIT_SAME \
  "v_bitop3_b32 v48, v32, v36, v40 bitop3:0x28\n v_bitop3_b32 v49, v33, v37, v41 bitop3:0x28\n" \
  "v_bitop3_b32 v50, v34, v38, v42 bitop3:0x28\n v_bitop3_b32 v51, v35, v39, v43 bitop3:0x28\n" \
  "v_bitop3_b32 v52, v44, v32, v36 bitop3:0x28\n v_bitop3_b32 v53, v45, v33, v37 bitop3:0x28\n" \
  "v_bitop3_b32 v54, v46, v34, v38 bitop3:0x28\n v_bitop3_b32 v55, v47, v35, v39 bitop3:0x28\n"
#define BIT_DIFF \
  "v_bitop3_b32 v48, v32, v37, v42 bitop3:0x28\n v_bitop3_b32 v49, v33, v38, v43 bitop3:0x28\n" \
  "v_bitop3_b32 v50, v34, v39, v40 bitop3:0x28\n v_bitop3_b32 v51, v35, v36, v41 bitop3:0x28\n" \
  "v_bitop3_b32 v52, v44, v33, v38 bitop3:0x28\n v_bitop3_b32 v53, v45, v34, v39 bitop3:0x28\n" \
  "v_bitop3_b32 v54, v46, v35, v36 bitop3:0x28\n v_bitop3_b32 v55, v47, v32, v37 bitop3:0x28\n"

#define KERNEL(NAME, BODY)                                                     \
  __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed) { \
    asm volatile("v_mov_b32 v32, %0\n v_mov_b32 v33, %0\n v_mov_b32 v34, %0\n v_mov_b32 v35, %0\n" \
                 "v_mov_b32 v36, %0\n v_mov_b32 v37, %0\n v_mov_b32 v38, %0\n v_mov_b32 v39, %0\n" \
                 "v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n" \
                 "v_mov_b32 v44, %0\n v_mov_b32 v45, %0\n v_mov_b32 v46, %0\n v_mov_b32 v47, %0\n" \
                 "v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n" \
                 "v_mov_b32 v52, 0\n v_mov_b32 v53, 0\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0\n" \
                 "v_mov_b32 v56, 0\n v_mov_b32 v57, 0\n v_mov_b32 v58, 0\n v_mov_b32 v59, 0\n" \
                 "v_mov_b32 v60, 0\n v_mov_b32 v61, 0\n v_mov_b32 v62, 0\n v_mov_b32 v63, 0\n" \
                 :: "v"(seed + threadIdx.x) : CLOB);                           \
    for (int it = 0; it < ITERS; ++it) { asm volatile(BODY BODY BODY BODY ::: CLOB); } \
    uint32_t r; asm volatile("v_add_u32 %0, v56, v63\n v_add_u32 %0, %0, v48" : "=v"(r) :: CLOB); \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                            \
  }
KERNEL(k_pair_same, BODY_SAME)
KERNEL(k_pair_diff, BODY_DIFF)
KERNEL(k_and_same, AND_SAME)
KERNEL(k_and_diff, AND_DIFF)
KERNEL(k_bcnt_same, BCNT_SAME)
KERNEL(k_bcnt_diff, BCNT_DIFF)
KERNEL(k_bit_same, BIT_SAME)
KERNEL(k_bit_diff, BIT_DIFF)

template <typename K>
int run(const char *name, K kern, uint32_t *d, int instrs_per_iter) {
  printf("%-12s", name);
  for (int blocks_per_cu : {1, 2, 4, 8}) {
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    kern<<<grid, 256>>>(d, 12345); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) kern<<<grid, 256>>>(d, 12345 + r);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double per_simd = (double)grid * 4 * ITERS * instrs_per_iter / 1024.0;
    printf("  w%d: %5.2f", blocks_per_cu, ms * 1e-3 * 2.4e9 / per_simd);
  }
  printf("   cyc/instr\n");
  return 0;
}
int main() {
  uint32_t *d; CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
  if (run("pair_same", k_pair_same, d, 64)) return 1;
  if (run("pair_diff", k_pair_diff, d, 64)) return 1;
  if (run("and_same", k_and_same, d, 32)) return 1;
  if (run("and_diff", k_and_diff, d, 32)) return 1;
  if (run("bcnt_same", k_bcnt_same, d, 32)) return 1;
  if (run("bcnt_diff", k_bcnt_diff, d, 32)) return 1;
  if (run("bitop3_same", k_bit_same, d, 32)) return 1;
  if (run("bitop3_diff", k_bit_diff, d, 32)) return 1;
  return 0;
}
