// Microbenchmark: sustained wave64 VALU issue rate on gfx950 for the ops the
// KING kernel uses (v_and_b32, v_bcnt_u32_b32, v_bitop3_b32) next to v_fma_f32,
// at 1/2/4/8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;  // independent chains

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
  uint32_t a[UNROLL], acc[UNROLL];
  float f[UNROLL];
  for (int u = 0; u < UNROLL; ++u) { a[u] = seed * (u + 1) + threadIdx.x; acc[u] = u; f[u] = (float)u; }
  const uint32_t b = seed ^ 0x5555AAAAu;
  const float fb = 1.0001f, fc = 0.5f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (OP == 0) {  // and + bcnt (2 VALU)
        uint32_t t = a[u] & (b + it);
        acc[u] += __builtin_popcount(t);
      } else if (OP == 1) {  // bcnt only (1 VALU)
        acc[u] += __builtin_popcount(a[u] + it);   // add folds? keep 2 ops at worst
      } else if (OP == 2) {  // fma (1 VALU)
        f[u] = __builtin_fmaf(f[u], fb, fc);
      } else if (OP == 3) {  // bitop3 + bcnt
        uint32_t t = __builtin_amdgcn_bitop3_b32(a[u], b, acc[u], 0x28);
        acc[u] += __builtin_popcount(t);
      } else if (OP == 4) {  // and only chain
        a[u] = (a[u] & b) ^ it;
      }
    }
  }
  uint32_t r = 0;
  for (int u = 0; u < UNROLL; ++u) r += acc[u] + a[u] + (uint32_t)f[u];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
int run(const char *name, int valu_per_iter_per_chain, uint32_t *d) {
  for (int blocks_per_cu : {1, 2, 4, 8}) {
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<OP><<<grid, 256>>>(d, 12345);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) k<OP><<<grid, 256>>>(d, 12345 + r);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    const double wave_instrs = (double)grid * 4 * ITERS * UNROLL * valu_per_iter_per_chain;
    const double per_simd = wave_instrs / 1024.0;
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-14s waves/SIMD=%d  %.3f ms  %.2f cycles/VALU-instr/SIMD (at 2.4 GHz)  %.2f T lane-ops/s\n",
           name, blocks_per_cu, ms, cycles / per_simd, wave_instrs * 64 / (ms * 1e-3) / 1e12);
  }
  return 0;
}

int main() {
  uint32_t *d; CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
  if (run<0>("and+bcnt", 2, d)) return 1;
  if (run<1>("add+bcnt", 2, d)) return 1;
  if (run<2>("fma_f32", 1, d)) return 1;
  if (run<3>("bitop3+bcnt", 2, d)) return 1;
  if (run<4>("and+xor", 2, d)) return 1;
  return 0;
}
