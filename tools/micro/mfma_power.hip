// Does the DATA in the fp4 operands change the clock the chip holds?  MFMAs only
// (loop-invariant fragments), one wave per SIMD on every CU, 20 accumulators
// per wave like the pair kernel; the B operand of 8 of every 20 MFMAs (the
// share of the hi / hj products) is either "defined" (99 % ones), its
// complement "missing" (1 % ones) or random, everything else random at the
// densities of real genotype planes.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void loop_kernel(const uint4 *frag, float *out, int iters,
                                                   unsigned long long *stamps) {
  const int l = threadIdx.x & 63;
  // 8 fragments per lane: 0..3 "dense" operands (A side), 4..5 random B, 6..7 the B under test
  v8i f[8];
  for (int k = 0; k < 8; ++k) {
    const uint4 w = frag[k * 64 + l];
    f[k] = v8i{(int)w.x, (int)w.y, (int)w.z, (int)w.w, 0, 0, 0, 0};
  }
  v16f acc[16] = {};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int b = q < 10 ? 4 + (q & 1) : 6 + (q & 1);   // 10 random-B, 6 tested-B MFMAs
      acc[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(f[q & 3], f[b], acc[q], 4, 4, 0, 0, 0, 0);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
  float s = 0;
  for (int q = 0; q < 16; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static uint32_t nibbles(double density, int code) {
  uint32_t w = 0;
  for (int n = 0; n < 8; ++n) if ((double)rand() / RAND_MAX < density) w |= (uint32_t)code << (4 * n);
  return w;
}

int main() {
  const int iters = 4000, grid = 256;
  uint4 *d_frag; float *d_out; unsigned long long *d_st;
  CHECK(hipMalloc(&d_frag, 8 * 64 * 16)); CHECK(hipMalloc(&d_out, grid * 256 * 4)); CHECK(hipMalloc(&d_st, grid * 16));
  const char *names[] = {"B = defined (99 % ones)", "B = missing (1 % ones)", "B = random 30 %", "all operands zero"};
  for (int mode = 0; mode < 4; ++mode) {
    std::vector<uint32_t> h(8 * 64 * 4);
    srand(5);
    for (int k = 0; k < 8; ++k)
      for (int i = 0; i < 64 * 4; ++i) {
        double dens = k < 4 ? (k == 0 ? 0.1 : k == 1 ? 0.55 : 0.3) : k < 6 ? 0.3 : (mode == 0 ? 0.99 : mode == 1 ? 0.01 : 0.3);
        if (mode == 3) dens = 0;
        h[(k * 64) * 4 + i] = nibbles(dens, 2);
      }
    CHECK(hipMemcpy(d_frag, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int r = 0; r < 300; ++r) loop_kernel<<<grid, 256>>>(d_frag, d_out, iters, d_st);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 20; ++r) loop_kernel<<<grid, 256>>>(d_frag, d_out, iters, d_st);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    std::vector<unsigned long long> st(grid * 2);
    CHECK(hipMemcpy(st.data(), d_st, grid * 16, hipMemcpyDeviceToHost));
    std::vector<double> clk;
    for (int b = 0; b < grid; ++b) clk.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 100e6);
    std::sort(clk.begin(), clk.end());
    printf("%-28s %.3f ms  in-kernel clock %.3f GHz  (%.2f cycles per MFMA)\n", names[mode], ms,
           clk[grid / 2] / 1e9, (double)st[0] / ((double)iters * 16));
  }
  return 0;
}
