#!/usr/bin/env python3
"""Generates tools/micro/valu_ratio.hip: streams of R v_and per 1 v_bcnt (R = 0,1,3,7,15,inf)
and phase-separated streams, to see how a few half-rate ops change the issue cost of the
full-rate ones.  Explicit registers, one run, several waves per SIMD."""
def and_(k): return f"v_and_b32 v{48 + k % 16}, v{32 + k % 8}, v{40 + (k + 1) % 8}"
def bcnt(k): return f"v_bcnt_u32_b32 v{64 + k % 16}, v{48 + (k + 5) % 16}, v{64 + k % 16}"
variants = {}
for r in (1, 3, 7, 15, 31):
    seq, a, b = [], 0, 0
    for blk in range(128 // (r + 1)):
        for _ in range(r):
            seq.append(and_(a)); a += 1
        seq.append(bcnt(b)); b += 1
    variants[f"and{r}_bcnt1"] = seq
variants["and_only"] = [and_(k) for k in range(128)]
variants["bcnt_only"] = [bcnt(k) for k in range(128)]
variants["phase64"] = [and_(k) for k in range(64)] + [bcnt(k) for k in range(64)]
clob = ",".join(f'"v{r}"' for r in range(32, 80))
src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdint>',
'#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\\n", #x, hipGetErrorString(e)); return 1; } } while (0)',
'constexpr int ITERS = 2048;', f'#define CLOB {clob}']
init = "\\n ".join([f"v_mov_b32 v{r}, %0" for r in range(32, 48)] + [f"v_mov_b32 v{r}, 0" for r in range(48, 80)])
for name, seq in variants.items():
    body = "\\n ".join(seq)
    src.append(f'''__global__ __launch_bounds__(256) void k_{name}(uint32_t *out, uint32_t seed) {{
  asm volatile("{init}" :: "v"(seed + threadIdx.x) : CLOB);
  for (int it = 0; it < ITERS; ++it) {{ asm volatile("{body}" ::: CLOB); }}
  uint32_t r; asm volatile("v_add_u32 %0, v64, v79\\n v_add_u32 %0, %0, v48" : "=v"(r) :: CLOB);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}}''')
src.append('''template <typename K>
int run(const char *name, K kern, uint32_t *d, int n_and, int n_bcnt) {
  printf("%-12s and=%3d bcnt=%3d", name, n_and, n_bcnt);
  for (int blocks_per_cu : {1, 2, 4, 8}) {
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) kern<<<grid, 256>>>(d, 12345);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 10; ++r) kern<<<grid, 256>>>(d, 12345 + r);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    const double blocks_per_simd = (double)grid * 4 * ITERS / 1024.0;
    printf("  w%d: %6.1f cyc/block", blocks_per_cu, ms * 1e-3 * 2.4e9 / blocks_per_simd);
  }
  printf("   additive model: %.0f\\n", n_and * 2.07 + n_bcnt * 4.19);
  return 0;
}
int main() {
  uint32_t *d; CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));''')
for name, seq in variants.items():
    na = sum(1 for x in seq if x.startswith("v_and")); nb = len(seq) - na
    src.append(f'  if (run("{name}", k_{name}, d, {na}, {nb})) return 1;')
src.append('  return 0;\n}')
open(__import__('os').path.dirname(__import__('os').path.abspath(__file__)) + '/valu_ratio.hip', 'w').write("\n".join(src) + "\n")
