#!/usr/bin/env python3
"""Generates tools/micro/valu_phase.hip: does separating full-rate (v_and) and half-rate
(v_bcnt) work into barrier-synchronised PHASES (all waves of a SIMD in the same phase)
recover the additive issue cost?  One workgroup per CU; 512 threads (2 waves/SIMD) or
1024 threads (4 waves/SIMD)."""
def and_(k, nt): return f"v_and_b32 v{24 + k % min(nt, 40)}, v{8 + k % 8}, v{16 + (k + 1) % 8}"
def bcnt(k, nt): return f"v_bcnt_u32_b32 v{64 + k % 64}, v{24 + k % min(nt, 40)}, v{64 + k % 64}"
def kernel(name, threads, body_a, body_b, barrier):
    a = "\\n ".join(body_a); b = "\\n ".join(body_b)
    bar = "s_barrier" if barrier else "s_nop 0"
    return f'''__global__ __launch_bounds__({threads}) void k_{name}(uint32_t *out, uint32_t seed) {{
  asm volatile("{init}" :: "v"(seed + threadIdx.x) : CLOB);
  for (int it = 0; it < ITERS; ++it) {{
    asm volatile("{a}\\n {bar}\\n {b}\\n {bar}" ::: CLOB);
  }}
  uint32_t r; asm volatile("v_add_u32 %0, v64, v127\\n v_add_u32 %0, %0, v24" : "=v"(r) :: CLOB);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}}'''
clob = ",".join(f'"v{r}"' for r in range(8, 128))
init = "\\n ".join([f"v_mov_b32 v{r}, %0" for r in range(8, 24)] + [f"v_mov_b32 v{r}, 0" for r in range(24, 128)])
src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdint>',
'#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\\n", #x, hipGetErrorString(e)); return 1; } } while (0)',
'constexpr int ITERS = 2048;', f'#define CLOB {clob}']
runs = []
for threads in (256, 512, 1024):
    for na, nb in ((40, 32), (20, 16), (10, 8)):
        nt = 80 if na > 40 else 40
        A = [and_(k, nt) for k in range(na)]; B = [bcnt(k, nt) for k in range(nb)]
        src.append(kernel(f"phase_{threads}_{na}_{nb}", threads, A, B, True)); runs.append((f"phase_{threads}_{na}_{nb}", threads, na, nb))
        src.append(kernel(f"nobar_{threads}_{na}_{nb}", threads, A, B, False)); runs.append((f"nobar_{threads}_{na}_{nb}", threads, na, nb))
    # interleaved reference (pairwise), no barrier
    mixed = []
    for k in range(32):
        mixed += [and_(k, 40), bcnt(k, 40)]
    mixed += [and_(k, 40) for k in range(32, 40)]
    src.append(kernel(f"mixed_{threads}", threads, mixed, [], False)); runs.append((f"mixed_{threads}", threads, 40, 32))
src.append('''template <typename K>
int run(const char *name, K kern, uint32_t *d, int threads, int na, int nb, int wg_per_cu) {
  const int grid = 256 * wg_per_cu;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int r = 0; r < 3; ++r) kern<<<grid, threads>>>(d, 12345);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 10; ++r) kern<<<grid, threads>>>(d, 12345 + r);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
  const int waves_per_simd = threads / 256 * wg_per_cu;
  const double cyc_per_iter_per_simd = ms * 1e-3 * 2.4e9 / ITERS;
  printf("%-22s WG/CU=%d waves/SIMD=%d  %7.1f cyc per iteration per SIMD  = %5.2f cyc/instr   additive %.0f\\n", name,
         wg_per_cu, waves_per_simd, cyc_per_iter_per_simd, cyc_per_iter_per_simd / (waves_per_simd * (na + nb)),
         waves_per_simd * (na * 2.07 + nb * 4.19));
  return 0;
}
int main() {
  uint32_t *d; CHECK(hipMalloc(&d, 1024 * 1024 * 4));''')
for name, threads, na, nb in runs:
    for wpc in ({256: (2, 4), 512: (1, 2), 1024: (1,)}[threads]):
        src.append(f'  if (run("{name}", k_{name}, d, {threads}, {na}, {nb}, {wpc})) return 1;')
src.append('  return 0;\n}')
open(__import__('os').path.dirname(__import__('os').path.abspath(__file__)) + '/valu_phase.hip', 'w').write("\n".join(src).replace("{init}", init) + "\n")
