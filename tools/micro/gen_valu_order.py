# generates tools/micro/valu_order.hip: orderings of 16 AND + 16 BCNT (explicit registers)
ands = [f"v_and_b32 v{48+k}, v{32+k}, v{40+(k+1)%8}" if k < 8 else f"v_and_b32 v{64+k-8}, v{32+(k-8)}, v{40+(k+3)%8}" for k in range(16)]
tmps = [48+k if k < 8 else 64+k-8 for k in range(16)]
bcnts = [f"v_bcnt_u32_b32 v{72+k}, v{tmps[k]}, v{72+k}" for k in range(16)]
def order(group):
    out = []
    for g in range(0, 16, group):
        out += ands[g:g+group] + bcnts[g:g+group]
    return out
def skew(dist):
    # and k+dist issued before bcnt k  (software pipelined, needs prologue; approximate in-loop)
    out = []
    for k in range(16):
        out.append(ands[(k+dist) % 16]); out.append(bcnts[k])
    return out
variants = {"g1": order(1), "g2": order(2), "g4": order(4), "g8": order(8), "g16": order(16),
            "skew2": skew(2), "skew4": skew(4), "skew8": skew(8)}
clob = ",".join(f'"v{r}"' for r in range(32, 88))
src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdint>',
'#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\\n", #x, hipGetErrorString(e)); return 1; } } while (0)',
'constexpr int ITERS = 2048;', f'#define CLOB {clob}']
init = "\\n ".join([f"v_mov_b32 v{r}, %0" for r in range(32, 48)] + [f"v_mov_b32 v{r}, 0" for r in range(48, 88)])
for name, seq in variants.items():
    body = "\\n ".join(seq)
    src.append(f'''__global__ __launch_bounds__(256) void k_{name}(uint32_t *out, uint32_t seed) {{
  asm volatile("{init}" :: "v"(seed + threadIdx.x) : CLOB);
  for (int it = 0; it < ITERS; ++it) {{ asm volatile("{body}\\n {body}" ::: CLOB); }}
  uint32_t r; asm volatile("v_add_u32 %0, v72, v87\\n v_add_u32 %0, %0, v80" : "=v"(r) :: CLOB);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}}''')
src.append('''template <typename K>
int run(const char *name, K kern, uint32_t *d) {
  printf("%-8s", name);
  for (int blocks_per_cu : {1, 2, 3, 4, 8}) {
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    kern<<<grid, 256>>>(d, 12345); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) kern<<<grid, 256>>>(d, 12345 + r);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double per_simd = (double)grid * 4 * ITERS * 64 / 1024.0;
    printf("  w%d: %5.2f", blocks_per_cu, ms * 1e-3 * 2.4e9 / per_simd);
  }
  printf("   cyc/instr (16 and + 16 bcnt)\\n");
  return 0;
}
int main() {
  uint32_t *d; CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));''')
for name in variants:
    src.append(f'  if (run("{name}", k_{name}, d)) return 1;')
src.append('  return 0;\n}')
open(__import__('os').path.dirname(__import__('os').path.abspath(__file__)) + '/valu_order.hip', 'w').write("\n".join(src) + "\n")
