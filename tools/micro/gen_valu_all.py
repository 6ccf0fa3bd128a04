# generates tools/micro/valu_all.hip: pure streams + mixes with explicit registers and a
# measured shader clock (s_memtime / s_memrealtime), all in ONE run.
def and_(d, a, b): return f"v_and_b32 v{d}, v{a}, v{b}"
def bcnt(acc, s): return f"v_bcnt_u32_b32 v{acc}, v{s}, v{acc}"
def bit3(d, a, b, c): return f"v_bitop3_b32 v{d}, v{a}, v{b}, v{c} bitop3:0x28"
variants = {}
variants["and"] = ([and_(48 + k % 8, 32 + k % 8, 40 + (k + 1) % 8) for k in range(32)], 32)
variants["bitop3"] = ([bit3(48 + k % 8, 32 + k % 8, 40 + (k + 1) % 8, 32 + (k + 3) % 8) for k in range(32)], 32)
variants["bcnt"] = ([bcnt(72 + k % 16, 32 + k % 16) for k in range(32)], 32)
# KING mix per pair-word: t=and; hh+=bcnt(t); u=bitop3(..t); opp+=bcnt(u); 3x(and,bcnt)
def king_pair(p, five=True):
    t, u = 48 + (p % 4) * 2, 49 + (p % 4) * 2
    acc = 56 + p * 5
    a, b = 32 + (p % 4), 40 + (p // 4)
    seq = [and_(t, a, b)]
    if five: seq.append(bcnt(acc, t))
    seq += [bit3(u, a, b, t), bcnt(acc + 1, u), and_(t, a, b + 1 if b < 47 else 40), bcnt(acc + 2, t),
            and_(u, a + 4 if a < 36 else 32, b), bcnt(acc + 3, u), and_(t, a, b), bcnt(acc + 4, t)]
    return seq
k55 = sum((king_pair(p) for p in range(4)), [])
k54 = sum((king_pair(p, False) for p in range(4)), [])
variants["king5+5"] = (k55, len(k55))
variants["king5+4"] = (k54, len(k54))
# same 5+5 multiset, all logic first then all bcnt
logic = [x for x in k55 if not x.startswith("v_bcnt")]
cnts = [x for x in k55 if x.startswith("v_bcnt")]
variants["king5+5_batched"] = (logic + cnts, len(k55))
variants["and+bcnt_alt"] = (sum(([and_(48 + k % 8, 32 + k % 8, 40 + (k + 1) % 8), bcnt(72 + k % 16, 48 + k % 8)] for k in range(16)), []), 32)
clob = ",".join(f'"v{r}"' for r in range(32, 96))
src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdint>', '#include <vector>',
'#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\\n", #x, hipGetErrorString(e)); return 1; } } while (0)',
'constexpr int ITERS = 4096;', f'#define CLOB {clob}']
init = "\\n ".join([f"v_mov_b32 v{r}, %0" for r in range(32, 48)] + [f"v_mov_b32 v{r}, 0" for r in range(48, 96)])
names = []
for name, (seq, n) in variants.items():
    cname = name.replace("+", "p")
    names.append((name, cname, n))
    body = "\\n ".join(seq)
    src.append(f'''__global__ __launch_bounds__(256) void k_{cname}(uint64_t *clk, uint32_t *out, uint32_t seed) {{
  asm volatile("{init}" :: "v"(seed + threadIdx.x) : CLOB);
  uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; ++it) {{ asm volatile("{body}" ::: CLOB); }}
  uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t r; asm volatile("v_add_u32 %0, v72, v95\\n v_add_u32 %0, %0, v56" : "=v"(r) :: CLOB);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0) {{ clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }}
}}''')
src.append('''template <typename K>
int run(const char *name, K kern, uint64_t *dclk, uint32_t *d, int n_instr) {
  printf("%-18s", name);
  for (int blocks_per_cu : {2, 4, 8}) {
    const int grid = 256 * blocks_per_cu;
    for (int r = 0; r < 3; ++r) kern<<<grid, 256>>>(dclk, d, 12345 + r);
    CHECK(hipDeviceSynchronize());
    std::vector<uint64_t> h(2 * grid);
    CHECK(hipMemcpy(h.data(), dclk, h.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (int b = 0; b < grid; ++b) { cyc += h[2 * b]; real += h[2 * b + 1]; }
    const double ghz = cyc / real * 0.1;          // s_memrealtime ticks at 100 MHz
    // per-SIMD cycles per instruction: a block's 4 waves sit on 4 SIMDs; with blocks_per_cu
    // waves per SIMD, in-kernel cycles / (instrs per wave * waves per SIMD)
    const double per_wave = cyc / grid / ((double)ITERS * n_instr);
    printf("  w%d: %5.2f cyc/instr @%.2f GHz", blocks_per_cu, per_wave / blocks_per_cu, ghz);
  }
  printf("\\n");
  return 0;
}
int main() {
  uint32_t *d; uint64_t *dclk;
  CHECK(hipMalloc(&d, 256 * 8 * 256 * 4)); CHECK(hipMalloc(&dclk, 256 * 8 * 2 * 8));
  // warm the clocks up
  for (int r = 0; r < 200; ++r) k_and<<<2048, 256>>>(dclk, d, r);
  CHECK(hipDeviceSynchronize());''')
for name, cname, n in names:
    src.append(f'  if (run("{name}", k_{cname}, dclk, d, {n})) return 1;')
src.append('  return 0;\n}')
open(__import__('os').path.dirname(__import__('os').path.abspath(__file__)) + '/valu_all.hip', 'w').write("\n".join(src) + "\n")
