#!/usr/bin/env python3
"""Generates tools/micro/king_step.hip: the tiled kernel's k-step (16 pairs x 10 VALU,
80 accumulators, operands in 8 register quads) in several instruction orders, plus pure
streams, all timed in one run.  Registers: acc v0..v79, ri[x] = v[80+4x..], cj[y] =
v[96+4y..], temps v112..v127.  Component order in a quad: x=H, y=A, z=Hom, w=D."""
H, A, HOM, D = 0, 1, 2, 3
ri = lambda x, c: 80 + 4 * x + c
cj = lambda y, c: 96 + 4 * y + c
acc = lambda x, y, k: (x * 4 + y) * 5 + k     # hh, opp, bh, hi, hj
def and_(d, a, b): return f"v_and_b32 v{d}, v{a}, v{b}"
def bcnt(a, s): return f"v_bcnt_u32_b32 v{a}, v{s}, v{a}"
def bit3(d, a, b, c): return f"v_bitop3_b32 v{d}, v{a}, v{b}, v{c} bitop3:0x28"

def pair_ops(x, y, t0, t1):
    """the 10 ops of one pair, as (kind, text) in the compiler's order"""
    return [and_(t0, cj(y, HOM), ri(x, HOM)), bcnt(acc(x, y, 0), t0),
            bit3(t0, ri(x, A), cj(y, A), t0), bcnt(acc(x, y, 1), t0),
            and_(t1, cj(y, H), ri(x, H)), bcnt(acc(x, y, 2), t1),
            and_(t0, cj(y, D), ri(x, H)), bcnt(acc(x, y, 3), t0),
            and_(t1, ri(x, D), cj(y, H)), bcnt(acc(x, y, 4), t1)]

orders = {}
# A: pair by pair (compiler-like), 2 temps alternating per pair
seq = []
for x in range(4):
    for y in range(4):
        p = x * 4 + y
        seq += pair_ops(x, y, 112 + 2 * (p % 6), 113 + 2 * (p % 6))
orders["pairwise"] = seq
# B: by counter type across the 16 pairs: 16 logic ops (distinct temps) then 16 bcnts
seq = []
for k, half in [(k, h) for h in range(2) for k in range(5)]:
    logic, cnt = [], []
    for x in range(2 * half, 2 * half + 2):
        for y in range(4):
            p = (x % 2) * 4 + y
            t = 112 + p
            if k == 0: logic.append(and_(t, cj(y, HOM), ri(x, HOM)))
            # opp needs hom_both again: recompute via bitop3 on (A_i, A_j, t) -> keep t from k==0?  no: use 2 ops
            if k == 1: logic += [and_(t, cj(y, HOM), ri(x, HOM)), bit3(t, ri(x, A), cj(y, A), t)]
            if k == 2: logic.append(and_(t, cj(y, H), ri(x, H)))
            if k == 3: logic.append(and_(t, cj(y, D), ri(x, H)))
            if k == 4: logic.append(and_(t, ri(x, D), cj(y, H)))
            cnt.append(bcnt(acc(x, y, k), t))
    seq += logic + cnt
orders["by_counter_11ops"] = seq          # 11 ops per pair (hom AND done twice)
# C: row-stationary: for each x, for each counter, the 4 y's
seq = []
for x in range(4):
    ts = [112 + y for y in range(4)]
    us = [116 + y for y in range(4)]
    seq += [and_(ts[y], cj(y, HOM), ri(x, HOM)) for y in range(4)]
    seq += [bcnt(acc(x, y, 0), ts[y]) for y in range(4)]
    seq += [bit3(ts[y], ri(x, A), cj(y, A), ts[y]) for y in range(4)]
    seq += [and_(us[y], cj(y, H), ri(x, H)) for y in range(4)]
    seq += [bcnt(acc(x, y, 1), ts[y]) for y in range(4)]
    seq += [and_(ts[y], cj(y, D), ri(x, H)) for y in range(4)]
    seq += [bcnt(acc(x, y, 2), us[y]) for y in range(4)]
    seq += [and_(us[y], ri(x, D), cj(y, H)) for y in range(4)]
    seq += [bcnt(acc(x, y, 3), ts[y]) for y in range(4)]
    seq += [bcnt(acc(x, y, 4), us[y]) for y in range(4)]
orders["row_groups_of_4"] = seq
# D: groups of 8 pairs: 8 logic ops then 8 bcnts, per counter
seq = []
tt = lambda x, y: 112 + (x % 2) * 4 + y
for half in range(2):
    xs = range(2 * half, 2 * half + 2)
    seq += [and_(tt(x, y), cj(y, HOM), ri(x, HOM)) for x in xs for y in range(4)]
    seq += [bcnt(acc(x, y, 0), tt(x, y)) for x in xs for y in range(4)]
    seq += [bit3(tt(x, y), ri(x, A), cj(y, A), tt(x, y)) for x in xs for y in range(4)]
    seq += [bcnt(acc(x, y, 1), tt(x, y)) for x in xs for y in range(4)]
    for k, ca, cb in ((2, H, H), (3, H, D), (4, D, H)):
        seq += [and_(tt(x, y), ri(x, ca), cj(y, cb)) for x in xs for y in range(4)]
        seq += [bcnt(acc(x, y, k), tt(x, y)) for x in xs for y in range(4)]
orders["groups_of_8"] = seq
# E: the 9-op form (no hh popcount): what a main pass without IBS2 would issue
seq = []
for x in range(4):
    for y in range(4):
        p = x * 4 + y
        ops = pair_ops(x, y, 112 + 2 * (p % 6), 113 + 2 * (p % 6))
        seq += [ops[0]] + ops[2:]
orders["pairwise_9ops"] = seq
# F: 8 ops: only two AND+BCNT pairs replaced... (bh, opp, hi, hj without reusing): same as E
# pure references (160 instrs each)
orders["pure_and"] = [and_(112 + k % 8, ri(k % 4, k % 4), cj((k // 4) % 4, (k + 1) % 4)) for k in range(160)]
orders["pure_bcnt"] = [bcnt(k % 80, 80 + k % 32) for k in range(160)]
orders["pure_bitop3"] = [bit3(112 + k % 8, ri(k % 4, k % 4), cj((k // 4) % 4, (k + 1) % 4), ri((k + 2) % 4, (k + 3) % 4)) for k in range(160)]

clob = ",".join(f'"v{r}"' for r in range(0, 124))
src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdint>',
'#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\\n", #x, hipGetErrorString(e)); return 1; } } while (0)',
'constexpr int ITERS = 1024;', f'#define CLOB {clob}']
init = "\\n ".join([f"v_mov_b32 v{r}, 0" for r in range(0, 80)] + [f"v_mov_b32 v{r}, s0" for r in range(80, 124)])
for name, seq in orders.items():
    body = "\\n ".join(seq)
    src.append(f'''__global__ __launch_bounds__(256, 4) void k_{name}(uint32_t *out, uint32_t seed) {{
  uint32_t tid = threadIdx.x;
  asm volatile("s_mov_b32 s0, %0\\n {init}\\n v_add_u32 v80, v80, %1\\n v_add_u32 v97, v97, %1" :: "s"(seed), "v"(tid) : CLOB, "s0");
  for (int it = 0; it < ITERS; ++it) {{ asm volatile("{body}" ::: CLOB); }}
  uint32_t r; asm volatile("v_add_u32 %0, v0, v79\\n v_add_u32 %0, %0, v40\\n v_add_u32 %0, %0, v112" : "=v"(r) :: CLOB);
  out[blockIdx.x * blockDim.x + tid] = r;
}}''')
src.append('''template <typename K>
int run(const char *name, K kern, uint32_t *d, int n_instr) {
  printf("%-18s %3d instr", name, n_instr);
  for (int blocks_per_cu : {2, 4}) {
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) kern<<<grid, 256>>>(d, 12345);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 10; ++r) kern<<<grid, 256>>>(d, 12345 + r);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    const double per_simd = (double)grid * 4 * ITERS / 1024.0;   // k-steps per SIMD
    printf("  w%d: %7.1f cyc/step %5.2f cyc/instr", blocks_per_cu, ms * 1e-3 * 2.4e9 / per_simd,
           ms * 1e-3 * 2.4e9 / per_simd / n_instr);
  }
  printf("\\n");
  return 0;
}
int main() {
  uint32_t *d; CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));''')
for name, seq in orders.items():
    src.append(f'  if (run("{name}", k_{name}, d, {len(seq)})) return 1;')
src.append('  return 0;\n}')
open(__import__('os').path.dirname(__import__('os').path.abspath(__file__)) + '/king_step.hip', 'w').write("\n".join(src) + "\n")
print({k: len(v) for k, v in orders.items()})
