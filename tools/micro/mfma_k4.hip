// Four plane products per site instead of five.
//
// Kinship needs het_i (hi), het_j (hj) and N = 2 bh - 4 opp - hi - hj.  As a
// bilinear form over the genotype states (R hom-ref, H het, A hom-alt, missing
// = 0) the table of N - hi + hj has rank 2:
//     N - hi + hj = -2 * S3,   S3 = sum_s (H + 2A)_i R_j + (2R + H)_i A_j
// so S1 = H_i.D_j (= hi), S2 = D_i.H_j (= hj) and S3 are FOUR MFMAs per site
// block (S3 takes two into one accumulator) instead of five -- if the two
// multi-valued fragments U = H + 2A and V = 2R + H can be built within the VALU
// budget.  fp4 (E2M1) codes 0001 / 0010 / 0100 are 0.5 / 1 / 2, so a site whose
// H bit sits at nibble bit q and whose A (or R) bit sits at bit q + 1 is the
// value v (H) or 2v (A), v = 2^(q-1): U = t | (A_shifted & mask), where
// t = H & mask is the H fragment the S1 product needs anyway -- one extra VALU
// instruction per dword for U and one for V.  The column side stores its
// nibbles with bits 0<->1 and 2<->3 exchanged, so that one shift puts a site at
// the position whose value is 1/v: every product is exactly 1 or 2 and no MFMA
// needs the block scale (the unscaled form holds the issue port 8 cycles
// instead of 13).
//
//  (A) exactness: 32 x 32 pairs x 256 sites of random genotypes against a CPU
//      count of hi, hj and N.
//  (B) rate: the k-step of a 64 x 64-pairs-per-wave tile (8 b128 LDS reads, the
//      precomputed shifted words, 4 site sets x 16 MFMAs) software pipelined,
//      one wave per SIMD on every CU, against the five-product k-step of the
//      shipped kernel.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_k4.hip -o mfma_k4
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// v_bitop3_b32 truth tables over (a, b, c), index = 4a + 2b + c.
constexpr int kA = 0x08;    // ~a &  b & c   hom-alt  of (het, hom, mask)
constexpr int kR = 0x02;    // ~a & ~b & c   hom-ref
constexpr int kH = 0x20;    //  a & ~b & c   het
constexpr int kD = 0x2A;    // ~(a & b) & c  defined
constexpr int kOrAnd = 0xF8;  // a | (b & c)

template <int KIND>
__device__ __forceinline__ v8i frag(const uint4 x, const uint4 y, uint32_t m) {
  v8i r = {0, 0, 0, 0, 0, 0, 0, 0};
  r[0] = (int)__builtin_amdgcn_bitop3_b32(x.x, y.x, m, KIND);
  r[1] = (int)__builtin_amdgcn_bitop3_b32(x.y, y.y, m, KIND);
  r[2] = (int)__builtin_amdgcn_bitop3_b32(x.z, y.z, m, KIND);
  r[3] = (int)__builtin_amdgcn_bitop3_b32(x.w, y.w, m, KIND);
  return r;
}
// t | (w & m)
__device__ __forceinline__ v8i frag_or(const v8i t, const uint4 w, uint32_t m) {
  v8i r = {0, 0, 0, 0, 0, 0, 0, 0};
  r[0] = (int)__builtin_amdgcn_bitop3_b32((uint32_t)t[0], w.x, m, kOrAnd);
  r[1] = (int)__builtin_amdgcn_bitop3_b32((uint32_t)t[1], w.y, m, kOrAnd);
  r[2] = (int)__builtin_amdgcn_bitop3_b32((uint32_t)t[2], w.z, m, kOrAnd);
  r[3] = (int)__builtin_amdgcn_bitop3_b32((uint32_t)t[3], w.w, m, kOrAnd);
  return r;
}
template <int KIND>
__device__ __forceinline__ uint4 ind(const uint4 x, const uint4 y, uint32_t ones) {
  return make_uint4(__builtin_amdgcn_bitop3_b32(x.x, y.x, ones, KIND),
                    __builtin_amdgcn_bitop3_b32(x.y, y.y, ones, KIND),
                    __builtin_amdgcn_bitop3_b32(x.z, y.z, ones, KIND),
                    __builtin_amdgcn_bitop3_b32(x.w, y.w, ones, KIND));
}
__device__ __forceinline__ uint4 shl1(const uint4 w) { return make_uint4(w.x << 1, w.y << 1, w.z << 1, w.w << 1); }
__device__ __forceinline__ uint4 shr1(const uint4 w) { return make_uint4(w.x >> 1, w.y >> 1, w.z >> 1, w.w >> 1); }
__device__ __forceinline__ uint4 shr2(const uint4 w) { return make_uint4(w.x >> 2, w.y >> 2, w.z >> 2, w.w >> 2); }

__device__ __forceinline__ v16f mma(const v8i a, const v8i b, const v16f c) {
  // fp4 x fp4, scale operands 0 = the unscaled instruction
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 0, 0, 0);
}

// Row side of one 32-row block: the raw planes and what is derived from them
// once per k-step.
struct RowPre {
  uint4 het, hom;      // sites at their natural nibble positions
  uint4 hetB, homB;    // >> 2: sites 2, 3 at positions 0, 1
  uint4 awL, awR, rwL, rwR;  // hom-alt / hom-ref indicator words << 1, >> 1
};
__device__ __forceinline__ void row_pre(RowPre &p, uint32_t ones) {
  const uint4 aw = ind<kA>(p.het, p.hom, ones), rw = ind<kR>(p.het, p.hom, ones);
  p.awL = shl1(aw); p.awR = shr1(aw);
  p.rwL = shl1(rw); p.rwR = shr1(rw);
  p.hetB = shr2(p.het); p.homB = shr2(p.hom);
}
// The same in the two halves the balanced pipeline (MODE 2) spreads over phases:
// A = what site sets 0, 1 need, B = what sets 2, 3 need.
__device__ __forceinline__ void row_pre_a(RowPre &p, uint32_t ones) {
  const uint4 aw = ind<kA>(p.het, p.hom, ones), rw = ind<kR>(p.het, p.hom, ones);
  p.awL = shl1(aw);
  p.rwL = shl1(rw);
}
__device__ __forceinline__ void row_pre_b(RowPre &p, uint32_t ones) {
  // (recomputes the indicator words: 8 more v_bitop3 per block and k-step than
  //  row_pre, in exchange for not keeping them alive across two phases)
  const uint4 aw = ind<kA>(p.het, p.hom, ones), rw = ind<kR>(p.het, p.hom, ones);
  p.awR = shr1(aw);
  p.rwR = shr1(rw);
  p.hetB = shr2(p.het);
  p.homB = shr2(p.hom);
}
// Column side: stored with nibble bits 0<->1, 2<->3 exchanged.
struct ColPre {
  uint4 het, hom;
  uint4 hetA, homA;    // << 1: sites 0, 1 at positions 2, 1
  uint4 hetB, homB;    // >> 1: sites 2, 3 at positions 2, 1
};
__device__ __forceinline__ void col_pre(ColPre &p) {
  p.hetA = shl1(p.het); p.homA = shl1(p.hom);
  p.hetB = shr1(p.het); p.homB = shr1(p.hom);
}
__device__ __forceinline__ void col_pre_a(ColPre &p) { p.hetA = shl1(p.het); p.homA = shl1(p.hom); }
__device__ __forceinline__ void col_pre_b(ColPre &p) { p.hetB = shr1(p.het); p.homB = shr1(p.hom); }
// Fragment sets of site p (0..3) of every nibble.  Row: [0] H (= t), [1] D,
// [2] U = H + 2A, [3] V = 2R + H.  Column: [0] R, [1] A, [2] D, [3] H.
template <int P>
__device__ __forceinline__ void row_frags(v8i f[4], const RowPre &p, uint32_t m0, uint32_t m1,
                                          uint32_t m2) {
  const uint4 het = P < 2 ? p.het : p.hetB, hom = P < 2 ? p.hom : p.homB;
  const uint4 aw = P < 2 ? p.awL : p.awR, rw = P < 2 ? p.rwL : p.rwR;
  const uint32_t mq = (P & 1) ? m1 : m0, mq1 = (P & 1) ? m2 : m1;
  f[0] = frag<kH>(het, hom, mq);
  f[1] = frag<kD>(het, hom, mq);
  f[2] = frag_or(f[0], aw, mq1);
  f[3] = frag_or(f[0], rw, mq1);
}
template <int P>
__device__ __forceinline__ void col_frags(v8i f[4], const ColPre &p, uint32_t m1, uint32_t m2) {
  const uint4 het = P < 2 ? p.hetA : p.hetB, hom = P < 2 ? p.homA : p.homB;
  const uint32_t m = (P & 1) ? m1 : m2;   // position 2 - q
  f[0] = frag<kR>(het, hom, m);
  f[1] = frag<kA>(het, hom, m);
  f[2] = frag<kD>(het, hom, m);
  f[3] = frag<kH>(het, hom, m);
}

// (A) one wave.  rows: [32][2 groups] {het uint4, hom uint4} natural layout;
// cols the same in the column layout.  out: [3][32][32].
__global__ void probe_kernel(const uint4 *rows, const uint4 *cols, float *out) {
  const int l = threadIdx.x;
  const uint32_t m0 = 0x11111111u, m1 = 0x22222222u, m2 = 0x44444444u;
  RowPre rp; ColPre cp;
  rp.het = rows[((l & 31) * 2 + (l >> 5)) * 2]; rp.hom = rows[((l & 31) * 2 + (l >> 5)) * 2 + 1];
  cp.het = cols[((l & 31) * 2 + (l >> 5)) * 2]; cp.hom = cols[((l & 31) * 2 + (l >> 5)) * 2 + 1];
  row_pre(rp, 0xFFFFFFFFu); col_pre(cp);
  v16f s1 = {}, s2 = {}, s3 = {};
#define SET(P) { v8i a[4], b[4]; row_frags<P>(a, rp, m0, m1, m2); col_frags<P>(b, cp, m1, m2); \
    s1 = mma(a[0], b[2], s1); s2 = mma(a[1], b[3], s2); s3 = mma(a[2], b[0], s3); s3 = mma(a[3], b[1], s3); }
  SET(0) SET(1) SET(2) SET(3)
#undef SET
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    out[0 * 1024 + row * 32 + (l & 31)] = s1[r];
    out[1 * 1024 + row * 32 + (l & 31)] = s2[r];
    out[2 * 1024 + row * 32 + (l & 31)] = s3[r];
  }
}

// n MFMAs each followed by v VALU
#define PACE(n, v)                                                            \
  _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {                        \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        \
    if ((v) > 0) __builtin_amdgcn_sched_group_barrier(0x002, (v), 0);         \
  }
#define MMA16(X)                                                              \
  _Pragma("unroll") for (int bi = 0; bi < 2; ++bi)                            \
  _Pragma("unroll") for (int bj = 0; bj < 2; ++bj) {                          \
    acc[bi][bj][0] = mma(X##a[bi][0], X##b[bj][2], acc[bi][bj][0]);           \
    acc[bi][bj][1] = mma(X##a[bi][1], X##b[bj][3], acc[bi][bj][1]);           \
    acc[bi][bj][2] = mma(X##a[bi][2], X##b[bj][0], acc[bi][bj][2]);           \
    acc[bi][bj][2] = mma(X##a[bi][3], X##b[bj][1], acc[bi][bj][2]);           \
  }
#define FRAGS(P, X)                                                           \
  _Pragma("unroll") for (int b = 0; b < 2; ++b) {                             \
    row_frags<P>(X##a[b], R[b], m0, m1, m2);                                  \
    col_frags<P>(X##b[b], C[b], m1, m2);                                      \
  }
// (B) MODE 0: compiler's order.  MODE 1: hand pipeline -- the fragments of site
// set p + 1 are built while the MFMAs of set p issue; the next k-step's LDS
// reads and derived words sit behind sets 2 and 3.
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(const uint4 *src, float *out, int iters,
                                                   unsigned long long *stamps) {
  extern __shared__ uint4 lds[];   // [2 bufs][2 groups][2 planes][256 samples]
  const int tid = threadIdx.x, l = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 2 * 2 * 256; i += 256) lds[i] = src[i];
  __syncthreads();
  const int wr = (wave >> 1) * 64, wc = 128 + (wave & 1) * 64;
  const int g = l >> 5, lr = l & 31;
  uint32_t m0, m1, m2, ones;
  asm volatile("s_mov_b32 %0, 0x11111111" : "=s"(m0));
  asm volatile("s_mov_b32 %0, 0x22222222" : "=s"(m1));
  asm volatile("s_mov_b32 %0, 0x44444444" : "=s"(m2));
  asm volatile("s_mov_b32 %0, -1" : "=s"(ones));
  v16f acc[2][2][3] = {};
  RowPre R[2]; ColPre C[2];
  v8i Xa[2][4], Xb[2][4], Ya[2][4], Yb[2][4];
#define LOAD_RAW(T)                                                           \
  _Pragma("unroll") for (int b = 0; b < 2; ++b) {                             \
    R[b].het = (T)[0 * 256 + wr + b * 32 + lr]; R[b].hom = (T)[1 * 256 + wr + b * 32 + lr]; \
    C[b].het = (T)[0 * 256 + wc + b * 32 + lr]; C[b].hom = (T)[1 * 256 + wc + b * 32 + lr]; \
  }
#define PRE()                                                                 \
  _Pragma("unroll") for (int b = 0; b < 2; ++b) { row_pre(R[b], ones); col_pre(C[b]); }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
      const uint4 *t = lds + (it & 1) * (2 * 2 * 256) + g * (2 * 256);
      LOAD_RAW(t)
      PRE()
      FRAGS(0, X) MMA16(X)
      FRAGS(1, Y) MMA16(Y)
      FRAGS(2, X) MMA16(X)
      FRAGS(3, Y) MMA16(Y)
    }
  } else if (MODE == 2 || MODE == 3) {
    // Balanced pipeline: every phase carries 16 MFMAs and ~88-96 VALU.  The words
    // sets 2, 3 need are derived while set 0 multiplies; the raw words are dead
    // after that, so the next k-step's LDS reads go out while set 2 multiplies
    // and what its sets 0, 1 need is derived behind them.
    RowPre Rn[2]; ColPre Cn[2];
    {
      const uint4 *t = lds + g * (2 * 256);
      LOAD_RAW(t)
      _Pragma("unroll") for (int b = 0; b < 2; ++b) { row_pre_a(R[b], ones); col_pre_a(C[b]); }
      FRAGS(0, X)
    }
    for (int it = 0; it < iters; ++it) {
      const uint4 *t = lds + ((it + 1) & 1) * (2 * 2 * 256) + g * (2 * 256);
      // set 0 multiplies; B-side words, then set 1
      _Pragma("unroll") for (int b = 0; b < 2; ++b) { row_pre_b(R[b], ones); col_pre_b(C[b]); }
      FRAGS(1, Y)
      MMA16(X)
      PACE(8, 6) PACE(8, 7)
      __builtin_amdgcn_sched_barrier(0);
      // set 1 multiplies; set 2
      FRAGS(2, X)
      MMA16(Y)
      PACE(16, 4)
      __builtin_amdgcn_sched_barrier(0);
      // set 2 multiplies; next k-step's raw words, set 3
      if (MODE == 3) __syncthreads();
      _Pragma("unroll") for (int b = 0; b < 2; ++b) {
        Rn[b].het = t[0 * 256 + wr + b * 32 + lr]; Rn[b].hom = t[1 * 256 + wr + b * 32 + lr];
        Cn[b].het = t[0 * 256 + wc + b * 32 + lr]; Cn[b].hom = t[1 * 256 + wc + b * 32 + lr];
      }
      FRAGS(3, Y)
      MMA16(X)
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
      PACE(16, 4)
      __builtin_amdgcn_sched_barrier(0);
      // set 3 multiplies; A-side words of the next k-step, its set 0
      _Pragma("unroll") for (int b = 0; b < 2; ++b) {
        R[b].het = Rn[b].het; R[b].hom = Rn[b].hom; C[b].het = Cn[b].het; C[b].hom = Cn[b].hom;
        row_pre_a(R[b], ones); col_pre_a(C[b]);
      }
      FRAGS(0, X)
      MMA16(Y)
      PACE(8, 6) PACE(8, 7)
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    {
      const uint4 *t = lds + g * (2 * 256);
      LOAD_RAW(t)
      PRE()
      FRAGS(0, X)
    }
    for (int it = 0; it < iters; ++it) {
      const uint4 *t = lds + ((it + 1) & 1) * (2 * 2 * 256) + g * (2 * 256);
      // set 0 multiplies, set 1 is built
      FRAGS(1, Y)
      MMA16(X)
      PACE(16, 4)
      __builtin_amdgcn_sched_barrier(0);
      // set 1 multiplies, set 2 is built
      FRAGS(2, X)
      MMA16(Y)
      PACE(16, 4)
      __builtin_amdgcn_sched_barrier(0);
      // set 2 multiplies, set 3 is built; the raw words are dead after this
      FRAGS(3, Y)
      MMA16(X)
      PACE(16, 4)
      __builtin_amdgcn_sched_barrier(0);
      // set 3 multiplies; next k-step: LDS reads, derived words, set 0
      LOAD_RAW(t)
      PRE()
      FRAGS(0, X)
      MMA16(Y)
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
      PACE(2, 0) PACE(14, 12)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
  float s = 0;
  for (int bi = 0; bi < 2; ++bi)
    for (int bj = 0; bj < 2; ++bj)
      for (int q = 0; q < 3; ++q)
        for (int r = 0; r < 16; ++r) s += acc[bi][bj][q][r];
  for (int bi = 0; bi < 2; ++bi) for (int p = 0; p < 4; ++p) s += (float)Xa[bi][p][0];
  out[blockIdx.x * 256 + tid] = s;
}

typedef void (*kernel_t)(const uint4 *, float *, int, unsigned long long *);
int rate_of(kernel_t kern, int mode, const uint4 *d_src, float *d_out) {
  const int iters = 2000, grid = 256;
  const size_t lds_bytes = 2 * 2 * 2 * 256 * sizeof(uint4);
  unsigned long long *d_stamps;
  CHECK(hipMalloc(&d_stamps, grid * 16));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int r = 0; r < 400; ++r) kern<<<grid, 256, lds_bytes>>>(d_src, d_out, iters, d_stamps);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 20; ++r) kern<<<grid, 256, lds_bytes>>>(d_src, d_out, iters, d_stamps);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 20;
  std::vector<unsigned long long> st(grid * 2);
  CHECK(hipMemcpy(st.data(), d_stamps, grid * 16, hipMemcpyDeviceToHost));
  std::vector<double> clk, cyc;
  for (int b = 0; b < grid; ++b) {
    clk.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 100e6);
    cyc.push_back((double)st[2 * b] / iters);
  }
  std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
  // one k-step = 64 x 64 pairs x 256 sites per wave
  const double pair_sites = (double)iters * 64 * 64 * 256 * grid * 4;
  printf("k4 mode=%d: %.3f ms  in-kernel clock %.3f GHz  %.0f shader cycles per k-step "
         "(%.1f per MFMA, 64 MFMAs)  = %.2fe9 pairs/s at 100032 sites\n",
         mode, ms, clk[grid / 2] / 1e9, cyc[grid / 2], cyc[grid / 2] / 64,
         pair_sites / (ms * 1e-3) / 100032 / 1e9);
  CHECK(hipFree(d_stamps));
  return 0;
}

int main() {
  // (A) random genotypes: 0 R, 1 H, 2 A, 3 missing
  const int K = 256;
  std::vector<uint8_t> gi(32 * K), gj(32 * K);
  srand(11);
  for (auto &g : gi) g = rand() & 3;
  for (auto &g : gj) g = (rand() >> 3) & 3;
  // reference encoding (cuking.cu:688-697): (het, hom) = 00 R, 10 H, 01 A, 11 missing
  auto het_bit = [](int g) { return g == 1 || g == 3; };
  auto hom_bit = [](int g) { return g == 2 || g == 3; };
  // lane (sample s, group g) holds sites [g*128, g*128+128): dword d, bit b = site g*128 + 32d + b
  std::vector<uint32_t> rows(32 * 2 * 8), cols(32 * 2 * 8);
  for (int s = 0; s < 32; ++s)
    for (int g = 0; g < 2; ++g)
      for (int d = 0; d < 4; ++d) {
        uint32_t rh = 0, rm = 0, ch = 0, cm = 0;
        for (int b = 0; b < 32; ++b) {
          const int site = g * 128 + 32 * d + b;
          rh |= (uint32_t)het_bit(gi[s * K + site]) << b;
          rm |= (uint32_t)hom_bit(gi[s * K + site]) << b;
          const int cb = b ^ 1;   // column layout: bits 0<->1, 2<->3 of every nibble
          ch |= (uint32_t)het_bit(gj[s * K + site]) << cb;
          cm |= (uint32_t)hom_bit(gj[s * K + site]) << cb;
        }
        rows[((s * 2 + g) * 2 + 0) * 4 + d] = rh; rows[((s * 2 + g) * 2 + 1) * 4 + d] = rm;
        cols[((s * 2 + g) * 2 + 0) * 4 + d] = ch; cols[((s * 2 + g) * 2 + 1) * 4 + d] = cm;
      }
  uint4 *d_rows, *d_cols; float *d_out;
  CHECK(hipMalloc(&d_rows, rows.size() * 4));
  CHECK(hipMalloc(&d_cols, cols.size() * 4));
  CHECK(hipMalloc(&d_out, 256 * 256 * 4));
  CHECK(hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_cols, cols.data(), cols.size() * 4, hipMemcpyHostToDevice));
  probe_kernel<<<1, 64>>>(d_rows, d_cols, d_out);
  CHECK(hipDeviceSynchronize());
  std::vector<float> out(3 * 1024);
  CHECK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      int hi = 0, hj = 0, bh = 0, opp = 0;
      for (int s = 0; s < K; ++s) {
        const int a = gi[i * K + s], b = gj[j * K + s];
        if (a == 3 || b == 3) continue;
        hi += a == 1; hj += b == 1; bh += (a == 1 && b == 1);
        opp += (a == 0 && b == 2) || (a == 2 && b == 0);
      }
      const int n = 2 * bh - 4 * opp - hi - hj;
      const float s1 = out[i * 32 + j], s2 = out[1024 + i * 32 + j], s3 = out[2048 + i * 32 + j];
      if (s1 != (float)hi || s2 != (float)hj || -2.f * s3 + s1 - s2 != (float)n) {
        if (bad < 8) printf("mismatch (%d,%d): s1 %g hi %d  s2 %g hj %d  s3 %g N %d\n", i, j, s1, hi, s2, hj, s3, n);
        ++bad;
      }
    }
  printf("(A) four-product exactness: %s (%d of 1024 pairs wrong)\n", bad ? "FAIL" : "OK", bad);

  // (B)
  std::vector<uint32_t> tile(2 * 2 * 2 * 256 * 4);
  for (auto &w : tile) w = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
  uint4 *d_src;
  CHECK(hipMalloc(&d_src, tile.size() * 4));
  CHECK(hipMemcpy(d_src, tile.data(), tile.size() * 4, hipMemcpyHostToDevice));
  const size_t lds_bytes = 2 * 2 * 2 * 256 * sizeof(uint4);
  CHECK(hipFuncSetAttribute((const void *)rate_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  CHECK(hipFuncSetAttribute((const void *)rate_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  CHECK(hipFuncSetAttribute((const void *)rate_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  CHECK(hipFuncSetAttribute((const void *)rate_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  if (rate_of(rate_kernel<0>, 0, d_src, d_out)) return 1;
  if (rate_of(rate_kernel<1>, 1, d_src, d_out)) return 1;
  if (rate_of(rate_kernel<2>, 2, d_src, d_out)) return 1;
  if (rate_of(rate_kernel<3>, 3, d_src, d_out)) return 1;
  return bad != 0;
}
