// Probe for the block-scaled fp4 MFMA as a bit-plane AND/popcount engine on
// gfx950: popcount(x & y) over K sites == dot product of the two bit vectors.
//  (A) exactness + operand layout: 32 rows x 32 cols x 256 sites of random
//      bits through v_mfma_scale_f32_32x32x64_f8f6f4 with the one-AND
//      expansion (bit at nibble position f -> fp4 value 2^(f-1), undone by the
//      E8M0 block scale), checked against a CPU popcount.
//  (B) rate: the k-step of a 64x64-pairs-per-wave tile (16 b128 LDS reads,
//      the expansions, 5 plane products x 4 block pairs x FPW MFMAs) in a
//      loop, one wave per SIMD on every CU.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_fp4.hip -o mfma_fp4
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// fragment f of four 32-site words: nibble q of dword d holds site 4q+f of
// word d as fp4 (E2M1) code 1<<f (f<3: 0.5, 1, 2) or, for f == 3, shifted
// down to code 1 (0.5).
template <int F>
__device__ __forceinline__ v8i expand(const uint4 w) {
  v8i r = {0, 0, 0, 0, 0, 0, 0, 0};
  if (F < 3) {
    const uint32_t m = 0x11111111u << F;
    r[0] = w.x & m; r[1] = w.y & m; r[2] = w.z & m; r[3] = w.w & m;
  } else {
    const uint32_t m = 0x11111111u;
    r[0] = (w.x >> 3) & m; r[1] = (w.y >> 3) & m;
    r[2] = (w.z >> 3) & m; r[3] = (w.w >> 3) & m;
  }
  return r;
}

// E8M0 scale that brings fragment F's values back to 1.0: 2^(1-F), F==3 -> 2.
template <int F>
__device__ __forceinline__ int scale_of() {
  return F == 0 ? 128 : F == 1 ? 127 : F == 2 ? 126 : 128;
}

template <int F>
__device__ __forceinline__ v16f mma(const v8i a, const v8i b, const v16f c) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
      a, b, c, 4 /*A fp4*/, 4 /*B fp4*/, 0, scale_of<F>(), 0, scale_of<F>());
}

// (A) one wave.  rows/cols: [32][2 groups][4 words].
__global__ void probe_kernel(const uint4 *rows, const uint4 *cols, float *out) {
  const int l = threadIdx.x;
  const uint4 a = rows[(l & 31) * 2 + (l >> 5)];
  const uint4 b = cols[(l & 31) * 2 + (l >> 5)];
  v16f acc = {};
  acc = mma<0>(expand<0>(a), expand<0>(b), acc);
  acc = mma<1>(expand<1>(a), expand<1>(b), acc);
  acc = mma<2>(expand<2>(a), expand<2>(b), acc);
  acc = mma<3>(expand<3>(a), expand<3>(b), acc);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    out[row * 32 + (l & 31)] = acc[r];
  }
}

// (B) 4 waves, wave tile 64x64 (2x2 blocks of 32x32), LDS tile
// [buf][group][plane][256 samples] uint4.
// MODE 0: compiler's order.  MODE 1: MFMAs only (loop-invariant fragments): the
// matrix-pipe ceiling at the clock the chip holds.  MODE 2: one MFMA, then
// four VALU, ... requested with sched_group_barrier.
template <int FPW, int MODE>
__global__ __launch_bounds__(256) void rate_kernel(const uint4 *src, float *out, int iters,
                                                   unsigned long long *stamps) {
  extern __shared__ uint4 lds[];
  const int tid = threadIdx.x, l = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 2 * 4 * 256; i += 256) lds[i] = src[i];
  __syncthreads();
  const int wr = (wave >> 1) * 64, wc = 128 + (wave & 1) * 64;
  const int g = l >> 5, lr = l & 31;
  v16f acc[2][2][4] = {};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    const uint4 *t = lds + (MODE == 1 ? 0 : (it & 1)) * (2 * 4 * 256) + g * (4 * 256);
    uint4 A[2][4], B[2][4];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        A[bi][p] = t[p * 256 + wr + bi * 32 + lr];
        B[bi][p] = t[p * 256 + wc + bi * 32 + lr];
      }
#define STEP(F)                                                              \
    if (F < FPW) {                                                           \
      v8i Af[2][4], Bf[2][4];                                                \
      _Pragma("unroll") for (int bi = 0; bi < 2; ++bi)                       \
      _Pragma("unroll") for (int p = 0; p < 4; ++p) {                        \
        Af[bi][p] = expand<F>(A[bi][p]);                                     \
        Bf[bi][p] = expand<F>(B[bi][p]);                                     \
      }                                                                      \
      _Pragma("unroll") for (int bi = 0; bi < 2; ++bi)                       \
      _Pragma("unroll") for (int bj = 0; bj < 2; ++bj) {                     \
        acc[bi][bj][0] = mma<F>(Af[bi][0], Bf[bj][1], acc[bi][bj][0]);       \
        acc[bi][bj][0] = mma<F>(Af[bi][1], Bf[bj][0], acc[bi][bj][0]);       \
        acc[bi][bj][1] = mma<F>(Af[bi][2], Bf[bj][2], acc[bi][bj][1]);       \
        acc[bi][bj][2] = mma<F>(Af[bi][2], Bf[bj][3], acc[bi][bj][2]);       \
        acc[bi][bj][3] = mma<F>(Af[bi][3], Bf[bj][2], acc[bi][bj][3]);       \
      }                                                                      \
      if (MODE == 2) {                                                       \
        _Pragma("unroll") for (int i = 0; i < 20; ++i) {                     \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 \
          __builtin_amdgcn_sched_group_barrier(0x002, F == 3 ? 7 : 3, 0);    \
        }                                                                    \
      }                                                                      \
    }
    STEP(0) STEP(1) STEP(2) STEP(3)
#undef STEP
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
      for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) s += acc[bi][bj][q][r];
  out[blockIdx.x * 256 + tid] = s;
}


// MODE 3 (FPW = 4 only): software pipelined by hand.  The fragments of step
// f + 1 are produced while the MFMAs of step f issue (two fragment sets), so
// no MFMA reads a register written just before it; the next iteration's LDS
// reads and its f = 0 expansion sit behind the MFMAs of f = 3.
#define MMA1(F, X, PA, PB, Q)                                                 \
  _Pragma("unroll") for (int bi = 0; bi < 2; ++bi)                            \
  _Pragma("unroll") for (int bj = 0; bj < 2; ++bj)                            \
    acc[bi][bj][Q] = mma<F>(X##a[bi][PA], X##b[bj][PB], acc[bi][bj][Q]);
#define MMA5(F, X)                                                            \
  MMA1(F, X, 0, 1, 0) MMA1(F, X, 2, 2, 1) MMA1(F, X, 2, 3, 2)                 \
  MMA1(F, X, 3, 2, 3) MMA1(F, X, 1, 0, 0)
#define EXPAND(F, X)                                                          \
  _Pragma("unroll") for (int bi = 0; bi < 2; ++bi)                            \
  _Pragma("unroll") for (int p = 0; p < 4; ++p) {                             \
    X##a[bi][p] = expand<F>(A[bi][p]);                                        \
    X##b[bi][p] = expand<F>(B[bi][p]);                                        \
  }
#define LOAD_RAW(T)                                                           \
  _Pragma("unroll") for (int bi = 0; bi < 2; ++bi)                            \
  _Pragma("unroll") for (int p = 0; p < 4; ++p) {                             \
    A[bi][p] = (T)[p * 256 + wr + bi * 32 + lr];                              \
    B[bi][p] = (T)[p * 256 + wc + bi * 32 + lr];                              \
  }
// n MFMAs each followed by v VALU
#define PACE(n, v)                                                            \
  _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {                        \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        \
    if ((v) > 0) __builtin_amdgcn_sched_group_barrier(0x002, (v), 0);         \
  }
template <int VARIANT>
__global__ __launch_bounds__(256) void piped_kernel(const uint4 *src, float *out, int iters,
                                                    unsigned long long *stamps) {
  extern __shared__ uint4 lds[];
  const int tid = threadIdx.x, l = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 2 * 4 * 256; i += 256) lds[i] = src[i];
  __syncthreads();
  const int wr = (wave >> 1) * 64, wc = 128 + (wave & 1) * 64;
  const int g = l >> 5, lr = l & 31;
  v16f acc[2][2][4] = {};
  uint4 A[2][4], B[2][4];
  v8i Xa[2][4], Xb[2][4], Ya[2][4], Yb[2][4];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  {
    const uint4 *t = lds + g * (4 * 256);
    LOAD_RAW(t)
    EXPAND(0, X)
  }
  for (int it = 0; it < iters; ++it) {
    const uint4 *t = lds + ((it + 1) & 1) * (2 * 4 * 256) + g * (4 * 256);
    EXPAND(1, Y)
    MMA5(0, X)
    PACE(16, 4) PACE(4, 0)
    EXPAND(2, X)
    MMA5(1, Y)
    PACE(16, 4) PACE(4, 0)
    EXPAND(3, Y)
    MMA5(2, X)
    if (VARIANT == 0) { PACE(20, 6) PACE(0, 0) } else { PACE(16, 8) PACE(4, 0) }
    LOAD_RAW(t)
    EXPAND(0, X)
    MMA5(3, Y)
    __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
    PACE(6, 0) PACE(13, 5) PACE(1, 0)
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
      for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) s += acc[bi][bj][q][r];
  for (int bi = 0; bi < 2; ++bi) for (int p = 0; p < 4; ++p) s += (float)Xa[bi][p][0];
  out[blockIdx.x * 256 + tid] = s;
}

typedef void (*kernel_t)(const uint4 *, float *, int, unsigned long long *);
int rate_of(kernel_t kern, int FPW, int MODE, const uint4 *d_src, float *d_out) {
  const int iters = 2000, grid = 256;
  const size_t lds_bytes = 2 * 2 * 4 * 256 * sizeof(uint4);
  CHECK(hipFuncSetAttribute((const void *)kern,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  unsigned long long *d_stamps;
  CHECK(hipMalloc(&d_stamps, grid * 16));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  // >= 1.5 s of back-to-back launches first: the clock the chip holds under
  // this load, not the one it had when idle.
  for (int r = 0; r < 500; ++r) kern<<<grid, 256, lds_bytes>>>(d_src, d_out, iters, d_stamps);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 20; ++r) kern<<<grid, 256, lds_bytes>>>(d_src, d_out, iters, d_stamps);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 20;
  const double mfma_per_wave = (double)iters * 20 * FPW;
  const double macs = mfma_per_wave * 32 * 32 * 64 * grid * 4;
  const double pair_sites = macs / 5;
  std::vector<unsigned long long> st(grid * 2);
  CHECK(hipMemcpy(st.data(), d_stamps, grid * 16, hipMemcpyDeviceToHost));
  std::vector<double> clk, cyc;
  for (int b = 0; b < grid; ++b) {
    clk.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 100e6);
    cyc.push_back((double)st[2 * b] / mfma_per_wave);
  }
  std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
  printf("FPW=%d mode=%d: %.3f ms  in-kernel clock %.3f GHz  %.1f shader cycles/MFMA  "
         "%.2f PMAC/s  (= %.2fe9 pairs/s at 100032 sites)\n",
         FPW, MODE, ms, clk[grid / 2] / 1e9, cyc[grid / 2], macs / (ms * 1e-3) / 1e15,
         pair_sites / (ms * 1e-3) / 100032 / 1e9);
  CHECK(hipFree(d_stamps));
  return 0;
}

int main() {
  // (A)
  std::vector<uint32_t> rows(32 * 8), cols(32 * 8);
  srand(7);
  for (auto &w : rows) w = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
  for (auto &w : cols) w = ((uint32_t)rand() << 17) ^ (uint32_t)rand();
  uint4 *d_rows, *d_cols; float *d_out;
  CHECK(hipMalloc(&d_rows, rows.size() * 4));
  CHECK(hipMalloc(&d_cols, cols.size() * 4));
  CHECK(hipMalloc(&d_out, 256 * 256 * 4));
  CHECK(hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_cols, cols.data(), cols.size() * 4, hipMemcpyHostToDevice));
  probe_kernel<<<1, 64>>>(d_rows, d_cols, d_out);
  CHECK(hipDeviceSynchronize());
  std::vector<float> out(32 * 32);
  CHECK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      int want = 0;
      for (int w = 0; w < 8; ++w) want += __builtin_popcount(rows[i * 8 + w] & cols[j * 8 + w]);
      if (out[i * 32 + j] != (float)want) {
        if (bad < 8) printf("mismatch (%d,%d): got %g want %d\n", i, j, out[i * 32 + j], want);
        ++bad;
      }
    }
  printf("(A) layout/exactness: %s (%d of 1024 wrong)\n", bad ? "FAIL" : "OK", bad);

  // (B)
  std::vector<uint32_t> tile(2 * 2 * 4 * 256 * 4);
  for (auto &w : tile) w = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
  uint4 *d_src;
  CHECK(hipMalloc(&d_src, tile.size() * 4));
  CHECK(hipMemcpy(d_src, tile.data(), tile.size() * 4, hipMemcpyHostToDevice));
  if (rate_of(rate_kernel<4, 1>, 4, 1, d_src, d_out)) return 1;
  if (rate_of(rate_kernel<4, 0>, 4, 0, d_src, d_out)) return 1;
  if (rate_of(piped_kernel<0>, 4, 30, d_src, d_out)) return 1;
  if (rate_of(piped_kernel<1>, 4, 31, d_src, d_out)) return 1;
  if (rate_of(rate_kernel<3, 0>, 3, 0, d_src, d_out)) return 1;
  return bad != 0;
}
