// Would TWO wavefronts per SIMD hide the LDS-DMA requests of the four-product
// loop (king_mfma.hip, "Four products")?  Timing model of one k-step slice, no
// real data: per wavefront and slice
//   BJ = 2 (shipped shape, 4 wavefronts of 64 x 64 pairs): 16 unscaled fp4 MFMAs,
//          48 v_and, 4 ds_read_b128, 2 LDS-DMA requests of 1 KiB
//   BJ = 1 (8 wavefronts of 64 x 32 pairs, two per SIMD): 8 MFMAs, 36 v_and,
//          3 ds_read_b128, 1 request
// in the same group structure as the kernel (builds pinned to their MFMA group),
// one stage barrier per four slices, requests served from an L2-resident buffer.
// Both shapes issue the same MFMAs per SIMD; the question is cycles per k-step.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_two_waves.hip -o mfma_two_waves
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#pragma clang diagnostic ignored "-Winline-asm"
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void *lds_void_ptr;

__device__ __forceinline__ v8i nfrag(const uint4 w, uint32_t mask) {
  v8i r = {0, 0, 0, 0, 0, 0, 0, 0};
  r[0] = (int)(w.x & mask); r[1] = (int)(w.y & mask); r[2] = (int)(w.z & mask); r[3] = (int)(w.w & mask);
  return r;
}
__device__ __forceinline__ v16f mma(const v8i a, const v8i b, const v16f c) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 0, 0, 0);
}
#define PIN4(W) asm volatile("" : "+v"((W).x), "+v"((W).y), "+v"((W).z), "+v"((W).w));
#define PINF(F) asm volatile("" : "+v"((F)[0]), "+v"((F)[1]), "+v"((F)[2]), "+v"((F)[3]));
#define PACE(n, v)                                                             \
  _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {                         \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                         \
    if ((v) > 0) __builtin_amdgcn_sched_group_barrier(0x002, (v), 0);          \
  }

// DMA: 0 = none, 1 = as shipped.  BAR: stage barrier per four slices.
template <int BJ, int DMA, int BAR>
__global__ __launch_bounds__(BJ == 2 ? 256 : 512, 1) void slice_kernel(const uint4 *src, float *out,
                                                                      int ksteps,
                                                                      unsigned long long *stamps) {
  extern __shared__ uint4 lds[];
  constexpr int BI = 2;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t lane16 = lane * 16;
  uint32_t mH, mD, mT;
  asm volatile("s_mov_b32 %0, 0x11111111" : "=s"(mH));
  asm volatile("s_mov_b32 %0, 0x22222222" : "=s"(mD));
  asm volatile("s_mov_b32 %0, 0xcccccccc" : "=s"(mT));
  const uint4 *g = src + (size_t)blockIdx.x * 4096 + wave * 256;   // 64 KiB per workgroup, L2-resident
  const uint32_t l_base = (uint32_t)(uintptr_t)(lds_void_ptr)(lds + wave * 1024);
  const uint4 *l_rd = lds + wave * 1024 + lane;
  v16f acc[BI][BJ][4];
  for (int bi = 0; bi < BI; ++bi) for (int bj = 0; bj < BJ; ++bj) for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) acc[bi][bj][q][r] = 0.f;
  uint4 RA[2][BI], RB[2][BJ];
  v8i Fa[3][BI], Fb[3][BJ];
  for (int k = 0; k < 2; ++k) {
    for (int b = 0; b < BI; ++b) RA[k][b] = src[lane + 64 * (k * 4 + b)];
    for (int b = 0; b < BJ; ++b) RB[k][b] = src[lane + 64 * (k * 4 + 2 + b)];
  }
  for (int k = 0; k < 3; ++k) {
    for (int b = 0; b < BI; ++b) Fa[k][b] = nfrag(RA[0][b], k == 0 ? mH : k == 1 ? mD : mT);
    for (int b = 0; b < BJ; ++b) Fb[k][b] = nfrag(RB[0][b], k == 0 ? mH : k == 1 ? mD : mT);
  }
  auto issue = [&](uint32_t slot, int half) {
    if (!DMA) return;
    const uint32_t dst = l_base + (slot & 7) * 2048;
    const uint4 *s = g + (slot & 3) * 64;
    if (half)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024"
                   : : "s"(dst), "v"(lane16), "s"(s) : "memory", "m0");
    else
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                   : : "s"(dst), "v"(lane16), "s"(s) : "memory", "m0");
  };
#define MMA(Q, KA, KB)                                                         \
  _Pragma("unroll") for (int bi = 0; bi < BI; ++bi)                            \
  _Pragma("unroll") for (int bj = 0; bj < BJ; ++bj)                            \
    acc[bi][bj][Q] = mma(Fa[KA][bi], Fb[KB][bj], acc[bi][bj][Q]);
#define BUILD_A(K, RB_, M) _Pragma("unroll") for (int b = 0; b < BI; ++b) Fa[K][b] = nfrag(RA[RB_][b], M);
#define BUILD_B(K, RB_, M) _Pragma("unroll") for (int b = 0; b < BJ; ++b) Fb[K][b] = nfrag(RB[RB_][b], M);
#define PIN_RA(RB_) _Pragma("unroll") for (int b = 0; b < BI; ++b) PIN4(RA[RB_][b])
#define PIN_RB(RB_) _Pragma("unroll") for (int b = 0; b < BJ; ++b) PIN4(RB[RB_][b])
#define PIN_A(K) _Pragma("unroll") for (int b = 0; b < BI; ++b) PINF(Fa[K][b])
#define PIN_B(K) _Pragma("unroll") for (int b = 0; b < BJ; ++b) PINF(Fb[K][b])
#define SLICE(CUR, NXT, SYNC, SLOT)                                            \
  {                                                                            \
    PIN_RA(CUR) PIN_RB(CUR)                                                    \
    BUILD_A(2, CUR, mT) BUILD_B(2, CUR, mT)                                    \
    MMA(0, 0, 1)                                                               \
    PACE(BI * BJ, (BI + BJ) * 4 / (BI * BJ))                                   \
    PIN_A(2) PIN_B(2)                                                          \
    __builtin_amdgcn_sched_barrier(0);                                         \
    if (SYNC && BAR) {                                                         \
      if (DMA) __builtin_amdgcn_s_waitcnt(0x0F70 | (BJ == 2 ? 4 : 2));         \
      __syncthreads();                                                         \
    }                                                                          \
    _Pragma("unroll") for (int b = 0; b < BI; ++b) RA[CUR][b] = l_rd[((SLOT) & 7) * 128 + b * 64]; \
    _Pragma("unroll") for (int b = 0; b < BJ; ++b) RB[CUR][b] = l_rd[512 + ((SLOT) & 3) * 128 + b * 64]; \
    PIN_RA(NXT)                                                                \
    BUILD_A(0, NXT, mH)                                                        \
    MMA(1, 1, 0)                                                               \
    PIN_A(0)                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                         \
    _Pragma("unroll") for (int r = 0; r < BJ; ++r) {                           \
      issue(SLOT, r);                                                          \
      acc[0][r][2] = mma(Fa[1][0], Fb[1][r], acc[0][r][2]);                    \
      __builtin_amdgcn_sched_barrier(0);                                       \
    }                                                                          \
    PIN_RB(NXT)                                                                \
    BUILD_B(0, NXT, mH)                                                        \
    _Pragma("unroll") for (int r = 0; r < BJ; ++r)                             \
      acc[1][r][2] = mma(Fa[1][1], Fb[1][r], acc[1][r][2]);                    \
    PIN_B(0)                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                         \
    PIN_RA(NXT) PIN_RB(NXT)                                                    \
    BUILD_B(1, NXT, mD) BUILD_A(1, NXT, mD)                                    \
    MMA(3, 2, 2)                                                               \
    PACE(BI * BJ, (BI + BJ) * 4 / (BI * BJ))                                   \
    PIN_B(1) PIN_A(1)                                                          \
    __builtin_amdgcn_sched_barrier(0);                                         \
  }
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int k = 0; k < ksteps; ++k) {
    SLICE(0, 1, false, 4 * k)
    SLICE(1, 0, false, 4 * k + 1)
    SLICE(0, 1, true, 4 * k + 2)
    SLICE(1, 0, false, 4 * k + 3)
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
  float s = 0;
  for (int bi = 0; bi < BI; ++bi) for (int bj = 0; bj < BJ; ++bj) for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 16; ++r) s += acc[bi][bj][q][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int BJ, int DMA, int BAR>
int run(const char *name, const uint4 *d_src, float *d_out, unsigned long long *d_st, int ksteps) {
  const int grid = 256, threads = BJ == 2 ? 256 : 512;
  auto k = slice_kernel<BJ, DMA, BAR>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    k<<<grid, threads, 160 * 1024>>>(d_src, d_out, ksteps, d_st);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> st(2 * grid);
  CHECK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> cyc, clk;
  for (int b = 0; b < grid; ++b) { cyc.push_back((double)st[2 * b] / ksteps); clk.push_back(100.0 * st[2 * b] / st[2 * b + 1]); }
  std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
  // MFMAs per SIMD and k-step: 64 in both shapes
  printf("%-58s %7.1f cycles per k-step (wave 0 of the median WG; 2048 = matrix pipe), %7.3f ms = %6.1f ns per k-step, "
         "clock %.0f MHz\n", name, cyc[grid / 2], ms, ms * 1e6 / ksteps, clk[grid / 2]);
  return 0;
}

int main(int argc, char **argv) {
  const bool reversed = argc > 1 && argv[1][0] == 'r';
  const int ksteps = 20000;
  uint4 *d_src; float *d_out; unsigned long long *d_st;
  const size_t src_bytes = (size_t)256 * 4096 * 16 + 65536;
  CHECK(hipMalloc(&d_src, src_bytes)); CHECK(hipMalloc(&d_out, 256 * 512 * 4)); CHECK(hipMalloc(&d_st, 256 * 16));
  std::vector<uint32_t> h(src_bytes / 4);
  srand(7);
  for (auto &w : h) { uint32_t x = 0; for (int n = 0; n < 8; ++n) { const int r = rand() % 100; x |= (uint32_t)(r < 60 ? 6 : r < 85 ? 3 : r < 99 ? 14 : 0) << (4 * n); } w = x; }
  CHECK(hipMemcpy(d_src, h.data(), src_bytes, hipMemcpyHostToDevice));
  // (argument "r": the two-wave shapes first -- does the order of the runs move the clock?)
  for (int pass = 0; pass < 2; ++pass) {
    const bool two = (pass == 0) == reversed;
    if (!two) {
      if (run<2, 1, 1>("1 wave/SIMD (64x64 per wave), requests + barrier", d_src, d_out, d_st, ksteps)) return 1;
      if (run<2, 0, 1>("1 wave/SIMD, no requests", d_src, d_out, d_st, ksteps)) return 1;
      if (run<2, 0, 0>("1 wave/SIMD, no requests, no barrier", d_src, d_out, d_st, ksteps)) return 1;
    } else {
      if (run<1, 1, 1>("2 waves/SIMD (64x32 per wave), requests + barrier", d_src, d_out, d_st, ksteps)) return 1;
      if (run<1, 0, 1>("2 waves/SIMD, no requests", d_src, d_out, d_st, ksteps)) return 1;
      if (run<1, 0, 0>("2 waves/SIMD, no requests, no barrier", d_src, d_out, d_st, ksteps)) return 1;
    }
  }
  return 0;
}
