// The KING pair kernel on the gfx950 matrix cores.
//
// popcount(x & y) over the sites of two bit planes is the dot product of the
// two 0/1 vectors, so the four sums kinship needs (cuking.cu:232-239) are five
// plane products per pair:
//     opp = A_i.R_j + R_i.A_j      bh = H_i.H_j
//     hi  = H_i.D_j                hj = D_i.H_j
// (A hom-alt, R hom-ref, H het, D defined; the full form adds
// hom_hom = (A|R)_i.(A|R)_j).  v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 (E2M1)
// operands does 32 x 32 pairs x 64 sites per instruction in 32 cycles, four
// times the bf16 rate; products and sums are small integers, exact in the
// float32 accumulators while every sum stays below 2^24 (kMfmaMaxSites).
//
// Operand expansion costs ONE v_and per dword: a lane's fragment is 32 fp4
// values = 4 dwords; AND-ing four 32-site words with 0x11111111 << f leaves
// site 4q+f of each word in nibble q as the fp4 code 1 << f, i.e. the value
// 2^(f-1) (0.5, 1, 2).  Both operands carry the same factor, and the
// instruction's E8M0 block scale (2^(1-f) on each side) takes it out again, so
// every product is exactly 1.0.  f = 3 would be the sign bit: those sites are
// shifted down to f = 0 first (two ops).  The order of the sites inside the
// k dimension is irrelevant as long as both operands use the same one.
//
// Workgroup = 128 x 128 pairs, 4 wavefronts (one per SIMD, up to 512
// registers each), each 64 x 64 pairs = 2 x 2 MFMA blocks x 4 (5) float32
// accumulator sets.  One k-step = 256 sites = for every lane one uint4 (four
// 32-site words) per plane and block, read from LDS with ds_read_b128 and
// expanded four times (f = 0..3): 80 (96) MFMAs per k-step and wavefront.
// The planes come from the quad layout (king_common.h) by LDS-DMA, 32 KiB per
// k-step, three stages deep.
#include <hip/hip_runtime.h>

#include "king_common.h"
#include "king_device.h"

namespace cuking {

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kTile = 128;
constexpr int kStageU4 = 2 * 2 * 4 * kTile;  // sides x k-groups x planes x samples
constexpr int kPiecesPerWave = 8;            // 32 x 1 KiB per stage, 4 wavefronts

// Plane indices of the quad layout.
constexpr int kA = 0, kR = 1, kH = 2, kD = 3;

// Fragment f of four 32-site words (see the header comment).
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

__device__ __forceinline__ v8i or_frag(const v8i a, const v8i b) {
  v8i r = {0, 0, 0, 0, 0, 0, 0, 0};
  r[0] = a[0] | b[0]; r[1] = a[1] | b[1]; r[2] = a[2] | b[2]; r[3] = a[3] | b[3];
  return r;
}

// acc += sum over the 64 sites of the fragment of a_site * b_site.  The E8M0
// scale 2^(1-F) on each side (F == 3 sits at position 0 again) undoes the
// 2^(F-1) of the expansion.
template <int F>
__device__ __forceinline__ v16f mma(const v8i a, const v8i b, const v16f c) {
  constexpr int scale = F == 0 ? 128 : F == 1 ? 127 : F == 2 ? 126 : 128;
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
      a, b, c, 4 /* A is fp4 */, 4 /* B is fp4 */, 0, scale, 0, scale);
}

template <int NSTAGE, bool FULL>
__global__ __launch_bounds__(256, 1) void king_mfma_kernel(const TiledArgs a) {
  static_assert(NSTAGE == 2 || NSTAGE == 3, "two or three LDS stages");
  constexpr int NQ = FULL ? 5 : 4;
  extern __shared__ uint4 lds[];  // [NSTAGE][side][k-group][plane][128]

  uint32_t tr, tc;
  if (!decode_tile(a, a.tile_begin + blockIdx.x, &tr, &tc)) return;

  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t wr = (wave >> 1) * 64;  // wavefront's rows inside the tile
  const uint32_t wc = (wave & 1) * 64;   // ... and columns
  const uint32_t g = lane >> 5;          // k-group of the MFMA operand
  const uint32_t lr = lane & 31;         // row / column inside the block

  const uint4 *g_rows = a.planes + (uint64_t)tr * kTile;
  const uint4 *g_cols = a.planes + a.geo.col_base + (uint64_t)tc * kTile;
  const uint32_t s_stride = a.geo.s_stride;
  const uint32_t num_steps = a.geo.k_words / 8;

  auto issue_stage = [&](uint32_t step, uint32_t buf) {
#pragma unroll
    for (int r = 0; r < kPiecesPerWave; ++r) {
      const uint32_t piece = wave * kPiecesPerWave + r;  // 0..31
      const uint32_t side = piece >> 4, kg = (piece >> 3) & 1;
      const uint32_t p = (piece >> 1) & 3, seg = piece & 1;
      const uint4 *src = (side ? g_cols : g_rows) +
                         ((uint64_t)(2 * step + kg) * 4 + p) * s_stride +
                         seg * 64 + lane;
      uint4 *dst = lds + buf * kStageU4 + ((side * 2 + kg) * 4 + p) * kTile +
                   seg * 64;
      // LDS-DMA, lane l's 16 bytes land at dst + 16 * l.  Inline asm keeps it
      // out of the compiler's wait-count bookkeeping (king_kernels.hip).
      const uint32_t lds_addr = (uint32_t)(uintptr_t)(lds_void_ptr)dst;
      asm volatile(
          "s_mov_b32 m0, %0\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %1, off"
          :
          : "s"(lds_addr), "v"(src)
          : "memory");
    }
  };

  v16f acc[2][2][NQ];
#pragma unroll
  for (int bi = 0; bi < 2; ++bi)
#pragma unroll
    for (int bj = 0; bj < 2; ++bj)
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[bi][bj][q][r] = 0.f;

#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if ((uint32_t)s < num_steps) issue_stage(s, s);

  uint32_t buf = 0;
  for (uint32_t step = 0; step < num_steps; ++step) {
    // Stage `step` has landed for this wavefront: at most the NSTAGE - 2
    // younger stages (8 DMAs each) may still be in flight.
    if (NSTAGE == 3 && step + 1 < num_steps)
      __builtin_amdgcn_s_waitcnt(0x0F78);  // vmcnt(8)
    else
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    // ... and for every wavefront, and all of them are done with the buffer
    // the next request overwrites.
    __syncthreads();
    {
      const uint32_t ahead = step + NSTAGE - 1;
      uint32_t nbuf = buf + NSTAGE - 1;
      if (nbuf >= NSTAGE) nbuf -= NSTAGE;
      if (ahead < num_steps) issue_stage(ahead, nbuf);
    }

    const uint4 *l_rows = lds + buf * kStageU4 + (0 * 2 + g) * 4 * kTile + wr + lr;
    const uint4 *l_cols = lds + buf * kStageU4 + (1 * 2 + g) * 4 * kTile + wc + lr;
    uint4 A[2][4], B[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        A[b][p] = l_rows[p * kTile + b * 32];
        B[b][p] = l_cols[p * kTile + b * 32];
      }

#define CUKING_MFMA_STEP(F)                                                    \
    {                                                                          \
      v8i Af[2][4], Bf[2][4];                                                  \
      _Pragma("unroll") for (int b = 0; b < 2; ++b)                            \
      _Pragma("unroll") for (int p = 0; p < 4; ++p) {                          \
        Af[b][p] = expand<F>(A[b][p]);                                         \
        Bf[b][p] = expand<F>(B[b][p]);                                         \
      }                                                                        \
      _Pragma("unroll") for (int bi = 0; bi < 2; ++bi)                         \
      _Pragma("unroll") for (int bj = 0; bj < 2; ++bj) {                       \
        acc[bi][bj][0] = mma<F>(Af[bi][kA], Bf[bj][kR], acc[bi][bj][0]);       \
        acc[bi][bj][0] = mma<F>(Af[bi][kR], Bf[bj][kA], acc[bi][bj][0]);       \
        acc[bi][bj][1] = mma<F>(Af[bi][kH], Bf[bj][kH], acc[bi][bj][1]);       \
        acc[bi][bj][2] = mma<F>(Af[bi][kH], Bf[bj][kD], acc[bi][bj][2]);       \
        acc[bi][bj][3] = mma<F>(Af[bi][kD], Bf[bj][kH], acc[bi][bj][3]);       \
        if (FULL)                                                              \
          acc[bi][bj][NQ - 1] =                                                \
              mma<F>(or_frag(Af[bi][kA], Af[bi][kR]),                          \
                     or_frag(Bf[bj][kA], Bf[bj][kR]), acc[bi][bj][NQ - 1]);    \
      }                                                                        \
    }
    CUKING_MFMA_STEP(0)
    CUKING_MFMA_STEP(1)
    CUKING_MFMA_STEP(2)
    CUKING_MFMA_STEP(3)
#undef CUKING_MFMA_STEP

    if (++buf == NSTAGE) buf = 0;
  }

  // --- epilogue: kinship, threshold, append (cuking.cu:284-313).  C layout of
  // the 32 x 32 MFMA: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
#pragma unroll
  for (int bi = 0; bi < 2; ++bi) {
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
      const uint32_t lj = tc * kTile + wc + bj * 32 + lr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t li =
            tr * kTile + wr + bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        // cuking.cu:199 plus the tile padding
        const bool valid = li < a.geo.num_rows && lj < a.geo.num_cols &&
                           a.i_begin + li < a.j_begin + lj;
        const uint32_t opp = (uint32_t)acc[bi][bj][0][r];
        const uint32_t bh = (uint32_t)acc[bi][bj][1][r];
        const uint32_t hi = (uint32_t)acc[bi][bj][2][r];
        const uint32_t hj = (uint32_t)acc[bi][bj][3][r];
        if (FULL)
          full_epilogue_pair(a, valid, li, lj, hi, hj, bh, opp,
                             (uint32_t)acc[bi][bj][NQ - 1][r]);
        else
          lean_epilogue_pair(a, valid, li, lj, hi, hj, bh, opp, lane);
      }
    }
  }
}

template <int NSTAGE, bool FULL>
hipError_t launch_shape(const TiledArgs &args, uint64_t num_tiles,
                        uint32_t lds_bytes, hipStream_t stream) {
  auto kernel = king_mfma_kernel<NSTAGE, FULL>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void *>(kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const uint64_t cap = max_blocks_per_launch(256);
  uint64_t done = 0;
  while (done < num_tiles) {
    const uint64_t n = (num_tiles - done < cap) ? num_tiles - done : cap;
    TiledArgs a = args;
    a.tile_begin = args.tile_begin + done;
    kernel<<<dim3((uint32_t)n), dim3(256), lds_bytes, stream>>>(a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    done += n;
  }
  return hipSuccess;
}

}  // namespace

hipError_t launch_mfma(bool full, const TiledArgs &args, uint64_t num_tiles,
                       uint32_t lds_bytes, hipStream_t stream) {
  if ((uint64_t)args.geo.k_words * 32 > kMfmaMaxSites) return hipErrorInvalidValue;
  return full ? launch_shape<3, true>(args, num_tiles, lds_bytes, stream)
              : launch_shape<3, false>(args, num_tiles, lds_bytes, stream);
}

}  // namespace cuking
