// HIP kernels of the KING hot path, written for gfx950 (MI355X, wave64).
//
//   prepare_planes_kernel  reference bitset (cuking.cu:507-523) -> k-major
//                          4-plane layout (king_common.h)
//   prepare_quads_kernel   ... -> quad layout of the matrix-core kernel
//   king_tiled_kernel      the VALU pair kernel: LDS-staged, register-tiled
//                          AND+popcount over all pairs of a tile, issued as
//                          barrier-aligned logic / popcount phases
//                          (ComputeKingKernel, cuking.cu:191-314; variants
//                          0-4 -- the default is king_mfma.hip, variant 5)
//   king_stream_kernel     one pair per wavefront straight from the reference
//                          layout, wave-level reductions (same contract)
//   pack_kernel            cuking.cu:675-703 on the device
//
// In this file the work is AND / BITOP3 / BCNT on 32-bit words (VALU) fed
// from LDS; see DESIGN.md for the op count and the rooflines.
#include <hip/hip_runtime.h>

#include "king_common.h"
#include "king_device.h"

// The LDS-DMA statements below write M0 and say so in their clobber lists; the
// compiler notes that it keeps no value of its own there (M0 is reserved).
#pragma clang diagnostic ignored "-Winline-asm"

namespace cuking {

namespace {

// ---------------------------------------------------------------------------
// prepare_planes_kernel
// One workgroup: 64 plane-samples x 16 source words (= 32 k-rows).  Reads are
// coalesced along a sample's words, the transpose goes through LDS, writes are
// coalesced along samples (1 KiB per k-row).
// ---------------------------------------------------------------------------
constexpr int kPrepSamples = 64;
constexpr int kPrepWords = 16;

__global__ __launch_bounds__(256) void prepare_planes_kernel(
    const uint64_t *__restrict__ bits, uint32_t words_per_sample,
    PlaneGeometry geo, uint4 *__restrict__ planes, uint32_t s_tile_begin) {
  __shared__ uint64_t het_lds[kPrepSamples][kPrepWords + 1];
  __shared__ uint64_t hom_lds[kPrepSamples][kPrepWords + 1];

  const uint32_t plane_words = words_per_sample / 2;
  const uint32_t s0 = (s_tile_begin + blockIdx.x) * kPrepSamples;
  const uint32_t w0 = blockIdx.y * kPrepWords;

#pragma unroll
  for (int it = 0; it < kPrepSamples * kPrepWords / 256; ++it) {
    const uint32_t idx = it * 256 + threadIdx.x;
    const uint32_t s = idx / kPrepWords, w = idx % kPrepWords;
    const uint32_t ps = s0 + s;  // plane sample index
    // Which stored sample of the reference bitset, if any.
    uint32_t src = 0xFFFFFFFFu;
    if (geo.diag || ps < geo.rows_padded) {
      if (ps < geo.num_rows) src = ps;
    } else {
      const uint32_t c = ps - geo.col_base;
      if (c < geo.num_cols) src = geo.num_rows + c;
    }
    uint64_t het = ~0ull, hom = ~0ull;  // missing
    if (src != 0xFFFFFFFFu && w0 + w < plane_words) {
      const uint64_t *p = bits + (uint64_t)src * words_per_sample + (w0 + w);
      het = p[0];
      hom = p[plane_words];
    }
    het_lds[s][w] = het;
    hom_lds[s][w] = hom;
  }
  __syncthreads();

#pragma unroll
  for (int it = 0; it < kPrepSamples * kPrepWords * 2 / 256; ++it) {
    const uint32_t idx = it * 256 + threadIdx.x;
    const uint32_t krow = idx / kPrepSamples, s = idx % kPrepSamples;
    const uint32_t k = w0 * 2 + krow;
    if (k >= geo.k_words || s0 + s >= geo.s_stride) continue;
    const uint32_t shift = (krow & 1) * 32;
    const uint32_t het = (uint32_t)(het_lds[s][krow >> 1] >> shift);
    const uint32_t hom = (uint32_t)(hom_lds[s][krow >> 1] >> shift);
    uint4 v;
    v.x = het & ~hom;    // H
    v.y = hom & ~het;    // A
    v.z = ~het;          // Hom (hom-ref or hom-alt)
    v.w = ~(het & hom);  // D
    planes[(uint64_t)k * geo.s_stride + s0 + s] = v;
  }
}

// Same transpose into the quad layout of the matrix-core kernel: one uint4 =
// four consecutive 32-site words (two u64 source words) of one reference plane.
__global__ __launch_bounds__(256) void prepare_quads_kernel(
    const uint64_t *__restrict__ bits, uint32_t words_per_sample,
    PlaneGeometry geo, uint4 *__restrict__ planes, uint32_t s_tile_begin) {
  __shared__ uint64_t het_lds[kPrepSamples][kPrepWords + 1];
  __shared__ uint64_t hom_lds[kPrepSamples][kPrepWords + 1];

  const uint32_t plane_words = words_per_sample / 2;
  const uint32_t s0 = (s_tile_begin + blockIdx.x) * kPrepSamples;
  const uint32_t w0 = blockIdx.y * kPrepWords;

#pragma unroll
  for (int it = 0; it < kPrepSamples * kPrepWords / 256; ++it) {
    const uint32_t idx = it * 256 + threadIdx.x;
    const uint32_t s = idx / kPrepWords, w = idx % kPrepWords;
    const uint32_t ps = s0 + s;
    uint32_t src = 0xFFFFFFFFu;
    if (geo.diag || ps < geo.rows_padded) {
      if (ps < geo.num_rows) src = ps;
    } else {
      const uint32_t c = ps - geo.col_base;
      if (c < geo.num_cols) src = geo.num_rows + c;
    }
    uint64_t het = ~0ull, hom = ~0ull;  // missing
    if (src != 0xFFFFFFFFu && w0 + w < plane_words) {
      const uint64_t *p = bits + (uint64_t)src * words_per_sample + (w0 + w);
      het = p[0];
      hom = p[plane_words];
    }
    het_lds[s][w] = het;
    hom_lds[s][w] = hom;
  }
  __syncthreads();

  constexpr int kQuads = kPrepWords / 2;
#pragma unroll
  for (int it = 0; it < kQuads * 2 * kPrepSamples / 256; ++it) {
    const uint32_t idx = it * 256 + threadIdx.x;
    const uint32_t s = idx % kPrepSamples;
    const uint32_t p = (idx / kPrepSamples) % 2;
    const uint32_t ql = idx / (kPrepSamples * 2);
    const uint32_t q = w0 / 2 + ql;
    if (q * 4 >= geo.k_words || s0 + s >= geo.s_stride) continue;
    const uint64_t lo = p ? hom_lds[s][2 * ql] : het_lds[s][2 * ql];
    const uint64_t hi = p ? hom_lds[s][2 * ql + 1] : het_lds[s][2 * ql + 1];
    uint4 v;
    v.x = (uint32_t)lo;
    v.y = (uint32_t)(lo >> 32);
    v.z = (uint32_t)hi;
    v.w = (uint32_t)(hi >> 32);
    planes[((uint64_t)q * 2 + p) * geo.s_stride + s0 + s] = v;
  }
}

// Same transpose into the nibble layout of the four-product matrix-core kernel
// (king_common.h): one uint4 = the 32 sites of one 32-bit word of the reference
// planes, one fp4 code per site.
__device__ __forceinline__ uint32_t spread8(uint32_t b) {
  // bit t of the low byte -> bit 4 t
  uint32_t x = b & 0xFFu;
  x = (x | (x << 12)) & 0x000F000Fu;
  x = (x | (x << 6)) & 0x03030303u;
  x = (x | (x << 3)) & 0x11111111u;
  return x;
}

// One workgroup: 64 plane-samples x kNibWords source words of each plane, read
// 16 bytes per lane in runs of 256 contiguous bytes per sample and plane (the
// rows of the reference layout are words_per_sample x 8 bytes apart, so longer
// runs mean fewer DRAM pages per byte), transposed through LDS, written 1 KiB
// per k-row (64 consecutive samples).
#ifndef CUKING_NIB_WORDS
#define CUKING_NIB_WORDS 32  // (A/B: 16, 64)
#endif
constexpr int kNibWords = CUKING_NIB_WORDS;

// CODES: the fp4 codes and the het-only copy; T2: the filter kernel's two-bit layout
// (king_common.h, kLayoutNibbleStats).  `perm` (or nullptr): which stored sample sits at
// a plane sample (king_common.h).  `gate` (or nullptr): the filter's control words -- the
// launch converts only if a quadrant went dense or a tile left for the exact kernel, and
// the codes are not there yet (*ready).
template <bool CODES, bool T2>
__global__ __launch_bounds__(256) void prepare_nibbles_kernel(
    const uint64_t *__restrict__ bits, uint32_t words_per_sample,
    PlaneGeometry geo, uint4 *__restrict__ planes, uint32_t s_tile_begin,
    const uint32_t *__restrict__ perm, const uint32_t *__restrict__ gate,
    const uint32_t *__restrict__ ready) {
  __shared__ uint64_t het_lds[kPrepSamples][kNibWords + 1];
  __shared__ uint64_t hom_lds[kPrepSamples][kNibWords + 1];
  if (gate != nullptr && ((gate[kCtrlGate] | gate[kCtrlDense]) == 0 || *ready != 0)) return;

  const uint32_t plane_words = words_per_sample / 2;
  const uint32_t s0 = (s_tile_begin + blockIdx.x) * kPrepSamples;
  const uint32_t w0 = blockIdx.y * kNibWords;
  // (plane_words even <=> a sample's planes start 16-byte aligned whenever the
  //  bitset does; otherwise fall back to 8-byte loads)
  const bool wide = (plane_words & 1) == 0 && (words_per_sample & 1) == 0 &&
                    (reinterpret_cast<uintptr_t>(bits) & 15) == 0;

#pragma unroll
  for (int it = 0; it < kPrepSamples * kNibWords / 2 / 256; ++it) {
    const uint32_t idx = it * 256 + threadIdx.x;
    const uint32_t s = idx / (kNibWords / 2), w = (idx % (kNibWords / 2)) * 2;
    const uint32_t ps = s0 + s;  // plane sample index
    uint32_t src = 0xFFFFFFFFu;  // which stored sample of the reference bitset, if any
    if (perm != nullptr) {
      if (ps < geo.s_stride) src = perm[ps];
    } else if (geo.diag || ps < geo.rows_padded) {
      if (ps < geo.num_rows) src = ps;
    } else {
      const uint32_t c = ps - geo.col_base;
      if (c < geo.num_cols) src = geo.num_rows + c;
    }
    uint64_t het[2] = {~0ull, ~0ull}, hom[2] = {~0ull, ~0ull};  // missing
    if (src != 0xFFFFFFFFu && w0 + w < plane_words) {
      const uint64_t *p = bits + (uint64_t)src * words_per_sample + (w0 + w);
      if (wide && w0 + w + 1 < plane_words) {
        const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(p);
        const ulonglong2 b = *reinterpret_cast<const ulonglong2 *>(p + plane_words);
        het[0] = a.x; het[1] = a.y;
        hom[0] = b.x; hom[1] = b.y;
      } else {
        het[0] = p[0];
        hom[0] = p[plane_words];
        if (w0 + w + 1 < plane_words) {
          het[1] = p[1];
          hom[1] = p[plane_words + 1];
        }
      }
    }
    het_lds[s][w] = het[0];
    het_lds[s][w + 1] = het[1];
    hom_lds[s][w] = hom[0];
    hom_lds[s][w + 1] = hom[1];
  }
  __syncthreads();

  if constexpr (CODES) {
#pragma unroll
  for (int it = 0; it < kPrepSamples * kNibWords * 2 / 256; ++it) {
    const uint32_t idx = it * 256 + threadIdx.x;
    const uint32_t krow = idx / kPrepSamples, s = idx % kPrepSamples;
    const uint32_t k = w0 * 2 + krow;
    if (k >= geo.k_words || s0 + s >= geo.s_stride) continue;
    const uint32_t shift = (krow & 1) * 32;
    const uint32_t het = (uint32_t)(het_lds[s][krow >> 1] >> shift);
    const uint32_t hom = (uint32_t)(hom_lds[s][krow >> 1] >> shift);
    uint32_t out[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t hs = spread8(het >> (8 * d)), ms = spread8(hom >> (8 * d));
      const uint32_t H = hs & ~ms;                 // het and defined
      const uint32_t D = 0x11111111u ^ (hs & ms);  // defined
      const uint32_t Y = 0x11111111u ^ hs;         // hom-ref or hom-alt
      const uint32_t A = ms & ~hs;                 // hom-alt
      out[d] = H | (D << 1) | (Y << 2) | (A << 3);
    }
    planes[(uint64_t)k * geo.s_stride + s0 + s] = make_uint4(out[0], out[1], out[2], out[3]);
  }
  // ... and the het plane once more as it is, transposed (the quad layout's
  // plane 0), behind the codes: what the full form's hom_hom pass reads.
  uint4 *hetq = planes + (uint64_t)geo.k_words * geo.s_stride;
  constexpr int kQuads = kNibWords / 2;
#pragma unroll
  for (int it = 0; it < kQuads * kPrepSamples / 256; ++it) {
    const uint32_t idx = it * 256 + threadIdx.x;
    const uint32_t s = idx % kPrepSamples, ql = idx / kPrepSamples;
    const uint32_t q = w0 / 2 + ql;
    if (q * 4 >= geo.k_words || s0 + s >= geo.s_stride) continue;
    const uint64_t lo = het_lds[s][2 * ql], hi = het_lds[s][2 * ql + 1];
    hetq[(uint64_t)q * geo.s_stride + s0 + s] =
        make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
  }
  }  // CODES
  if constexpr (T2) {
    // One unit = one 64-bit word of the reference planes: its low 32 sites in
    // bits 2-3 of the nibbles, its high 32 sites in bits 0-1; per site (hom-alt,
    // hom) = the fp4 code's (sign, 2.0) bits.
    uint4 *t2 = const_cast<uint4 *>(plane_t2(planes, geo));
#pragma unroll
    for (int it = 0; it < kNibWords * kPrepSamples / 256; ++it) {
      const uint32_t idx = it * 256 + threadIdx.x;
      const uint32_t s = idx % kPrepSamples, ul = idx / kPrepSamples;
      const uint32_t u = w0 + ul;
      if (u * 2 >= geo.k_words || s0 + s >= geo.s_stride) continue;
      const uint64_t het = het_lds[s][ul], hom = hom_lds[s][ul];
      uint32_t out[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const uint32_t h_lo = (uint32_t)(het >> (8 * d)), m_lo = (uint32_t)(hom >> (8 * d));
        const uint32_t h_hi = (uint32_t)(het >> (32 + 8 * d)), m_hi = (uint32_t)(hom >> (32 + 8 * d));
        const uint32_t y_lo = spread8(~h_lo), a_lo = spread8(m_lo & ~h_lo);
        const uint32_t y_hi = spread8(~h_hi), a_hi = spread8(m_hi & ~h_hi);
        out[d] = (a_lo << 3) | (y_lo << 2) | (a_hi << 1) | y_hi;
      }
      t2[(uint64_t)u * geo.s_stride + s0 + s] = make_uint4(out[0], out[1], out[2], out[3]);
    }
  }
}

// ---------------------------------------------------------------------------
// king_tiled_kernel
//
// Workgroup = one TILE x TILE tile of sample pairs, TIT x TJT threads, each
// owning an RI x RJ micro-tile (rows q*TIT + ti, columns q*TJT + tj) with four
// (lean form) or five (full form) u32 accumulators per pair.  The shipped
// shape is 16 x 32 threads, 4 x 2 pairs per thread, phased (see PHASED below).
// K is streamed in chunks of KC 32-site words:
// every chunk is 2 * KC rows of TILE uint4 (row side and column side), copied
// global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave
// instruction, no VGPR staging) into a double buffer while the previous chunk
// is consumed with ds_read_b128.
//
// Per pair and 32-site word, 10 VALU ops (9 in the lean form, which skips hh
// and recounts it only for emitted pairs):
//     t    = Hom_i & Hom_j                       v_and
//     hh  += popc(t)                             v_bcnt (accumulating)
//     opp += popc((A_i ^ A_j) & t)               v_bitop3 + v_bcnt
//     bh  += popc(H_i & H_j)                     v_and + v_bcnt
//     hi  += popc(H_i & D_j)                     v_and + v_bcnt
//     hj  += popc(D_i & H_j)                     v_and + v_bcnt
// from which the reference's six sums (cuking.cu:232-239) follow exactly:
//     het_i = hi, het_j = hj, both_het = bh, opposing_hom = opp,
//     concordant_hom = hh - opp, shared = hi + hj - bh + hh.
// ---------------------------------------------------------------------------
// PHASED: a workgroup with two waves per SIMD runs every k-step as a logic
// phase (v_and / v_bitop3 into temporaries) followed by a popcount phase
// (v_bcnt), with one s_barrier in front of each logic phase.  Measured on
// gfx950 (tools/micro/valu_phase.hip, archive/profiles/r01_valu_microbench.txt): a
// wave issues at most one VALU instruction per 4 cycles; a full-rate
// instruction leaves half of that slot to ANOTHER wave's full-rate
// instruction, a half-rate v_bcnt takes all of it.  Unsynchronised waves mix
// the two kinds and every instruction costs ~4 cycles; two waves of one
// workgroup on the same SIMD that enter their logic phases together pair
// their full-rate instructions up, and the stream costs the sum of its parts
// (~3.05 cycles per instruction for this mix).  (PHASED 2 / 3: barrier
// placements that measured slower, tuning builds only.)
template <int TIT, int TJT, int RI, int RJ, int KC, int KU, int MINW, bool FULL,
          int ABLATE = 0, int PHASED = 0>  // ABLATE bit 2: one column operand at a time
__global__ __launch_bounds__(TIT *TJT, MINW) void king_tiled_kernel(
    const TiledArgs a) {
  constexpr int TILE = TIT * RI;
  static_assert(TILE == TJT * RJ, "square tiles only");
  constexpr int NT = TIT * TJT;
  constexpr int NW = NT / 64;
  constexpr int SEG = TILE / 64;            // 1 KiB wave-rows per tile row
  constexpr int WAVE_ROWS = 2 * KC * SEG;   // per chunk
  static_assert(WAVE_ROWS % NW == 0, "rows must split evenly over waves");
  constexpr int PER_WAVE = WAVE_ROWS / NW;

  extern __shared__ uint4 lds[];  // [2 buffers][2 sides][KC][TILE]

  // --- which tile (uniform across the workgroup) ---
  const uint64_t t = a.tile_begin + blockIdx.x;
  uint32_t tr, tc;
  if (!decode_tile(a, t, &tr, &tc)) return;  // before any barrier

  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t ti = threadIdx.x % TIT;
  const uint32_t tj = threadIdx.x / TIT;

  const uint4 *g_rows = a.planes + (uint64_t)tr * TILE;
  const uint4 *g_cols = a.planes + a.geo.col_base + (uint64_t)tc * TILE;
  const uint32_t s_stride = a.geo.s_stride;
  const uint32_t num_chunks = a.geo.k_words / KC;

  auto issue_chunk = [&](uint32_t chunk, uint32_t buf) {
#pragma unroll
    for (int r = 0; r < PER_WAVE; ++r) {
      const uint32_t wr = wave * PER_WAVE + r;
      const uint32_t side = wr / (KC * SEG);
      const uint32_t rem = wr % (KC * SEG);
      const uint32_t kc = rem / SEG, seg = rem % SEG;
      const uint4 *src = (side ? g_cols : g_rows) +
                         (uint64_t)(chunk * KC + kc) * s_stride + seg * 64 +
                         lane;
      uint4 *dst = lds + ((buf * 2 + side) * KC + kc) * TILE + seg * 64;
      // LDS-DMA: lane l's 16 bytes land at dst + 16 * l.
      if (PHASED) {
        // Issued through inline asm so that the compiler's wait-count pass does
        // not know about it: otherwise every ds_read that follows (they read
        // the OTHER buffer) is preceded by s_waitcnt vmcnt(0), i.e. waits for
        // the chunk that was just requested.  The hand-placed vmcnt(0) in
        // front of the chunk barrier below is the only wait this needs.
        const uint32_t lds_addr = (uint32_t)(uintptr_t)(lds_void_ptr)dst;
        asm volatile(
            "s_mov_b32 m0, %0\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, off"
            :
            : "s"(lds_addr), "v"(src)
            : "memory", "m0");
      } else {
        __builtin_amdgcn_global_load_lds((global_void_ptr)src, (lds_void_ptr)dst,
                                         16, 0, 0);
      }
    }
  };

  // hh is only accumulated by the FULL form (dead registers otherwise).
  uint32_t c_hh[RI][RJ], c_opp[RI][RJ], c_bh[RI][RJ], c_hi[RI][RJ],
      c_hj[RI][RJ];
#pragma unroll
  for (int x = 0; x < RI; ++x)
#pragma unroll
    for (int y = 0; y < RJ; ++y)
      c_hh[x][y] = c_opp[x][y] = c_bh[x][y] = c_hi[x][y] = c_hj[x][y] = 0;

  issue_chunk(0, 0);
  for (uint32_t chunk = 0; chunk < num_chunks; ++chunk) {
    const uint32_t buf = chunk & 1;
    // Chunk `chunk` has landed (vmcnt) for every wave, and every wave is done
    // reading the other buffer.
    // (ABLATE: timing-only experiments of the tuning build; results are
    // wrong.  bit 0 = no per-step LDS reads, bit 1 = no DMA and no barrier.)
    if (!(ABLATE & 2)) {
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), leaves lgkmcnt/expcnt
      __syncthreads();
      if (chunk + 1 < num_chunks) issue_chunk(chunk + 1, buf ^ 1);
    }

    const uint4 *l_rows = lds + (buf * 2 + 0) * KC * TILE;
    const uint4 *l_cols = lds + (buf * 2 + 1) * KC * TILE;
    if (PHASED) {
      // Software pipelined: the operands of k-step kc+1 are requested right
      // after the logic phase of kc has consumed the previous ones, so they
      // arrive during the popcount phase.
      uint4 ri[RI], cj[RJ];
#pragma unroll
      for (int x = 0; x < RI; ++x) ri[x] = l_rows[x * TIT + ti];
#pragma unroll
      for (int y = 0; y < RJ; ++y) cj[y] = l_cols[y * TJT + tj];
#pragma unroll KU
      for (int kc = 0; kc < KC; ++kc) {
        uint32_t v_opp[RI][RJ], v_bh[RI][RJ], v_hi[RI][RJ], v_hj[RI][RJ],
            v_hh[RI][RJ];
        // --- logic phase: full-rate instructions only ---
#pragma unroll
        for (int x = 0; x < RI; ++x) {
#pragma unroll
          for (int y = 0; y < RJ; ++y) {
            const uint32_t hom_both = ri[x].z & cj[y].z;
            v_hh[x][y] = hom_both;
            v_opp[x][y] =
                __builtin_amdgcn_bitop3_b32(ri[x].y, cj[y].y, hom_both, 0x28);
            v_bh[x][y] = ri[x].x & cj[y].x;
            v_hi[x][y] = ri[x].x & cj[y].w;
            v_hj[x][y] = ri[x].w & cj[y].x;
          }
        }
        if (kc + 1 < KC) {
#pragma unroll
          for (int x = 0; x < RI; ++x) ri[x] = l_rows[(kc + 1) * TILE + x * TIT + ti];
#pragma unroll
          for (int y = 0; y < RJ; ++y) cj[y] = l_cols[(kc + 1) * TILE + y * TJT + tj];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (PHASED >= 2) __builtin_amdgcn_s_barrier();  // tuning only: measured slower
        __builtin_amdgcn_sched_barrier(0);
        // --- popcount phase: half-rate instructions only.  Inline asm keeps
        // the accumulate fused (v_bcnt d, s, d): left to itself the compiler
        // re-associates two k-steps into bcnt + bcnt + v_add3. ---
#pragma unroll
        for (int x = 0; x < RI; ++x) {
#pragma unroll
          for (int y = 0; y < RJ; ++y) {
            if (FULL)
              asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(c_hh[x][y]) : "v"(v_hh[x][y]));
            asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(c_opp[x][y]) : "v"(v_opp[x][y]));
            asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(c_bh[x][y]) : "v"(v_bh[x][y]));
            asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(c_hi[x][y]) : "v"(v_hi[x][y]));
            asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(c_hj[x][y]) : "v"(v_hj[x][y]));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (PHASED != 3) __builtin_amdgcn_s_barrier();  // aligns the next logic phase
        __builtin_amdgcn_sched_barrier(0);
      }
      continue;
    }
    // Partial unroll only: a full unroll lets the scheduler hoist every
    // k-step's ds_reads and spills hundreds of VGPRs.
#pragma unroll KU
    for (int kc = 0; kc < KC; ++kc) {
      uint4 ri[RI], cj[RJ];
      const int kr = (ABLATE & 1) ? 0 : kc;
#pragma unroll
      for (int x = 0; x < RI; ++x) ri[x] = l_rows[kr * TILE + x * TIT + ti];
      if (!(ABLATE & 4)) {
#pragma unroll
        for (int y = 0; y < RJ; ++y) cj[y] = l_cols[kr * TILE + y * TJT + tj];
      }
      if (ABLATE & 1) {  // keep the logic ops from being hoisted
#pragma unroll
        for (int x = 0; x < RI; ++x)
          asm volatile("" : "+v"(ri[x].x), "+v"(ri[x].y), "+v"(ri[x].z), "+v"(ri[x].w));
#pragma unroll
        for (int y = 0; y < RJ; ++y)
          asm volatile("" : "+v"(cj[y].x), "+v"(cj[y].y), "+v"(cj[y].z), "+v"(cj[y].w));
      }
      if (ABLATE & 4) {
        // Register-lean order: one column operand live at a time.
#pragma unroll
        for (int y = 0; y < RJ; ++y) {
          const uint4 c = l_cols[kr * TILE + y * TJT + tj];
#pragma unroll
          for (int x = 0; x < RI; ++x) {
            const uint32_t hom_both = ri[x].z & c.z;
            if (FULL) c_hh[x][y] += __builtin_popcount(hom_both);
            c_opp[x][y] += __builtin_popcount(
                __builtin_amdgcn_bitop3_b32(ri[x].y, c.y, hom_both, 0x28));
            c_bh[x][y] += __builtin_popcount(ri[x].x & c.x);
            c_hi[x][y] += __builtin_popcount(ri[x].x & c.w);
            c_hj[x][y] += __builtin_popcount(ri[x].w & c.x);
          }
        }
        continue;
      }
#pragma unroll
      for (int x = 0; x < RI; ++x) {
#pragma unroll
        for (int y = 0; y < RJ; ++y) {
          const uint32_t hom_both = ri[x].z & cj[y].z;
          if (FULL) c_hh[x][y] += __builtin_popcount(hom_both);
          c_opp[x][y] += __builtin_popcount(
              __builtin_amdgcn_bitop3_b32(ri[x].y, cj[y].y, hom_both, 0x28));
          c_bh[x][y] += __builtin_popcount(ri[x].x & cj[y].x);
          c_hi[x][y] += __builtin_popcount(ri[x].x & cj[y].w);
          c_hj[x][y] += __builtin_popcount(ri[x].w & cj[y].x);
        }
      }
    }
  }

  // --- epilogue: kinship, threshold, append (cuking.cu:284-313) ---
  const EmitCtx emit_ctx = make_emit_ctx(a);
#pragma unroll
  for (int x = 0; x < RI; ++x) {
    const uint32_t li = tr * TILE + x * TIT + ti;  // row inside the block
#pragma unroll
    for (int y = 0; y < RJ; ++y) {
      const uint32_t lj = tc * TILE + y * TJT + tj;
      // cuking.cu:199 plus the tile padding
      const bool valid = li < a.geo.num_rows && lj < a.geo.num_cols &&
                         a.i_begin + li < a.j_begin + lj;
      if (FULL)
        full_epilogue_pair(a, valid, li, lj, c_hi[x][y], c_hj[x][y], c_bh[x][y],
                           c_opp[x][y], c_hh[x][y]);
      else
        lean_epilogue_pair(emit_ctx, valid, li, lj, c_hi[x][y], c_hj[x][y],
                           c_bh[x][y], c_opp[x][y], lane);
    }
  }
}

// ---------------------------------------------------------------------------
// king_stream_kernel: one pair per wavefront, straight from the reference
// layout; lanes stride over the 64-bit words (coalesced 512-byte rows), six
// sums with the reference's own masks (cuking.cu:219-239), wave64 butterfly
// reduction (replaces the 32-lane shuffle + shared-memory atomics of
// cuking.cu:242-282: a pair never spans more than one wavefront here).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void king_stream_kernel(
    const cuking_submatrix sm, const uint32_t words_per_sample,
    const uint64_t *__restrict__ bits, const float kin_threshold,
    const uint32_t max_results, cuking_result *results, uint32_t *result_index,
    uint32_t *result_overflow, cuking_counts *dense_counts,
    const uint64_t block_offset) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t num_cols = sm_num_cols(sm);
  const uint32_t col_groups = (num_cols + 3) / 4;
  const uint64_t b = block_offset + blockIdx.x;
  const uint32_t li = (uint32_t)(b / col_groups);
  const uint32_t lj = (uint32_t)(b % col_groups) * 4 + wave;
  const uint32_t i = sm.i_begin + li, j = sm.j_begin + lj;
  if (lj >= num_cols || i >= j) return;  // whole wavefront leaves together

  const uint32_t n = words_per_sample / 2;
  const uint64_t *het_i_w =
      bits + (uint64_t)sm_sample_offset(sm, i) * words_per_sample;
  const uint64_t *alt_i_w = het_i_w + n;
  const uint64_t *het_j_w =
      bits + (uint64_t)sm_sample_offset(sm, j) * words_per_sample;
  const uint64_t *alt_j_w = het_j_w + n;

  uint32_t s_het_i = 0, s_het_j = 0, s_both = 0, s_opp = 0, s_conc = 0,
           s_shared = 0;
  for (uint32_t k = lane; k < n; k += 64) {
    const uint64_t hi = het_i_w[k], ai = alt_i_w[k];
    const uint64_t hj = het_j_w[k], aj = alt_j_w[k];
    const uint64_t ri = ~(hi | ai), rj = ~(hj | aj);
    const uint64_t defined = ~((hi & ai) | (hj & aj));
    s_het_i += __popcll(hi & defined);
    s_het_j += __popcll(hj & defined);
    s_both += __popcll(hi & hj & defined);
    s_opp += __popcll(((ri & aj) | (ai & rj)) & defined);
    s_conc += __popcll(((ri & rj) | (ai & aj)) & defined);
    s_shared += __popcll(defined);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s_het_i += __shfl_xor(s_het_i, off);
    s_het_j += __shfl_xor(s_het_j, off);
    s_both += __shfl_xor(s_both, off);
    s_opp += __shfl_xor(s_opp, off);
    s_conc += __shfl_xor(s_conc, off);
    s_shared += __shfl_xor(s_shared, off);
  }
  if (lane != 0) return;
  if (dense_counts != nullptr) {
    cuking_counts c;
    c.het_i = s_het_i;
    c.het_j = s_het_j;
    c.both_het = s_both;
    c.opposing_hom = s_opp;
    c.concordant_hom = s_conc;
    c.shared = s_shared;
    dense_counts[(uint64_t)li * num_cols + lj] = c;
    return;
  }
  const float kin = king_kinship(s_het_i, s_het_j, s_both, s_opp);
  if (kin > kin_threshold) {
    const uint32_t ibs0 = s_opp, ibs2 = s_conc + s_both;
    emit_result(i, j, kin, ibs0, s_shared - ibs0 - ibs2, ibs2, max_results,
                results, result_index, result_overflow);
  }
}

// ---------------------------------------------------------------------------
// pack_kernel: cuking.cu:675-703 with device atomics (AtomicClearBit :317-323).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_kernel(
    const cuking_submatrix sm, const uint32_t words_per_sample,
    uint64_t *bit_set, const int64_t *__restrict__ row_idx,
    const int64_t *__restrict__ col_idx, const int32_t *__restrict__ n_alt,
    const uint64_t num_triples, uint32_t *status) {
  const uint32_t plane_words = words_per_sample / 2;
  const uint64_t plane_bits = (uint64_t)plane_words * 64;
  uint32_t bad = 0;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
       t < num_triples; t += (uint64_t)gridDim.x * blockDim.x) {
    const int64_t col = col_idx[t];
    if (col < 0 || col > 0xFFFFFFFFll || !sm_contains(sm, (uint32_t)col))
      continue;  // :677-679
    const int64_t row = row_idx[t];
    if (row < 0 || (uint64_t)row >= plane_bits) {
      bad |= 2u;
      continue;
    }
    unsigned long long *het = reinterpret_cast<unsigned long long *>(
        bit_set + (uint64_t)sm_sample_offset(sm, (uint32_t)col) *
                      words_per_sample);
    unsigned long long *hom = het + plane_words;
    const unsigned long long clear = ~(1ull << (row & 63));
    const uint64_t word = (uint64_t)row >> 6;
    const int32_t g = n_alt[t];
    if (g == 0) {  // hom-ref
      atomicAnd(het + word, clear);
      atomicAnd(hom + word, clear);
    } else if (g == 1) {  // het keeps its het bit
      atomicAnd(hom + word, clear);
    } else if (g == 2) {  // hom-var keeps its hom_var bit
      atomicAnd(het + word, clear);
    } else {
      bad |= 1u;  // :698-702
    }
  }
  if (bad) atomicOr(status, bad);
}

// The same for triples a host has already filtered to the block and narrowed:
// site index, and block-local sample offset (bits 0..29) with the allele count
// in bits 30..31 -- 8 bytes per genotype on the wire instead of 20.
__global__ __launch_bounds__(256) void pack_compact_kernel(
    const uint32_t words_per_sample, const uint32_t num_samples, uint64_t *bit_set,
    const uint32_t *__restrict__ site, const uint32_t *__restrict__ sample_alt,
    const uint64_t num_triples, uint32_t *status) {
  const uint32_t plane_words = words_per_sample / 2;
  const uint64_t plane_bits = (uint64_t)plane_words * 64;
  uint32_t bad = 0;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
       t < num_triples; t += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t sa = sample_alt[t], row = site[t];
    const uint32_t sample = sa & 0x3FFFFFFFu, g = sa >> 30;
    if (row >= plane_bits || sample >= num_samples) {
      bad |= 2u;
      continue;
    }
    unsigned long long *het = reinterpret_cast<unsigned long long *>(
        bit_set + (uint64_t)sample * words_per_sample);
    unsigned long long *hom = het + plane_words;
    const unsigned long long clear = ~(1ull << (row & 63));
    const uint32_t word = row >> 6;
    if (g == 0) {
      atomicAnd(het + word, clear);
      atomicAnd(hom + word, clear);
    } else if (g == 1) {
      atomicAnd(hom + word, clear);
    } else if (g == 2) {
      atomicAnd(het + word, clear);
    } else {
      bad |= 1u;
    }
  }
  if (bad) atomicOr(status, bad);
}

uint64_t g_max_blocks_override = 0;  // tests: force splitting at small sizes

template <int TIT, int TJT, int RI, int RJ, int KC, int KU, int MINW, bool FULL,
          int ABLATE = 0, int PHASED = 0>
hipError_t launch_variant(const TiledArgs &args, uint64_t num_tiles,
                          uint32_t lds_bytes, hipStream_t stream) {
  auto kernel = king_tiled_kernel<TIT, TJT, RI, RJ, KC, KU, MINW, FULL, ABLATE, PHASED>;
  // The attribute belongs to the function object of ONE device: a host with
  // contexts on several GPUs has to set it on each of them.
  static DeviceOnce attr_set;
  if (!attr_set.done()) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void *>(kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    attr_set.mark();
  }
  // One launch may not exceed 2^32 - 1 threads in x (HIP truncates silently
  // beyond that), so long tile ranges go out as several launches.
  const uint64_t cap = max_blocks_per_launch(TIT * TJT);
  uint64_t done = 0;
  while (done < num_tiles) {
    const uint64_t n = (num_tiles - done < cap) ? num_tiles - done : cap;
    TiledArgs a = args;
    a.tile_begin = args.tile_begin + done;
    kernel<<<dim3((uint32_t)n), dim3(TIT * TJT), lds_bytes, stream>>>(a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    done += n;
  }
  return hipSuccess;
}

const TiledVariant kVariants[kNumTiledVariants] = {
    {"t64_r4x2_k16_phased", 64, 16, 512, 2 * 2 * 16 * 64 * 16, kLayoutWord},
    {"t64_r4x4_k8_w4", 64, 8, 256, 2 * 2 * 8 * 64 * 16, kLayoutWord},
    {"t128_r8x4_k8_w2", 128, 8, 512, 2 * 2 * 8 * 128 * 16, kLayoutWord},
    {"t128_r4x4_k8_w4", 128, 8, 1024, 2 * 2 * 8 * 128 * 16, kLayoutWord},
    {"t64_r4x4_k16_w4", 64, 16, 256, 2 * 2 * 16 * 64 * 16, kLayoutWord},
    // Matrix cores: 128 x 128 pairs per workgroup, 256 sites per k-step,
    // 16 KiB LDS stages (king_mfma.hip).
    {"t128_mfma_fp4", 128, 8, 256, kMfmaLdsBytes, kLayoutQuad},
    // Matrix cores, four plane products on one fp4 code per site: 32 KiB LDS
    // stages (king_mfma.hip).
    {"t128_mfma_fp4_n4", 128, 8, 256, kMfmaN4LdsBytes, kLayoutNibble},
    // Matrix cores, ONE plane product as a bound on kinship + exact recount of
    // what it lets through: 256 x 256 pairs per workgroup, 128 sites per k-step
    // (king_filter.hip).
    {"t256_mfma_fp4_filter", kFilterTile, 8, 256, kFilterLdsBytes, kLayoutNibbleStats},
#ifdef CUKING_TUNING
    {"phased_both_barriers", 64, 16, 512, 2 * 2 * 16 * 64 * 16, kLayoutWord},
    {"phased_bar_before_popcount_only", 64, 16, 512, 2 * 2 * 16 * 64 * 16, kLayoutWord},
    {"phased_k8", 64, 8, 512, 2 * 2 * 8 * 64 * 16, kLayoutWord},
    {"unphased_noldsread", 64, 8, 256, 2 * 2 * 8 * 64 * 16, kLayoutWord},
    {"unphased_nodma_nobarrier", 64, 8, 256, 2 * 2 * 8 * 64 * 16, kLayoutWord},
    {"unphased_w5", 64, 8, 256, 2 * 2 * 8 * 64 * 16, kLayoutWord},
#endif
};


}  // namespace

void set_max_blocks_per_launch(uint64_t blocks) { g_max_blocks_override = blocks; }

uint64_t max_blocks_per_launch(uint32_t threads) {
  const uint64_t hw = 0xFFFFFFFFull / threads;
  return (g_max_blocks_override && g_max_blocks_override < hw) ? g_max_blocks_override : hw;
}

const TiledVariant &tiled_variant(int v) { return kVariants[v]; }

hipError_t launch_tiled(int variant, bool full, const TiledArgs &args,
                        uint64_t num_tiles, hipStream_t stream) {
  if (num_tiles == 0) return hipSuccess;
  const uint32_t lds = kVariants[variant].lds_bytes;
#define CUKING_SHAPE(...)                                                      \
  (full ? launch_variant<__VA_ARGS__, true>(args, num_tiles, lds, stream)      \
        : launch_variant<__VA_ARGS__, false>(args, num_tiles, lds, stream))
#define CUKING_PHASED(KC_, MODE_)                                              \
  (full ? launch_variant<16, 32, 4, 2, KC_, KC_, 4, true, 0, MODE_>(args, num_tiles, lds, stream) \
        : launch_variant<16, 32, 4, 2, KC_, KC_, 4, false, 0, MODE_>(args, num_tiles, lds, stream))
  switch (variant) {
    case 0: return CUKING_PHASED(16, 1);
    case 1: return CUKING_SHAPE(16, 16, 4, 4, 8, 2, 4);
    case 2: return CUKING_SHAPE(16, 32, 8, 4, 8, 2, 2);
    case 3: return CUKING_SHAPE(32, 32, 4, 4, 8, 2, 4);
    case 4: return CUKING_SHAPE(16, 16, 4, 4, 16, 2, 4);
    case kMfmaVariant: return launch_mfma(full, false, args, num_tiles, lds, stream);
    case kMfmaN4Variant: return launch_mfma(full, true, args, num_tiles, lds, stream);
    case kMfmaFilterVariant: {
      // The bound only helps the lean form with a threshold inside (0, 1/2);
      // otherwise the four-product kernel runs on the quadrants of the same
      // 256-sample tiles.
      const bool filter = !full && args.dense_counts == nullptr && args.quad == 0 &&
                          args.kin_threshold > 0.f && args.kin_threshold < 0.5f &&
                          args.filter_ctrl != nullptr;
      if (filter) return launch_filter(args, num_tiles, stream);
      TiledArgs a = args;
      if (a.quad == 0) {
        a.quad = 1;
        a.tile_begin = args.tile_begin * 4;
        num_tiles *= 4;
      }
      return launch_mfma(full, true, a, num_tiles, kMfmaN4LdsBytes, stream);
    }
#ifdef CUKING_TUNING
    case 8: return CUKING_PHASED(16, 2);
    case 9: return CUKING_PHASED(16, 3);
    case 10: return CUKING_PHASED(8, 1);
    case 11: return launch_variant<16, 16, 4, 4, 8, 2, 4, false, 1>(args, num_tiles, lds, stream);
    case 12: return launch_variant<16, 16, 4, 4, 8, 2, 4, false, 2>(args, num_tiles, lds, stream);
    case 13: return launch_variant<16, 16, 4, 4, 8, 1, 5, false, 4>(args, num_tiles, lds, stream);
#endif
    default: return hipErrorInvalidValue;
  }
#undef CUKING_SHAPE
#undef CUKING_PHASED
}

hipError_t launch_prepare_planes(uint32_t layout, const uint64_t *d_bit_sets,
                                 uint32_t words_per_sample,
                                 const PlaneGeometry &geo, uint4 *d_planes,
                                 uint32_t s_tile_begin, uint32_t s_tile_end,
                                 hipStream_t stream) {
  const uint32_t all = (geo.s_stride + kPrepSamples - 1) / kPrepSamples;
  if (s_tile_end > all) s_tile_end = all;
  if (s_tile_begin >= s_tile_end) return hipSuccess;
  const bool nibble = layout == kLayoutNibble || layout == kLayoutNibbleStats;
  const uint32_t block_words = nibble ? kNibWords : kPrepWords;
  const dim3 grid(s_tile_end - s_tile_begin,
                  (geo.k_words + 2 * block_words - 1) / (2 * block_words));
  if (grid.y == 0) return hipSuccess;
  if (layout == kLayoutQuad)
    prepare_quads_kernel<<<grid, dim3(256), 0, stream>>>(
        d_bit_sets, words_per_sample, geo, d_planes, s_tile_begin);
  else if (layout == kLayoutNibbleStats)
    return hipErrorInvalidValue;  // (king_abi.hip prepares that layout step by step)
  else if (nibble)
    prepare_nibbles_kernel<true, false><<<grid, dim3(256), 0, stream>>>(
        d_bit_sets, words_per_sample, geo, d_planes, s_tile_begin, nullptr, nullptr, nullptr);
  else
    prepare_planes_kernel<<<grid, dim3(256), 0, stream>>>(
        d_bit_sets, words_per_sample, geo, d_planes, s_tile_begin);
  return hipGetLastError();
}

hipError_t launch_prepare_nibbles(bool codes, bool t2, const uint64_t *d_bit_sets,
                                  uint32_t words_per_sample, const PlaneGeometry &geo,
                                  uint4 *d_planes, const uint32_t *perm, uint32_t s_tile_begin,
                                  uint32_t s_tile_end, const uint32_t *gate,
                                  const uint32_t *ready, hipStream_t stream) {
  const uint32_t all = (geo.s_stride + kPrepSamples - 1) / kPrepSamples;
  if (s_tile_end > all) s_tile_end = all;
  if (s_tile_begin >= s_tile_end || (!codes && !t2)) return hipSuccess;
  const dim3 grid(s_tile_end - s_tile_begin, (geo.k_words + 2 * kNibWords - 1) / (2 * kNibWords));
  if (grid.y == 0) return hipSuccess;
  if (codes && t2)
    prepare_nibbles_kernel<true, true><<<grid, dim3(256), 0, stream>>>(
        d_bit_sets, words_per_sample, geo, d_planes, s_tile_begin, perm, gate, ready);
  else if (codes)
    prepare_nibbles_kernel<true, false><<<grid, dim3(256), 0, stream>>>(
        d_bit_sets, words_per_sample, geo, d_planes, s_tile_begin, perm, gate, ready);
  else
    prepare_nibbles_kernel<false, true><<<grid, dim3(256), 0, stream>>>(
        d_bit_sets, words_per_sample, geo, d_planes, s_tile_begin, perm, gate, ready);
  return hipGetLastError();
}

namespace {
__global__ void mark_codes_ready_kernel(const uint32_t *gate, uint32_t *ready) {
  if ((gate[kCtrlGate] | gate[kCtrlDense]) != 0) *ready = 1;
}
}  // namespace

hipError_t launch_mark_codes_ready(const uint32_t *gate, uint32_t *ready, hipStream_t stream) {
  mark_codes_ready_kernel<<<dim3(1), dim3(1), 0, stream>>>(gate, ready);
  return hipGetLastError();
}

hipError_t launch_stream(const cuking_submatrix &sm, uint32_t words_per_sample,
                         const uint64_t *d_bit_sets, float kin_threshold,
                         uint32_t max_results, cuking_result *d_results,
                         uint32_t *d_result_index, uint32_t *d_result_overflow,
                         cuking_counts *d_dense_counts, hipStream_t stream) {
  const uint64_t rows = sm_num_rows(sm);
  const uint64_t col_groups = ((uint64_t)sm_num_cols(sm) + 3) / 4;
  const uint64_t blocks = rows * col_groups;
  const uint64_t cap = max_blocks_per_launch(256);
  for (uint64_t done = 0; done < blocks; done += cap) {
    const uint64_t n = blocks - done < cap ? blocks - done : cap;
    king_stream_kernel<<<dim3((uint32_t)n), dim3(256), 0, stream>>>(
        sm, words_per_sample, d_bit_sets, kin_threshold, max_results, d_results,
        d_result_index, d_result_overflow, d_dense_counts, done);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t launch_pack(const cuking_submatrix &sm, uint32_t words_per_sample,
                       uint64_t *d_bit_set, const int64_t *d_row_idx,
                       const int64_t *d_col_idx, const int32_t *d_n_alt,
                       size_t num_triples, uint32_t *d_status,
                       hipStream_t stream) {
  if (num_triples == 0) return hipSuccess;
  uint64_t blocks = (num_triples + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride the rest
  pack_kernel<<<dim3((uint32_t)blocks), dim3(256), 0, stream>>>(
      sm, words_per_sample, d_bit_set, d_row_idx, d_col_idx, d_n_alt,
      num_triples, d_status);
  return hipGetLastError();
}

hipError_t launch_pack_compact(uint32_t words_per_sample, uint32_t num_samples,
                               uint64_t *d_bit_set, const uint32_t *d_site,
                               const uint32_t *d_sample_alt, size_t num_triples,
                               uint32_t *d_status, hipStream_t stream) {
  if (num_triples == 0) return hipSuccess;
  uint64_t blocks = (num_triples + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride the rest
  pack_compact_kernel<<<dim3((uint32_t)blocks), dim3(256), 0, stream>>>(
      words_per_sample, num_samples, d_bit_set, d_site, d_sample_alt, num_triples, d_status);
  return hipGetLastError();
}

}  // namespace cuking
