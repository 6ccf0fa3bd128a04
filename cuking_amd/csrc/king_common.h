// Shared host/device declarations of the KING hot path (gfx950 only).
#ifndef CUKING_AMD_KING_COMMON_H_
#define CUKING_AMD_KING_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cuking_amd.h"
#include "king_submatrix.h"

namespace cuking {

// ---------------------------------------------------------------------------
// Plane layout (device-internal; built by prepare_planes from the reference
// bitset of cuking.cu:507-523).
//
// For every 32-site word k and every stored sample s one uint4:
//     x = H    het and defined            (het & ~hom_var)
//     y = A    hom-alt                    (hom_var & ~het)
//     z = Hom  hom-ref or hom-alt         (~het)
//     w = D    defined (not missing)      ~(het & hom_var)
// laid out k-major: planes[k * s_stride + s].  Row samples occupy
// [0, rows_padded), column samples [col_base, col_base + cols_padded); for a
// diagonal block col_base = 0 and the samples are stored once (cuking.cu:161).
// Padding samples and padding k-words are all-zero (= everything missing),
// so they add nothing to any sum.
//
// Quad layout (kLayoutQuad, the five-product matrix-core kernel): the reference's two bit
// planes as they are, transposed: for every quad q of four consecutive 32-site
// words, plane p (0 = het, 1 = hom_var, cuking.cu:507-511) and sample s one
// uint4 holding those four words, planes[(q * 2 + p) * s_stride + s].  Half
// the bytes of the word layout (2 bits per sample and site); the kernel derives
// the A / R / H / D fragments with one three-input bit operation each.
// Padding samples and k-words have both bits set (= missing).
//
// Nibble layout (kLayoutNibble, the four-product matrix-core kernel): ONE fp4
// code per site, ready to be an MFMA operand after a single AND: for every
// group j of 32 sites and sample s one uint4 of 32 nibbles,
// planes[j * s_stride + s], site 32 j + 8 d + t in nibble t of dword d:
//     bit 0  H   het                       (E2M1 0001 = 0.5)
//     bit 1  D   defined                   (0010 = 1.0)
//     bit 2  Y   hom-ref or hom-alt        (0100 = 2.0)
//     bit 3  A   hom-alt: the SIGN of Y    (1100 = -2.0)
// i.e. hom-ref 0110, het 0011, hom-alt 1110, missing / padding 0000.  4 bits per
// sample and site = twice the bytes of the quad layout.  Behind the codes
// (at planes + k_words * s_stride) sits a het-only copy in quad form, one uint4
// per 128 sites and sample, hetq[q * s_stride + s], for the full form's hom_hom
// pass (1 bit per sample and site more).
// ---------------------------------------------------------------------------
constexpr uint32_t kLayoutWord = 0;
constexpr uint32_t kLayoutQuad = 1;
constexpr uint32_t kLayoutNibble = 2;
// ... plus what the filter kernel (king_filter.hip) reads, behind the het-only
// copy: the T2 layout -- T = hom-ref minus hom-alt alone, TWO bits per site (hom,
// hom-alt), for every unit u of 64 sites and sample s one uint4, t2[u * s_stride
// + s]: nibble t of dword d holds site 64 u + 8 d + t in bits 2-3 (code 0100 /
// 1100 = +-2.0 after `& 0xCCCCCCCC`) and site 64 u + 32 + 8 d + t in bits 0-1 (the
// same code after `<< 2`), half the bytes of the codes -- and one float2 per stored
// sample: (homozygous - missing site count, het count).
constexpr uint32_t kLayoutNibbleStats = 3;

struct PlaneGeometry {
  uint32_t num_rows, num_cols;        // block shape in samples
  uint32_t rows_padded, cols_padded;  // rounded up to the tile edge
  uint32_t col_base;                  // plane index of column sample 0
  uint32_t s_stride;                  // uint4 per k-row
  uint32_t k_words;                   // 32-site words, padded to the K chunk
  uint32_t diag;                      // rows and columns are the same samples
};

// Tiles of the block's pair space, enumerated band by band: a band is
// `band_rows` tile-rows high and is walked column-major, so that workgroups
// resident at the same time share few row/column strips (L2 reuse per XCD).
// A diagonal block only has tiles with tile_row <= tile_col.
struct TileSpace {
  uint32_t tiles_r, tiles_c;
  uint32_t band_rows;
  uint32_t diag;

  __host__ __device__ uint32_t num_bands() const {
    return (tiles_r + band_rows - 1) / band_rows;
  }
  // Tiles in band b.
  __host__ __device__ uint64_t band_tiles(uint32_t b) const {
    const uint32_t r0 = b * band_rows;
    const uint32_t h = (tiles_r - r0 < band_rows) ? tiles_r - r0 : band_rows;
    if (!diag) return (uint64_t)h * tiles_c;
    const uint32_t ncols = tiles_c - r0;  // columns r0 .. tiles_c-1
    return (uint64_t)h * (h + 1) / 2 + (uint64_t)(ncols - h) * h;
  }
  // Decodes index u inside band b.
  __host__ __device__ void decode(uint32_t b, uint64_t u, uint32_t *tr,
                                  uint32_t *tc) const {
    const uint32_t r0 = b * band_rows;
    const uint32_t h = (tiles_r - r0 < band_rows) ? tiles_r - r0 : band_rows;
    if (!diag) {
      *tc = (uint32_t)(u / h);
      *tr = r0 + (uint32_t)(u % h);
      return;
    }
    const uint64_t tri = (uint64_t)h * (h + 1) / 2;
    if (u < tri) {
      // Column r0 + c holds c + 1 tiles (rows r0 .. r0 + c).
      uint32_t c = 0;
      uint64_t before = 0;
      while (before + c + 1 <= u) {
        before += c + 1;
        ++c;
      }
      *tc = r0 + c;
      *tr = r0 + (uint32_t)(u - before);
    } else {
      const uint64_t v = u - tri;
      *tc = r0 + h + (uint32_t)(v / h);
      *tr = r0 + (uint32_t)(v % h);
    }
  }
};

// Arguments of the tiled pair kernel.
struct TiledArgs {
  const uint4 *planes;
  PlaneGeometry geo;
  TileSpace tiles;
  const uint64_t *band_prefix;  // device; num_bands + 1 entries
  uint64_t tile_begin;          // first tile of this launch
  // Rectangle mode (rect_rows != 0): the launch covers the rect_rows tile rows
  // rect_row0 + k * rect_row_stride x the rect_cols tile columns from
  // rect_col0, enumerated in bands of band_rows rows, column-major inside a
  // band.  Used to start on the columns whose samples
  // have already arrived while the rest of the bitset is still in flight; the
  // stride lets the ranks of a node take tile rows round-robin.
  uint32_t rect_rows, rect_cols, rect_row0, rect_col0, rect_row_stride;
  uint32_t i_begin, j_begin;    // global sample index of row / column 0
  float kin_threshold;
  uint32_t max_results;
  cuking_result *results;
  uint32_t *result_index;
  uint32_t *result_overflow;
  cuking_counts *dense_counts;  // non-null: diagnostic mode, no threshold
  // Reference-layout bitset of the block (the lean kernel recounts hom/hom
  // sites from it for the pairs it emits).
  const uint64_t *bits;
  uint32_t words_per_sample;
  // Matrix-core kernel, remainder launch (king_mfma.hip): split_tiles tiles
  // from tile_begin are cut into split_wgs equal pieces of k-steps; partial
  // results are parked in split_scratch, tickets in split_counters (zero
  // between launches).  split_wgs == 0 or no scratch: never split.
  uint32_t split_tiles, split_wgs;
  // ... preceded, in the same launch, by split_whole workgroups that take one
  // whole tile each (tiles tile_begin .. tile_begin + split_whole - 1).
  uint32_t split_whole;
  uint32_t *split_scratch, *split_counters;
  // XCD-aware order (matrix-core kernel, whole-tile launches): workgroups are
  // dealt round-robin over the 8 XCDs, each with its own L2.  With xcd_chunk != 0
  // the workgroups resident on one XCD at a time hold CONSECUTIVE tiles of the
  // band order (few row/column strips per L2) instead of every eighth one:
  // xcd_chunk == 1: patches of 32 consecutive tiles, patch p on XCD p % 8
  // (workgroup b = XCD b % 8, j = b / 8 takes tile ((j / 32) * 8 + b % 8) * 32 +
  // j % 32); xcd_chunk > 1: one contiguous chunk of xcd_chunk tiles per XCD
  // (workgroup b takes tile (b % 8) * xcd_chunk + b / 8).  On entry to a launch
  // function the field is the context's switch (0 off, 1 chunks, 2 patches).
  uint32_t xcd_chunk, launch_tiles;
  // Dynamic tail (matrix-core kernel, whole-tile launches of many rounds): the
  // XCDs of an MI355X run this kernel at rates 2-3 % apart and a launch's
  // workgroups are dealt to them statically, so the launch would end when the
  // slowest XCD does.  The last dyn_tiles tiles of a launch are therefore not
  // tied to a workgroup index: dyn_wgs (> dyn_tiles) workgroups behind the
  // launch_tiles statically mapped ones each take the next tile from a counter
  // (in split_counters, zero between launches) or leave at once when none is
  // left -- an XCD that gets through its static share early takes more of them.
  // On entry to a launch function dyn_tiles is the context's threshold: launches
  // of at least that many tiles get a dynamic tail (0 = never).
  uint32_t dyn_tiles, dyn_wgs;
  // Quadrant mode (quad != 0): the tile space above is one of 256-sample tiles
  // (the filter variant's geometry) and the launch enumerates their four
  // 128-sample quadrants: unit t = quadrant t % 4 (row half t / 2 % 2, column
  // half t % 2) of tile t / 4.  Lets every 128-tile kernel serve a context whose
  // tile indices, tile bounds and prepared ranges are in 256-sample units.
  uint32_t quad;
  // Tile-list mode (tile_list != nullptr; matrix-core kernels): the 128-sample
  // tiles to compute are the first min(*tile_list_count, tile_list_cap) entries
  // (x = tile row, y = tile column) of a device list written by an earlier
  // kernel of the stream; workgroup b takes entries b, b + grid, ...
  const uint2 *tile_list;
  const uint32_t *tile_list_count;
  uint32_t tile_list_cap;
  // Filter variant (king_filter.hip): per-sample statistics, control words
  // (candidate count, dense-quadrant count), the candidate pairs and the
  // quadrants handed to the exact kernel instead.
  const float2 *sample_stats;
  const uint4 *t2;         // the T2 layout (king_common.h, kLayoutNibbleStats)
  uint32_t *filter_ctrl;
  unsigned long long *filter_totals;
  uint2 *cand_list;
  uint32_t cand_cap;       // entries of cand_list
  uint32_t quadrant_cap;   // candidates per quadrant beyond which it is "dense"
  uint2 *dense_list;
  uint32_t dense_cap;
  // Filter kernel, remainder of a short launch (king_filter.hip): workgroups from
  // fsplit_first on take ONE of fsplit_parts equal pieces of the k range of tile
  // fsplit_tile0 + (index - fsplit_first) / fsplit_parts of the launch; a piece
  // parks its sums in its slab, and the workgroup that delivers a tile's last
  // piece adds the others to its own (tickets: zero between launches).
  uint32_t fsplit_parts, fsplit_first, fsplit_tile0;
  float4 *fsplit_slabs;
  uint32_t *fsplit_tickets;
  // Filter kernel, check points inside the k loop (king_filter.hip, "Check points"):
  // after a number of k-steps (even) the workgroup tests its 256 x 256 sums against the
  // bound evaluated on the sites so far -- per-sample counts over that prefix, one
  // float per plane sample.  Check 1 is rigorous (every term of X is non-negative): a
  // tile none of whose pairs can still become a candidate leaves there.  Check 0
  // is a forecast (the bound scaled to the prefix) for short launches: a tile most
  // of whose quadrants look dense leaves for the exact kernel at once.  The pipeline is
  // drained at a check, so the wavefronts exchange their findings through LDS;
  // tile_done[t] = 1 tells the fallback launch that tile t of the chunk needs nothing more.
  // check_steps[k] (device memory, written by the kernel that computed the counts): k-steps
  // behind share k of kCheckShares64 from the first site on (0 = bitset too short for
  // checks); prefix_u[(x - 1) * s_stride + s]: u of plane sample s over the k-steps in front
  // of phase boundary x = 1 .. kNumCum (king_common.h phase_step).
  // Entry 0 is the forecast's (used when check0 != 0: short launches), entries 1 .. the
  // rigorous check's: every workgroup picks the same one from the threshold and the
  // cohort's mean missing and het rates (cohort_sums: samples, missing calls, het calls).
  const uint32_t *check_steps;
  const float *prefix_u;
  const unsigned long long *cohort_sums;
  // switches (check1, bits 0-7: 0 off, 1 automatic, 2 + k: entry k forced; bits 8-15: live
  // pairs per quadrant up to which a tile that fails the rigorous check hands them to the
  // candidate list and leaves anyway, kCheckEmitCap unless a test says otherwise; bit 16:
  // the tiles may give up at once where the cohort's sums say the bound lets every pair
  // through -- the forecast's switch, whatever the launch's length)
  uint32_t check0, check1;
  // Rotated tiles (king_filter.hip): 0 every tile starts at its first k-step; 1 a whole tile
  // with check points starts at the phase boundary the other tiles of its XCD are at
  // (filter_ctrl + kCtrlPos: one position word per XCD) and wraps around; 2 test hook: a
  // phase drawn from the tile's index; 3 + j test hook: phase j.  rotate_min_steps: bitsets
  // of fewer k-steps are not rotated; nor are launches of fewer than rotate_min_tiles tiles
  // (the tiles of a few rounds have not drifted apart yet).
  uint32_t rotate, rotate_min_steps;
  uint32_t rotate_min_tiles;  // (host side: launches of fewer tiles are not rotated)
  // Persistent launch of the filter kernel (king_filter.hip king_filter_persistent_kernel):
  // on entry (host side) 1 = allowed; in a launch != 0: the indices of the one-tile-per-
  // workgroup grid that the resident workgroups take in turn before the dynamic tail.
  uint32_t persist_wgs;
  uint32_t persist_min_tiles;  // (host side: shorter launches go out one workgroup per tile)
  // one flag per tile of the launch chunk, directly behind the chunk's control words
  // (filter_ctrl + kCtrlChunkBytes: one memset clears both in front of a chunk)
  uint8_t *tile_done;
  // Persistent mode of the four-product kernel (king_mfma.hip), the filter's
  // fallback: when *gate != 0, workgroup b of a fixed grid takes units b, b + grid,
  // ... of the gate_count units from tile_begin on (same order as a whole-tile
  // launch: 32 consecutive units per XCD and round), skipping the quadrants of
  // tiles with skip_tiles[tile - skip_base] != 0; when *gate == 0 every workgroup
  // leaves at once.  gate == nullptr: not this mode.
  const uint32_t *gate;
  uint32_t gate_count;
  const uint8_t *skip_tiles;
  uint64_t skip_base;
  // Sample order of the kernel layout (kLayoutNibbleStats workspaces; nullptr: as
  // stored).  perm[p] = which stored sample of the block (row samples first, then the
  // columns of an off-diagonal block: the index into `bits`) sits at plane sample p, or
  // 0xFFFFFFFF for padding.  The conversion sorts the samples of a prepared range by
  // their share of missing calls (king_sort.hip), so that the few low-call-rate samples
  // of a cohort -- whose pairs the filter's bound cannot rule out -- fill a few tile rows
  // instead of spoiling every quadrant.  Pairs are enumerated in plane order (i < j there);
  // a record carries the original indices, smaller first.
  const uint32_t *perm;
  // Lazy codes (filter variant): the four-product kernel's nibble codes are converted by a
  // gated launch behind the filter kernel only if a quadrant went dense or a tile left
  // (codes_ready: device word, 1 once they are there for the prepared block).
  uint32_t *codes_ready;
};

// Prefix statistics: the k-steps (of 256 sites) a check may sit behind, as shares of the
// bitset in 1/64ths -- the per-sample prefix counts are computed for every entry when
// the layout is prepared, a launch picks the entries its threshold calls for.
// Live pairs per quadrant a tile may hand to the candidate list at the rigorous check
// (3.4 ns of recount each at 100k sites; the k-steps saved are ~50 us of a tile).
constexpr uint32_t kCheckEmitCap = 64;
constexpr uint32_t kNumCheckShares = 8;
constexpr uint32_t kCheckShares64[kNumCheckShares] = {8, 50, 53, 56, 58, 60, 61, 62};
// The k-steps of a bitset in kNumPhases PHASES: phase x is k-steps [phase_step(x), phase_step(x + 1)).
// A tile may start its k loop at any phase boundary and wrap around (king_filter.hip,
// "Rotated tiles": the tiles an XCD holds at a time read the same k-steps at the same time,
// whenever each of them started), so the per-sample prefix counts are kept CUMULATIVE at
// every phase boundary: u over phases [a, b) is the difference of two of them.
constexpr uint32_t kNumPhases = 128;  // (a multiple of 64: the checks' shares are 64ths)
constexpr uint32_t kPhasesPerShare = kNumPhases / 64;
constexpr uint32_t kNumCum = kNumPhases - 1;  // inner boundaries (0 is zero, the last the total)
__host__ __device__ inline uint32_t phase_step(uint32_t all_steps, uint32_t x) {
  return (uint32_t)((uint64_t)all_steps * x / kNumPhases);
}
// k-steps (of 256 sites) behind share k for a bitset of `all_steps` k-steps, counted from
// the bitset's first site; 0 when the bitset is too short for a check to pay.
__host__ __device__ inline uint32_t check_step_of(uint32_t all_steps, uint32_t k,
                                                  uint32_t min_steps = 64) {
  if (all_steps < min_steps) return 0;
  return phase_step(all_steps, kCheckShares64[k] * kPhasesPerShare);
}
// Remainder splitting: at most this many pieces per launch (one per CU), a slab of
// 256 x 256 float sums and a ticket word each.
constexpr uint32_t kFilterSplitSlabs = 256;
constexpr size_t kFilterSlabBytes = 256 * 256 * sizeof(float);
constexpr size_t kFilterTicketBytes = kFilterSplitSlabs * sizeof(uint32_t);
// Bytes of the plane workspace for a geometry.
__host__ __device__ inline size_t plane_bytes(const PlaneGeometry &g,
                                              uint32_t layout) {
  // word layout: 16 B per 32-site word and sample; quad layout: 2 x 16 B per
  // four words and sample; nibble layout: 16 B per 32 sites and sample.
  const size_t base = (size_t)g.k_words * g.s_stride * (layout == kLayoutQuad ? 8 : 16);
  if (layout == kLayoutNibble) return base + base / 4;  // + the het-only copy
  if (layout == kLayoutNibbleStats)
    return base + base / 4 + base / 2 + (size_t)g.s_stride * sizeof(float2) +
           (size_t)g.s_stride * kNumCum * sizeof(float) + 64 + 64 +
           // control block, perm, statistics before the sort, the sort's four arrays
           64 + (size_t)g.s_stride * (4 + 8 + 4 * kNumCum + 16);
  return base;
}
// Where the T2 layout and the per-sample statistics of kLayoutNibbleStats start
// (k_words is a multiple of 8: whole uint4 either way).
__host__ __device__ inline const uint4 *plane_t2(const uint4 *planes, const PlaneGeometry &g) {
  return planes + (size_t)g.k_words * g.s_stride * 5 / 4;
}
__host__ __device__ inline const float2 *plane_stats(const uint4 *planes, const PlaneGeometry &g) {
  return reinterpret_cast<const float2 *>(planes + (size_t)g.k_words * g.s_stride * 7 / 4);
}
// ... behind them the cumulative counts (kNumCum x s_stride floats) and the cohort's
// sums (three u64: samples counted, missing calls, het calls; s_stride is a multiple of
// the tile edge, so everything stays 16-byte aligned), then the k-steps behind each share
// as the prefix counts were computed for (kNumCheckShares u32).
__host__ __device__ inline const float *plane_prefix_u(const uint4 *planes, const PlaneGeometry &g) {
  return reinterpret_cast<const float *>(plane_stats(planes, g) + g.s_stride);
}
__host__ __device__ inline const unsigned long long *plane_cohort_sums(const uint4 *planes,
                                                                       const PlaneGeometry &g) {
  return reinterpret_cast<const unsigned long long *>(plane_prefix_u(planes, g) +
                                                      (size_t)g.s_stride * kNumCum);
}

// One compiled shape of the tiled kernel.
struct TiledVariant {
  const char *name;
  uint32_t tile;      // samples per tile edge
  uint32_t k_chunk;   // 32-site words staged per LDS buffer
  uint32_t threads;   // workgroup size
  uint32_t lds_bytes; // dynamic LDS
  uint32_t layout;    // kLayoutWord / kLayoutQuad
};

// The matrix-core variant accumulates in float32: exact while every sum stays
// below 2^24.
constexpr uint32_t kMfmaMaxSites = 1u << 24;

#ifdef CUKING_TUNING
constexpr int kNumTiledVariants = 14;  // + timing-only experiments
#else
constexpr int kNumTiledVariants = 8;
#endif
constexpr int kMfmaVariant = 5;    // five plane products, quad layout
constexpr int kMfmaN4Variant = 6;  // four plane products, nibble layout (king_mfma.hip)
// The four-product variant decides kinship on the integer num = hi + hj - 2 dd
// + 2 q, which equals the reference's float expression while every partial sum
// of that is exact, i.e. below 2^22 sites (include/cuking_amd.h, numerics
// contract); wider bitsets take the five-product variant.
constexpr uint32_t kMfmaN4MaxSites = 1u << 22;
constexpr int kMfmaN4Stages = 5;
constexpr uint32_t kMfmaN4LdsBytes = kMfmaN4Stages * 2 * 2 * 4 * 128 * 16;  // 5 x 32 KiB
static_assert(kMfmaN4LdsBytes <= 160 * 1024, "LDS of one CU");
// One plane product T_i.T_j per pair as a rigorous upper bound on kinship, exact
// recount of the few pairs it lets through (king_filter.hip); 256-sample tiles.
constexpr int kMfmaFilterVariant = 7;
constexpr uint32_t kFilterTile = 256;
constexpr uint32_t kFilterLdsBytes = 5 * 2 * 2 * 2 * 256 * 16;  // 5 x 32 KiB
static_assert(kFilterLdsBytes <= 160 * 1024, "LDS of one CU");
// Per launch chunk: at most this many 256-tiles (bounds the dense-quadrant list).
constexpr uint32_t kFilterChunkTiles = 1u << 17;
// Candidate pairs per chunk: 64 per quadrant of a full chunk.  (2^20 until the missing-rate
// curve, profiles/r03_missing_curve.txt: at configs[2] with 7 % missing the bound lets 4 x 10^6
// pairs through -- 14 ms of recounts -- but the list was full after the first million and
// 230,000 quadrants went to the exact kernel instead, 512 ms for a 160 ms pass.)
constexpr uint32_t kFilterCandCap = 1u << 25;
// Candidates per 128 x 128 quadrant beyond which the quadrant goes to the exact
// kernel: one wavefront per candidate costs 3.4 ns of chip time at 100k sites
// (1.65 M candidates in 5.6 ms), the four-product kernel 1.85 us per quadrant --
// break-even near 540, whatever the site count (both are linear in it;
// profiles/r03_filter_curve.txt).
constexpr uint32_t kFilterQuadrantCap = 384;
constexpr size_t kFilterCtrlBytes = 256;
// Control words (uint32 indices into filter_ctrl).  Zeroed in front of every launch
// chunk (kCtrlChunkBytes): 0 candidates of the chunk, 1 dense quadrants of the chunk
// (list slots), 2 dynamic-tail counter, 3 quadrants finished, 4 quadrants of tiles that
// left for the exact kernel at check 0, 5 "every remaining tile leaves" (most quadrants
// so far went dense), 6 gate of the fallback launch (some tile left).  Running totals
// since the scratch was allocated (u64 each, behind the tile flags: TiledArgs::
// filter_totals): 0 candidates, 1 quadrants handed to the exact kernel, 2 tiles that left
// at the rigorous check, 3 tiles that started at another phase than the first.
constexpr uint32_t kCtrlCand = 0, kCtrlDense = 1, kCtrlDyn = 2, kCtrlFinished = 3,
                   kCtrlLeft = 4, kCtrlAllLeave = 5, kCtrlGate = 6;
// ... and where the tiles of each XCD are (rotated tiles, king_filter.hip): word 8 + x: the
// 100 MHz counter's ticks per k-step x 16 as a tile of XCD x last measured them; from byte
// 64 on, per XCD kPosSlots u64 = (k-step, counted on through the tiles, a tile of the XCD
// had reached) << 32 | (when: low word of that counter) -- a tile writes slot (its
// workgroup's turn on the XCD) mod kPosSlots at its segment ends, a new tile joins the most
// advanced of them.
constexpr uint32_t kCtrlStepTicks = 8, kCtrlPos = 16, kPosSlots = 16;
// ... and behind the slots one ticket counter per XCD (persistent launch).
constexpr uint32_t kCtrlTickets = kCtrlPos + 8 * kPosSlots * 2;
constexpr size_t kCtrlChunkBytes = (kCtrlTickets + 16) * 4;
constexpr uint32_t kTotalCand = 0, kTotalDense = 1, kTotalEarly = 2, kTotalRotated = 3;
constexpr uint32_t kNumTotals = 4;
__host__ __device__ inline const uint32_t *plane_check_steps(const uint4 *planes,
                                                             const PlaneGeometry &g) {
  return reinterpret_cast<const uint32_t *>(plane_cohort_sums(planes, g) + 8);
}
// ... one control block (16 u32; word 0: codes_ready), then the sample order and what
// building it needs, every array s_stride entries long: perm (u32), the statistics in
// stored order before the sort (float2, then kNumCum floats), keys in / out and
// values in / out of the sort (u32 each).
__host__ __device__ inline uint32_t *plane_flags(const uint4 *planes, const PlaneGeometry &g) {
  return const_cast<uint32_t *>(plane_check_steps(planes, g)) + 16;
}
__host__ __device__ inline uint32_t *plane_perm(const uint4 *planes, const PlaneGeometry &g) {
  return plane_flags(planes, g) + 16;
}
__host__ __device__ inline float2 *plane_tmp_stats(const uint4 *planes, const PlaneGeometry &g) {
  return reinterpret_cast<float2 *>(plane_perm(planes, g) + g.s_stride);
}
__host__ __device__ inline float *plane_tmp_prefix(const uint4 *planes, const PlaneGeometry &g) {
  return reinterpret_cast<float *>(plane_tmp_stats(planes, g) + g.s_stride);
}
__host__ __device__ inline uint32_t *plane_sort_words(const uint4 *planes, const PlaneGeometry &g) {
  return reinterpret_cast<uint32_t *>(plane_tmp_prefix(planes, g) +
                                      (size_t)g.s_stride * kNumCum);
}
constexpr uint32_t kNoSample = 0xFFFFFFFFu;

// Scratch of one stream: the chunk's control words and one flag per tile of a launch chunk
// (cleared together in front of every chunk), the running totals, tickets, the candidate
// list, the dense-quadrant list, the remainder slabs.  The lists are sized for the block the
// scratch serves (`tiles` 256-sample tiles in its enumeration): 64 candidates per quadrant
// of a chunk, at most kFilterCandCap.
struct FilterScratchLayout {
  size_t tile_done, totals, tickets, cand, dense, slabs, bytes;
  uint32_t chunk_tiles, cand_entries;
};
inline FilterScratchLayout filter_scratch_layout(uint64_t tiles) {
  FilterScratchLayout l;
  l.chunk_tiles = (uint32_t)(tiles < kFilterChunkTiles ? (tiles ? tiles : 1) : kFilterChunkTiles);
  const uint64_t cand = (uint64_t)l.chunk_tiles * 4 * 64;
  l.cand_entries = (uint32_t)(cand < kFilterCandCap ? cand : kFilterCandCap);
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  l.tile_done = kCtrlChunkBytes;
  l.totals = up(l.tile_done + l.chunk_tiles);
  l.tickets = l.totals + kFilterCtrlBytes;
  l.cand = l.tickets + kFilterTicketBytes;
  l.dense = l.cand + up((size_t)l.cand_entries * sizeof(uint2));
  l.slabs = l.dense + up((size_t)l.chunk_tiles * 4 * sizeof(uint2));
  l.bytes = l.slabs + kFilterSplitSlabs * kFilterSlabBytes;
  return l;
}
inline bool is_mfma_variant(int v) {
  return v == kMfmaVariant || v == kMfmaN4Variant || v == kMfmaFilterVariant;
}
#ifndef CUKING_MFMA_STAGES
#define CUKING_MFMA_STAGES 6
#endif
constexpr uint32_t kMfmaLdsBytes = CUKING_MFMA_STAGES * 2 * 2 * 2 * 128 * 16;
// The full form parks its fifth sum (64 registers per lane) behind the stages.
constexpr uint32_t kMfmaParkBytes = 4 * 64 * 64 * 4;
static_assert(kMfmaLdsBytes + kMfmaParkBytes <= 160 * 1024, "LDS of one CU");
const TiledVariant &tiled_variant(int v);
// Enqueues tiles [args.tile_begin, args.tile_begin + num_tiles).
// full = accumulate all five sums for every pair (needed for the diagnostic
// counts; chosen when nearly every pair is expected to pass the threshold);
// otherwise the lean form: four sums in the main loop, IBS2 recounted for
// emitted pairs only.  Same records either way.
hipError_t launch_tiled(int variant, bool full, const TiledArgs &args,
                        uint64_t num_tiles, hipStream_t stream);
// The matrix-core kernel (king_mfma.hip); reached through launch_tiled.
hipError_t launch_mfma(bool full, bool nibble, const TiledArgs &args, uint64_t num_tiles,
                       uint32_t lds_bytes, hipStream_t stream);
// The four-product kernel, lean form, over args.tile_list with `grid` workgroups.
hipError_t launch_mfma_list(const TiledArgs &args, uint32_t grid, hipStream_t stream);
// The four-product kernel, lean form, persistent mode (TiledArgs::gate): `grid`
// workgroups walk num_units units from args.tile_begin on, if the gate is open.
hipError_t launch_mfma_gated(const TiledArgs &args, uint64_t num_units, uint32_t grid,
                             hipStream_t stream);
// The filter variant (king_filter.hip): num_tiles 256-sample tiles from
// args.tile_begin; needs args.filter_ctrl etc. (king_abi.hip: filter scratch).
hipError_t launch_filter(const TiledArgs &args, uint64_t num_tiles, hipStream_t stream);
// Statistics of plane samples [s_begin, s_end) of a kLayoutNibbleStats workspace.
hipError_t launch_sample_stats(const uint64_t *d_bit_sets, uint32_t words_per_sample,
                               const PlaneGeometry &geo, uint4 *d_planes, uint32_t s_begin,
                               uint32_t s_end, hipStream_t stream);
// Test hook, process-wide: bitsets of fewer k-steps get no check points (default 64).
void set_filter_check_min_steps(uint32_t steps);
// The sample order of plane samples [s_begin, s_end) of a kLayoutNibbleStats workspace
// (king_sort.hip): statistics in stored order -> keys -> stable sort (or none) -> perm and
// the statistics in plane order.  `sort_temp` / `sort_temp_bytes`: device scratch of at
// least sort_temp_bytes_for(s_end - s_begin).
size_t sort_temp_bytes_for(uint32_t n);
hipError_t launch_sample_order(const PlaneGeometry &geo, uint32_t words_per_sample,
                               uint4 *d_planes, uint32_t s_begin, uint32_t s_end, bool sort,
                               void *sort_temp, size_t sort_temp_bytes, hipStream_t stream);
// Codes (+ het-only copy) and / or T2 of plane tiles [s_tile_begin, s_tile_end) of a
// nibble workspace; `gate` != nullptr: only if a quadrant went dense or a tile left
// (filter control words) and the codes are not there yet (*ready == 0).
hipError_t launch_prepare_nibbles(bool codes, bool t2, const uint64_t *d_bit_sets,
                                  uint32_t words_per_sample, const PlaneGeometry &geo,
                                  uint4 *d_planes, const uint32_t *perm, uint32_t s_tile_begin,
                                  uint32_t s_tile_end, const uint32_t *gate,
                                  const uint32_t *ready, hipStream_t stream);
// *ready = 1 if the gate is open (behind a gated launch_prepare_nibbles).
hipError_t launch_mark_codes_ready(const uint32_t *gate, uint32_t *ready, hipStream_t stream);
// Bytes of split scratch (counters, then slabs) for `wgs` workgroups, and of
// the counter part alone (the only part that must start out zero).
size_t mfma_split_scratch_bytes(uint32_t wgs);
size_t mfma_split_counter_bytes(uint32_t wgs);
#ifdef CUKING_MFMA_TIMELINE
void mfma_timeline_dump();  // diagnostic build (king_mfma.hip)
#endif

// Converts plane-sample tiles [s_tile_begin, s_tile_end) (units of 64 plane
// samples) of the block.
// Test hook: cap the workgroups per launch (0 = hardware limit only).
void set_max_blocks_per_launch(uint64_t blocks);

hipError_t launch_prepare_planes(uint32_t layout, const uint64_t *d_bit_sets,
                                 uint32_t words_per_sample,
                                 const PlaneGeometry &geo, uint4 *d_planes,
                                 uint32_t s_tile_begin, uint32_t s_tile_end,
                                 hipStream_t stream);

hipError_t launch_stream(const cuking_submatrix &sm, uint32_t words_per_sample,
                         const uint64_t *d_bit_sets, float kin_threshold,
                         uint32_t max_results, cuking_result *d_results,
                         uint32_t *d_result_index, uint32_t *d_result_overflow,
                         cuking_counts *d_dense_counts, hipStream_t stream);

hipError_t launch_pack(const cuking_submatrix &sm, uint32_t words_per_sample,
                       uint64_t *d_bit_set, const int64_t *d_row_idx,
                       const int64_t *d_col_idx, const int32_t *d_n_alt,
                       size_t num_triples, uint32_t *d_status,
                       hipStream_t stream);

hipError_t launch_pack_compact(uint32_t words_per_sample, uint32_t num_samples,
                               uint64_t *d_bit_set, const uint32_t *d_site,
                               const uint32_t *d_sample_alt, size_t num_triples,
                               uint32_t *d_status, hipStream_t stream);

hipError_t launch_synth(uint64_t seed, const uint32_t *d_kind,
                        const uint32_t *d_pa, const uint32_t *d_pb,
                        uint32_t sample_begin, uint32_t sample_end,
                        uint32_t num_sites, uint32_t words_per_sample,
                        uint64_t *d_bit_set, hipStream_t stream);

hipError_t launch_clock_probe(uint64_t microseconds, uint64_t *d_out,
                              hipStream_t stream);

}  // namespace cuking

#endif  // CUKING_AMD_KING_COMMON_H_
