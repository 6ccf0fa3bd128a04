// C ABI of the KING hot path (include/cuking_amd.h): context, device memory,
// layout preparation + kernel launch, timing hooks.  gfx950 only; no CPU
// fallback -- every device entry point fails with CUKING_ERR_DEVICE when HIP
// cannot provide a gfx950 device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <tuple>
#include <vector>

#include "king_common.h"
#include "king_host.h"

using namespace cuking;

namespace {

#define HIP_TRY(expr)                                                      \
  do {                                                                     \
    const hipError_t _e = (expr);                                          \
    if (_e != hipSuccess) {                                                \
      (void)hipGetLastError(); /* do not leave it sticky for other users */ \
      return cuking_fail(_e == hipErrorOutOfMemory ? CUKING_ERR_OUT_OF_MEMORY     \
                                            : CUKING_ERR_DEVICE,           \
                  "%s failed: %s", #expr, hipGetErrorString(_e));          \
    }                                                                      \
  } while (0)

uint32_t ceil_div(uint32_t a, uint32_t b) {
  return (uint32_t)(((uint64_t)a + b - 1) / b);
}
uint32_t round_up(uint32_t a, uint32_t b) { return ceil_div(a, b) * b; }

struct EventPair {
  hipEvent_t start = nullptr, stop = nullptr;
};

struct Timer {
  std::vector<EventPair> pool;
  size_t used = 0;

  hipError_t begin(hipStream_t s, EventPair **out) {
    if (used == pool.size()) {
      EventPair p;
      hipError_t e = hipEventCreate(&p.start);
      if (e != hipSuccess) return e;
      e = hipEventCreate(&p.stop);
      if (e != hipSuccess) return e;
      pool.push_back(p);
    }
    *out = &pool[used++];
    return hipEventRecord((*out)->start, s);
  }
  hipError_t collect(double *ms, uint64_t *n) {
    *ms = 0;
    *n = used;
    for (size_t k = 0; k < used; ++k) {
      hipError_t e = hipEventSynchronize(pool[k].stop);
      if (e != hipSuccess) return e;
      float t = 0;
      e = hipEventElapsedTime(&t, pool[k].start, pool[k].stop);
      if (e != hipSuccess) return e;
      *ms += t;
    }
    return hipSuccess;
  }
  void destroy() {
    for (auto &p : pool) {
      if (p.start) (void)hipEventDestroy(p.start);
      if (p.stop) (void)hipEventDestroy(p.stop);
    }
    pool.clear();
    used = 0;
  }
};

}  // namespace

struct cuking_ctx {
  int device = 0;
  cuking_kernel kernel = CUKING_KERNEL_TILED;
  int variant = 0;
  // Tile rows per scheduling band.  NOT a multiple of 8: workgroups are dealt
  // round-robin over the 8 XCDs, so with 16 rows per band every XCD would see
  // the same rows in every column -- and rows differ in how many of their
  // tiles are real (triangle, strided rectangles): measured 6.7 ms vs 4.3 ms
  // for a rank's strided launch (tools/rect_probe.py).
  // 0 = by block size (band_rows_for below); tests and tuning runs pin a value.
  uint32_t band_rows = 0;
  int counts_mode = -1;  // -1 auto, 0 lean (4 sums + recount), 1 full (5 sums)
  int xcd_swizzle = 2;   // matrix-core kernel: XCD-aware order, 0 off / 1 chunks / 2 patches (king_common.h)
  uint32_t dyn_tail_tiles = 16384;  // launches of at least this many tiles get a dynamic tail; 0 = never

  // Workspace of the tiled kernel: the k-major planes and the band prefix.
  uint4 *planes = nullptr;
  size_t planes_bytes = 0;
  uint64_t *band_prefix = nullptr;
  size_t band_prefix_entries = 0;
  TileSpace prefix_for = {0, 0, 0, 0};  // tile space the prefix was built for

  // Remainder splitting of the matrix-core kernel (king_mfma.hip): one zeroed
  // scratch slab per stream that launches it (launches on different streams
  // may overlap), split_wgs workgroups = one per CU.  0 = never split.
  uint32_t split_wgs = 0;
  std::vector<std::pair<hipStream_t, uint32_t *>> split_scratch;
  // Filter variant (king_filter.hip): control words, candidate list and dense-
  // quadrant list, one set per stream like the split scratch.  The two caps are
  // options so that tests can force the dense and the list-full paths.
  struct FilterScratch {
    hipStream_t stream;
    uint8_t *base;
    uint64_t tiles;  // the block size (tiles of its enumeration) the lists are sized for
  };
  std::vector<FilterScratch> filter_scratch;
  // The running totals of scratch blocks that have been freed since (a larger block took
  // their place, their stream made room): the diagnostics count over the context's life.
  unsigned long long filter_totals_retired[kNumTotals] = {};
  uint32_t filter_quadrant_cap = kFilterQuadrantCap;
  uint32_t filter_cand_cap = kFilterCandCap;
  uint32_t filter_split_min_steps = 8;  // k-steps per remainder piece, at least
  // Check points of the filter kernel (king_common.h): check 0 (forecast) 0 off, 1 for
  // launches of fewer than 16 rounds, 2 always; check 1 (rigorous) 0 off, 1 the entry the
  // kernel picks from threshold and cohort, 2 + k entry k of the share menu forced.
  int filter_check0 = 1, filter_check1 = 1, filter_check_emit = (int)kCheckEmitCap;
  // Rotated tiles (king_common.h TiledArgs::rotate): 0 off, 1 on, 2 / 3 + j test hooks;
  // bitsets of fewer k-steps (of 256 sites) than the minimum are not rotated.
  int filter_rotate = 1;
  uint32_t filter_rotate_min_steps = 128;
  uint32_t filter_rotate_min_tiles = 2048;  // 8 rounds of one tile per CU
  // one resident workgroup per CU takes tile after tile (king_filter.hip): measured 1.2 % behind
  // one workgroup per tile at configs[2] -- the chip is power-bound, a CU that waits for the
  // dispatcher lends its share to the others -- so off unless asked for
  bool filter_persistent = false;
  uint32_t filter_persistent_min_tiles = 2048;
  // The kernel layout's samples sorted by their share of missing calls (king_sort.hip);
  // the four-product kernel's codes converted only when the filter needs them
  // (0: with every conversion).
  // filter_sort: 0 never; 1 (default) whole-block conversions -- cuking_compute_king and
  // cuking_compute_king_tiles, whose tiles are opaque units of work; 2 also the ranges of
  // cuking_prepare_samples (for hosts whose rectangles cover every prepared range in full
  // or in a union, like the staged multi-GPU schedules: inside a sorted range the samples
  // behind a sub-rectangle are not the ones its bounds name).
  int filter_sort = 1;
  bool filter_lazy_codes = true;
  void *sort_temp = nullptr;
  size_t sort_temp_bytes = 0;

  // What the plane workspace holds: the block it was converted for and which
  // 64-sample plane tiles of it have been converted (cuking_compute_king_rect
  // refuses to read anything else).
  struct Prepared {
    bool valid = false;
    cuking_submatrix sm = {0, 0, 0, 0};
    uint32_t words_per_sample = 0;
    int variant = -1;
    uint32_t tile = 0;  // tile edge of the geometry (the context variant's)
    const uint64_t *bits = nullptr;
    bool codes = true;  // every converted tile has its nibble codes (false: T2 only, lazy)
    // per 64 plane samples: 0 = not converted, 1 = converted (kernels that
    // read it, if any, all precede the tail of `ordered_on`), 2 = converted
    // and read by kernels enqueued since
    std::vector<uint8_t> tiles;
    hipStream_t ordered_on = nullptr;
    bool ordered_valid = false;
  } prepared;
  // Streams with pair kernels that may still be reading the workspace, each
  // with an event to order a later rewrite after them.
  std::vector<std::pair<hipStream_t, hipEvent_t>> readers;

  // Hosts that promise not to rewrite a bitset in place without telling
  // (cuking_invalidate) may skip the conversion when the workspace already
  // holds this block ("reuse_prepared"); off by default: like the reference's
  // kernel, a call then reads whatever the bitset holds when it runs.
  bool reuse_prepared = false;
  // Book-keeping for hosts that must not allocate or synchronise while
  // collectives of other devices are in flight (cuking_ctx_reserve): device
  // allocations made for the workspace, and host-side waits, so far.
  uint64_t workspace_allocations = 0, host_syncs = 0, conversions_skipped = 0;

  bool timing = false;
  Timer king_timer, prepare_timer;
};

namespace {

int default_variant() {
  if (const char *v = getenv("CUKING_AMD_VARIANT")) {
    const int k = atoi(v);
    if (k >= 0 && k < kNumTiledVariants) return k;
  }
  return kMfmaFilterVariant;
}

// The variant that runs for a bitset of this width: the matrix-core variant
// counts in float32 (exact below 2^24 sites); wider bitsets take the VALU
// variant with the same tile edge and k padding, so tile indices, tile bounds
// and prepared ranges mean the same either way.
int effective_variant(const cuking_ctx *ctx, uint32_t words_per_sample) {
  const uint64_t sites = (uint64_t)round_up(words_per_sample, 8) * 32;
  int v = ctx->variant;
  // (the four-product variant decides on an integer that matches the
  //  reference's float expression below 2^22 sites only)
  if ((v == kMfmaN4Variant || v == kMfmaFilterVariant) && sites > kMfmaN4MaxSites) v = kMfmaVariant;
  if (v == kMfmaVariant && sites > kMfmaMaxSites) v = 2;
  return v;
}

// The shape everything about a call is laid out for: the kernel variant that
// runs (layout, k padding) with the tile edge of the CONTEXT's variant, so that
// tile indices, tile bounds and prepared ranges never depend on the width of
// the bitset.  When the two differ (filter variant, 256-sample tiles, falling
// back to a 128-tile kernel for a wide bitset) the kernel runs in quadrant mode.
TiledVariant plan_variant(const cuking_ctx *ctx, uint32_t words_per_sample) {
  TiledVariant v = tiled_variant(effective_variant(ctx, words_per_sample));
  v.tile = tiled_variant(ctx->variant).tile;
  return v;
}

// Which form of the tiled kernel.  The lean form keeps four sums per pair but
// recounts hom/hom sites for every EMITTED pair; the full form keeps five sums
// for every pair.  Measured at 10k x 100k sites: VALU kernels lean 27.8 ms +
// 3 ms per 10^6 emitted pairs against full 34.4 ms flat; matrix-core kernel
// lean 7.0 ms + 10 ms per 10^6 emitted pairs against full 8.8 ms + 0.9 ms per
// 10^6 (the fifth sum in a pass of its own, king_mfma.hip).  For unrelated
// samples kinship scatters around 0 with a spread ~ 1/sqrt(sites) (at 100k
// sites 2 % of the pairs exceed 0.005, 0.3 % exceed 0.007), so the automatic
// choice is lean iff kin_threshold > c / sqrt(sites) with c = 1.6 (VALU) or
// 2.05 (matrix cores: break-even near 0.4 % of the pairs emitted).  Either
// form gives the same records.
bool use_full_counts(const cuking_ctx *ctx, float kin_threshold, bool dense,
                     uint32_t words_per_sample) {
  if (dense || ctx->counts_mode == 1) return true;
  if (ctx->counts_mode == 0) return false;
  if (!(kin_threshold > 0.0f)) return true;
  const double sites = 32.0 * words_per_sample;
  const double c = is_mfma_variant(effective_variant(ctx, words_per_sample)) ? 2.05 : 1.6;
  return (double)kin_threshold * kin_threshold * sites < c * c;
}

cuking_status bind(cuking_ctx *ctx) {
  if (ctx == nullptr) return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null context");
  HIP_TRY(hipSetDevice(ctx->device));
  return CUKING_OK;
}

PlaneGeometry make_geometry(const cuking_submatrix &sm,
                            uint32_t words_per_sample,
                            const TiledVariant &v) {
  PlaneGeometry g;
  g.num_rows = sm_num_rows(sm);
  g.num_cols = sm_num_cols(sm);
  g.diag = sm_is_diag(sm) ? 1u : 0u;
  g.rows_padded = round_up(g.num_rows, v.tile);
  g.cols_padded = round_up(g.num_cols, v.tile);
  g.col_base = g.diag ? 0u : g.rows_padded;
  g.s_stride = g.diag ? g.rows_padded : g.rows_padded + g.cols_padded;
  g.k_words = round_up(words_per_sample, v.k_chunk);  // 2 x 32-bit per u64 / 2 planes
  return g;
}

// Band height when the caller has not pinned one.  Measured on MI355X with the
// XCD-aware order (archive/profiles/r02_xcd_order.txt): the matrix-core kernel wants
// the 32 tiles an XCD holds at a time to be a compact patch, 5 rows x ~6
// columns (11 strips through one L2 instead of 32): 100k x 100k 645 -> 621 ms,
// 300k x 150k 9.33 -> 8.69 s.  Below ~128 tile rows the bitset sits in the
// Infinity Cache anyway and 17 rows measured 2-3 % better (10k samples:
// 7.04 -> 6.77 ms with the XCD order, 6.96 with 5 rows).  The VALU kernels
// keep 17 (see cuking_ctx::band_rows in the round-1 notes, tools/rect_probe.py).
uint32_t band_rows_for(uint32_t pinned, const PlaneGeometry &g, const TiledVariant &v) {
  if (pinned != 0) return pinned;
  if (v.layout == kLayoutWord) return 17;
  // 256-sample tiles (filter variant): 8 rows x 4 columns per XCD patch at every size
  // (configs[2]: 146.8 ms against 147.4 with 5 rows and 147.2 with 17; configs[1]: 1.78
  // against 1.81 with 17; profiles/r03_ablation.txt)
  if (v.tile == kFilterTile) return 8;
  return g.rows_padded / v.tile >= 128 ? 5 : 17;
}

TileSpace make_tiles(const PlaneGeometry &g, const TiledVariant &v,
                     uint32_t band_rows) {
  band_rows = band_rows_for(band_rows, g, v);
  TileSpace t;
  t.tiles_r = g.rows_padded / v.tile;
  t.tiles_c = g.cols_padded / v.tile;
  t.band_rows = band_rows;
  t.diag = g.diag;
  return t;
}

uint64_t total_tiles(const TileSpace &t) {
  uint64_t n = 0;
  for (uint32_t b = 0; b < t.num_bands(); ++b) n += t.band_tiles(b);
  return n;
}

// Split scratch of `stream` (allocated and zeroed on first use), or nullptr
// when splitting is off / does not apply to the variant.
cuking_status split_scratch_for(cuking_ctx *ctx, hipStream_t stream,
                                uint32_t **scratch, uint32_t **counters) {
  *scratch = *counters = nullptr;
  if (ctx->split_wgs == 0 || !is_mfma_variant(ctx->variant)) return CUKING_OK;
  const size_t bytes = mfma_split_scratch_bytes(ctx->split_wgs);
  uint32_t *base = nullptr;
  for (auto &e : ctx->split_scratch)
    if (e.first == stream) base = e.second;
  if (base == nullptr) {
    if (ctx->split_scratch.size() >= 8) {
      // Streams come and go (torch hands out new handles): keep the cache
      // small.  Nothing may still be using a slab we free.
      ++ctx->host_syncs;
      HIP_TRY(hipDeviceSynchronize());
      for (auto &e : ctx->split_scratch) (void)hipFree(e.second);
      ctx->split_scratch.clear();
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&base), bytes));
    ++ctx->workspace_allocations;
    // The tickets must read zero when the first launch on this stream starts:
    // zero them ON that stream (a null-stream memset is not ordered against a
    // non-blocking stream, and fresh memory holds whatever was there before).
    hipError_t e = hipMemsetAsync(base, 0, mfma_split_counter_bytes(ctx->split_wgs), stream);
    if (e != hipSuccess) {
      (void)hipFree(base);
      HIP_TRY(e);
    }
    ctx->split_scratch.emplace_back(stream, base);
  }
  *counters = base;
  *scratch = base + mfma_split_counter_bytes(ctx->split_wgs) / sizeof(uint32_t);
  return CUKING_OK;
}

// Filter scratch of `stream` for a block of `tiles` tiles (allocated on first use, again
// when a larger block comes; the control words are zeroed by every launch chunk), or
// nothing when the context's variant is not the filter variant.  Fills the filter
// fields of `a`.
// Before a scratch block is freed (nothing runs on its stream any more): its running totals
// join the context's.
static void retire_filter_totals(cuking_ctx *ctx, const cuking_ctx::FilterScratch &e) {
  if (e.base == nullptr) return;
  unsigned long long v[kNumTotals] = {};
  if (hipMemcpy(v, e.base + filter_scratch_layout(e.tiles).totals, sizeof v,
                hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  for (uint32_t k = 0; k < kNumTotals; ++k) ctx->filter_totals_retired[k] += v[k];
}

cuking_status filter_scratch_for(cuking_ctx *ctx, hipStream_t stream, const PlaneGeometry &geo,
                                 uint64_t tiles, TiledArgs *a) {
  a->sample_stats = nullptr;
  a->t2 = nullptr;
  a->filter_ctrl = nullptr;
  a->cand_list = nullptr;
  a->dense_list = nullptr;
  a->cand_cap = a->dense_cap = a->quadrant_cap = 0;
  a->fsplit_parts = a->fsplit_first = a->fsplit_tile0 = 0;
  a->fsplit_slabs = nullptr;
  a->fsplit_tickets = nullptr;
  a->filter_totals = nullptr;
  a->tile_done = nullptr;
  a->prefix_u = nullptr;
  a->cohort_sums = nullptr;
  a->check_steps = nullptr;
  a->check0 = a->check1 = 0;
  a->rotate = a->rotate_min_steps = a->rotate_min_tiles = 0;
  a->persist_wgs = a->persist_min_tiles = 0;
  if (ctx->variant != kMfmaFilterVariant) return CUKING_OK;
  const FilterScratchLayout want = filter_scratch_layout(tiles);
  cuking_ctx::FilterScratch *entry = nullptr;
  for (auto &e : ctx->filter_scratch)
    if (e.stream == stream) entry = &e;
  if (entry != nullptr && filter_scratch_layout(entry->tiles).bytes < want.bytes) {
    // a larger block: kernels of this stream may still use the old lists
    ++ctx->host_syncs;
    HIP_TRY(hipStreamSynchronize(stream));
    retire_filter_totals(ctx, *entry);
    (void)hipFree(entry->base);
    entry->base = nullptr;
  }
  if (entry == nullptr || entry->base == nullptr) {
    if (entry == nullptr && ctx->filter_scratch.size() >= 8) {
      // Streams come and go (torch hands out new handles): the oldest entry makes room.
      ++ctx->host_syncs;
      HIP_TRY(hipStreamSynchronize(ctx->filter_scratch.front().stream));
      (void)hipGetLastError();  // (a stream its owner has destroyed meanwhile: nothing runs on it)
      retire_filter_totals(ctx, ctx->filter_scratch.front());
      (void)hipFree(ctx->filter_scratch.front().base);
      ctx->filter_scratch.erase(ctx->filter_scratch.begin());
    }
    uint8_t *base = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&base), want.bytes));
    ++ctx->workspace_allocations;
    // (the running totals behind "filter_candidates" / "filter_dense_quadrants")
    // ... and the tickets of the remainder pieces, zero between launches
    hipError_t e = hipMemsetAsync(base + want.totals, 0, kFilterCtrlBytes + kFilterTicketBytes, stream);
    if (e != hipSuccess) {
      (void)hipFree(base);
      HIP_TRY(e);
    }
    if (entry == nullptr) {
      ctx->filter_scratch.push_back({stream, base, tiles});
      entry = &ctx->filter_scratch.back();
    } else {
      entry->base = base;
      entry->tiles = tiles;
    }
  }
  uint8_t *base = entry->base;
  const FilterScratchLayout l = filter_scratch_layout(entry->tiles);
  a->sample_stats = plane_stats(ctx->planes, geo);
  a->t2 = plane_t2(ctx->planes, geo);
  a->prefix_u = plane_prefix_u(ctx->planes, geo);
  a->cohort_sums = plane_cohort_sums(ctx->planes, geo);
  a->filter_ctrl = reinterpret_cast<uint32_t *>(base);
  a->filter_totals = reinterpret_cast<unsigned long long *>(base + l.totals);
  a->fsplit_tickets = reinterpret_cast<uint32_t *>(base + l.tickets);
  a->tile_done = base + l.tile_done;
  a->cand_list = reinterpret_cast<uint2 *>(base + l.cand);
  a->cand_cap = ctx->filter_cand_cap < l.cand_entries ? ctx->filter_cand_cap : l.cand_entries;
  a->quadrant_cap = ctx->filter_quadrant_cap;
  a->dense_list = reinterpret_cast<uint2 *>(base + l.dense);
  a->dense_cap = l.chunk_tiles * 4;
  // (remainder splitting follows the matrix-core kernels' switch: "split_wgs" 0 = never)
  a->fsplit_first = ctx->filter_split_min_steps;  // (on entry: launch_filter)
  a->fsplit_slabs = ctx->split_wgs != 0 ? reinterpret_cast<float4 *>(base + l.slabs) : nullptr;
  a->check0 = (uint32_t)ctx->filter_check0;  // (switches on entry: launch_filter)
  a->check1 = (uint32_t)ctx->filter_check1 | ((uint32_t)ctx->filter_check_emit << 8) |
              (ctx->filter_check0 != 0 ? 1u << 16 : 0u);
  a->rotate = (uint32_t)ctx->filter_rotate;
  a->rotate_min_steps = ctx->filter_rotate_min_steps;
  a->rotate_min_tiles = ctx->filter_rotate_min_tiles;
  a->persist_wgs = ctx->filter_persistent ? 1u : 0u;
  a->persist_min_tiles = ctx->filter_persistent_min_tiles;
  a->check_steps = plane_check_steps(ctx->planes, geo);
  return CUKING_OK;
}

// The sample order of the workspace's layout, and the lazy-codes word, for the launch.
void layout_order_for(const cuking_ctx *ctx, uint32_t words_per_sample, const PlaneGeometry &geo,
                      TiledArgs *a) {
  a->perm = nullptr;
  a->codes_ready = nullptr;
  if (plan_variant(ctx, words_per_sample).layout != kLayoutNibbleStats) return;
  a->perm = plane_perm(ctx->planes, geo);
  if (!ctx->prepared.codes) a->codes_ready = plane_flags(ctx->planes, geo);
}

// Enqueues `num_tiles` tiles of the planned geometry: as they are when the
// kernel's tile edge is the geometry's, as four quadrants each otherwise.
hipError_t launch_planned(const cuking_ctx *ctx, uint32_t words_per_sample, bool full,
                          TiledArgs a, uint64_t num_tiles, hipStream_t stream) {
  const int kv = effective_variant(ctx, words_per_sample);
  a.quad = 0;
  a.tile_list = nullptr;
  a.tile_list_count = nullptr;
  a.tile_list_cap = 0;
  if (tiled_variant(kv).tile != tiled_variant(ctx->variant).tile) {
    // (only the filter variant's 256-sample geometry over a 128-tile kernel)
    if (tiled_variant(kv).tile * 2 != tiled_variant(ctx->variant).tile) return hipErrorInvalidValue;
    a.quad = 1;
    a.tile_begin *= 4;
    num_tiles *= 4;
  }
  return launch_tiled(kv, full, a, num_tiles, stream);
}

// A pair kernel that reads the workspace has been enqueued on `stream`.
void note_reader(cuking_ctx *ctx, hipStream_t stream) {
  for (auto &r : ctx->readers)
    if (r.first == stream) return;
  ctx->readers.emplace_back(stream, nullptr);
}

// Orders `stream` behind every pair kernel enqueued so far on OTHER streams
// (same-stream work is ordered anyway): the caller is about to rewrite plane
// data they may still read.  One context serves one host thread at a time, so
// "enqueued so far" is everything there is.  Afterwards the tail of `stream`
// stands for all of them (it stays in the list as their proxy), and no plane
// tile counts as "being read" any more.
cuking_status wait_for_readers(cuking_ctx *ctx, hipStream_t stream) {
  bool failed = false;
  for (auto &r : ctx->readers) {
    if (r.first == stream) continue;
    if (r.second == nullptr &&
        hipEventCreateWithFlags(&r.second, hipEventDisableTiming) != hipSuccess) {
      failed = true;
      break;
    }
    if (hipEventRecord(r.second, r.first) != hipSuccess ||
        hipStreamWaitEvent(stream, r.second, 0) != hipSuccess) {
      failed = true;  // e.g. a stream the caller has destroyed meanwhile
      break;
    }
  }
  if (failed) {
    (void)hipGetLastError();
    ++ctx->host_syncs;
    HIP_TRY(hipDeviceSynchronize());
  }
  for (auto &r : ctx->readers)
    if (r.second) (void)hipEventDestroy(r.second);
  ctx->readers.clear();
  ctx->readers.emplace_back(stream, nullptr);
  ctx->prepared.ordered_on = stream;
  ctx->prepared.ordered_valid = true;
  for (auto &t : ctx->prepared.tiles)
    if (t == 2) t = 1;
  return CUKING_OK;
}

// Marks plane tiles [begin, end) (units of 64 samples) as read by a kernel
// that has just been enqueued.
void mark_read(cuking_ctx *ctx, uint32_t begin, uint32_t end) {
  std::vector<uint8_t> &t = ctx->prepared.tiles;
  if (end > t.size()) end = (uint32_t)t.size();
  for (uint32_t k = begin; k < end; ++k) t[k] = 2;
}

bool same_block(const cuking_ctx::Prepared &p, const cuking_submatrix &sm,
                uint32_t words_per_sample, int variant, uint32_t tile, const uint64_t *bits) {
  return p.valid && p.sm.i_begin == sm.i_begin && p.sm.i_end == sm.i_end &&
         p.sm.j_begin == sm.j_begin && p.sm.j_end == sm.j_end &&
         p.words_per_sample == words_per_sample && p.variant == variant &&
         p.tile == tile && p.bits == bits;
}

// Makes the plane workspace at least `need` bytes and the band prefix at least
// `entries` long.  Replacing either waits for the whole device first: kernels
// of earlier calls (possibly on other streams) may still read the old one.
cuking_status ensure_workspace(cuking_ctx *ctx, size_t need, size_t entries) {
  if (need > ctx->planes_bytes) {
    ++ctx->host_syncs;
    HIP_TRY(hipDeviceSynchronize());
    if (ctx->planes) HIP_TRY(hipFree(ctx->planes));
    ctx->planes = nullptr;
    ctx->planes_bytes = 0;
    ctx->prepared.valid = false;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->planes), need));
    ++ctx->workspace_allocations;
    ctx->planes_bytes = need;
  }
  if (entries > ctx->band_prefix_entries) {
    ++ctx->host_syncs;
    HIP_TRY(hipDeviceSynchronize());
    if (ctx->band_prefix) HIP_TRY(hipFree(ctx->band_prefix));
    ctx->band_prefix = nullptr;
    ctx->band_prefix_entries = 0;
    ctx->prefix_for = TileSpace{0, 0, 0, 0};
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->band_prefix), entries * sizeof(uint64_t)));
    ++ctx->workspace_allocations;
    ctx->band_prefix_entries = entries;
  }
  return CUKING_OK;
}

// Device scratch of the sample sort for ranges of up to `n` samples.
cuking_status ensure_sort_temp(cuking_ctx *ctx, uint32_t n) {
  const size_t need = sort_temp_bytes_for(n);
  if (need <= ctx->sort_temp_bytes) return CUKING_OK;
  ++ctx->host_syncs;
  HIP_TRY(hipDeviceSynchronize());
  if (ctx->sort_temp) HIP_TRY(hipFree(ctx->sort_temp));
  ctx->sort_temp = nullptr;
  ctx->sort_temp_bytes = 0;
  HIP_TRY(hipMalloc(&ctx->sort_temp, need));
  ++ctx->workspace_allocations;
  ctx->sort_temp_bytes = need;
  return CUKING_OK;
}

bool same_tile_space(const TileSpace &a, const TileSpace &b) {
  return a.tiles_r == b.tiles_r && a.tiles_c == b.tiles_c && a.band_rows == b.band_rows &&
         a.diag == b.diag;
}

// Uploads the band prefix of `tiles` (the caller has ordered `stream` behind
// the readers of the old one).
cuking_status upload_prefix(cuking_ctx *ctx, const TileSpace &tiles, hipStream_t stream) {
  const uint32_t nb = tiles.num_bands();
  std::vector<uint64_t> prefix((size_t)nb + 1, 0);
  for (uint32_t b = 0; b < nb; ++b) prefix[b + 1] = prefix[b] + tiles.band_tiles(b);
  // Small and pageable: wait until the host buffer may go away.
  HIP_TRY(hipMemcpyAsync(ctx->band_prefix, prefix.data(), prefix.size() * sizeof(uint64_t),
                         hipMemcpyHostToDevice, stream));
  ++ctx->host_syncs;
  HIP_TRY(hipStreamSynchronize(stream));
  ctx->prefix_for = tiles;
  return CUKING_OK;
}

// Builds planes + band prefix for `sm` in the context workspace.
// The nibble codes (+ het-only copy) of every converted tile of a kLayoutNibbleStats
// workspace whose last conversion left them out (lazy codes): for calls that run the
// four-product kernel directly.
cuking_status convert_codes_now(cuking_ctx *ctx, const PlaneGeometry &geo,
                                uint32_t words_per_sample, const uint64_t *d_bit_sets,
                                hipStream_t stream) {
  cuking_ctx::Prepared &pr = ctx->prepared;
  const uint32_t all = (uint32_t)pr.tiles.size();
  for (uint32_t t = 0; t < all;) {
    if (pr.tiles[t] == 0) {
      ++t;
      continue;
    }
    uint32_t e = t;
    while (e < all && pr.tiles[e] != 0) ++e;
    HIP_TRY(launch_prepare_nibbles(true, false, d_bit_sets, words_per_sample, geo, ctx->planes,
                                   plane_perm(ctx->planes, geo), t, e, nullptr, nullptr, stream));
    t = e;
  }
  HIP_TRY(hipMemsetAsync(plane_flags(ctx->planes, geo), 1, sizeof(uint32_t), stream));
  pr.codes = true;
  return CUKING_OK;
}

cuking_status prepare(cuking_ctx *ctx, const cuking_submatrix &sm,
                      uint32_t words_per_sample, const uint64_t *d_bit_sets,
                      hipStream_t stream, PlaneGeometry *geo_out,
                      TileSpace *tiles_out, bool need_codes, uint32_t s_tile_begin = 0,
                      uint32_t s_tile_end = 0xFFFFFFFFu) {
  const int variant = effective_variant(ctx, words_per_sample);
  const TiledVariant v = plan_variant(ctx, words_per_sample);
  const PlaneGeometry geo = make_geometry(sm, words_per_sample, v);
  const TileSpace tiles = make_tiles(geo, v, ctx->band_rows);
  *geo_out = geo;
  *tiles_out = tiles;

  const size_t need = plane_bytes(geo, v.layout);
  const uint32_t nb = tiles.num_bands();
  cuking_status st = ensure_workspace(ctx, need, (size_t)nb + 1);
  if (st != CUKING_OK) return st;
  if (v.layout == kLayoutNibbleStats && ctx->filter_sort != 0) {
    st = ensure_sort_temp(ctx, geo.s_stride);
    if (st != CUKING_OK) return st;
  }
  // Book-keeping of what the workspace holds, and ordering against kernels on
  // other streams that still read what is about to be overwritten -- the planes
  // AND the band prefix below, so this comes before either is touched.
  const uint32_t all_tiles = (geo.s_stride + 63) / 64;
  const uint32_t t_end = s_tile_end < all_tiles ? s_tile_end : all_tiles;
  cuking_ctx::Prepared &pr = ctx->prepared;
  const bool new_prefix = !same_tile_space(ctx->prefix_for, tiles);
  const bool same = same_block(pr, sm, words_per_sample, variant, v.tile, d_bit_sets);
  if (ctx->reuse_prepared && same && !new_prefix) {
    // The host has promised that the bitset behind this pointer is unchanged
    // since it was converted (cuking_invalidate otherwise): nothing to do when
    // every tile asked for is there.
    bool all_there = true;
    for (uint32_t t = s_tile_begin; t < t_end && all_there; ++t) all_there = pr.tiles[t] != 0;
    if (all_there) {
      ++ctx->conversions_skipped;
      if (v.layout == kLayoutNibbleStats && need_codes && !pr.codes)
        return convert_codes_now(ctx, geo, words_per_sample, d_bit_sets, stream);
      return CUKING_OK;
    }
  }
  // Needs ordering: a conversion for another block while anything may still
  // read the old one; a repeated conversion of tiles that kernels enqueued
  // since have read (2); or of tiles whose readers were last ordered behind a
  // different stream (1).  Fresh tiles of the same block have no readers.
  const bool other_stream = pr.ordered_valid && pr.ordered_on != stream;
  bool must_wait = new_prefix && !ctx->readers.empty();
  if (!same) {
    must_wait = must_wait || !ctx->readers.empty();
    pr.valid = true;
    pr.sm = sm;
    pr.words_per_sample = words_per_sample;
    pr.variant = variant;
    pr.tile = v.tile;
    pr.bits = d_bit_sets;
    pr.tiles.assign(all_tiles, 0);
    pr.codes = true;
  }
  for (uint32_t t = s_tile_begin; t < t_end; ++t)
    must_wait = must_wait || pr.tiles[t] == 2 || (pr.tiles[t] == 1 && other_stream);
  if (must_wait) {
    st = wait_for_readers(ctx, stream);
    if (st != CUKING_OK) return st;
  }
  for (uint32_t t = s_tile_begin; t < t_end; ++t) pr.tiles[t] = 1;
  if (new_prefix) {
    st = upload_prefix(ctx, tiles, stream);
    if (st != CUKING_OK) return st;
  }

  if (need == 0) return CUKING_OK;
  EventPair *ev = nullptr;
  if (ctx->timing) HIP_TRY(ctx->prepare_timer.begin(stream, &ev));
  if (v.layout == kLayoutNibbleStats) {
    // The filter variant's layout, step by step: statistics in stored order, the sample
    // order (king_sort.hip), T2 -- and the four-product kernel's codes now, or by a gated
    // launch behind the filter kernel if that turns out to need them (lazy codes: whole
    // blocks only; the ranges of a staged pass are converted in full).
    const bool whole = s_tile_begin == 0 && t_end == all_tiles;
    const bool codes = need_codes || !ctx->filter_lazy_codes || !whole;
    const uint32_t sb = s_tile_begin * 64, se = t_end * 64;
    if (s_tile_begin == 0)  // a conversion that starts at plane sample 0 starts the cohort's sums afresh
      HIP_TRY(hipMemsetAsync(const_cast<unsigned long long *>(plane_cohort_sums(ctx->planes, geo)),
                             0, 64, stream));
    // (whatever is converted now, the codes of the block as a whole are not "there" unless
    //  this conversion or convert_codes_now() below says so)
    HIP_TRY(hipMemsetAsync(plane_flags(ctx->planes, geo), 0, sizeof(uint32_t), stream));
    HIP_TRY(launch_sample_stats(d_bit_sets, words_per_sample, geo, ctx->planes, sb, se, stream));
    HIP_TRY(launch_sample_order(geo, words_per_sample, ctx->planes, sb, se,
                                ctx->filter_sort == 2 || (ctx->filter_sort == 1 && whole),
                                ctx->sort_temp, ctx->sort_temp_bytes, stream));
    HIP_TRY(launch_prepare_nibbles(codes, true, d_bit_sets, words_per_sample, geo, ctx->planes,
                                   plane_perm(ctx->planes, geo), s_tile_begin, t_end, nullptr,
                                   nullptr, stream));
    if (!codes) {
      pr.codes = false;
    } else if (!pr.codes) {
      // earlier tiles of this block were converted without codes: complete them
      st = convert_codes_now(ctx, geo, words_per_sample, d_bit_sets, stream);
      if (st != CUKING_OK) return st;
    } else {
      HIP_TRY(hipMemsetAsync(plane_flags(ctx->planes, geo), 1, sizeof(uint32_t), stream));
    }
  } else {
    HIP_TRY(launch_prepare_planes(v.layout, d_bit_sets, words_per_sample, geo, ctx->planes,
                                  s_tile_begin, s_tile_end, stream));
  }
  if (ev) HIP_TRY(hipEventRecord(ev->stop, stream));
  return CUKING_OK;
}

cuking_status run_tiled(cuking_ctx *ctx, const cuking_submatrix &sm,
                        uint32_t words_per_sample, const uint64_t *d_bit_sets,
                        uint64_t tile_begin, uint64_t tile_end, bool whole,
                        float kin_threshold, uint32_t max_results,
                        cuking_result *d_results, uint32_t *d_result_index,
                        uint32_t *d_result_overflow, cuking_counts *d_counts,
                        hipStream_t stream) {
  PlaneGeometry geo;
  TileSpace tiles;
  const bool full = use_full_counts(ctx, kin_threshold, d_counts != nullptr, words_per_sample);
  // (the filter's bound applies: the four-product kernel's codes may stay unconverted)
  const bool filter_runs = effective_variant(ctx, words_per_sample) == kMfmaFilterVariant &&
                           !full && kin_threshold > 0.f && kin_threshold < 0.5f;
  cuking_status st =
      prepare(ctx, sm, words_per_sample, d_bit_sets, stream, &geo, &tiles, !filter_runs);
  if (st != CUKING_OK) return st;
  const uint64_t n_tiles = total_tiles(tiles);
  if (whole) {
    tile_begin = 0;
    tile_end = n_tiles;
  }
  if (tile_begin > tile_end || tile_end > n_tiles)
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT,
                "tile range [%llu, %llu) outside [0, %llu)",
                (unsigned long long)tile_begin, (unsigned long long)tile_end,
                (unsigned long long)n_tiles);
  if (tile_begin == tile_end) return CUKING_OK;

  TiledArgs a = {};
  a.planes = ctx->planes;
  a.geo = geo;
  a.tiles = tiles;
  a.band_prefix = ctx->band_prefix;
  a.tile_begin = tile_begin;
  a.i_begin = sm.i_begin;
  a.j_begin = sm.j_begin;
  a.kin_threshold = kin_threshold;
  a.max_results = max_results;
  a.results = d_results;
  a.result_index = d_result_index;
  a.result_overflow = d_result_overflow;
  a.dense_counts = d_counts;
  a.rect_rows = a.rect_cols = a.rect_row0 = a.rect_col0 = 0;
  a.rect_row_stride = 1;
  a.bits = d_bit_sets;
  a.words_per_sample = words_per_sample;
  a.split_tiles = 0;
  a.split_whole = 0;
  a.split_wgs = ctx->split_wgs;
  a.xcd_chunk = (uint32_t)ctx->xcd_swizzle;  // (switch: 1 chunks, 2 patches; the launch sets the value)
  a.launch_tiles = 0;
  a.dyn_tiles = ctx->dyn_tail_tiles;  // (threshold; the launch sets the count)
  a.dyn_wgs = 0;
  st = split_scratch_for(ctx, stream, &a.split_scratch, &a.split_counters);
  if (st != CUKING_OK) return st;
  st = filter_scratch_for(ctx, stream, geo, n_tiles, &a);
  if (st != CUKING_OK) return st;
  layout_order_for(ctx, words_per_sample, geo, &a);

  EventPair *ev = nullptr;
  if (ctx->timing) HIP_TRY(ctx->king_timer.begin(stream, &ev));
  HIP_TRY(launch_planned(ctx, words_per_sample, full, a, tile_end - tile_begin, stream));
  note_reader(ctx, stream);
  mark_read(ctx, 0, 0xFFFFFFFFu);
  if (ev) HIP_TRY(hipEventRecord(ev->stop, stream));
  return CUKING_OK;
}

cuking_status run_stream(cuking_ctx *ctx, const cuking_submatrix &sm,
                         uint32_t words_per_sample, const uint64_t *d_bit_sets,
                         float kin_threshold, uint32_t max_results,
                         cuking_result *d_results, uint32_t *d_result_index,
                         uint32_t *d_result_overflow, cuking_counts *d_counts,
                         hipStream_t stream) {
  EventPair *ev = nullptr;
  if (ctx->timing) HIP_TRY(ctx->king_timer.begin(stream, &ev));
  HIP_TRY(launch_stream(sm, words_per_sample, d_bit_sets, kin_threshold,
                        max_results, d_results, d_result_index,
                        d_result_overflow, d_counts, stream));
  if (ev) HIP_TRY(hipEventRecord(ev->stop, stream));
  return CUKING_OK;
}

}  // namespace

extern "C" {

// ---- context and memory ---------------------------------------------------

int cuking_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

cuking_status cuking_ctx_create(int device, cuking_ctx **out) {
  if (out == nullptr) return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null out pointer");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0)
    return cuking_fail(CUKING_ERR_DEVICE,
                "no HIP device available (%s); this library has no CPU path",
                e == hipSuccess ? "0 devices" : hipGetErrorString(e));
  if (device < 0 || device >= n)
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "device %d outside [0, %d)", device, n);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return cuking_fail(CUKING_ERR_DEVICE,
                "device %d is %s; this library only carries gfx950 code",
                device, prop.gcnArchName);
  HIP_TRY(hipSetDevice(device));
  cuking_ctx *ctx = new cuking_ctx();
  ctx->device = device;
  ctx->variant = default_variant();
  ctx->split_wgs = (uint32_t)prop.multiProcessorCount;
  if (const char *v = getenv("CUKING_AMD_SPLIT_WGS")) {
    const int k = atoi(v);
    if (k >= 0 && k <= 4096) ctx->split_wgs = (uint32_t)k;
  }
  if (const char *v = getenv("CUKING_AMD_XCD_SWIZZLE")) {
    const int k = atoi(v);
    if (k >= 0 && k <= 2) ctx->xcd_swizzle = k;
  }
  if (const char *v = getenv("CUKING_AMD_DYN_TAIL_TILES")) {
    const long long k = atoll(v);
    if (k >= 0 && k <= 0x7FFFFFFF) ctx->dyn_tail_tiles = (uint32_t)k;
  }
  if (const char *v = getenv("CUKING_AMD_BAND_ROWS")) {
    const int k = atoi(v);
    if (k >= 1 && k <= 64) ctx->band_rows = (uint32_t)k;
  }
  *out = ctx;
  return CUKING_OK;
}

void cuking_ctx_destroy(cuking_ctx *ctx) {
  if (ctx == nullptr) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->planes) (void)hipFree(ctx->planes);
  if (ctx->band_prefix) (void)hipFree(ctx->band_prefix);
  if (ctx->sort_temp) (void)hipFree(ctx->sort_temp);
  for (auto &e : ctx->split_scratch) (void)hipFree(e.second);
  for (auto &e : ctx->filter_scratch) (void)hipFree(e.base);
  for (auto &r : ctx->readers)
    if (r.second) (void)hipEventDestroy(r.second);
  ctx->king_timer.destroy();
  ctx->prepare_timer.destroy();
  delete ctx;
}

cuking_status cuking_ctx_set_kernel(cuking_ctx *ctx, cuking_kernel kernel) {
  if (ctx == nullptr) return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null context");
  if (kernel != CUKING_KERNEL_TILED && kernel != CUKING_KERNEL_STREAM)
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "unknown kernel %d", (int)kernel);
  ctx->kernel = kernel;
  return CUKING_OK;
}

cuking_status cuking_ctx_set_option(cuking_ctx *ctx, const char *key,
                                    int64_t value) {
  if (ctx == nullptr || key == nullptr)
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null argument");
  if (strcmp(key, "variant") == 0) {
    if (value < 0 || value >= kNumTiledVariants)
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "variant outside [0, %d)",
                  kNumTiledVariants);
    ctx->variant = (int)value;
    return CUKING_OK;
  }
  if (strcmp(key, "band_rows") == 0) {
    if (value < 0 || value > 64)
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "band_rows outside [0, 64]");
    ctx->band_rows = (uint32_t)value;
    return CUKING_OK;
  }
  if (strcmp(key, "split_wgs") == 0) {  // 0 = never split the remainder
    if (value < 0 || value > 4096)
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "split_wgs outside [0, 4096]");
    if ((uint32_t)value != ctx->split_wgs) {
      // slabs are sized by the workgroup count
      HIP_TRY(hipDeviceSynchronize());
      for (auto &e : ctx->split_scratch) (void)hipFree(e.second);
      ctx->split_scratch.clear();
    }
    ctx->split_wgs = (uint32_t)value;
    return CUKING_OK;
  }
  if (strcmp(key, "max_launch_blocks") == 0) {  // test hook, process-wide
    if (value < 0) return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "negative block cap");
    set_max_blocks_per_launch((uint64_t)value);
    return CUKING_OK;
  }
  if (strcmp(key, "dyn_tail_tiles") == 0) {  // 0 = never; tests lower it
    if (value < 0 || value > 0x7FFFFFFF)
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "dyn_tail_tiles outside [0, 2^31)");
    ctx->dyn_tail_tiles = (uint32_t)value;
    return CUKING_OK;
  }
  if (strcmp(key, "xcd_swizzle") == 0) {
    if (value < 0 || value > 2)
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "xcd_swizzle outside [0, 2]");
    ctx->xcd_swizzle = (int)value;
    return CUKING_OK;
  }
  if (strcmp(key, "counts_mode") == 0) {
    if (value < -1 || value > 1)
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "counts_mode outside [-1, 1]");
    ctx->counts_mode = (int)value;
    return CUKING_OK;
  }
  if (strcmp(key, "filter_quadrant_cap") == 0) {  // tests: 0 sends every quadrant with a candidate to the exact kernel
    if (value < 0 || value > 16384)
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "filter_quadrant_cap outside [0, 16384]");
    ctx->filter_quadrant_cap = (uint32_t)value;
    return CUKING_OK;
  }
  if (strcmp(key, "filter_split_min_steps") == 0) {  // tests: remainder pieces of short bitsets
    if (value < 1 || value > 4096)
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "filter_split_min_steps outside [1, 4096]");
    ctx->filter_split_min_steps = (uint32_t)value;
    return CUKING_OK;
  }
  if (strcmp(key, "filter_check_min_steps") == 0) {  // test hook, process-wide: checks for short bitsets
    if (value < 4 || value > 1 << 20)
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "filter_check_min_steps outside [4, 2^20]");
    set_filter_check_min_steps((uint32_t)value);
    ctx->prepared.valid = false;  // (prefix counts of the workspace belong to the old value)
    return CUKING_OK;
  }
  if (strcmp(key, "filter_sort") == 0) {
    if (value < 0 || value > 2)
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "filter_sort outside [0, 2]");
    ctx->filter_sort = (int)value;
    ctx->prepared.valid = false;  // (the workspace was laid out under the old setting)
    return CUKING_OK;
  }
  if (strcmp(key, "filter_lazy_codes") == 0) {
    if (value < 0 || value > 1)
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "filter_lazy_codes outside [0, 1]");
    ctx->filter_lazy_codes = value != 0;
    ctx->prepared.valid = false;
    return CUKING_OK;
  }
  if (strcmp(key, "filter_check0") == 0) {  // forecast check: 0 off, 1 short launches, 2 always
    if (value < 0 || value > 2)
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "filter_check0 outside [0, 2]");
    ctx->filter_check0 = (int)value;
    return CUKING_OK;
  }
  if (strcmp(key, "filter_rotate") == 0) {  // rotated tiles: 0 off, 1 on, 2 / 3 + phase: test hooks
    if (value < 0 || value > 2 + (int64_t)kNumPhases)
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "filter_rotate outside [0, %u]", 2 + kNumPhases);
    ctx->filter_rotate = (int)value;
    return CUKING_OK;
  }
  if (strcmp(key, "filter_persistent_min_tiles") == 0) {
    if (value < 0 || value > (1 << 30))
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "filter_persistent_min_tiles outside [0, 2^30]");
    ctx->filter_persistent_min_tiles = (uint32_t)value;
    return CUKING_OK;
  }
  if (strcmp(key, "filter_persistent") == 0) {
    ctx->filter_persistent = value != 0;
    return CUKING_OK;
  }
  if (strcmp(key, "filter_rotate_min_tiles") == 0) {
    if (value < 0 || value > (1 << 30))
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "filter_rotate_min_tiles outside [0, 2^30]");
    ctx->filter_rotate_min_tiles = (uint32_t)value;
    return CUKING_OK;
  }
  if (strcmp(key, "filter_rotate_min_steps") == 0) {
    if (value < 1 || value > (1 << 20))
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "filter_rotate_min_steps outside [1, 2^20]");
    ctx->filter_rotate_min_steps = (uint32_t)value;
    return CUKING_OK;
  }
  if (strcmp(key, "filter_check_emit") == 0) {  // live pairs per quadrant handed over at the check
    if (value < 0 || value > 255)
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "filter_check_emit outside [0, 255]");
    ctx->filter_check_emit = (int)value;
    return CUKING_OK;
  }
  if (strcmp(key, "filter_check1") == 0) {  // rigorous check: 0 off, 1 automatic, 2 + k entry k
    if (value < 0 || value > 1 + (int64_t)kNumCheckShares || value == 2)
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "filter_check1 outside {0, 1, 3 .. %u}",
                  1 + kNumCheckShares);
    ctx->filter_check1 = (int)value;
    return CUKING_OK;
  }
  if (strcmp(key, "filter_cand_cap") == 0) {  // tests: a short candidate list
    if (value < 0 || value > (int64_t)kFilterCandCap)
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "filter_cand_cap outside [0, %u]", kFilterCandCap);
    ctx->filter_cand_cap = (uint32_t)value;
    return CUKING_OK;
  }
  if (strcmp(key, "reuse_prepared") == 0) {
    if (value < 0 || value > 1)
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "reuse_prepared outside [0, 1]");
    ctx->reuse_prepared = value != 0;
    return CUKING_OK;
  }
  return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "unknown option %s", key);
}

cuking_status cuking_device_alloc(cuking_ctx *ctx, size_t bytes, void **d_ptr) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  if (d_ptr == nullptr) return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null out pointer");
  *d_ptr = nullptr;
  if (bytes == 0) return CUKING_OK;
  HIP_TRY(hipMalloc(d_ptr, bytes));
  return CUKING_OK;
}

cuking_status cuking_device_free(cuking_ctx *ctx, void *d_ptr) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  if (d_ptr) HIP_TRY(hipFree(d_ptr));
  return CUKING_OK;
}

cuking_status cuking_memset_async(cuking_ctx *ctx, void *d_ptr, int byte_value,
                                  size_t bytes, void *stream) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  if (bytes) HIP_TRY(hipMemsetAsync(d_ptr, byte_value, bytes, (hipStream_t)stream));
  return CUKING_OK;
}

cuking_status cuking_copy_to_device(cuking_ctx *ctx, void *d_dst,
                                    const void *src, size_t bytes,
                                    void *stream) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  if (bytes)
    HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice,
                           (hipStream_t)stream));
  return CUKING_OK;
}

cuking_status cuking_copy_to_host(cuking_ctx *ctx, void *dst, const void *d_src,
                                  size_t bytes, void *stream) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  if (bytes)
    HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost,
                           (hipStream_t)stream));
  return CUKING_OK;
}

cuking_status cuking_stream_synchronize(cuking_ctx *ctx, void *stream) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return CUKING_OK;
}

cuking_status cuking_stream_create(cuking_ctx *ctx, void **stream) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  if (stream == nullptr) return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null out pointer");
  hipStream_t s = nullptr;
  HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = s;
  return CUKING_OK;
}

cuking_status cuking_stream_destroy(cuking_ctx *ctx, void *stream) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  if (stream) {
    // hipStreamDestroy lets the stream's work finish; nothing may name it later.
    auto &rs = ctx->readers;
    for (size_t k = 0; k < rs.size(); ++k)
      if (rs[k].first == (hipStream_t)stream) {
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        if (rs[k].second) (void)hipEventDestroy(rs[k].second);
        rs.erase(rs.begin() + k);
        break;
      }
    HIP_TRY(hipStreamDestroy((hipStream_t)stream));
  }
  return CUKING_OK;
}

cuking_status cuking_host_alloc(cuking_ctx *ctx, size_t bytes, void **ptr) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  if (ptr == nullptr) return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null out pointer");
  *ptr = nullptr;
  if (bytes == 0) return CUKING_OK;
  HIP_TRY(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
  return CUKING_OK;
}

cuking_status cuking_host_free(cuking_ctx *ctx, void *ptr) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  if (ptr) HIP_TRY(hipHostFree(ptr));
  return CUKING_OK;
}

// ---- hot path -------------------------------------------------------------

cuking_status cuking_pack_device(cuking_ctx *ctx, const cuking_submatrix *sm,
                                 uint32_t words_per_sample, uint64_t *d_bit_set,
                                 const int64_t *d_row_idx,
                                 const int64_t *d_col_idx,
                                 const int32_t *d_n_alt_alleles,
                                 size_t num_triples, uint32_t *d_status,
                                 void *stream) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  st = cuking_check_block(sm, words_per_sample);
  if (st != CUKING_OK) return st;
  if (num_triples &&
      (!d_bit_set || !d_row_idx || !d_col_idx || !d_n_alt_alleles || !d_status))
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null device pointer");
  HIP_TRY(launch_pack(*sm, words_per_sample, d_bit_set, d_row_idx, d_col_idx,
                      d_n_alt_alleles, num_triples, d_status,
                      (hipStream_t)stream));
  return CUKING_OK;
}

cuking_status cuking_pack_device_compact(cuking_ctx *ctx, const cuking_submatrix *sm,
                                         uint32_t words_per_sample, uint64_t *d_bit_set,
                                         const uint32_t *d_site,
                                         const uint32_t *d_sample_alt, size_t num_triples,
                                         uint32_t *d_status, void *stream) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  st = cuking_check_block(sm, words_per_sample);
  if (st != CUKING_OK) return st;
  if (num_triples && (!d_bit_set || !d_site || !d_sample_alt || !d_status))
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null device pointer");
  HIP_TRY(launch_pack_compact(words_per_sample, sm_num_samples(*sm), d_bit_set, d_site,
                              d_sample_alt, num_triples, d_status, (hipStream_t)stream));
  return CUKING_OK;
}

cuking_status cuking_event_create(cuking_ctx *ctx, void **event) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  if (event == nullptr) return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null out pointer");
  hipEvent_t e = nullptr;
  HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *event = e;
  return CUKING_OK;
}

cuking_status cuking_event_record(cuking_ctx *ctx, void *event, void *stream) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return CUKING_OK;
}

cuking_status cuking_event_synchronize(cuking_ctx *ctx, void *event) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  HIP_TRY(hipEventSynchronize((hipEvent_t)event));
  return CUKING_OK;
}

cuking_status cuking_event_destroy(cuking_ctx *ctx, void *event) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  if (event) HIP_TRY(hipEventDestroy((hipEvent_t)event));
  return CUKING_OK;
}

uint32_t cuking_tile_samples(const cuking_ctx *ctx) {
  return tiled_variant(ctx ? ctx->variant : default_variant()).tile;
}

uint64_t cuking_num_tiles(const cuking_ctx *ctx, const cuking_submatrix *sm) {
  if (sm == nullptr) return 0;
  const TiledVariant &v = tiled_variant(ctx ? ctx->variant : default_variant());
  const PlaneGeometry g = make_geometry(*sm, 2, v);
  if (g.num_rows == 0 || g.num_cols == 0) return 0;
  return total_tiles(make_tiles(g, v, ctx ? ctx->band_rows : 0));
}

cuking_status cuking_tile_bounds(const cuking_ctx *ctx,
                                 const cuking_submatrix *sm, uint64_t tile,
                                 uint32_t *row_begin, uint32_t *row_end,
                                 uint32_t *col_begin, uint32_t *col_end) {
  cuking_status st = cuking_check_block(sm, 2);
  if (st != CUKING_OK) return st;
  const TiledVariant &v = tiled_variant(ctx ? ctx->variant : default_variant());
  const PlaneGeometry g = make_geometry(*sm, 2, v);
  const TileSpace ts = make_tiles(g, v, ctx ? ctx->band_rows : 0);
  if (g.num_rows == 0 || g.num_cols == 0 || tile >= total_tiles(ts))
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "tile index out of range");
  uint32_t b = 0;
  uint64_t first = 0;
  while (first + ts.band_tiles(b) <= tile) first += ts.band_tiles(b++);
  uint32_t tr, tc;
  ts.decode(b, tile - first, &tr, &tc);
  auto clampr = [&](uint64_t x) { return (uint32_t)std::min<uint64_t>(x, g.num_rows); };
  auto clampc = [&](uint64_t x) { return (uint32_t)std::min<uint64_t>(x, g.num_cols); };
  if (row_begin) *row_begin = sm->i_begin + clampr((uint64_t)tr * v.tile);
  if (row_end) *row_end = sm->i_begin + clampr((uint64_t)(tr + 1) * v.tile);
  if (col_begin) *col_begin = sm->j_begin + clampc((uint64_t)tc * v.tile);
  if (col_end) *col_end = sm->j_begin + clampc((uint64_t)(tc + 1) * v.tile);
  return CUKING_OK;
}

cuking_status cuking_ctx_get_option(const cuking_ctx *ctx, const char *key,
                                    int64_t *value) {
  if (ctx == nullptr || key == nullptr || value == nullptr)
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null argument");
  if (strcmp(key, "variant") == 0) *value = ctx->variant;
  else if (strcmp(key, "split_wgs") == 0) *value = ctx->split_wgs;
  else if (strcmp(key, "band_rows") == 0) *value = ctx->band_rows;
  else if (strcmp(key, "counts_mode") == 0) *value = ctx->counts_mode;
  else if (strcmp(key, "xcd_swizzle") == 0) *value = ctx->xcd_swizzle;
  else if (strcmp(key, "dyn_tail_tiles") == 0) *value = ctx->dyn_tail_tiles;
  else if (strcmp(key, "reuse_prepared") == 0) *value = ctx->reuse_prepared ? 1 : 0;
  else if (strcmp(key, "filter_quadrant_cap") == 0) *value = ctx->filter_quadrant_cap;
  else if (strcmp(key, "filter_cand_cap") == 0) *value = ctx->filter_cand_cap;
  else if (strcmp(key, "filter_split_min_steps") == 0) *value = ctx->filter_split_min_steps;
  else if (strcmp(key, "filter_sort") == 0) *value = ctx->filter_sort;
  else if (strcmp(key, "filter_lazy_codes") == 0) *value = ctx->filter_lazy_codes ? 1 : 0;
  else if (strcmp(key, "filter_check0") == 0) *value = ctx->filter_check0;
  else if (strcmp(key, "filter_check1") == 0) *value = ctx->filter_check1;
  else if (strcmp(key, "filter_check_emit") == 0) *value = ctx->filter_check_emit;
  else if (strcmp(key, "filter_rotate") == 0) *value = ctx->filter_rotate;
  else if (strcmp(key, "filter_rotate_min_steps") == 0) *value = ctx->filter_rotate_min_steps;
  else if (strcmp(key, "filter_rotate_min_tiles") == 0) *value = ctx->filter_rotate_min_tiles;
  else if (strcmp(key, "filter_persistent") == 0) *value = ctx->filter_persistent ? 1 : 0;
  else if (strcmp(key, "filter_persistent_min_tiles") == 0) *value = ctx->filter_persistent_min_tiles;
  else if (strcmp(key, "filter_candidates") == 0 || strcmp(key, "filter_dense_quadrants") == 0 ||
           strcmp(key, "filter_early_exits") == 0 || strcmp(key, "filter_rotated_tiles") == 0) {
    // Diagnostics (they WAIT for the device): pairs the bound let through, quadrants
    // handed to the exact kernel, and tiles that left at the rigorous check point, summed
    // over the context's streams, since the context was created.
    const size_t word = strcmp(key, "filter_candidates") == 0        ? kTotalCand
                        : strcmp(key, "filter_dense_quadrants") == 0 ? kTotalDense
                        : strcmp(key, "filter_early_exits") == 0     ? kTotalEarly
                                                                     : kTotalRotated;
    unsigned long long total = 0;
    if (hipSetDevice(ctx->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
      return cuking_fail(CUKING_ERR_DEVICE, "device wait failed");
    for (auto &e : ctx->filter_scratch) {
      unsigned long long v = 0;
      if (hipMemcpy(&v, e.base + filter_scratch_layout(e.tiles).totals + word * 8, 8,
                    hipMemcpyDeviceToHost) != hipSuccess)
        return cuking_fail(CUKING_ERR_DEVICE, "reading the filter counters failed");
      total += v;
    }
    *value = (int64_t)(total + ctx->filter_totals_retired[word]);
  }
  else if (strncmp(key, "filter_total_", 13) == 0) {
    // Diagnostic (WAITS for the device): word N < 32 of the filter's running totals, summed
    // over the context's streams -- the words behind the named ones are the timing build's
    // (-DCUKING_FILTER_TIMING=1, king_filter.hip).
    const int word = atoi(key + 13);
    if (word < 0 || word >= 32) return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "unknown option %s", key);
    if (hipSetDevice(ctx->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
      return cuking_fail(CUKING_ERR_DEVICE, "device wait failed");
    unsigned long long total = 0;
    for (auto &e : ctx->filter_scratch) {
      unsigned long long v = 0;
      if (hipMemcpy(&v, e.base + filter_scratch_layout(e.tiles).totals + (size_t)word * 8, 8,
                    hipMemcpyDeviceToHost) != hipSuccess)
        return cuking_fail(CUKING_ERR_DEVICE, "reading the filter counters failed");
      total += v;
    }
    *value = (int64_t)total;
  }
  else if (strcmp(key, "filter_step_ticks16") == 0) {
    // Diagnostic (WAITS for the device): the 100 MHz counter's ticks per k-step x 16 as the
    // tiles of the last launch chunk measured them (rotated tiles, king_filter.hip), averaged
    // over the XCDs that said so and the context's streams; 0 = nobody did.
    if (hipSetDevice(ctx->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
      return cuking_fail(CUKING_ERR_DEVICE, "device wait failed");
    uint64_t sum = 0, count = 0;
    for (auto &e : ctx->filter_scratch) {
      uint32_t w[8];
      if (hipMemcpy(w, e.base + kCtrlStepTicks * 4, sizeof w, hipMemcpyDeviceToHost) != hipSuccess)
        return cuking_fail(CUKING_ERR_DEVICE, "reading the filter counters failed");
      for (uint32_t v : w)
        if (v != 0) {
          sum += v;
          ++count;
        }
    }
    *value = count != 0 ? (int64_t)(sum / count) : 0;
  }
  else if (strcmp(key, "workspace_allocations") == 0) *value = (int64_t)ctx->workspace_allocations;
  else if (strcmp(key, "host_syncs") == 0) *value = (int64_t)ctx->host_syncs;
  else if (strcmp(key, "conversions_skipped") == 0) *value = (int64_t)ctx->conversions_skipped;
  else return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "unknown option %s", key);
  return CUKING_OK;
}

int cuking_num_variants(void) { return kNumTiledVariants; }
const char *cuking_variant_name(int variant) {
  if (variant < 0 || variant >= kNumTiledVariants) return "";
  return tiled_variant(variant).name;
}

static cuking_status check_compute_args(const cuking_submatrix *sm,
                                        uint32_t words_per_sample,
                                        const uint64_t *d_bit_sets) {
  cuking_status st = cuking_check_block(sm, words_per_sample);
  if (st != CUKING_OK) return st;
  if (sm_num_samples(*sm) != 0 && d_bit_sets == nullptr)
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null bitset pointer");
  return CUKING_OK;
}

cuking_status cuking_compute_king(cuking_ctx *ctx, const cuking_submatrix *sm,
                                  uint32_t words_per_sample,
                                  const uint64_t *d_bit_sets,
                                  float kin_threshold, uint32_t max_results,
                                  cuking_result *d_results,
                                  uint32_t *d_result_index,
                                  uint32_t *d_result_overflow, void *stream) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  st = check_compute_args(sm, words_per_sample, d_bit_sets);
  if (st != CUKING_OK) return st;
  if (!d_result_index || !d_result_overflow || (max_results && !d_results))
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null result pointer");
  if (sm_num_rows(*sm) == 0 || sm_num_cols(*sm) == 0) return CUKING_OK;
  if (ctx->kernel == CUKING_KERNEL_STREAM)
    return run_stream(ctx, *sm, words_per_sample, d_bit_sets, kin_threshold,
                      max_results, d_results, d_result_index, d_result_overflow,
                      nullptr, (hipStream_t)stream);
  return run_tiled(ctx, *sm, words_per_sample, d_bit_sets, 0, 0, true,
                   kin_threshold, max_results, d_results, d_result_index,
                   d_result_overflow, nullptr, (hipStream_t)stream);
}

cuking_status cuking_compute_king_tiles(
    cuking_ctx *ctx, const cuking_submatrix *sm, uint32_t words_per_sample,
    const uint64_t *d_bit_sets, uint64_t tile_begin, uint64_t tile_end,
    float kin_threshold, uint32_t max_results, cuking_result *d_results,
    uint32_t *d_result_index, uint32_t *d_result_overflow, void *stream) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  st = check_compute_args(sm, words_per_sample, d_bit_sets);
  if (st != CUKING_OK) return st;
  if (!d_result_index || !d_result_overflow || (max_results && !d_results))
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null result pointer");
  if (sm_num_rows(*sm) == 0 || sm_num_cols(*sm) == 0) {
    if (tile_begin != 0 || tile_end != 0)
      return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "empty block has no tiles");
    return CUKING_OK;
  }
  return run_tiled(ctx, *sm, words_per_sample, d_bit_sets, tile_begin, tile_end,
                   false, kin_threshold, max_results, d_results, d_result_index,
                   d_result_overflow, nullptr, (hipStream_t)stream);
}

// Offsets of a sample range inside a diagonal block, in tiles.
static cuking_status tile_span(const cuking_submatrix &sm, uint32_t tile,
                               uint32_t begin, uint32_t end, const char *what,
                               uint32_t *t0, uint32_t *t1) {
  if (begin < sm.i_begin || end > sm.i_end || begin > end)
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "%s range [%u, %u) outside the block",
                what, begin, end);
  const uint32_t b = begin - sm.i_begin, e = end - sm.i_begin;
  if (b % tile != 0 || (e % tile != 0 && end != sm.i_end))
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT,
                "%s range [%u, %u) is not aligned to the %u-sample tile edge", what,
                begin, end, tile);
  *t0 = b / tile;
  *t1 = (e + tile - 1) / tile;
  return CUKING_OK;
}

cuking_status cuking_prepare_samples(cuking_ctx *ctx, const cuking_submatrix *sm,
                                     uint32_t words_per_sample,
                                     const uint64_t *d_bit_sets,
                                     uint32_t sample_begin, uint32_t sample_end,
                                     void *stream) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  st = check_compute_args(sm, words_per_sample, d_bit_sets);
  if (st != CUKING_OK) return st;
  if (!sm_is_diag(*sm))
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT,
                "staged preparation needs a diagonal block (rows == columns)");
  const uint32_t tile = plan_variant(ctx, words_per_sample).tile;
  uint32_t t0, t1;
  st = tile_span(*sm, tile, sample_begin, sample_end, "sample", &t0, &t1);
  if (st != CUKING_OK) return st;
  if (t0 == t1) return CUKING_OK;
  PlaneGeometry geo;
  TileSpace tiles;
  // prepare() works in units of 64 plane samples.
  return prepare(ctx, *sm, words_per_sample, d_bit_sets, (hipStream_t)stream, &geo,
                 &tiles, true, t0 * (tile / 64), t1 * (tile / 64));
}

cuking_status cuking_compute_king_rect(
    cuking_ctx *ctx, const cuking_submatrix *sm, uint32_t words_per_sample,
    const uint64_t *d_bit_sets, uint32_t row_begin, uint32_t row_end,
    uint32_t row_step, uint32_t col_begin, uint32_t col_end, float kin_threshold,
    uint32_t max_results, cuking_result *d_results, uint32_t *d_result_index,
    uint32_t *d_result_overflow, void *stream) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  st = check_compute_args(sm, words_per_sample, d_bit_sets);
  if (st != CUKING_OK) return st;
  if (!sm_is_diag(*sm))
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT,
                "rectangle launches need a diagonal block (rows == columns)");
  if (!d_result_index || !d_result_overflow || (max_results && !d_results))
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null result pointer");
  const TiledVariant v = plan_variant(ctx, words_per_sample);
  const PlaneGeometry geo = make_geometry(*sm, words_per_sample, v);
  // Rectangles of a diagonal block contain slots below the diagonal that leave
  // at once; one contiguous chunk of the enumeration per XCD then leaves some
  // XCDs with little real work (100k x 100k in 8 staged rectangles: 0.69 s
  // against 0.63 s round-robin), so rectangles take the XCD order only in its
  // patch form (32 consecutive slots per patch, patches dealt round-robin:
  // 0.66 s), with 17-row bands.
  const TileSpace tiles = make_tiles(geo, v, ctx->band_rows ? ctx->band_rows : 17);
  const size_t need = plane_bytes(geo, v.layout);
  const int variant = effective_variant(ctx, words_per_sample);
  if (ctx->planes == nullptr || ctx->planes_bytes < need ||
      !same_block(ctx->prepared, *sm, words_per_sample, variant, v.tile, d_bit_sets))
    return cuking_fail(CUKING_ERR_FAILED_PRECONDITION,
                "cuking_prepare_samples() has not been called for this block "
                "(or the workspace has been converted for another one since)");
  uint32_t r0, r1, c0, c1;
  st = tile_span(*sm, v.tile, row_begin, row_end, "row", &r0, &r1);
  if (st != CUKING_OK) return st;
  st = tile_span(*sm, v.tile, col_begin, col_end, "column", &c0, &c1);
  if (st != CUKING_OK) return st;
  if (r0 == r1 || c0 == c1) return CUKING_OK;
  if (row_step == 0) row_step = v.tile;
  if (row_step % v.tile != 0)
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT,
                "row_step %u is not a multiple of the %u-sample tile edge", row_step,
                v.tile);
  const uint32_t stride = row_step / v.tile;
  const uint32_t n_rows = (r1 - r0 + stride - 1) / stride;
  {
    // Every sample the rectangle reads must have been converted.
    const uint32_t per = v.tile / 64;
    const std::vector<uint8_t> &done = ctx->prepared.tiles;
    auto converted = [&](uint32_t tile) {
      for (uint32_t t = tile * per; t < (tile + 1) * per; ++t)
        if (t >= done.size() || !done[t]) return false;
      return true;
    };
    for (uint32_t r = r0; r < r1; r += stride)
      if (!converted(r))
        return cuking_fail(CUKING_ERR_FAILED_PRECONDITION,
                    "row samples from %u on have not been prepared",
                    sm->i_begin + r * v.tile);
    for (uint32_t c = c0; c < c1; ++c)
      if (!converted(c))
        return cuking_fail(CUKING_ERR_FAILED_PRECONDITION,
                    "column samples from %u on have not been prepared",
                    sm->i_begin + c * v.tile);
  }

  TiledArgs a = {};
  a.planes = ctx->planes;
  a.geo = geo;
  a.tiles = tiles;
  a.band_prefix = ctx->band_prefix;  // unused in rectangle mode
  a.tile_begin = 0;
  a.rect_rows = n_rows;
  a.rect_cols = c1 - c0;
  a.rect_row0 = r0;
  a.rect_col0 = c0;
  a.rect_row_stride = stride;
  a.i_begin = sm->i_begin;
  a.j_begin = sm->j_begin;
  a.kin_threshold = kin_threshold;
  a.max_results = max_results;
  a.results = d_results;
  a.result_index = d_result_index;
  a.result_overflow = d_result_overflow;
  a.dense_counts = nullptr;
  a.bits = d_bit_sets;
  a.words_per_sample = words_per_sample;
  a.split_tiles = 0;
  a.split_whole = 0;
  a.split_wgs = ctx->split_wgs;
  a.xcd_chunk = ctx->xcd_swizzle == 2 ? 2u : 0u;  // (see above; patches keep the balance)
  a.launch_tiles = 0;
  a.dyn_tiles = ctx->dyn_tail_tiles;
  a.dyn_wgs = 0;
  st = split_scratch_for(ctx, (hipStream_t)stream, &a.split_scratch, &a.split_counters);
  if (st != CUKING_OK) return st;
  st = filter_scratch_for(ctx, (hipStream_t)stream, geo, total_tiles(make_tiles(geo, v, ctx->band_rows)), &a);
  if (st != CUKING_OK) return st;
  layout_order_for(ctx, words_per_sample, geo, &a);
  const bool full = use_full_counts(ctx, kin_threshold, false, words_per_sample);
  if (v.layout == kLayoutNibbleStats && !ctx->prepared.codes &&
      !(variant == kMfmaFilterVariant && !full && kin_threshold > 0.f && kin_threshold < 0.5f)) {
    // (a block converted without the four-product kernel's codes, and a call that runs
    //  that kernel directly)
    st = convert_codes_now(ctx, geo, words_per_sample, d_bit_sets, (hipStream_t)stream);
    if (st != CUKING_OK) return st;
    a.codes_ready = nullptr;
  }
  EventPair *ev = nullptr;
  if (ctx->timing) HIP_TRY(ctx->king_timer.begin((hipStream_t)stream, &ev));
  HIP_TRY(launch_planned(ctx, words_per_sample, full, a, (uint64_t)n_rows * (c1 - c0),
                         (hipStream_t)stream));
  note_reader(ctx, (hipStream_t)stream);
  {
    const uint32_t per = v.tile / 64;
    for (uint32_t r = r0; r < r1; r += stride) mark_read(ctx, r * per, (r + 1) * per);
    mark_read(ctx, c0 * per, c1 * per);
  }
  if (ev) HIP_TRY(hipEventRecord(ev->stop, (hipStream_t)stream));
  return CUKING_OK;
}

cuking_status cuking_ctx_reserve(cuking_ctx *ctx, const cuking_submatrix *sm,
                                 uint32_t words_per_sample, void *const *streams,
                                 size_t num_streams) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  st = cuking_check_block(sm, words_per_sample);
  if (st != CUKING_OK) return st;
  if (num_streams != 0 && streams == nullptr)
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null stream list");
  if (sm_num_rows(*sm) == 0 || sm_num_cols(*sm) == 0) return CUKING_OK;
  const TiledVariant v = plan_variant(ctx, words_per_sample);
  const PlaneGeometry geo = make_geometry(*sm, words_per_sample, v);
  const TileSpace tiles = make_tiles(geo, v, ctx->band_rows);
  st = ensure_workspace(ctx, plane_bytes(geo, v.layout), (size_t)tiles.num_bands() + 1);
  if (st != CUKING_OK) return st;
  if (v.layout == kLayoutNibbleStats && ctx->filter_sort != 0) {
    st = ensure_sort_temp(ctx, geo.s_stride);
    if (st != CUKING_OK) return st;
  }
  if (!same_tile_space(ctx->prefix_for, tiles)) {
    // (nothing may be reading another block's prefix: the caller reserves
    //  before it enqueues work for this block)
    if (!ctx->readers.empty()) {
      ++ctx->host_syncs;
      HIP_TRY(hipDeviceSynchronize());
    }
    st = upload_prefix(ctx, tiles, nullptr);
    if (st != CUKING_OK) return st;
  }
  if (num_streams > 8)
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "at most 8 streams per context can be reserved");
  if (ctx->split_wgs != 0 && is_mfma_variant(ctx->variant)) {
    // (the slab cache holds 8 streams and drops ALL of them when a ninth comes:
    //  make room now rather than lose a slab reserved a moment ago)
    size_t missing = 0;
    for (size_t k = 0; k < num_streams; ++k) {
      bool have = false;
      for (auto &e : ctx->split_scratch) have = have || e.first == (hipStream_t)streams[k];
      missing += have ? 0 : 1;
    }
    if (missing != 0 && ctx->split_scratch.size() + missing > 8) {
      ++ctx->host_syncs;
      HIP_TRY(hipDeviceSynchronize());
      for (auto &e : ctx->split_scratch) (void)hipFree(e.second);
      ctx->split_scratch.clear();
    }
  }
  if (ctx->variant == kMfmaFilterVariant) {
    // (the scratch cache holds 8 streams and evicts the oldest for a ninth: make room
    //  now, so that nothing reserved here is the one evicted)
    size_t missing = 0;
    for (size_t k = 0; k < num_streams; ++k) {
      bool have = false;
      for (auto &e : ctx->filter_scratch) have = have || e.stream == (hipStream_t)streams[k];
      missing += have ? 0 : 1;
    }
    if (missing != 0 && ctx->filter_scratch.size() + missing > 8) {
      // drop what this reservation does not name
      ++ctx->host_syncs;
      HIP_TRY(hipDeviceSynchronize());
      std::vector<cuking_ctx::FilterScratch> keep;
      for (auto &e : ctx->filter_scratch) {
        bool named = false;
        for (size_t k = 0; k < num_streams; ++k) named = named || e.stream == (hipStream_t)streams[k];
        if (named) {
          keep.push_back(e);
        } else {
          retire_filter_totals(ctx, e);
          (void)hipFree(e.base);
        }
      }
      ctx->filter_scratch.swap(keep);
    }
  }
  for (size_t k = 0; k < num_streams; ++k) {
    uint32_t *scratch, *counters;
    st = split_scratch_for(ctx, (hipStream_t)streams[k], &scratch, &counters);
    if (st != CUKING_OK) return st;
    TiledArgs unused = {};
    st = filter_scratch_for(ctx, (hipStream_t)streams[k], geo, total_tiles(tiles), &unused);
    if (st != CUKING_OK) return st;
  }
  return CUKING_OK;
}

cuking_status cuking_invalidate(cuking_ctx *ctx) {
  if (ctx == nullptr) return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null context");
  // The next conversion treats the workspace as another block's: it is ordered
  // behind every kernel that may still read it.
  ctx->prepared.valid = false;
  return CUKING_OK;
}

cuking_status cuking_compute_counts(cuking_ctx *ctx, const cuking_submatrix *sm,
                                    uint32_t words_per_sample,
                                    const uint64_t *d_bit_sets,
                                    cuking_counts *d_counts, void *stream) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  st = check_compute_args(sm, words_per_sample, d_bit_sets);
  if (st != CUKING_OK) return st;
  if (sm_num_rows(*sm) == 0 || sm_num_cols(*sm) == 0) return CUKING_OK;
  if (d_counts == nullptr)
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null counts pointer");
  if (ctx->kernel == CUKING_KERNEL_STREAM)
    return run_stream(ctx, *sm, words_per_sample, d_bit_sets, 0.f, 0, nullptr,
                      nullptr, nullptr, d_counts, (hipStream_t)stream);
  return run_tiled(ctx, *sm, words_per_sample, d_bit_sets, 0, 0, true, 0.f, 0,
                   nullptr, nullptr, nullptr, d_counts, (hipStream_t)stream);
}

// ---- timing ---------------------------------------------------------------

cuking_status cuking_timing_enable(cuking_ctx *ctx, int enabled) {
  if (ctx == nullptr) return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null context");
  ctx->timing = enabled != 0;
  return CUKING_OK;
}

cuking_status cuking_timing_reset(cuking_ctx *ctx) {
  if (ctx == nullptr) return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null context");
  ctx->king_timer.used = 0;
  ctx->prepare_timer.used = 0;
  return CUKING_OK;
}

cuking_status cuking_timing_collect(cuking_ctx *ctx, double *king_ms,
                                    uint64_t *king_launches, double *prepare_ms,
                                    uint64_t *prepare_launches) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  double a = 0, b = 0;
  uint64_t na = 0, nb = 0;
  HIP_TRY(ctx->king_timer.collect(&a, &na));
  HIP_TRY(ctx->prepare_timer.collect(&b, &nb));
#ifdef CUKING_MFMA_TIMELINE
  mfma_timeline_dump();  // diagnostic build: the last matrix-core launch
#endif
#ifdef CUKING_MFMA_STAMPS
  // diagnostic build: per-phase cycles of the matrix-core kernel's k-step, for
  // k-steps without (row 0) and with (row 1) a stage hand-over
  for (auto &e : ctx->split_scratch) {
    std::vector<unsigned long long> h(1024 * 16);
    uint32_t *scratch = e.second + mfma_split_counter_bytes(ctx->split_wgs) / sizeof(uint32_t);
    if (hipMemcpy(h.data(), scratch, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) break;
    double sum[2][6] = {}, steps = 0;
    int n = 0;
    for (int b = 0; b < 1024; ++b)
      if (h[b * 16 + 7] == 0x5354414D50ull) {
        for (int k = 0; k < 6; ++k) sum[0][k] += (double)h[b * 16 + k];
        for (int k = 0; k < 6; ++k) sum[1][k] += (double)h[b * 16 + 8 + k];
        steps += (double)h[b * 16 + 6];
        ++n;
      }
    if (n) {
      // five-product loop: alternating k-steps without / with a hand-over (6 phases);
      // four-product loop: per k-step three slices without and one with (4 groups)
      for (int r = 0; r < 2; ++r) {
        double t = 0;
        for (int k = 0; k < 6; ++k) t += sum[r][k];
        if (t == 0) continue;
        fprintf(stderr,
                "mfma stamps (%d workgroups, row %d = %s hand-over): cycles per k-step of the "
                "tile  p0 %.0f | p1 %.0f | p2 %.0f | p3 %.0f | p4 %.0f | p5 %.0f | total %.0f\n",
                n, r, r ? "with" : "without", sum[r][0] / steps, sum[r][1] / steps,
                sum[r][2] / steps, sum[r][3] / steps, sum[r][4] / steps, sum[r][5] / steps,
                t / steps);
      }
    }
  }
  // four-product loop: 4 slices x 4 groups (hi | hj + reads | dd + requests | q)
  for (auto &e : ctx->split_scratch) {
    std::vector<unsigned long long> h(1024 * 32);
    uint32_t *scratch = e.second + mfma_split_counter_bytes(ctx->split_wgs) / sizeof(uint32_t);
    if (hipMemcpy(h.data(), scratch, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) break;
    double sum[16] = {}, steps = 0;
    int n = 0;
    for (int b = 0; b < 1024; ++b)
      if (h[b * 32 + 25] == 0x5354414D5034ull) {
        for (int k = 0; k < 16; ++k) sum[k] += (double)h[b * 32 + (k < 7 ? k : k + 1)];
        steps += (double)h[b * 32 + 24];
        ++n;
      }
    if (n)
      for (int c = 0; c < 4; ++c)
        fprintf(stderr,
                "mfma4 stamps (%d workgroups) slice %d: cycles per k-step  hi %.0f | hj+reads%s %.0f | "
                "dd+requests %.0f | q %.0f\n",
                n, c, sum[4 * c] / steps, c == 2 ? "+hand-over" : "", sum[4 * c + 1] / steps,
                sum[4 * c + 2] / steps, sum[4 * c + 3] / steps);
  }
#endif
  if (king_ms) *king_ms = a;
  if (king_launches) *king_launches = na;
  if (prepare_ms) *prepare_ms = b;
  if (prepare_launches) *prepare_launches = nb;
  return CUKING_OK;
}

cuking_status cuking_clock_probe(cuking_ctx *ctx, uint64_t microseconds,
                                 uint64_t *d_ticks, void *stream) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  if (d_ticks == nullptr) return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null out pointer");
  if (microseconds == 0 || microseconds > 60ull * 1000 * 1000)
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "probe time outside (0, 60 s]");
  HIP_TRY(launch_clock_probe(microseconds, d_ticks, (hipStream_t)stream));
  return CUKING_OK;
}

// ---- synthetic inputs -----------------------------------------------------

cuking_status cuking_synth_bitset(cuking_ctx *ctx, uint64_t seed,
                                  const uint32_t *d_kind, const uint32_t *d_pa,
                                  const uint32_t *d_pb, uint32_t sample_begin,
                                  uint32_t sample_end, uint32_t num_sites,
                                  uint32_t words_per_sample,
                                  uint64_t *d_bit_set, void *stream) {
  cuking_status st = bind(ctx);
  if (st != CUKING_OK) return st;
  if (sample_end < sample_begin)
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "sample range reversed");
  if (words_per_sample != cuking_words_per_sample(num_sites))
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT,
                "words_per_sample %u does not match %u sites", words_per_sample,
                num_sites);
  if (sample_end > sample_begin && (!d_kind || !d_pa || !d_pb || !d_bit_set))
    return cuking_fail(CUKING_ERR_INVALID_ARGUMENT, "null device pointer");
  HIP_TRY(launch_synth(seed, d_kind, d_pa, d_pb, sample_begin, sample_end,
                       num_sites, words_per_sample, d_bit_set,
                       (hipStream_t)stream));
  return CUKING_OK;
}

}  // extern "C"
