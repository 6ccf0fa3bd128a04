// Host-side scheduling of one block over the GPUs of a node: pure integer
// functions (no GPU, no RCCL), shared by multi_gpu.cc and `--print_schedule`.
//
// Replaces the reference's one-VM-per-shard fan-out (cloud_batch_submit.py:45,
// :73; README.md:94-102) inside a node.  Two schedules:
//   simple   every rank takes an equal contiguous range of the kernel's tile
//            enumeration (cuking_num_tiles / cuking_compute_king_tiles)
//   staged   diagonal blocks: the bitset travels in ascending tile-aligned
//            sample chunks; tile rows are dealt round-robin (row r -> rank
//            r mod W); as chunk c lands a rank evaluates (its rows below the
//            chunk end) x (chunk c).  A pair (i < j) only needs the chunk of
//            j and the chunks before it, so everybody computes while later
//            chunks are still on the wire.
// The Python driver has the same functions (cuking_amd/dist.py); the tests
// check the two against each other and by brute force.
#ifndef CUKING_AMD_HOST_SCHEDULE_H_
#define CUKING_AMD_HOST_SCHEDULE_H_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace cuking_host {

struct TileRange {
  uint64_t begin, end;
};

// Contiguous ranges whose sizes differ by at most one tile.
inline std::vector<TileRange> TilePartition(uint64_t num_tiles, uint32_t world) {
  std::vector<TileRange> out;
  const uint64_t base = num_tiles / world, extra = num_tiles % world;
  uint64_t begin = 0;
  for (uint32_t r = 0; r < world; ++r) {
    const uint64_t end = begin + base + (r < extra ? 1 : 0);
    out.push_back({begin, end});
    begin = end;
  }
  return out;
}

// Contiguous ranges proportional to `weights` (a rank's measured speed, tiles
// per millisecond): the GPUs of one node sustain clocks several percent apart
// under this kernel, and with equal ranges the slowest one sets the pace.
// Exact cover of [0, num_tiles), monotone.  Same arithmetic as
// cuking_amd/dist.py weighted_tile_partition (round half to even).
inline std::vector<TileRange> WeightedTilePartition(uint64_t num_tiles,
                                                    const std::vector<double> &weights) {
  std::vector<TileRange> out;
  double total = 0;
  for (double w : weights) total += w;
  uint64_t begin = 0;
  double acc = 0;
  for (size_t r = 0; r < weights.size(); ++r) {
    acc += weights[r];
    uint64_t end = num_tiles;
    if (r + 1 != weights.size()) {
      const double cut = std::nearbyint((double)num_tiles * acc / total);
      end = cut <= (double)begin ? begin : (uint64_t)cut;
      if (end > num_tiles) end = num_tiles;
    }
    out.push_back({begin, end});
    begin = end;
  }
  return out;
}

// Calibration of the simple schedule: every rank first evaluates
// CalibrationTiles() tiles of its own (rank r: [r * c, (r + 1) * c)), timed
// with events; the rates are all-gathered and the remaining tiles
// [world * c, num_tiles) are cut in proportion (WeightedTilePartition).  About
// 2 % of a rank's share, whole rounds of 256 workgroups, at least 8 rounds --
// and none (0) for jobs under 64 rounds per rank, where a few percent of
// imbalance cost less than the extra launch and the rate exchange.
inline uint64_t CalibrationTiles(uint64_t num_tiles, uint32_t world) {
  if (world < 2) return 0;
  const uint64_t share = num_tiles / world;
  if (share < 64 * 256) return 0;
  uint64_t c = share / 50 / 256 * 256;
  if (c < 8 * 256) c = 8 * 256;
  return c;
}

struct SampleChunk {
  uint32_t begin, end;  // block-local sample indices
};

// Ascending, tile-aligned chunks covering [0, num_samples).
inline std::vector<SampleChunk> ChunkRanges(uint32_t num_samples, uint32_t tile,
                                            uint32_t num_chunks) {
  std::vector<SampleChunk> out;
  const uint64_t tiles = ((uint64_t)num_samples + tile - 1) / tile;
  uint64_t n = num_chunks < 1 ? 1 : num_chunks;
  if (n > tiles) n = tiles;
  for (uint64_t c = 0; c < n; ++c) {
    const uint64_t b = tiles * c / n * tile;
    uint64_t e = tiles * (c + 1) / n * tile;
    if (e > num_samples) e = num_samples;
    if (e > b) out.push_back({(uint32_t)b, (uint32_t)e});
  }
  return out;
}

// Which tile rows a rank owns in the staged schedule: row r belongs to the rank
// that owns position r mod period of the deal.  Unweighted: period = world,
// position p -> rank p (round-robin).  Weighted (the GPUs of a node differ by
// several percent): period = 16 x world positions, rank r gets a share of them
// in proportion to its weight (largest remainders), spread evenly over the
// period -- evenly, because the rows of a chunk's
// own triangle differ in length and a rank with the first rows of every period
// would get more of it.
struct RowDeal {
  uint32_t period = 1;
  std::vector<std::vector<uint32_t>> offsets;  // per rank: its positions in [0, period)
};

inline RowDeal MakeRowDeal(uint32_t world, const std::vector<double> &weights) {
  RowDeal d;
  d.offsets.assign(world, {});
  if (weights.size() != world) {
    d.period = world;
    for (uint32_t r = 0; r < world; ++r) d.offsets[r].push_back(r);
    return d;
  }
  d.period = 16 * world;
  double total = 0;
  for (double w : weights) total += w;
  // positions per rank: floor of the exact share, the rest by largest remainder
  std::vector<uint32_t> n(world);
  std::vector<std::pair<double, uint32_t>> rem;
  uint32_t given = 0;
  for (uint32_t r = 0; r < world; ++r) {
    const double exact = d.period * weights[r] / total;
    n[r] = (uint32_t)exact;
    if (n[r] == 0) n[r] = 1;  // every rank takes part
    given += n[r];
    rem.push_back({exact - std::floor(exact), r});
  }
  for (uint32_t k = 0; given < d.period; ++k) {
    uint32_t best = 0;
    for (uint32_t r = 1; r < world; ++r)
      if (rem[r].first > rem[best].first) best = r;
    ++n[best];
    rem[best].first = -1;
    ++given;
    if (k > 4 * world) break;
  }
  while (given > d.period) {  // (the "at least one" rule overshot: take from the largest)
    uint32_t big = 0;
    for (uint32_t r = 1; r < world; ++r)
      if (n[r] > n[big]) big = r;
    --n[big];
    --given;
  }
  // every rank's positions evenly spread: its j-th of n sits nearest to
  // (j + 1/2) x period / n; positions are handed out in the order of those targets
  std::vector<std::pair<double, uint32_t>> targets;
  for (uint32_t r = 0; r < world; ++r)
    for (uint32_t j = 0; j < n[r]; ++j)
      targets.push_back({(j + 0.5) * d.period / n[r], r});
  std::stable_sort(targets.begin(), targets.end(),
                   [](const std::pair<double, uint32_t> &a, const std::pair<double, uint32_t> &b) {
                     return a.first < b.first;
                   });
  for (uint32_t p = 0; p < d.period; ++p) d.offsets[targets[p].second].push_back(p);
  return d;
}

struct RowStride {
  uint32_t row_begin, row_end, row_step;  // rows row_begin, +row_step, ... < row_end (samples)
};

struct StagedStep {
  SampleChunk chunk;  // arrives in this step (= the rectangle's columns)
  bool has_rect;      // false: none of the rank's rows lie below the chunk end
  uint32_t row_begin, row_end, row_step;  // the first (unweighted: the only) row set
  std::vector<RowStride> rects;           // every row set of the rank for this chunk
};

inline std::vector<StagedStep> StagedSchedule(uint32_t num_samples, uint32_t tile,
                                              uint32_t world, uint32_t rank,
                                              uint32_t num_chunks,
                                              const RowDeal *deal = nullptr) {
  RowDeal plain;
  if (deal == nullptr) {
    plain = MakeRowDeal(world, {});
    deal = &plain;
  }
  std::vector<StagedStep> out;
  for (const SampleChunk &c : ChunkRanges(num_samples, tile, num_chunks)) {
    StagedStep s;
    s.chunk = c;
    s.has_rect = false;
    s.row_begin = s.row_end = 0;
    s.row_step = deal->period * tile;
    for (uint32_t off : deal->offsets[rank]) {
      const uint64_t first_row = (uint64_t)off * tile;
      if (first_row >= c.end) continue;
      s.rects.push_back({(uint32_t)first_row, c.end, deal->period * tile});
    }
    if (!s.rects.empty()) {
      s.has_rect = true;
      s.row_begin = s.rects[0].row_begin;
      s.row_end = s.rects[0].row_end;
    }
    out.push_back(s);
  }
  return out;
}

// Where each rank's records land in rank 0's gather buffer.
struct GatherPlan {
  std::vector<uint64_t> offset;  // in records
  uint64_t total;
};

inline GatherPlan PlanGather(const std::vector<uint32_t> &counts) {
  GatherPlan p;
  p.total = 0;
  for (uint32_t c : counts) {
    p.offset.push_back(p.total);
    p.total += c;
  }
  return p;
}

}  // namespace cuking_host

#endif  // CUKING_AMD_HOST_SCHEDULE_H_
