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

struct StagedStep {
  SampleChunk chunk;  // arrives in this step (= the rectangle's columns)
  bool has_rect;      // false: none of the rank's rows lie below the chunk end
  uint32_t row_begin, row_end, row_step;  // rows row_begin, +row_step, ... < row_end
};

inline std::vector<StagedStep> StagedSchedule(uint32_t num_samples, uint32_t tile,
                                              uint32_t world, uint32_t rank,
                                              uint32_t num_chunks) {
  std::vector<StagedStep> out;
  const uint64_t first_row = (uint64_t)rank * tile;
  for (const SampleChunk &c : ChunkRanges(num_samples, tile, num_chunks)) {
    StagedStep s;
    s.chunk = c;
    s.has_rect = first_row < c.end;
    s.row_begin = s.has_rect ? (uint32_t)first_row : 0;
    s.row_end = s.has_rect ? c.end : 0;
    s.row_step = world * tile;
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
