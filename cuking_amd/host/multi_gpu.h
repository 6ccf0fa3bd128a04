// One block over N GPUs of this node from ONE process: a host thread, a
// context and three streams per GPU, RCCL (behind collectives.h) for the two
// exchange steps -- chunked broadcast of the packed bitset from the GPU that
// holds it, gather of the thresholded records on rank 0.  north_star: "block-partitioned
// across the 8 GPUs of one node, bitset halves broadcast with RCCL over xGMI,
// thresholded pair lists gathered at the end"; reference anchor: Run() is C++
// end to end (cuking.cu:435-882) and fans shards out over VMs instead
// (cloud_batch_submit.py:45,73).
#ifndef CUKING_AMD_HOST_MULTI_GPU_H_
#define CUKING_AMD_HOST_MULTI_GPU_H_

#include <cstdint>
#include <string>
#include <vector>

#include "cuking_amd.h"

namespace cuking_host {

struct MultiGpuInput {
  int num_gpus = 1;
  int first_device = 0;            // ranks use devices first_device .. + num_gpus - 1
  std::string kernel = "tiled";    // tiled | stream (stream: simple schedule only)
  std::string mode = "auto";       // auto | staged | simple
  uint32_t chunks = 8;             // broadcast chunks
  cuking_submatrix sm = {0, 0, 0, 0};
  uint32_t words_per_sample = 0;
  // The packed bitset of the block: host memory (page-locked or not), or, if
  // host_bits is null, device memory on first_device (--pack=device).
  const uint64_t *host_bits = nullptr;
  uint64_t *d_bits_rank0 = nullptr;
  float kin_threshold = 0.f;
  uint32_t max_results = 0;
  // "rccl" (product) or "loopback" (TEST ONLY, collectives.h: rank threads may
  // then share a GPU -- every rank runs on first_device).
  std::string collectives = "rccl";
  // Simple schedule: tile ranges in proportion to these per-rank weights; empty
  // = measure them (one calibration launch per rank, schedule.h) unless
  // `calibrate` is off or the job is small, then equal ranges.
  std::vector<double> rank_weights;
  bool calibrate = true;
  uint64_t calibration_tiles = 0;  // per rank; 0 = schedule.h CalibrationTiles()
  // A rank that stays in one phase longer than this (seconds; 0 = no limit) ends
  // the PROCESS: "\nError: DEADLINE_EXCEEDED: ..." with every rank's phase on stderr,
  // exit code 1.  (Threads stuck inside a collective or a device wait cannot be
  // cancelled, so there is no status to return.)
  double phase_timeout_seconds = 0;
  // TEST ONLY: rank `inject_failure_rank` reports a failure in phase "setup",
  // "compute" or "gather" (exercises the agreement on failures), or never comes
  // back from it ("hang_compute", "hang_gather": exercises the watchdog).
  int inject_failure_rank = -1;
  std::string inject_failure_phase;
};

struct MultiGpuOutput {
  std::vector<cuking_result> results;  // all ranks', unsorted
  std::string mode;                    // the schedule that ran
  double exchange_and_compute_seconds = 0, gather_seconds = 0;
  double comm_init_seconds = 0;        // ncclCommInitAll (seconds on a cold process)
  std::vector<double> rank_kernel_ms, rank_prepare_ms;
  std::vector<uint32_t> rank_results;
  uint64_t bytes_broadcast = 0;
  std::string collectives;             // implementation that ran
  // Simple schedule: the tile range each rank evaluated after calibration and
  // the rates (tiles / ms) behind the cut (empty: equal ranges).
  std::vector<std::pair<uint64_t, uint64_t>> rank_tile_ranges;
  std::vector<double> rank_rates;
  uint64_t calibration_tiles = 0;      // per rank
  // Workspace allocations / host-side waits the library made on a rank AFTER
  // cuking_ctx_reserve, i.e. while collectives may be in flight: must be 0.
  std::vector<uint64_t> rank_allocations_after_reserve, rank_host_syncs_after_reserve;
};

// Returns "" on success; otherwise the message, with *code set to the
// absl-style status name (RESOURCE_EXHAUSTED for result overflow).
std::string RunMultiGpu(const MultiGpuInput &in, MultiGpuOutput *out,
                        std::string *code);

}  // namespace cuking_host

#endif  // CUKING_AMD_HOST_MULTI_GPU_H_
