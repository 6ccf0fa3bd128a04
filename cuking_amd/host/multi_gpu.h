// One block over N GPUs of this node from ONE process: a host thread, a
// context and two streams per GPU, RCCL (ncclCommInitAll) for the two exchange
// steps -- chunked broadcast of the packed bitset from the GPU that holds it,
// gather of the thresholded records on rank 0.  north_star: "block-partitioned
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
};

struct MultiGpuOutput {
  std::vector<cuking_result> results;  // all ranks', unsorted
  std::string mode;                    // the schedule that ran
  double exchange_and_compute_seconds = 0, gather_seconds = 0;
  double comm_init_seconds = 0;        // ncclCommInitAll (seconds on a cold process)
  std::vector<double> rank_kernel_ms, rank_prepare_ms;
  std::vector<uint32_t> rank_results;
  uint64_t bytes_broadcast = 0;
};

// Returns "" on success; otherwise the message, with *code set to the
// absl-style status name (RESOURCE_EXHAUSTED for result overflow).
std::string RunMultiGpu(const MultiGpuInput &in, MultiGpuOutput *out,
                        std::string *code);

}  // namespace cuking_host

#endif  // CUKING_AMD_HOST_MULTI_GPU_H_
