// Command-line flags of the `cuking` binary: same names, defaults and
// validation as the reference's Abseil flags (cuking.cu:27-52, :436-465),
// without Abseil.  Both spellings are accepted: the binary's own
// `--kin_threshold` and the wrapper's `--kin-threshold`
// (cloud_batch_submit.py:28-32), as `--flag=value` or `--flag value`.
#ifndef CUKING_AMD_HOST_FLAGS_H_
#define CUKING_AMD_HOST_FLAGS_H_

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace cuking_host {

struct Flags {
  std::string input_uri;                // cuking.cu:27
  std::string output_uri;               // :30
  std::string requester_pays_project;   // :33 (accepted, unused: no GCS here)
  size_t num_reader_threads = 36;       // :36
  uint32_t max_results = 10u << 20;     // :40
  float kin_threshold = 0.0884f;        // :43
  uint32_t split_factor = 1;            // :46
  uint32_t shard_index = 0;             // :49
  // Additions (no reference counterpart).
  int device = 0;                       // HIP device index
  std::string kernel = "tiled";         // tiled | stream
  int variant = -1;                     // tiled kernel variant (library option "variant"); -1: the default
  std::string pack = "auto";            // host | device | auto (device for large inputs when
                                        // few reader threads can run at once, cuking_main.cc)
  std::string decode = "auto";          // table: a whole table (or row group) decoded, then
                                        // packed; stream: batches of triples packed as they
                                        // are decoded (parquet_io.h StreamTriples); auto:
                                        // stream
  size_t decode_batch = 0;              // test hook: triples per batch of --decode=stream
                                        // (0: 32 Ki for the host pack, a staging slot's
                                        // worth for the device pack)
  std::string dump_bitset;              // diagnostic: write the packed host
                                        // bitset here and exit (no GPU used)
  // Several GPUs of this node share the shard (multi_gpu.h).  0 = the classic
  // one-GPU path of the reference; N >= 1 = N ranks over RCCL (N = 1 runs the
  // same collectives on a one-rank communicator).
  uint32_t num_gpus = 0;
  std::string multi_gpu_mode = "auto";  // auto | staged | simple
  uint32_t bcast_chunks = 8;
  // Simple schedule: per-rank weights of the tile ranges ("1,1.05,..."); empty =
  // measured by a calibration launch per rank (--calibrate, on by default).
  std::string rank_weights;
  std::vector<double> rank_weight_values;
  bool calibrate = true;
  uint64_t calibration_tiles = 0;       // per rank; 0 = automatic
  std::string collectives = "rccl";     // rccl | loopback (TEST ONLY: rank threads share one GPU)
  std::string inject_failure;           // TEST ONLY: "rank:phase" (setup | compute | gather fail;
                                        // hang_compute | hang_gather: the rank never comes back)
  int inject_failure_rank = -1;
  std::string inject_failure_phase;
  // Wall-clock limit of every phase of a --num_gpus run (multi_gpu.cc: a rank that
  // sits in one phase -- inside a collective, at a phase barrier, waiting for its
  // device -- longer than this ends the process with exit code 1 and a message that
  // names every rank's phase); 0 = no limit.
  double phase_timeout_seconds = 1800;
  // "N,M[,seed]": no input tables; the cohort of synth_plan.h is generated on
  // the GPU (BASELINE configs without their 1e9 .. 1.5e11-row Parquet form).
  std::string synthetic;
  uint32_t synth_samples = 0, synth_sites = 0;
  uint64_t synth_seed = 20240229;
  bool print_schedule = false;          // diagnostic: print the multi-GPU
                                        // schedule as JSON and exit (no GPU used)
  bool help = false;
};

// Returns "" on success, otherwise the error message.
std::string ParseFlags(int argc, char **argv, Flags *flags);
// The reference's validation order and messages (cuking.cu:437-462).
std::string ValidateFlags(const Flags &flags);
std::string Usage();

}  // namespace cuking_host

#endif  // CUKING_AMD_HOST_FLAGS_H_
