#include "flags.h"

#include <algorithm>
#include <cerrno>
#include <cstdlib>
#include <cstring>

namespace cuking_host {

namespace {

std::string Normalise(std::string name) {
  std::replace(name.begin(), name.end(), '-', '_');
  return name;
}

bool ParseUnsigned(const std::string &text, uint64_t max, uint64_t *out) {
  if (text.empty() || text[0] == '-' || text[0] == '+') return false;
  errno = 0;
  char *end = nullptr;
  const unsigned long long v = strtoull(text.c_str(), &end, 10);
  if (errno != 0 || end == text.c_str() || *end != '\0' || v > max) return false;
  *out = v;
  return true;
}

bool ParseFloat(const std::string &text, float *out) {
  if (text.empty()) return false;
  errno = 0;
  char *end = nullptr;
  const float v = strtof(text.c_str(), &end);
  if (end == text.c_str() || *end != '\0') return false;
  *out = v;
  return true;
}

}  // namespace

std::string Usage() {
  return "cuking (MI355X) -- KING-robust kinship for all sample pairs\n"
         "  --input_uri=DIR        directory with metadata.json + *.parquet "
         "(row_idx, col_idx, n_alt_alleles)\n"
         "  --output_uri=DIR       receives part-<shard>.snappy.parquet\n"
         "  --kin_threshold=F      only store kin > F (default 0.0884)\n"
         "  --split_factor=K       split the relatedness matrix into K(K+1)/2 "
         "shards (default 1)\n"
         "  --shard_index=I        which shard to compute (default 0)\n"
         "  --max_results=N        result records to reserve (default 10485760)\n"
         "  --num_reader_threads=N Parquet reader threads (default 36)\n"
         "  --requester_pays_project=P  accepted for compatibility, unused\n"
         "  --device=D             GPU index (default 0)\n"
         "  --kernel=tiled|stream  device kernel (default tiled)\n"
         "  --variant=N            tiled kernel variant (default: the library's, 7 = one-product "
         "filter + exact recount; 6 = four products for every pair: the choice when more than "
         "~10 % of the calls are missing or the threshold sits inside the noise of unrelated "
         "pairs; same records either way)\n"
         "  --pack=host|device|auto  where triples are packed (default auto: device for inputs "
         "of 1 GiB of Parquet or more when at most 32 reader threads can run at once -- the "
         "smaller of --num_reader_threads and the hardware threads this process sees)\n"
         "  --decode=table|stream|auto  table: decode a whole table (or row group), then pack "
         "it; stream: pack batches of triples as they are decoded (no column-sized buffers); "
         "default auto = stream\n"
         "  --num_gpus=N           share the shard among N GPUs of this node over "
         "RCCL (default 0: one GPU, no RCCL)\n"
         "  --multi_gpu_mode=auto|staged|simple  broadcast overlapped with compute "
         "(diagonal shards) or broadcast then tile ranges\n"
         "  --bcast_chunks=N       pieces the bitset broadcast is cut into (default 8)\n"
         "  --rank_weights=W0,W1,..  simple schedule: tile ranges in proportion to these "
         "weights (default: measured by one calibration launch per GPU)\n"
         "  --calibrate=true|false measure the GPUs' rates before cutting the tile ranges "
         "(default true; large jobs only)\n"
         "  --calibration_tiles=N  tiles per GPU in the calibration launch (default 0: about "
         "2 % of a GPU's share)\n"
         "  --collectives=rccl|loopback  loopback is for tests: the ranks of --num_gpus=N "
         "share ONE GPU, copies instead of RCCL\n"
         "  --inject_failure=R:PHASE  tests: rank R fails in phase setup|compute|gather, or "
         "never comes back from it (hang_compute|hang_gather)\n"
         "  --phase_timeout_seconds=T  --num_gpus runs: a rank that stays in one phase (a "
         "collective, a phase barrier, a wait for its GPU) longer than T seconds ends the "
         "process with exit code 1, naming every rank's phase (default 1800; 0 = no limit)\n"
         "  --synthetic=N,M[,SEED] instead of --input_uri: synthetic cohort of N samples x M "
         "sites generated on the GPU (founders + planted relatives)\n"
         "  --print_schedule       diagnostic: print the multi-GPU schedule (JSON) and "
         "exit before any GPU work\n"
         "  --dump_bitset=FILE     diagnostic: write the packed bitset (raw "
         "little-endian u64) and exit before any GPU work\n"
         "Dashes and underscores are interchangeable in flag names.\n";
}

std::string ParseFlags(int argc, char **argv, Flags *flags) {
  for (int a = 1; a < argc; ++a) {
    std::string arg = argv[a];
    if (arg == "--help" || arg == "-h" || arg == "-help") {
      flags->help = true;
      continue;
    }
    if (arg.size() < 3 || arg[0] != '-') return "Unexpected argument: " + arg;
    // Abseil accepts -flag and --flag.
    arg = arg.substr(arg[1] == '-' ? 2 : 1);
    std::string name, value;
    bool has_value = false;
    const size_t eq = arg.find('=');
    if (eq != std::string::npos) {
      name = arg.substr(0, eq);
      value = arg.substr(eq + 1);
      has_value = true;
    } else {
      name = arg;
    }
    name = Normalise(name);
    auto need_value = [&]() -> bool {
      if (has_value) return true;
      if (a + 1 >= argc) return false;
      value = argv[++a];
      return true;
    };
    uint64_t u = 0;
    if (name == "input_uri") {
      if (!need_value()) return "Missing value for --input_uri";
      flags->input_uri = value;
    } else if (name == "output_uri") {
      if (!need_value()) return "Missing value for --output_uri";
      flags->output_uri = value;
    } else if (name == "requester_pays_project") {
      if (!need_value()) return "Missing value for --requester_pays_project";
      flags->requester_pays_project = value;
    } else if (name == "num_reader_threads") {
      if (!need_value() || !ParseUnsigned(value, SIZE_MAX, &u))
        return "Illegal value '" + value + "' specified for flag 'num_reader_threads'";
      flags->num_reader_threads = (size_t)u;
    } else if (name == "max_results") {
      if (!need_value() || !ParseUnsigned(value, UINT32_MAX, &u))
        return "Illegal value '" + value + "' specified for flag 'max_results'";
      flags->max_results = (uint32_t)u;
    } else if (name == "kin_threshold") {
      if (!need_value() || !ParseFloat(value, &flags->kin_threshold))
        return "Illegal value '" + value + "' specified for flag 'kin_threshold'";
    } else if (name == "split_factor") {
      if (!need_value() || !ParseUnsigned(value, UINT32_MAX, &u))
        return "Illegal value '" + value + "' specified for flag 'split_factor'";
      flags->split_factor = (uint32_t)u;
    } else if (name == "shard_index") {
      if (!need_value() || !ParseUnsigned(value, UINT32_MAX, &u))
        return "Illegal value '" + value + "' specified for flag 'shard_index'";
      flags->shard_index = (uint32_t)u;
    } else if (name == "device") {
      if (!need_value() || !ParseUnsigned(value, 1023, &u))
        return "Illegal value '" + value + "' specified for flag 'device'";
      flags->device = (int)u;
    } else if (name == "kernel") {
      if (!need_value() || (value != "tiled" && value != "stream"))
        return "Illegal value '" + value + "' specified for flag 'kernel'";
      flags->kernel = value;
    } else if (name == "variant") {
      if (!need_value() || !ParseUnsigned(value, 63, &u))
        return "Illegal value '" + value + "' specified for flag 'variant'";
      flags->variant = (int)u;
    } else if (name == "dump_bitset") {
      if (!need_value()) return "Missing value for --dump_bitset";
      flags->dump_bitset = value;
    } else if (name == "num_gpus") {
      if (!need_value() || !ParseUnsigned(value, 64, &u))
        return "Illegal value '" + value + "' specified for flag 'num_gpus'";
      flags->num_gpus = (uint32_t)u;
    } else if (name == "bcast_chunks") {
      if (!need_value() || !ParseUnsigned(value, 1u << 20, &u) || u == 0)
        return "Illegal value '" + value + "' specified for flag 'bcast_chunks'";
      flags->bcast_chunks = (uint32_t)u;
    } else if (name == "multi_gpu_mode") {
      if (!need_value() || (value != "auto" && value != "staged" && value != "simple"))
        return "Illegal value '" + value + "' specified for flag 'multi_gpu_mode'";
      flags->multi_gpu_mode = value;
    } else if (name == "synthetic") {
      if (!need_value()) return "Missing value for --synthetic";
      uint64_t parts[3] = {0, 0, flags->synth_seed};
      size_t count = 0, pos = 0;
      bool ok = true;
      while (ok && pos <= value.size() && count < 3) {
        const size_t comma = value.find(',', pos);
        const std::string item =
            value.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
        ok = ParseUnsigned(item, count < 2 ? 0xFFFFFFFFull : UINT64_MAX, &parts[count]);
        ++count;
        if (comma == std::string::npos) break;
        pos = comma + 1;
        if (count == 3) ok = false;  // a fourth item
      }
      if (!ok || count < 2 || parts[0] == 0 || parts[1] == 0)
        return "Illegal value '" + value + "' specified for flag 'synthetic' (N,M[,seed])";
      flags->synthetic = value;
      flags->synth_samples = (uint32_t)parts[0];
      flags->synth_sites = (uint32_t)parts[1];
      flags->synth_seed = parts[2];
    } else if (name == "print_schedule") {
      if (has_value && value != "true" && value != "1")
        return "Illegal value '" + value + "' specified for flag 'print_schedule'";
      flags->print_schedule = true;
    } else if (name == "pack") {
      if (!need_value() || (value != "host" && value != "device" && value != "auto"))
        return "Illegal value '" + value + "' specified for flag 'pack'";
      flags->pack = value;
    } else if (name == "decode") {
      if (!need_value() || (value != "table" && value != "stream" && value != "auto"))
        return "Illegal value '" + value + "' specified for flag 'decode'";
      flags->decode = value;
    } else if (name == "decode_batch") {
      if (!need_value() || !ParseUnsigned(value, 1u << 24, &u))
        return "Illegal value '" + value + "' specified for flag 'decode_batch'";
      flags->decode_batch = (size_t)u;
    } else if (name == "collectives") {
      if (!need_value() || (value != "rccl" && value != "loopback"))
        return "Illegal value '" + value + "' specified for flag 'collectives'";
      flags->collectives = value;
    } else if (name == "calibrate") {
      if (has_value && value != "true" && value != "false" && value != "1" && value != "0")
        return "Illegal value '" + value + "' specified for flag 'calibrate'";
      flags->calibrate = !has_value || value == "true" || value == "1";
    } else if (name == "calibration_tiles") {
      if (!need_value() || !ParseUnsigned(value, UINT64_MAX, &u))
        return "Illegal value '" + value + "' specified for flag 'calibration_tiles'";
      flags->calibration_tiles = u;
    } else if (name == "rank_weights") {
      if (!need_value()) return "Missing value for --rank_weights";
      flags->rank_weights = value;
      flags->rank_weight_values.clear();
      size_t pos = 0;
      while (pos <= value.size()) {
        const size_t comma = value.find(',', pos);
        const std::string item =
            value.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
        errno = 0;
        char *end = nullptr;
        const double w = strtod(item.c_str(), &end);
        if (item.empty() || end == item.c_str() || *end != '\0' || errno != 0 || !(w > 0))
          return "Illegal value '" + value + "' specified for flag 'rank_weights'";
        flags->rank_weight_values.push_back(w);
        if (comma == std::string::npos) break;
        pos = comma + 1;
      }
    } else if (name == "inject_failure") {
      if (!need_value()) return "Missing value for --inject_failure";
      const size_t colon = value.find(':');
      uint64_t r = 0;
      if (colon == std::string::npos || !ParseUnsigned(value.substr(0, colon), 63, &r))
        return "Illegal value '" + value + "' specified for flag 'inject_failure'";
      const std::string phase = value.substr(colon + 1);
      if (phase != "setup" && phase != "compute" && phase != "gather" &&
          phase != "hang_compute" && phase != "hang_gather")
        return "Illegal value '" + value + "' specified for flag 'inject_failure'";
      flags->inject_failure = value;
      flags->inject_failure_rank = (int)r;
      flags->inject_failure_phase = phase;
    } else if (name == "phase_timeout_seconds") {
      errno = 0;
      char *end = nullptr;
      const double t = need_value() ? strtod(value.c_str(), &end) : -1.0;
      if (value.empty() || end == value.c_str() || *end != '\0' || errno != 0 || !(t >= 0))
        return "Illegal value '" + value + "' specified for flag 'phase_timeout_seconds'";
      flags->phase_timeout_seconds = t;
    } else {
      return "Unknown command line flag '" + name + "'";
    }
  }
  return "";
}

std::string ValidateFlags(const Flags &f) {
  if (f.input_uri.empty() && f.synthetic.empty())
    return "No input URI specified";                                 // :438-440
  if (f.output_uri.empty()) return "No output URI specified";        // :444-446
  if (f.num_reader_threads == 0) return "Invalid number of reader threads";  // :450-452
  if (f.split_factor == 0) return "Invalid split factor";            // :455-457
  const uint64_t shards = (uint64_t)f.split_factor * ((uint64_t)f.split_factor + 1) / 2;
  if (f.shard_index >= shards) return "Invalid shard index";         // :460-462
  return "";
}

}  // namespace cuking_host
