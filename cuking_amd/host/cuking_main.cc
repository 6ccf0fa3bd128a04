// `cuking` for MI355X: drop-in for the reference binary's Parquet-in /
// Parquet-out path (cuking.cu:435-895) -- same flags, same input layout, same
// output file name, schema and ordering -- with the device work behind the
// C ABI of include/cuking_amd.h.  C++ host only; no GPU code in this file.
//
// Differences from the reference, all at the storage edge: URIs are local
// paths (or file://) because no GCS client exists in this image; gs:// is
// rejected with a clear error.  Everything after the bytes are read is the
// same pipeline: metadata -> Submatrix -> all-ones bitset -> parallel Parquet
// decode + pack -> kernel -> overflow check -> sort -> Snappy Parquet.
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <future>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "cuking_amd.h"
#include "flags.h"
#include "metadata.h"
#include "multi_gpu.h"
#include "parquet_io.h"
#include "schedule.h"
#include "synth_plan.h"
#include "thread_pool.h"

namespace {

using cuking_host::Flags;

struct Status {
  std::string code;  // "" = OK; otherwise absl-style code name
  std::string message;
  bool ok() const { return code.empty(); }
  static Status Ok() { return {}; }
};

Status InvalidArgument(std::string m) { return {"INVALID_ARGUMENT", std::move(m)}; }
Status FailedPrecondition(std::string m) { return {"FAILED_PRECONDITION", std::move(m)}; }
Status ResourceExhausted(std::string m) { return {"RESOURCE_EXHAUSTED", std::move(m)}; }
Status Unknown(std::string m) { return {"UNKNOWN", std::move(m)}; }

Status FromAbi(cuking_status st) {
  if (st == CUKING_OK) return Status::Ok();
  const std::string msg = cuking_last_error();
  switch (st) {
    case CUKING_ERR_INVALID_ARGUMENT: return InvalidArgument(msg);
    case CUKING_ERR_FAILED_PRECONDITION: return FailedPrecondition(msg);
    case CUKING_ERR_RESOURCE_EXHAUSTED: return ResourceExhausted(msg);
    case CUKING_ERR_OUT_OF_MEMORY: return {"RESOURCE_EXHAUSTED", msg};
    default: return {"INTERNAL", msg};
  }
}

#define RETURN_IF_ERROR(expr)          \
  do {                                 \
    Status _s = (expr);                \
    if (!_s.ok()) return _s;           \
  } while (0)

class StopWatch {  // cuking.cu:325-337
 public:
  double ElapsedAndReset() {
    const auto now = std::chrono::steady_clock::now();
    const double s = std::chrono::duration<double>(now - last_).count();
    last_ = now;
    return s;
  }

 private:
  std::chrono::steady_clock::time_point last_ = std::chrono::steady_clock::now();
};

void Done(StopWatch *sw) {
  std::cout << " (" << std::fixed << std::setprecision(3) << sw->ElapsedAndReset()
            << "s)" << std::endl;
}

// The reference only takes gs:// URIs (cuking.cu:340-353).  Here: local.
Status ResolveUri(const std::string &uri, std::string *path) {
  if (uri.rfind("gs://", 0) == 0)
    return InvalidArgument("Unsupported URI: " + uri +
                           " (this build has no GCS client; pass a local "
                           "directory or file:// URI)");
  *path = uri.rfind("file://", 0) == 0 ? uri.substr(7) : uri;
  while (path->size() > 1 && path->back() == '/') path->pop_back();
  if (path->empty()) return InvalidArgument("Incomplete URI " + uri);
  return Status::Ok();
}

// Non-recursive listing of *.parquet (cuking.cu:529-545: the "/" delimiter
// skips Spark's _temporary/ folders).
Status ListParquetFiles(const std::string &dir,
                        std::vector<std::pair<std::string, size_t>> *files) {
  DIR *d = opendir(dir.c_str());
  if (d == nullptr)
    return FailedPrecondition("Cannot list " + dir + ": " + strerror(errno));
  while (dirent *e = readdir(d)) {
    const std::string name = e->d_name;
    const std::string suffix = ".parquet";
    if (name.size() < suffix.size() ||
        name.compare(name.size() - suffix.size(), suffix.size(), suffix) != 0)
      continue;
    const std::string full = dir + "/" + name;
    struct stat st;
    if (stat(full.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) continue;
    files->emplace_back(full, (size_t)st.st_size);
  }
  closedir(d);
  std::sort(files->begin(), files->end());
  return Status::Ok();
}

Status MakeDirs(const std::string &path) {
  std::string cur;
  std::istringstream parts(path);
  std::string part;
  if (!path.empty() && path[0] == '/') cur = "/";
  while (std::getline(parts, part, '/')) {
    if (part.empty()) continue;
    cur += part + "/";
    if (mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST)
      return Unknown("Cannot create " + cur + ": " + strerror(errno));
  }
  return Status::Ok();
}

uint64_t CeilMiB(uint64_t bytes) { return (bytes + (1u << 20) - 1) >> 20; }

struct DeviceBuffers {
  cuking_ctx *ctx = nullptr;
  void *host_bits = nullptr;
  void *d_bits = nullptr, *d_results = nullptr, *d_counters = nullptr;
  void *pack_host = nullptr, *pack_dev = nullptr;  // --pack=device staging rings
  ~DeviceBuffers() {
    if (ctx == nullptr) return;
    if (pack_host) cuking_host_free(ctx, pack_host);
    if (pack_dev) cuking_device_free(ctx, pack_dev);
    if (host_bits) cuking_host_free(ctx, host_bits);
    if (d_bits) cuking_device_free(ctx, d_bits);
    if (d_results) cuking_device_free(ctx, d_results);
    if (d_counters) cuking_device_free(ctx, d_counters);
    cuking_ctx_destroy(ctx);
  }
};

uint64_t NowMicros() {
  return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(
             std::chrono::steady_clock::now().time_since_epoch()).count();
}

// --decode=stream: hands every decoded batch to the pack of the run (host or device),
// counts the triples and the time spent packing.
struct BatchSink : cuking_host::TripleSink {
  using PackFn = std::function<std::string(const int64_t *, const int64_t *, const int32_t *,
                                           size_t)>;
  PackFn pack_batch;
  size_t triples = 0;
  uint64_t pack_time_us = 0;
  explicit BatchSink(PackFn p) : pack_batch(std::move(p)) {}
  std::string Consume(const int64_t *row, const int64_t *col, const int32_t *alt,
                      size_t count) override {
    const uint64_t t0 = NowMicros();
    std::string err = pack_batch(row, col, alt, count);
    pack_time_us += NowMicros() - t0;
    triples += count;
    return err;
  }
};

// --pack=device: one per reader thread.  A decoded table goes to the GPU in
// pieces of kChunkTriples through a ring of kSlots page-locked buffers: each
// piece is filtered to the shard and narrowed to 8 bytes per genotype on the
// host (cuking_narrow_triples; the reference's own validation, cuking.cu:677-702,
// happens there), copied asynchronously and packed by pack_compact_kernel on
// the thread's own stream.  Before a slot is refilled the thread waits for THAT
// slot's previous piece only (an event), so the host fills piece n + 2 while
// pieces n and n + 1 are on the wire or in the kernel, and the next table is
// being decoded while the tail of this one is still in flight.
// Where the reader threads' device-pack time goes (summed over threads).
struct PackerStats {
  std::atomic<uint64_t> setup_us{0}, wait_us{0}, narrow_us{0}, enqueue_us{0};
};

class DevicePacker {
 public:
  static constexpr size_t kChunkTriples = size_t(512) << 10;  // 4 MiB per slot
  static constexpr int kSlots = 3;
  static constexpr size_t kSlotBytes = kChunkTriples * 8;
  static constexpr size_t kRingBytes = kSlots * kSlotBytes;
  PackerStats *stats = nullptr;

  // `host` / `dev`: this packer's kRingBytes of page-locked and of device
  // memory, carved out of ONE allocation each by the caller: 64 reader threads
  // each pinning their own ring spent 0.65 s apiece inside the driver's locks.
  DevicePacker(cuking_ctx *ctx, void *host, void *dev) : ctx_(ctx), host_(host), dev_(dev) {}
  ~DevicePacker() {
    if (stream_) cuking_stream_synchronize(ctx_, stream_);
    for (int b = 0; b < kSlots; ++b)
      if (event_[b]) cuking_event_destroy(ctx_, event_[b]);
    if (stream_) cuking_stream_destroy(ctx_, stream_);
  }

  // Stream and events (main thread, before the readers start).
  std::string Init() {
    cuking_status st = cuking_stream_create(ctx_, &stream_);
    for (int b = 0; b < kSlots && st == CUKING_OK; ++b)
      st = cuking_event_create(ctx_, &event_[b]);
    return st == CUKING_OK ? "" : cuking_last_error();
  }

  // Returns "" or "<CODE>\n<message>".
  std::string Pack(const cuking_submatrix &sm, uint32_t words_per_sample,
                   uint64_t *d_bits, const int64_t *row_idx, const int64_t *col_idx,
                   const int32_t *n_alt_alleles, size_t n, uint32_t *d_status) {
    auto abi_error = [](cuking_status st) {
      const char *code = st == CUKING_ERR_FAILED_PRECONDITION ? "FAILED_PRECONDITION"
                         : st == CUKING_ERR_INVALID_ARGUMENT  ? "INVALID_ARGUMENT"
                                                              : "INTERNAL";
      return std::string(code) + "\n" + cuking_last_error();
    };
    auto now = []() {
      return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(
                 std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    for (size_t done = 0; done < n; done += kChunkTriples) {
      const size_t m = std::min(kChunkTriples, n - done);
      const int b = next_;
      next_ = (next_ + 1) % kSlots;
      uint64_t t0 = now();
      if (used_[b]) {
        const cuking_status st = cuking_event_synchronize(ctx_, event_[b]);
        if (st != CUKING_OK) return abi_error(st);
      }
      uint64_t t1 = now();
      if (stats) stats->wait_us += t1 - t0;
      // [site x m][sample_alt x m]: one contiguous piece, one copy
      uint32_t *h_site = reinterpret_cast<uint32_t *>(static_cast<char *>(host_) + b * kSlotBytes);
      uint32_t *h_sa = h_site + m;
      size_t kept = 0;
      cuking_status st = cuking_narrow_triples(
          &sm, words_per_sample, row_idx + done, col_idx + done, n_alt_alleles + done, m,
          h_site, h_sa, &kept);
      if (st != CUKING_OK) return abi_error(st);
      t0 = now();
      if (stats) stats->narrow_us += t0 - t1;
      if (kept == 0) continue;
      uint32_t *d_site = reinterpret_cast<uint32_t *>(static_cast<char *>(dev_) + b * kSlotBytes);
      st = cuking_copy_to_device(ctx_, d_site, h_site, (m + kept) * 4, stream_);
      if (st == CUKING_OK)
        st = cuking_pack_device_compact(ctx_, &sm, words_per_sample, d_bits, d_site, d_site + m,
                                        kept, d_status, stream_);
      if (st == CUKING_OK) st = cuking_event_record(ctx_, event_[b], stream_);
      if (st != CUKING_OK) return abi_error(st);
      used_[b] = true;
      if (stats) stats->enqueue_us += now() - t0;
    }
    return "";
  }

  std::string Finish() {
    if (stream_ && cuking_stream_synchronize(ctx_, stream_) != CUKING_OK)
      return cuking_last_error();
    return "";
  }

 private:
  cuking_ctx *ctx_;
  void *stream_ = nullptr;
  void *host_ = nullptr, *dev_ = nullptr;
  void *event_[kSlots] = {nullptr, nullptr, nullptr};
  bool used_[kSlots] = {false, false, false};
  int next_ = 0;
};

// --print_schedule: what each rank of a --num_gpus run would do, as JSON (host
// arithmetic only; the tests compare it with cuking_amd/dist.py and check by
// brute force that every pair is covered exactly once).
void PrintSchedule(const Flags &flags, const cuking_submatrix &sm) {
  const uint32_t world = flags.num_gpus ? flags.num_gpus : 1;
  const uint32_t tile = cuking_tile_samples(nullptr);
  const uint64_t num_tiles = cuking_num_tiles(nullptr, &sm);
  const uint32_t stored = cuking_submatrix_num_samples(&sm);
  const bool diag = sm.i_begin == sm.j_begin;
  const bool staged = diag && flags.kernel != "stream" && flags.multi_gpu_mode != "simple";
  std::cout << "{\"world\": " << world << ", \"tile\": " << tile << ", \"num_tiles\": "
            << num_tiles << ", \"stored_samples\": " << stored << ", \"mode\": \""
            << (staged ? "staged" : "simple") << "\", \"tile_ranges\": [";
  // (with --rank_weights the ranges the simple schedule would use; otherwise the
  //  equal ranges it falls back to when it does not calibrate)
  const auto parts = flags.rank_weight_values.size() == world
                         ? cuking_host::WeightedTilePartition(num_tiles, flags.rank_weight_values)
                         : cuking_host::TilePartition(num_tiles, world);
  for (uint32_t r = 0; r < world; ++r)
    std::cout << (r ? ", " : "") << "[" << parts[r].begin << ", " << parts[r].end << "]";
  const uint64_t cal =
      !flags.calibrate || !flags.rank_weight_values.empty() || world < 2 ? 0
      : flags.calibration_tiles != 0 && flags.calibration_tiles * world <= num_tiles
          ? flags.calibration_tiles
          : cuking_host::CalibrationTiles(num_tiles, world);
  std::cout << "], \"weighted\": " << (flags.rank_weight_values.size() == world ? "true" : "false")
            << ", \"calibration_tiles\": " << cal << ", \"chunks\": [";
  const auto chunks = cuking_host::ChunkRanges(stored, tile, flags.bcast_chunks);
  for (size_t c = 0; c < chunks.size(); ++c)
    std::cout << (c ? ", " : "") << "[" << chunks[c].begin << ", " << chunks[c].end << "]";
  std::cout << "], \"staged\": [";
  const cuking_host::RowDeal deal = cuking_host::MakeRowDeal(
      world, staged ? flags.rank_weight_values : std::vector<double>());
  for (uint32_t r = 0; r < world; ++r) {
    std::cout << (r ? ", " : "") << "[";
    const auto steps =
        cuking_host::StagedSchedule(stored, tile, world, r, flags.bcast_chunks, &deal);
    for (size_t k = 0; k < steps.size(); ++k) {
      const auto &s = steps[k];
      std::cout << (k ? ", " : "") << "{\"chunk\": [" << s.chunk.begin << ", " << s.chunk.end
                << "], \"rows\": ";
      if (s.has_rect)
        std::cout << "[" << s.row_begin << ", " << s.row_end << ", " << s.row_step << "]";
      else
        std::cout << "null";
      std::cout << ", \"rects\": [";
      for (size_t q = 0; q < s.rects.size(); ++q)
        std::cout << (q ? ", " : "") << "[" << s.rects[q].row_begin << ", " << s.rects[q].row_end
                  << ", " << s.rects[q].row_step << "]";
      std::cout << "]}";
    }
    std::cout << "]";
  }
  std::cout << "], \"row_deal_period\": " << deal.period << ", \"row_deal\": [";
  for (uint32_t r = 0; r < world; ++r) {
    std::cout << (r ? ", " : "") << "[";
    for (size_t q = 0; q < deal.offsets[r].size(); ++q)
      std::cout << (q ? ", " : "") << deal.offsets[r][q];
    std::cout << "]";
  }
  std::cout << "]}" << std::endl;
}

// One run of the binary, phase by phase (cuking.cu:435-882 is one function; here the
// phases are functions over this state).
struct Job {
  explicit Job(const Flags &f) : flags(f) {}
  const Flags &flags;
  bool synthetic = false, multi_gpu = false, dump_only = false;
  std::string input_dir, output_dir;
  cuking_host::Metadata metadata;
  uint32_t num_samples = 0, words_per_sample = 0;
  cuking_submatrix sm = {0, 0, 0, 0};  // cuking.cu:505
  size_t bit_set_words = 0, bit_set_bytes = 0;
  // Input: the tables, the decode tasks ({file, row group or -1 = the whole file}) and
  // where the triples are packed.
  std::vector<std::pair<std::string, size_t>> input_files;
  std::vector<std::pair<size_t, int>> tasks;
  std::string pack_mode;
  bool pack_on_device = false;  // the bitset is built in device memory (device pack, synthetic)
  bool device_pack = false;     // ... from triples, by the GPU
  DeviceBuffers buf;
  std::vector<uint64_t> dump_bits;  // --dump_bitset: plain host memory, no GPU
  uint64_t *host_bits = nullptr;
  // Measurements for the summary line.
  std::atomic<size_t> num_triples{0};
  std::atomic<uint64_t> decode_us{0}, pack_us{0};  // summed over reader threads
  PackerStats packer_stats;
  double read_pack_seconds = 0, kernel_seconds = 0;
  std::ostringstream multi_summary;
  std::vector<cuking_result> results;
  StopWatch sw;
};

// Metadata, the shard's Submatrix and the sizes that follow from them (cuking.cu:475-515).
Status DescribeInput(Job *job) {
  const Flags &flags = job->flags;
  {
    const std::string err = cuking_host::ValidateFlags(flags);
    if (!err.empty()) return InvalidArgument(err);
  }
  job->synthetic = !flags.synthetic.empty();
  if (!job->synthetic) RETURN_IF_ERROR(ResolveUri(flags.input_uri, &job->input_dir));
  RETURN_IF_ERROR(ResolveUri(flags.output_uri, &job->output_dir));

  std::cout << "Reading metadata..." << std::flush;
  cuking_host::Metadata &metadata = job->metadata;
  if (job->synthetic) {
    metadata.num_sites = flags.synth_sites;
    metadata.samples.reserve(flags.synth_samples);
    char name[16];
    for (uint32_t k = 0; k < flags.synth_samples; ++k) {
      snprintf(name, sizeof(name), "S%07u", k);
      metadata.samples.emplace_back(name);
    }
  } else {
    const std::string err =
        cuking_host::ReadMetadataFile(job->input_dir + "/metadata.json", &metadata);
    if (!err.empty()) return FailedPrecondition(err);
  }
  if (metadata.samples.size() > 0xFFFFFFFFull)
    return FailedPrecondition("too many samples");
  job->num_samples = (uint32_t)metadata.samples.size();
  job->words_per_sample = cuking_words_per_sample(metadata.num_sites);
  Done(&job->sw);

  RETURN_IF_ERROR(FromAbi(cuking_submatrix_init(&job->sm, job->num_samples, flags.split_factor,
                                                flags.shard_index)));
  job->multi_gpu = flags.num_gpus > 0;
  job->bit_set_words = (size_t)job->words_per_sample * cuking_submatrix_num_samples(&job->sm);
  job->bit_set_bytes = job->bit_set_words * sizeof(uint64_t);
  job->dump_only = !flags.dump_bitset.empty();
  return Status::Ok();
}

// The input tables (cuking.cu:529-545), where their triples are packed, and the decode
// tasks: one per (file, row group) when the files are fewer than the reader threads -- the
// reference hands out whole files (cuking.cu:550-553), which caps the decode at one thread
// per file; a table written with several row groups decodes on several.
Status PlanInput(Job *job) {
  const Flags &flags = job->flags;
  if (job->dump_only && job->synthetic)
    return InvalidArgument("--dump_bitset needs input tables, not --synthetic");
  // (a synthetic cohort is generated on the GPU: from here on it is a bitset
  //  that already sits in device memory, like a device-packed one)
  job->pack_mode = flags.pack;
  if (!job->synthetic) {  // (listed before anything is allocated: the pack mode depends on it)
    std::cout << "Listing input files..." << std::flush;
    RETURN_IF_ERROR(ListParquetFiles(job->input_dir, &job->input_files));
    Done(&job->sw);
    if (job->input_files.empty()) return FailedPrecondition("No input files found");  // :542-544
    std::cout << "Found " << job->input_files.size() << " input files." << std::endl;
  }
  if (job->pack_mode == "auto") {
    // The device pack pays ~0.1 s of set-up (page-locked rings, streams) and wins while
    // the GPU's atomics keep up with the readers.  Measured on MI355X boxes
    // (profiles/r03_pack_pipeline.txt): with up to ~32 reader threads per GPU and an input
    // of a gigabyte or more the pipelined device pack is ahead (1e9 triples: +37 %); below
    // that its set-up is not amortised (1e8 triples: 0.27 s against 0.20 s), and beyond ~32
    // threads the host's cores outrun one GPU's atomic units.
    // (the reader threads that can run at once: the flag's default is the reference's 36,
    //  cuking.cu:36, whatever the box -- a GPU's share of a node is 16 hardware threads)
    size_t input_bytes = 0;
    for (const auto &f : job->input_files) input_bytes += f.second;
    const size_t hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t running = std::min(flags.num_reader_threads, hw);
    job->pack_mode = (running <= 32 && input_bytes >= (size_t(1) << 30)) ? "device" : "host";
  }
  job->pack_on_device = (job->pack_mode == "device" || job->synthetic) && !job->dump_only;
  job->device_pack = job->pack_on_device && !job->synthetic;
  if (!job->synthetic) {
    const auto &files = job->input_files;
    if (files.size() >= flags.num_reader_threads) {
      for (size_t f = 0; f < files.size(); ++f) job->tasks.emplace_back(f, -1);
    } else {
      std::vector<int> groups(files.size(), 1);
      const std::string err = cuking_host::ParallelFor(
          flags.num_reader_threads, 0, files.size(), [&](size_t f) -> std::string {
            return cuking_host::CountRowGroups(files[f].first, &groups[f]);
          });
      if (!err.empty()) return FailedPrecondition(err);
      for (size_t f = 0; f < files.size(); ++f) {
        if (groups[f] <= 1) job->tasks.emplace_back(f, -1);
        else for (int g = 0; g < groups[f]; ++g) job->tasks.emplace_back(f, g);
      }
    }
  }
  return Status::Ok();
}

// The bitset, everything missing (cuking.cu:516-523): device memory for the device pack
// and the synthetic cohort, a page-locked staging buffer for the host pack.
Status AllocateBitSet(Job *job) {
  const Flags &flags = job->flags;
  DeviceBuffers &buf = job->buf;
  std::cout << "Allocating " << CeilMiB(job->bit_set_bytes)
            << " MiB of memory for bit set..." << std::flush;
  if (job->dump_only) {
    job->dump_bits.assign(job->bit_set_words, ~0ull);
  } else {
    RETURN_IF_ERROR(FromAbi(cuking_ctx_create(flags.device, &buf.ctx)));
    RETURN_IF_ERROR(FromAbi(cuking_ctx_set_kernel(
        buf.ctx,
        flags.kernel == "stream" ? CUKING_KERNEL_STREAM : CUKING_KERNEL_TILED)));
    // (with --num_gpus and host pack every rank allocates its own copy later)
    if (!job->multi_gpu || job->pack_on_device)
      RETURN_IF_ERROR(FromAbi(cuking_device_alloc(buf.ctx, job->bit_set_bytes, &buf.d_bits)));
    if (job->synthetic) {
      // every word is written by the generator
    } else if (job->pack_on_device) {
      RETURN_IF_ERROR(FromAbi(
          cuking_memset_async(buf.ctx, buf.d_bits, 0xFF, job->bit_set_bytes, nullptr)));
    } else {
      RETURN_IF_ERROR(FromAbi(cuking_host_alloc(buf.ctx, job->bit_set_bytes, &buf.host_bits)));
      if (job->bit_set_bytes) memset(buf.host_bits, 0xFF, job->bit_set_bytes);
    }
  }
  job->host_bits =
      job->dump_only ? job->dump_bits.data() : static_cast<uint64_t *>(buf.host_bits);
  Done(&job->sw);
  return Status::Ok();
}

// --synthetic: founders + planted relatives (synth_plan.h), genotypes from the device
// generator, straight into the reference layout: rows of the shard first, then its
// columns (cuking.cu:171-175).
Status SynthesiseCohort(Job *job) {
  const Flags &flags = job->flags;
  DeviceBuffers &buf = job->buf;
  const cuking_submatrix &sm = job->sm;
  const uint32_t num_samples = job->num_samples, words_per_sample = job->words_per_sample;
  const cuking_host::CohortPlan plan = cuking_host::PlanCohort(num_samples, flags.synth_seed);
  void *d_plan = nullptr;
  const size_t plan_bytes = (size_t)num_samples * sizeof(uint32_t);
  RETURN_IF_ERROR(FromAbi(cuking_device_alloc(buf.ctx, 3 * plan_bytes, &d_plan)));
  uint32_t *d_kind = static_cast<uint32_t *>(d_plan);
  uint32_t *d_pa = d_kind + num_samples, *d_pb = d_pa + num_samples;
  cuking_status st = cuking_copy_to_device(buf.ctx, d_kind, plan.kind.data(), plan_bytes, nullptr);
  if (st == CUKING_OK) st = cuking_copy_to_device(buf.ctx, d_pa, plan.pa.data(), plan_bytes, nullptr);
  if (st == CUKING_OK) st = cuking_copy_to_device(buf.ctx, d_pb, plan.pb.data(), plan_bytes, nullptr);
  uint64_t *d_bits = static_cast<uint64_t *>(buf.d_bits);
  if (st == CUKING_OK)
    st = cuking_synth_bitset(buf.ctx, flags.synth_seed, d_kind, d_pa, d_pb, sm.i_begin, sm.i_end,
                             job->metadata.num_sites, words_per_sample, d_bits, nullptr);
  if (st == CUKING_OK && sm.i_begin != sm.j_begin)
    st = cuking_synth_bitset(buf.ctx, flags.synth_seed, d_kind, d_pa, d_pb, sm.j_begin, sm.j_end,
                             job->metadata.num_sites, words_per_sample,
                             d_bits + (size_t)(sm.i_end - sm.i_begin) * words_per_sample, nullptr);
  if (st == CUKING_OK) st = cuking_stream_synchronize(buf.ctx, nullptr);
  cuking_device_free(buf.ctx, d_plan);
  return FromAbi(st);
}

// Decode + pack on the reader threads (cuking.cu:547-711): every task's triples into the
// bitset, by the host's relaxed atomics or through a reader thread's device packer; then
// the bitset is where the kernel (or the broadcast) expects it.
Status DecodeAndPack(Job *job) {
  const Flags &flags = job->flags;
  DeviceBuffers &buf = job->buf;
  const cuking_submatrix &sm = job->sm;
  const uint32_t words_per_sample = job->words_per_sample;
  const bool device_pack = job->device_pack;
  uint64_t *const host_bits = job->host_bits;
  const auto &tasks = job->tasks;
  const auto &input_files = job->input_files;
  std::atomic<size_t> num_processed(0);
  auto now_us = []() { return NowMicros(); };
  PackerStats &packer_stats = job->packer_stats;
  std::vector<std::unique_ptr<DevicePacker>> packers;
  uint32_t *d_pack_status = nullptr;
  std::atomic<size_t> next_packer(0);
  if (device_pack) {
    RETURN_IF_ERROR(FromAbi(cuking_device_alloc(buf.ctx, sizeof(uint32_t),
                                                reinterpret_cast<void **>(&d_pack_status))));
    RETURN_IF_ERROR(FromAbi(
        cuking_memset_async(buf.ctx, d_pack_status, 0, sizeof(uint32_t), nullptr)));
  }
  // One staging ring per reader thread that can be busy at once, carved out of
  // one page-locked and one device allocation -- made on a helper thread while
  // the readers already decode their first tables (page-locking 100-200 MB
  // takes about as long as decoding one table); a reader waits for it before
  // its first piece.
  std::string setup_error;
  std::thread setup_thread;
  std::shared_future<void> setup_done;
  if (device_pack) {
    std::promise<void> promise;
    setup_done = promise.get_future().share();
    setup_thread = std::thread([&, promise = std::move(promise)]() mutable {
      const size_t rings = std::min(flags.num_reader_threads, tasks.size());
      const uint64_t setup_begin = now_us();
      auto check = [&](cuking_status st) {
        if (st != CUKING_OK && setup_error.empty()) setup_error = cuking_last_error();
        return st == CUKING_OK;
      };
      if (check(cuking_host_alloc(buf.ctx, rings * DevicePacker::kRingBytes, &buf.pack_host)) &&
          check(cuking_device_alloc(buf.ctx, rings * DevicePacker::kRingBytes, &buf.pack_dev))) {
        for (size_t r = 0; r < rings && setup_error.empty(); ++r) {
          packers.emplace_back(new DevicePacker(
              buf.ctx, static_cast<char *>(buf.pack_host) + r * DevicePacker::kRingBytes,
              static_cast<char *>(buf.pack_dev) + r * DevicePacker::kRingBytes));
          packers.back()->stats = &packer_stats;
          setup_error = packers.back()->Init();
        }
      }
      // The packers' streams are non-blocking, i.e. NOT ordered behind the null
      // stream: the all-ones fill of the bitset (tens of ms at cohort scale) and
      // the zeroed status word must be complete before the first pack kernel.
      check(cuking_stream_synchronize(buf.ctx, nullptr));
      packer_stats.setup_us += now_us() - setup_begin;
      promise.set_value();
    });
  }
  struct Joiner {  // (every exit path below joins the helper)
    std::thread &t;
    ~Joiner() {
      if (t.joinable()) t.join();
    }
  } joiner{setup_thread};
  // --decode=auto: batches of triples packed as they are decoded, for both packs (the
  // streaming form measured ahead of whole tables on a GPU box: profiles/r04_pack_pipeline.txt).
  const bool stream_decode = flags.decode != "table";
  const std::string pack_error = cuking_host::ParallelFor(
      flags.num_reader_threads, 0, tasks.size(), [&](size_t task) -> std::string {
        const size_t f = tasks[task].first;
        const std::string &path = input_files[f].first;
        // One batch of triples into the bitset: the host's, or through this reader
        // thread's packer (stream + staging ring) the device's.
        auto pack = [&](const int64_t *row, const int64_t *col, const int32_t *alt,
                        size_t count) -> std::string {
          if (!device_pack) {
            if (cuking_pack_host(&sm, words_per_sample, host_bits, row, col, alt, count) !=
                CUKING_OK)
              return std::string("FAILED_PRECONDITION\n") + cuking_last_error() + " in " + path;
            return "";
          }
          if (count == 0) return "";
          thread_local DevicePacker *packer = nullptr;
          if (packer == nullptr) {
            setup_done.wait();
            if (!setup_error.empty()) return "INTERNAL\n" + setup_error;
            packer = packers[next_packer.fetch_add(1) % packers.size()].get();
          }
          const std::string msg = packer->Pack(sm, words_per_sample,
                                               static_cast<uint64_t *>(buf.d_bits), row, col,
                                               alt, count, d_pack_status);
          return msg.empty() ? msg : msg + " in " + path;
        };
        const uint64_t t_begin = now_us();
        size_t n = 0;
        if (stream_decode) {
          // Batches small enough to stay in the cache for the host pack, one staging
          // slot's worth for the device pack.
          BatchSink sink(pack);
          thread_local cuking_host::TripleScratch scratch;
          const size_t batch = flags.decode_batch != 0 ? flags.decode_batch
                               : device_pack         ? DevicePacker::kChunkTriples
                                                     : size_t(32) << 10;
          std::string err = cuking_host::StreamTriples(path, tasks[task].second, batch, &scratch,
                                                       &sink);
          if (!err.empty())
            return err.rfind("FAILED_PRECONDITION\n", 0) == 0 || err.rfind("INTERNAL\n", 0) == 0 ||
                           err.rfind("INVALID_ARGUMENT\n", 0) == 0
                       ? err
                       : "FAILED_PRECONDITION\n" + err;
          n = sink.triples;
          const uint64_t total = now_us() - t_begin;
          job->pack_us += sink.pack_time_us;
          job->decode_us += total > sink.pack_time_us ? total - sink.pack_time_us : 0;
        } else {
          cuking_host::Triples t;
          std::string err = cuking_host::ReadTriples(path, tasks[task].second, &t);
          if (!err.empty()) return "FAILED_PRECONDITION\n" + err;
          n = t.row_idx.size();
          const uint64_t t_decoded = now_us();
          job->decode_us += t_decoded - t_begin;
          err = pack(t.row_idx.data(), t.col_idx.data(), t.n_alt_alleles.data(), n);
          if (!err.empty()) return err;
          job->pack_us += now_us() - t_decoded;
        }
        job->num_triples += n;
        if ((++num_processed & 1023) == 0) std::cout << "." << std::flush;  // :705-708
        return "";
      });
  // The helper thread owns `packers` and `setup_error` until its promise is
  // set; a run whose tables are all empty never waited for it inside a reader.
  if (device_pack) setup_done.wait();
  if (!pack_error.empty()) {
    const size_t nl = pack_error.find('\n');
    return {pack_error.substr(0, nl), pack_error.substr(nl + 1)};
  }
  if (device_pack && !setup_error.empty()) return {"INTERNAL", setup_error};
  if (job->dump_only) return Status::Ok();
  if (device_pack) {
    for (auto &p : packers) {
      const std::string msg = p->Finish();
      if (!msg.empty()) return {"INTERNAL", msg};
    }
    packers.clear();
    uint32_t pack_status = 0;
    RETURN_IF_ERROR(FromAbi(cuking_copy_to_host(buf.ctx, &pack_status, d_pack_status,
                                                sizeof(uint32_t), nullptr)));
    RETURN_IF_ERROR(FromAbi(cuking_stream_synchronize(buf.ctx, nullptr)));
    cuking_device_free(buf.ctx, d_pack_status);
    if (pack_status & 1u)
      return FailedPrecondition("Invalid value for n_alt_alleles encountered");
    if (pack_status & 2u)
      return InvalidArgument("row_idx outside the padded sites encountered");
  } else if (!job->multi_gpu && !job->synthetic) {
    RETURN_IF_ERROR(FromAbi(cuking_copy_to_device(buf.ctx, buf.d_bits, buf.host_bits,
                                                  job->bit_set_bytes, nullptr)));
    RETURN_IF_ERROR(FromAbi(cuking_stream_synchronize(buf.ctx, nullptr)));
    cuking_host_free(buf.ctx, buf.host_bits);
    buf.host_bits = nullptr;
  }
  return Status::Ok();
}

// --dump_bitset: the packed host bitset as raw little-endian u64, no GPU.
Status DumpBitSet(Job *job) {
  const Flags &flags = job->flags;
  Done(&job->sw);
  FILE *f = fopen(flags.dump_bitset.c_str(), "wb");
  if (f == nullptr) return Unknown("Cannot write " + flags.dump_bitset);
  const size_t wrote = fwrite(job->dump_bits.data(), sizeof(uint64_t), job->bit_set_words, f);
  fclose(f);
  if (wrote != job->bit_set_words) return Unknown("Short write to " + flags.dump_bitset);
  std::cout << "Dumped " << job->bit_set_words << " words (" << job->num_triples.load()
            << " triples, " << job->tasks.size() << " decode tasks) to " << flags.dump_bitset
            << std::endl;
  return Status::Ok();
}

// The shard over flags.num_gpus GPUs: chunked RCCL broadcast of the packed bitset,
// pair-space shares per rank, records gathered on rank 0 (multi_gpu.h).
Status ComputeOnSeveralGpus(Job *job) {
  const Flags &flags = job->flags;
  DeviceBuffers &buf = job->buf;
  std::cout << "Running KING HIP kernel for " << cuking_submatrix_num_rows(&job->sm) << " x "
            << cuking_submatrix_num_cols(&job->sm) << " matrix on " << flags.num_gpus
            << " GPU(s)..." << std::flush;
  cuking_host::MultiGpuInput in;
  in.num_gpus = (int)flags.num_gpus;
  in.first_device = flags.device;
  in.kernel = flags.kernel;
  in.mode = flags.multi_gpu_mode;
  in.chunks = flags.bcast_chunks;
  in.sm = job->sm;
  in.words_per_sample = job->words_per_sample;
  in.host_bits = job->pack_on_device ? nullptr : static_cast<const uint64_t *>(buf.host_bits);
  in.d_bits_rank0 = static_cast<uint64_t *>(buf.d_bits);
  in.kin_threshold = flags.kin_threshold;
  in.max_results = flags.max_results;
  in.collectives = flags.collectives;
  in.rank_weights = flags.rank_weight_values;
  in.calibrate = flags.calibrate;
  in.calibration_tiles = flags.calibration_tiles;
  in.inject_failure_rank = flags.inject_failure_rank;
  in.inject_failure_phase = flags.inject_failure_phase;
  in.phase_timeout_seconds = flags.phase_timeout_seconds;
  cuking_host::MultiGpuOutput mg;
  std::string code;
  const std::string err = cuking_host::RunMultiGpu(in, &mg, &code);
  if (!err.empty()) return {code, err};
  // kernel_seconds = exchange + compute + gather; the communicator set-up
  // (seconds in a cold process) and the per-GPU contexts are reported apart
  const double wall = job->sw.ElapsedAndReset();
  job->kernel_seconds = mg.exchange_and_compute_seconds + mg.gather_seconds;
  std::cout << " (" << std::fixed << std::setprecision(3) << job->kernel_seconds
            << "s; with RCCL set-up " << wall << "s)" << std::endl;
  job->results.swap(mg.results);
  std::ostringstream &out = job->multi_summary;
  auto list = [&](const char *key, const auto &values, int precision) {
    out << ", \"" << key << "\": [";
    for (size_t r = 0; r < values.size(); ++r)
      out << (r ? ", " : "") << std::setprecision(precision) << values[r];
    out << "]";
  };
  out << ", \"gpus\": " << flags.num_gpus << ", \"multi_gpu_wall_seconds\": "
      << std::setprecision(3) << wall << ", \"comm_init_seconds\": " << mg.comm_init_seconds
      << ", \"multi_gpu_mode\": \"" << mg.mode << "\", \"bytes_broadcast\": "
      << mg.bytes_broadcast << ", \"exchange_and_compute_seconds\": " << std::setprecision(6)
      << mg.exchange_and_compute_seconds << ", \"gather_seconds\": " << mg.gather_seconds;
  list("rank_kernel_ms", mg.rank_kernel_ms, 3);
  list("rank_results", mg.rank_results, 6);
  out << ", \"collectives\": \"" << mg.collectives << "\"";
  list("allocations_after_reserve", mg.rank_allocations_after_reserve, 6);
  list("host_syncs_after_reserve", mg.rank_host_syncs_after_reserve, 6);
  out << ", \"calibration_tiles\": " << mg.calibration_tiles;
  list("rank_rates_tiles_per_ms", mg.rank_rates, 4);
  out << ", \"rank_tile_ranges\": [";
  for (size_t r = 0; r < mg.rank_tile_ranges.size(); ++r)
    out << (r ? ", " : "") << "[" << mg.rank_tile_ranges[r].first << ", "
        << mg.rank_tile_ranges[r].second << "]";
  out << "]";
  if (buf.host_bits) {
    cuking_host_free(buf.ctx, buf.host_bits);
    buf.host_bits = nullptr;
  }
  std::cout << "Processing " << job->results.size() << " results..." << std::flush;
  return Status::Ok();
}

// The reference's path (cuking.cu:713-756): one kernel call through the C ABI, the
// overflow check, the records back on the host.
Status ComputeOnOneGpu(Job *job) {
  const Flags &flags = job->flags;
  DeviceBuffers &buf = job->buf;
  const uint32_t max_results = flags.max_results;
  std::cout << "Allocating " << CeilMiB((uint64_t)max_results * sizeof(cuking_result))
            << " MiB of memory for results..." << std::flush;
  RETURN_IF_ERROR(FromAbi(cuking_device_alloc(
      buf.ctx, (size_t)max_results * sizeof(cuking_result), &buf.d_results)));
  RETURN_IF_ERROR(
      FromAbi(cuking_device_alloc(buf.ctx, 2 * sizeof(uint32_t), &buf.d_counters)));
  RETURN_IF_ERROR(FromAbi(
      cuking_memset_async(buf.ctx, buf.d_counters, 0, 2 * sizeof(uint32_t), nullptr)));
  Done(&job->sw);

  std::cout << "Running KING HIP kernel for " << cuking_submatrix_num_rows(&job->sm) << " x "
            << cuking_submatrix_num_cols(&job->sm) << " matrix..." << std::flush;
  uint32_t *d_counters = static_cast<uint32_t *>(buf.d_counters);
  RETURN_IF_ERROR(FromAbi(cuking_compute_king(
      buf.ctx, &job->sm, job->words_per_sample, static_cast<uint64_t *>(buf.d_bits),
      flags.kin_threshold, max_results, static_cast<cuking_result *>(buf.d_results),
      d_counters, d_counters + 1, nullptr)));
  uint32_t counters[2] = {0, 0};
  RETURN_IF_ERROR(FromAbi(cuking_copy_to_host(buf.ctx, counters, d_counters,
                                              sizeof(counters), nullptr)));
  RETURN_IF_ERROR(FromAbi(cuking_stream_synchronize(buf.ctx, nullptr)));  // errors surface here
  job->kernel_seconds = job->sw.ElapsedAndReset();
  std::cout << " (" << std::fixed << std::setprecision(3) << job->kernel_seconds << "s)"
            << std::endl;

  if (counters[1] != 0)  // cuking.cu:747-751
    return ResourceExhausted(
        "Could not store all results: try increasing the --max_results parameter.");

  std::cout << "Processing " << counters[0] << " results..." << std::flush;
  job->results.resize(counters[0]);
  RETURN_IF_ERROR(FromAbi(cuking_copy_to_host(buf.ctx, job->results.data(), buf.d_results,
                                              job->results.size() * sizeof(cuking_result),
                                              nullptr)));
  return FromAbi(cuking_stream_synchronize(buf.ctx, nullptr));
}

// Sort, part-<shard>.snappy.parquet (cuking.cu:757-879), and the machine-readable summary
// (extends the reference's phase log).
Status WriteOutput(Job *job) {
  const Flags &flags = job->flags;
  DeviceBuffers &buf = job->buf;
  std::vector<cuking_result> &results = job->results;
  const size_t num_results = results.size();
  // Free device memory before post-processing (cuking.cu:757-758).
  if (buf.d_bits) {
    cuking_device_free(buf.ctx, buf.d_bits);
    buf.d_bits = nullptr;
  }
  cuking_sort_results(results.data(), results.size());  // :761-765

  RETURN_IF_ERROR(MakeDirs(job->output_dir));
  std::ostringstream name;  // :868-870
  name << job->output_dir << "/part-" << std::setw(5) << std::setfill('0') << flags.shard_index
       << ".snappy.parquet";
  uint64_t bytes_written = 0;
  {
    const std::string err = cuking_host::WriteResults(
        name.str(), results.data(), results.size(), job->metadata.samples, &bytes_written);
    if (!err.empty()) return Unknown(err);
  }
  Done(&job->sw);
  std::cout << "Wrote " << CeilMiB(bytes_written) << " MiB." << std::endl;

  const uint64_t pairs = cuking_submatrix_num_pairs(&job->sm);
  const double kernel_seconds = job->kernel_seconds, read_pack_seconds = job->read_pack_seconds;
  const double rate = kernel_seconds > 0 ? pairs / kernel_seconds : 0;
  const size_t num_triples = job->num_triples.load();
  const PackerStats &packer_stats = job->packer_stats;
  std::cout << "{\"pairs\": " << pairs << ", \"triples\": " << num_triples
            << ", \"results\": " << num_results << ", \"decode_thread_seconds\": "
            << std::setprecision(3) << job->decode_us.load() * 1e-6
            << ", \"pack_thread_seconds\": " << job->pack_us.load() * 1e-6
            << ", \"device_pack_thread_seconds\": {\"setup\": "
            << packer_stats.setup_us.load() * 1e-6 << ", \"slot_wait\": "
            << packer_stats.wait_us.load() * 1e-6 << ", \"narrow\": "
            << packer_stats.narrow_us.load() * 1e-6 << ", \"enqueue\": "
            << packer_stats.enqueue_us.load() * 1e-6 << "}"
            << ", \"pack\": \"" << (job->synthetic ? "synthetic" : job->pack_mode)
            << "\", \"decode\": \"" << (flags.decode != "table" ? "stream" : "table")
            << "\", \"decode_tasks\": " << job->tasks.size() << ", \"reader_threads\": "
            << flags.num_reader_threads << ", \"read_pack_seconds\": " << read_pack_seconds
            << ", \"triples_per_second\": " << std::setprecision(1)
            << (read_pack_seconds > 0 ? num_triples / read_pack_seconds : 0.0)
            << std::setprecision(3)
            << ", \"kernel_seconds\": "
            << std::setprecision(6) << kernel_seconds << ", \"pairs_per_second\": "
            << std::setprecision(1) << rate << ", \"algorithmic_GBps\": "
            << rate * cuking_bytes_per_pair(job->words_per_sample) / 1e9
            << ", \"hbm_roofline_fraction\": " << std::setprecision(3)
            << rate * cuking_bytes_per_pair(job->words_per_sample) / 8e12
            // what the five-product matrix-core form (king_mfma.hip, rounds 1-2: five
            // plane products per pair and site, 2 FLOP each) would have to sustain for
            // this rate; the default kernel (king_filter.hip) issues ONE product for
            // every pair and the exact sums for the pairs its bound admits, so this
            // may exceed the 10 PF the matrix cores have
            << ", \"five_product_equivalent_PFLOPs\": "
            << rate * 10.0 * job->metadata.num_sites / 1e15
            << job->multi_summary.str() << "}" << std::endl;
  return Status::Ok();
}

// cuking.cu:435-882, phase by phase: metadata -> Submatrix -> all-ones bitset -> parallel
// Parquet decode + pack -> kernel -> overflow check -> sort -> Snappy Parquet.
Status Run(const Flags &flags) {
  Job job(flags);
  RETURN_IF_ERROR(DescribeInput(&job));
  if (flags.print_schedule) {  // host arithmetic only: no GPU, no input tables
    PrintSchedule(flags, job.sm);
    return Status::Ok();
  }
  RETURN_IF_ERROR(PlanInput(&job));
  RETURN_IF_ERROR(AllocateBitSet(&job));
  std::cout << (job.synthetic ? "Synthesising genotypes on the GPU..."
                              : "Processing Parquet tables...")
            << std::flush;
  if (job.synthetic) RETURN_IF_ERROR(SynthesiseCohort(&job));
  RETURN_IF_ERROR(DecodeAndPack(&job));
  if (job.dump_only) return DumpBitSet(&job);
  // (list -> decode -> pack -> bitset on the GPU; with --num_gpus and host pack
  //  the upload belongs to the broadcast that follows)
  job.read_pack_seconds = job.sw.ElapsedAndReset();
  std::cout << " (" << std::fixed << std::setprecision(3) << job.read_pack_seconds << "s)"
            << std::endl;
  RETURN_IF_ERROR(job.multi_gpu ? ComputeOnSeveralGpus(&job) : ComputeOnOneGpu(&job));
  return WriteOutput(&job);
}

}  // namespace

int main(int argc, char **argv) {
  Flags flags;
  const std::string parse_error = cuking_host::ParseFlags(argc, argv, &flags);
  if (!parse_error.empty()) {
    std::cerr << "ERROR: " << parse_error << std::endl;
    return 1;
  }
  if (flags.help) {
    std::cout << cuking_host::Usage();
    return 0;
  }
  if (flags.variant >= 0) {
    // The library reads its default variant from the environment wherever it needs one --
    // contexts of every rank, and the tile geometry the schedules are cut from before any
    // context exists: one setting for all of them.
    if (flags.variant >= cuking_num_variants()) {
      std::cerr << "ERROR: Illegal value '" << flags.variant << "' specified for flag 'variant'"
                << std::endl;
      return 1;
    }
    setenv("CUKING_AMD_VARIANT", std::to_string(flags.variant).c_str(), 1);
  }
  const Status status = Run(flags);
  if (!status.ok()) {  // cuking.cu:889-892
    std::cerr << std::endl
              << "Error: " << status.code << ": " << status.message << std::endl;
    return 1;
  }
  return 0;
}
