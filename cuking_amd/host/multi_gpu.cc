// See multi_gpu.h.  Host code only: HIP runtime API for streams/events/copies,
// collectives.h (RCCL) for the exchange steps, the C ABI (include/cuking_amd.h)
// for every kernel.
#include "multi_gpu.h"

#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>

#include <unistd.h>

#include <atomic>
#include <barrier>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

#include "collectives.h"
#include "schedule.h"

namespace cuking_host {

namespace {

double Now() {
  return std::chrono::duration<double>(
             std::chrono::steady_clock::now().time_since_epoch()).count();
}

// What the rank threads share.
struct Shared {
  const MultiGpuInput *in;
  MultiGpuOutput *out;
  Collectives *coll = nullptr;
  bool staged = false;
  uint32_t tile = 0;
  uint64_t num_tiles = 0;
  size_t bit_set_bytes = 0;
  std::atomic<bool> failed{false};
  // The failure flag as it stood when the LAST rank arrived at the most recent
  // phase barrier: written by the barrier's completion step, read by every rank
  // after it -- one value for all of them.  (Reading `failed` itself after the
  // barrier is a race: a rank that has passed may fail and set it before a
  // slower peer has looked, and the two then disagree on whether to go on.)
  bool agreed_failed = false;
  std::mutex mu;
  std::string error, code;
  // The phase every rank is in and since when (watchdog below).
  struct Watch {
    std::atomic<const char *> phase{nullptr};  // nullptr: not started / through
    std::atomic<double> since{0};
  };
  std::unique_ptr<Watch[]> watch;

  void Enter(int rank, const char *phase) {
    watch[rank].since.store(Now());
    watch[rank].phase.store(phase);
  }

  void Fail(const std::string &c, const std::string &msg) {
    std::lock_guard<std::mutex> lock(mu);
    if (error.empty()) {
      code = c;
      error = msg;
    }
    failed.store(true);
  }
};

// Wall-clock watchdog of a --num_gpus run.  The ranks block in places nobody can
// cancel from outside -- inside an RCCL call that waits for a peer, at a phase
// barrier, in a wait for a device whose collective kernel never ends -- so a rank
// that stays in ONE phase longer than the limit ends the process: the reference's
// error line (cuking.cu:889-892) with every rank's phase, exit code 1.
class Watchdog {
 public:
  Watchdog(Shared *sh, int world, double limit) : sh_(sh), world_(world), limit_(limit) {
    if (limit_ > 0) thread_ = std::thread([this] { Loop(); });
  }
  ~Watchdog() {
    {
      std::lock_guard<std::mutex> lock(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    if (thread_.joinable()) thread_.join();
  }

 private:
  void Loop() {
    std::unique_lock<std::mutex> lock(mu_);
    while (!cv_.wait_for(lock, std::chrono::milliseconds(50), [this] { return stop_; })) {
      const double now = Now();
      int late = -1;
      for (int r = 0; r < world_; ++r)
        if (sh_->watch[r].phase.load() != nullptr && now - sh_->watch[r].since.load() > limit_ &&
            (late < 0 || sh_->watch[r].since.load() < sh_->watch[late].since.load()))
          late = r;
      if (late < 0) continue;
      std::string msg = "\nError: DEADLINE_EXCEEDED: rank " + std::to_string(late) +
                        " has been in phase '" + sh_->watch[late].phase.load() + "' for " +
                        std::to_string((int)(now - sh_->watch[late].since.load())) +
                        " s (--phase_timeout_seconds=" + std::to_string((int)limit_) +
                        "); every rank:";
      for (int r = 0; r < world_; ++r) {
        const char *ph = sh_->watch[r].phase.load();
        msg += " [" + std::to_string(r) + "] " + (ph ? ph : "through") +
               (ph ? " " + std::to_string((int)(now - sh_->watch[r].since.load())) + " s" : "");
      }
      msg += "\n";
      (void)!write(2, msg.data(), msg.size());
      _exit(1);  // (no destructors: the stuck threads hold the devices)
    }
  }
  Shared *sh_;
  int world_;
  double limit_;
  std::mutex mu_;
  std::condition_variable cv_;
  bool stop_ = false;
  std::thread thread_;
};

struct Snapshot {
  Shared *sh;
  void operator()() noexcept { sh->agreed_failed = sh->failed.load(); }
};
using PhaseBarrier = std::barrier<Snapshot>;

struct RankState {
  int rank = 0, device = 0;
  cuking_ctx *ctx = nullptr;
  hipStream_t comm = nullptr, compute = nullptr, copy = nullptr;
  uint64_t *d_bits = nullptr;
  bool owns_bits = false;
  cuking_result *d_results = nullptr, *d_gather = nullptr;
  uint32_t *d_counters = nullptr, *d_all = nullptr;
  double *d_rate = nullptr, *d_rates = nullptr;
  hipEvent_t cal_begin = nullptr, cal_end = nullptr;
  std::vector<hipEvent_t> events;

  // Nothing of this rank is in flight any more (peers may still be copying out
  // of its buffers until they have drained theirs: callers pair this with a
  // barrier where that matters).
  void Drain() {
    (void)hipSetDevice(device);
    if (comm) (void)hipStreamSynchronize(comm);
    if (compute) (void)hipStreamSynchronize(compute);
    if (copy) (void)hipStreamSynchronize(copy);
  }

  ~RankState() {
    Drain();
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
    if (cal_begin) (void)hipEventDestroy(cal_begin);
    if (cal_end) (void)hipEventDestroy(cal_end);
    if (comm) (void)hipStreamDestroy(comm);
    if (compute) (void)hipStreamDestroy(compute);
    if (copy) (void)hipStreamDestroy(copy);
    if (d_bits && owns_bits) (void)hipFree(d_bits);
    if (d_results) (void)hipFree(d_results);
    if (d_gather) (void)hipFree(d_gather);
    if (d_counters) (void)hipFree(d_counters);
    if (d_all) (void)hipFree(d_all);
    if (d_rate) (void)hipFree(d_rate);
    if (d_rates) (void)hipFree(d_rates);
    if (ctx) cuking_ctx_destroy(ctx);
  }
};

#define RANK_HIP(expr)                                                        \
  do {                                                                        \
    const hipError_t e_ = (expr);                                             \
    if (e_ != hipSuccess) {                                                   \
      sh->Fail(e_ == hipErrorOutOfMemory ? "RESOURCE_EXHAUSTED" : "INTERNAL", \
               std::string("rank ") + std::to_string(st.rank) + ": " + #expr + \
                   " failed: " + hipGetErrorString(e_));                      \
      ok = false;                                                             \
    }                                                                         \
  } while (0)
#define RANK_ABI(expr)                                                        \
  do {                                                                        \
    const cuking_status s_ = (expr);                                          \
    if (s_ != CUKING_OK) {                                                    \
      sh->Fail(s_ == CUKING_ERR_OUT_OF_MEMORY ? "RESOURCE_EXHAUSTED"          \
               : s_ == CUKING_ERR_INVALID_ARGUMENT ? "INVALID_ARGUMENT"       \
                                                   : "INTERNAL",              \
               std::string("rank ") + std::to_string(st.rank) + ": " +        \
                   cuking_last_error());                                      \
      ok = false;                                                             \
    }                                                                         \
  } while (0)
// Collectives are issued whatever `ok` says (every rank must make every call).
#define RANK_COLL(expr)                                                       \
  do {                                                                        \
    const std::string m_ = (expr);                                            \
    if (!m_.empty()) {                                                        \
      sh->Fail("INTERNAL", std::string("rank ") + std::to_string(st.rank) + ": " + m_); \
      ok = false;                                                             \
    }                                                                         \
  } while (0)

hipEvent_t NewEvent(RankState &st) {
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess)
    st.events.push_back(e);
  return e;
}

int64_t Option(cuking_ctx *ctx, const char *key) {
  int64_t v = 0;
  if (ctx == nullptr || cuking_ctx_get_option(ctx, key, &v) != CUKING_OK) return 0;
  return v;
}

// The whole job of one rank.  Every collective is issued by every rank in the
// same order whatever happens locally: a rank whose own work failed keeps
// taking part (with nothing to contribute) and the failure is agreed on at the
// phase barriers (Shared::agreed_failed), so nobody is left waiting inside RCCL
// or at a barrier for a rank that gave up.
void RankBody(Shared *sh, PhaseBarrier *barrier, int rank);
void RankMain(Shared *sh, PhaseBarrier *barrier, int rank) {
  RankBody(sh, barrier, rank);
  sh->Enter(rank, nullptr);  // through (also on the early exits of an agreed failure)
}
void RankBody(Shared *sh, PhaseBarrier *barrier, int rank) {
  const MultiGpuInput &in = *sh->in;
  const int world = in.num_gpus;
  Collectives *coll = sh->coll;
  RankState st;
  st.rank = rank;
  st.device = in.collectives == "loopback" ? in.first_device : in.first_device + rank;
  bool ok = true;
  const uint32_t wps = in.words_per_sample;
  const uint32_t stored = cuking_submatrix_num_samples(&in.sm);
  const bool tiled = in.kernel != "stream";
  auto inject = [&](const char *phase) {
    if (rank == in.inject_failure_rank && in.inject_failure_phase == phase) {
      sh->Fail("INTERNAL", std::string("rank ") + std::to_string(rank) +
                               ": injected failure in phase " + phase);
      ok = false;
    }
  };
  // TEST ONLY: the rank never comes back from `phase` (the watchdog ends the job).
  auto inject_hang = [&](const char *phase) {
    if (rank == in.inject_failure_rank && in.inject_failure_phase == std::string("hang_") + phase)
      for (;;) std::this_thread::sleep_for(std::chrono::seconds(1));
  };
  // Agreement at the end of a phase: true = somebody failed, everybody stops.
  auto phase_failed = [&]() {
    barrier->arrive_and_wait();
    if (!sh->agreed_failed) return false;
    st.Drain();
    barrier->arrive_and_wait();  // ... and nobody frees what a peer still reads
    return true;
  };

  // ---- setup ----------------------------------------------------------------
  sh->Enter(rank, "setup (context, streams, buffers, workspace reservation)");
  RANK_HIP(hipSetDevice(st.device));
  if (ok) RANK_ABI(cuking_ctx_create(st.device, &st.ctx));
  if (ok)
    RANK_ABI(cuking_ctx_set_kernel(st.ctx, tiled ? CUKING_KERNEL_TILED : CUKING_KERNEL_STREAM));
  if (ok) RANK_ABI(cuking_timing_enable(st.ctx, 1));
  // (the staged schedule's rectangles cover every broadcast chunk in a union over the
  //  ranks: the chunks may be laid out sorted by missing share, include/cuking_amd.h)
  if (ok && tiled) RANK_ABI(cuking_ctx_set_option(st.ctx, "filter_sort", 2));
  if (ok) RANK_HIP(hipStreamCreateWithFlags(&st.comm, hipStreamNonBlocking));
  if (ok) RANK_HIP(hipStreamCreateWithFlags(&st.compute, hipStreamNonBlocking));
  if (ok) RANK_HIP(hipStreamCreateWithFlags(&st.copy, hipStreamNonBlocking));
  if (ok) {
    if (rank == 0 && in.host_bits == nullptr) {
      st.d_bits = in.d_bits_rank0;  // packed on this device already
    } else {
      RANK_HIP(hipMalloc(reinterpret_cast<void **>(&st.d_bits),
                         sh->bit_set_bytes ? sh->bit_set_bytes : 8));
      st.owns_bits = ok;
    }
  }
  const size_t result_bytes = (size_t)(in.max_results ? in.max_results : 1) * sizeof(cuking_result);
  if (ok) RANK_HIP(hipMalloc(reinterpret_cast<void **>(&st.d_results), result_bytes));
  // (the total over the ranks is held to max_results, so this is all rank 0 can receive)
  if (ok && rank == 0 && world > 1)
    RANK_HIP(hipMalloc(reinterpret_cast<void **>(&st.d_gather), result_bytes));
  if (ok) RANK_HIP(hipMalloc(reinterpret_cast<void **>(&st.d_counters), 2 * sizeof(uint32_t)));
  if (ok) RANK_HIP(hipMalloc(reinterpret_cast<void **>(&st.d_all), 2 * sizeof(uint32_t) * world));
  if (ok) RANK_HIP(hipMalloc(reinterpret_cast<void **>(&st.d_rate), sizeof(double)));
  if (ok) RANK_HIP(hipMalloc(reinterpret_cast<void **>(&st.d_rates), sizeof(double) * world));
  if (ok) RANK_HIP(hipEventCreate(&st.cal_begin));
  if (ok) RANK_HIP(hipEventCreate(&st.cal_end));
  if (ok) RANK_HIP(hipMemsetAsync(st.d_counters, 0, 2 * sizeof(uint32_t), st.compute));
  // (a rank that fails before it has a rate contributes 0 to the rate exchange)
  if (ok) RANK_HIP(hipMemsetAsync(st.d_rate, 0, sizeof(double), st.compute));
  // Simple schedule: the bitset does not change once it has arrived, so the block
  // is converted ONCE -- in front of the calibration launch, whose timed range
  // then holds the pair kernel only -- and the main launch reuses the layout.
  if (ok && tiled && !sh->staged) RANK_ABI(cuking_ctx_set_option(st.ctx, "reuse_prepared", 1));
  // The kernel layout of the whole block, the tile prefix and the compute
  // stream's split slab, NOW: once the first broadcast is enqueued no rank may
  // allocate or wait for its device (a blocking allocation beside in-flight
  // RCCL kernels of the other devices of this process can deadlock them).
  if (ok && tiled) {
    void *streams[1] = {st.compute};
    RANK_ABI(cuking_ctx_reserve(st.ctx, &in.sm, wps, streams, 1));
  }
  const int64_t allocs_reserved = Option(st.ctx, "workspace_allocations");
  const int64_t syncs_reserved = Option(st.ctx, "host_syncs");
  inject("setup");
  if (phase_failed()) return;  // nobody has issued a collective yet

  // ---- exchange step 1 + compute ---------------------------------------------
  sh->Enter(rank, "exchange 1 + compute (broadcasts and kernels enqueued)");
  const double t0 = Now();
  // (staged schedule: --rank_weights deals the tile rows in proportion; the
  //  calibration launch belongs to the simple schedule)
  const RowDeal deal = MakeRowDeal(world, sh->staged ? in.rank_weights : std::vector<double>());
  const std::vector<StagedStep> steps =
      StagedSchedule(stored, sh->tile, world, rank, in.chunks, &deal);
  // Exchange first: every upload and broadcast is enqueued before any compute
  // call is made, so that nothing on the host (a calibration wait, a slow launch)
  // delays the chunks still to come; chunk c's kernels wait for its event.
  std::vector<hipEvent_t> arrived(steps.size(), nullptr);
  for (size_t k = 0; k < steps.size(); ++k) {
    const StagedStep &s = steps[k];
    const size_t off = (size_t)s.chunk.begin * wps;                       // words
    const size_t bytes = (size_t)(s.chunk.end - s.chunk.begin) * wps * 8;
    if (rank == 0 && in.host_bits != nullptr) {
      // upload on the copy stream; the broadcast of this chunk waits for it,
      // the upload of the next one runs beside it
      hipEvent_t up = NewEvent(st);
      if (ok) RANK_HIP(hipMemcpyAsync(st.d_bits + off, in.host_bits + off, bytes,
                                      hipMemcpyHostToDevice, st.copy));
      if (ok && up) RANK_HIP(hipEventRecord(up, st.copy));
      if (ok && up) RANK_HIP(hipStreamWaitEvent(st.comm, up, 0));
    }
    RANK_COLL(coll->Broadcast(rank, st.d_bits + off, bytes, 0, st.comm));
    arrived[k] = NewEvent(st);
    if (ok && arrived[k]) RANK_HIP(hipEventRecord(arrived[k], st.comm));
  }
  for (size_t k = 0; k < steps.size(); ++k) {
    const StagedStep &s = steps[k];
    if (ok && arrived[k]) RANK_HIP(hipStreamWaitEvent(st.compute, arrived[k], 0));
    if (sh->staged && s.has_rect && ok) {
      RANK_ABI(cuking_prepare_samples(st.ctx, &in.sm, wps, st.d_bits,
                                      in.sm.i_begin + s.chunk.begin,
                                      in.sm.i_begin + s.chunk.end, st.compute));
      for (const RowStride &r : s.rects)
        if (ok)
          RANK_ABI(cuking_compute_king_rect(
              st.ctx, &in.sm, wps, st.d_bits, in.sm.i_begin + r.row_begin,
              in.sm.i_begin + r.row_end, r.row_step, in.sm.i_begin + s.chunk.begin,
              in.sm.i_begin + s.chunk.end, in.kin_threshold, in.max_results, st.d_results,
              st.d_counters, st.d_counters + 1, st.compute));
    }
  }
  TileRange mine = {0, 0};
  std::vector<double> rates;
  uint64_t cal_tiles = 0;
  if (!sh->staged) {
    auto launch_tiles = [&](uint64_t begin, uint64_t end) {
      if (ok && end > begin)
        RANK_ABI(cuking_compute_king_tiles(st.ctx, &in.sm, wps, st.d_bits, begin, end,
                                           in.kin_threshold, in.max_results, st.d_results,
                                           st.d_counters, st.d_counters + 1, st.compute));
    };
    if (!tiled) {
      // the streaming kernel has no tile enumeration: rank 0 takes the block
      if (rank == 0 && ok)
        RANK_ABI(cuking_compute_king(st.ctx, &in.sm, wps, st.d_bits, in.kin_threshold,
                                     in.max_results, st.d_results, st.d_counters,
                                     st.d_counters + 1, st.compute));
    } else if (!in.rank_weights.empty()) {
      mine = WeightedTilePartition(sh->num_tiles, in.rank_weights)[rank];
      launch_tiles(mine.begin, mine.end);
    } else if (in.calibrate && world > 1 &&
               (cal_tiles = in.calibration_tiles != 0 &&
                                    in.calibration_tiles * (uint64_t)world <= sh->num_tiles
                                ? in.calibration_tiles
                                : CalibrationTiles(sh->num_tiles, world)) != 0) {
      // The GPUs of a node run this kernel several percent apart: every rank
      // times a small range of its own, the rates are exchanged, and the rest
      // of the enumeration is cut in proportion (schedule.h).  Same decision
      // and same collective on every rank.
      sh->Enter(rank, "calibration launch + rate exchange");
      // (an empty tile range converts the block and launches nothing: the timed
      //  range below is the pair kernel alone)
      if (ok)
        RANK_ABI(cuking_compute_king_tiles(st.ctx, &in.sm, wps, st.d_bits, 0, 0, in.kin_threshold,
                                           in.max_results, st.d_results, st.d_counters,
                                           st.d_counters + 1, st.compute));
      if (ok) RANK_HIP(hipEventRecord(st.cal_begin, st.compute));
      launch_tiles((uint64_t)rank * cal_tiles, (uint64_t)(rank + 1) * cal_tiles);
      if (ok) RANK_HIP(hipEventRecord(st.cal_end, st.compute));
      double rate = 0;  // a rank that failed contributes 0: equal ranges then
      float ms = 0;
      if (ok) RANK_HIP(hipEventSynchronize(st.cal_end));
      if (ok) RANK_HIP(hipEventElapsedTime(&ms, st.cal_begin, st.cal_end));
      if (ok && ms > 0) rate = (double)cal_tiles / ms;
      if (ok) RANK_HIP(hipMemcpyAsync(st.d_rate, &rate, sizeof(double), hipMemcpyHostToDevice,
                                      st.compute));
      RANK_COLL(coll->AllGather(rank, st.d_rate, st.d_rates, sizeof(double), st.compute));
      rates.assign(world, 0.0);
      if (ok) RANK_HIP(hipMemcpyAsync(rates.data(), st.d_rates, sizeof(double) * world,
                                      hipMemcpyDeviceToHost, st.compute));
      if (ok) RANK_HIP(hipStreamSynchronize(st.compute));
      bool usable = ok;
      for (double r : rates) usable = usable && std::isfinite(r) && r > 0;
      const uint64_t done = (uint64_t)world * cal_tiles, rest = sh->num_tiles - done;
      // (a rank that cannot use the rates has failed: what it launches no longer matters)
      mine = usable ? WeightedTilePartition(rest, rates)[rank] : TilePartition(rest, world)[rank];
      mine.begin += done;
      mine.end += done;
      launch_tiles(mine.begin, mine.end);
    } else {
      mine = TilePartition(sh->num_tiles, world)[rank];
      launch_tiles(mine.begin, mine.end);
    }
  }
  inject("compute");
  sh->Enter(rank, "exchange 2: counts (all-gather, wait for this rank's kernels)");
  inject_hang("compute");

  // ---- exchange step 2: counts, then records ---------------------------------
  RANK_COLL(coll->AllGather(rank, st.d_counters, st.d_all, 2 * sizeof(uint32_t), st.compute));
  std::vector<uint32_t> all(2 * (size_t)world, 0);
  if (ok) RANK_HIP(hipMemcpyAsync(all.data(), st.d_all, all.size() * sizeof(uint32_t),
                                  hipMemcpyDeviceToHost, st.compute));
  if (ok) RANK_HIP(hipStreamSynchronize(st.compute));   // kernel errors surface here
  if (ok) RANK_HIP(hipStreamSynchronize(st.comm));
  const double t1 = Now();
  if (phase_failed()) return;  // every rank has finished its collectives so far

  std::vector<uint32_t> counts(world);
  bool overflow = false;
  for (int r = 0; r < world; ++r) {
    counts[r] = all[2 * r];
    overflow = overflow || all[2 * r + 1] != 0;
  }
  const GatherPlan plan = PlanGather(counts);
  // One GPU reports overflow when its records exceed max_results; the job must
  // not depend on how many GPUs shared it, so the total is held to the same cap.
  if (overflow || plan.total > in.max_results) {
    if (rank == 0)
      sh->Fail("RESOURCE_EXHAUSTED",
               "Could not store all results: try increasing the --max_results parameter.");
    return;  // same decision on every rank (same data): no collective is left half-issued
  }
  inject("gather");
  if (phase_failed()) return;  // (a failure here is agreed on before anybody sends)
  sh->Enter(rank, "exchange 2: records (gather on rank 0)");
  inject_hang("gather");
  {
    std::vector<uint64_t> bytes(world), offset(world);
    for (int r = 0; r < world; ++r) {
      bytes[r] = (uint64_t)counts[r] * sizeof(cuking_result);
      offset[r] = plan.offset[r] * sizeof(cuking_result);
    }
    RANK_COLL(coll->GatherToRoot(rank, st.d_results, st.d_gather, bytes, offset, st.compute));
  }
  if (rank == 0) {
    sh->out->results.resize(plan.total);
    cuking_result *host = sh->out->results.data();
    if (ok && counts[0])
      RANK_HIP(hipMemcpyAsync(host, st.d_results, (size_t)counts[0] * sizeof(cuking_result),
                              hipMemcpyDeviceToHost, st.compute));
    if (ok && world > 1 && plan.total > counts[0])
      RANK_HIP(hipMemcpyAsync(host + counts[0], st.d_gather + counts[0],
                              (size_t)(plan.total - counts[0]) * sizeof(cuking_result),
                              hipMemcpyDeviceToHost, st.compute));
  }
  if (ok) RANK_HIP(hipStreamSynchronize(st.compute));
  const double t2 = Now();

  double king_ms = 0, prep_ms = 0;
  uint64_t nk = 0, np = 0;
  if (ok) RANK_ABI(cuking_timing_collect(st.ctx, &king_ms, &nk, &prep_ms, &np));
  {
    std::lock_guard<std::mutex> lock(sh->mu);
    sh->out->rank_kernel_ms[rank] = king_ms;
    sh->out->rank_prepare_ms[rank] = prep_ms;
    sh->out->rank_results[rank] = counts[rank];
    sh->out->rank_allocations_after_reserve[rank] =
        (uint64_t)(Option(st.ctx, "workspace_allocations") - allocs_reserved);
    sh->out->rank_host_syncs_after_reserve[rank] =
        (uint64_t)(Option(st.ctx, "host_syncs") - syncs_reserved);
    if (!sh->staged && tiled) sh->out->rank_tile_ranges[rank] = {mine.begin, mine.end};
    if (rank == 0) {
      sh->out->exchange_and_compute_seconds = t1 - t0;
      sh->out->gather_seconds = t2 - t1;
      sh->out->rank_rates = rates;
      sh->out->calibration_tiles = cal_tiles;
    }
  }
  sh->Enter(rank, "last barrier (nobody tears down while a peer still receives)");
  barrier->arrive_and_wait();
  sh->Enter(rank, nullptr);
}

}  // namespace

std::string RunMultiGpu(const MultiGpuInput &in, MultiGpuOutput *out, std::string *code) {
  Shared sh;
  sh.in = &in;
  sh.out = out;
  auto fail = [&](const char *c, const std::string &m) {
    *code = c;
    return m;
  };
  const bool loopback = in.collectives == "loopback";
  if (!loopback && in.collectives != "rccl")
    return fail("INVALID_ARGUMENT", "unknown collectives implementation " + in.collectives);
  const int available = cuking_device_count();
  if (available <= 0)
    return fail("INTERNAL", "no HIP device available; this program has no CPU path");
  if (in.num_gpus < 1 || in.first_device < 0 ||
      in.first_device + (loopback ? 1 : in.num_gpus) > available)
    return fail("INVALID_ARGUMENT",
                "--num_gpus=" + std::to_string(in.num_gpus) + " from device " +
                    std::to_string(in.first_device) + ", but " + std::to_string(available) +
                    " GPU(s) are visible");
  if (!in.rank_weights.empty()) {
    if ((int)in.rank_weights.size() != in.num_gpus)
      return fail("INVALID_ARGUMENT", "--rank_weights needs one weight per GPU");
    for (double w : in.rank_weights)
      if (!(w > 0) || !std::isfinite(w))
        return fail("INVALID_ARGUMENT", "--rank_weights must be positive");
  }
  const bool diag = in.sm.i_begin == in.sm.j_begin;
  if (in.mode == "staged" && (!diag || in.kernel == "stream"))
    return fail("INVALID_ARGUMENT",
                "--multi_gpu_mode=staged needs a diagonal block and the tiled kernel");
  sh.staged = in.kernel != "stream" && diag && in.mode != "simple";
  out->mode = sh.staged ? "staged" : "simple";
  sh.tile = cuking_tile_samples(nullptr);
  sh.num_tiles = cuking_num_tiles(nullptr, &in.sm);
  sh.bit_set_bytes = (size_t)cuking_submatrix_num_samples(&in.sm) * in.words_per_sample * 8;
  out->bytes_broadcast = sh.bit_set_bytes;
  out->rank_kernel_ms.assign(in.num_gpus, 0);
  out->rank_prepare_ms.assign(in.num_gpus, 0);
  out->rank_results.assign(in.num_gpus, 0);
  out->rank_allocations_after_reserve.assign(in.num_gpus, 0);
  out->rank_host_syncs_after_reserve.assign(in.num_gpus, 0);
  out->rank_tile_ranges.assign(in.num_gpus, {0, 0});

  std::vector<int> devices(in.num_gpus);
  for (int r = 0; r < in.num_gpus; ++r) devices[r] = loopback ? in.first_device : in.first_device + r;
  std::unique_ptr<Collectives> coll = loopback ? MakeLoopbackCollectives() : MakeRcclCollectives();
  sh.coll = coll.get();
  out->collectives = coll->name();
  sh.watch.reset(new Shared::Watch[in.num_gpus]);
  Watchdog watchdog(&sh, in.num_gpus, in.phase_timeout_seconds);
  for (int r = 0; r < in.num_gpus; ++r) sh.Enter(r, "communicator set-up (InitAll)");
  const auto init_t0 = std::chrono::steady_clock::now();
  const std::string init_error = coll->InitAll(devices);
  out->comm_init_seconds =
      std::chrono::duration<double>(std::chrono::steady_clock::now() - init_t0).count();
  if (!init_error.empty()) return fail("INTERNAL", init_error);

  PhaseBarrier barrier(in.num_gpus, Snapshot{&sh});
  std::vector<std::thread> threads;
  for (int r = 1; r < in.num_gpus; ++r) threads.emplace_back(RankMain, &sh, &barrier, r);
  RankMain(&sh, &barrier, 0);
  for (auto &t : threads) t.join();
  coll->Destroy();
  if (sh.failed.load()) {
    *code = sh.code;
    return sh.error;
  }
  return "";
}

}  // namespace cuking_host
