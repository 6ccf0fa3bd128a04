// See multi_gpu.h.  Host code only: HIP runtime API for streams/events/copies,
// RCCL for the collectives, the C ABI (include/cuking_amd.h) for every kernel.
#include "multi_gpu.h"

#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <barrier>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>

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
  bool staged = false;
  uint32_t tile = 0;
  uint64_t num_tiles = 0;
  size_t bit_set_bytes = 0;
  std::vector<ncclComm_t> comms;
  std::barrier<> *barrier = nullptr;
  std::atomic<bool> failed{false};
  std::mutex mu;
  std::string error, code;

  void Fail(const std::string &c, const std::string &msg) {
    std::lock_guard<std::mutex> lock(mu);
    if (error.empty()) {
      code = c;
      error = msg;
    }
    failed.store(true);
  }
};

struct RankState {
  int rank = 0, device = 0;
  cuking_ctx *ctx = nullptr;
  hipStream_t comm = nullptr, compute = nullptr, copy = nullptr;
  uint64_t *d_bits = nullptr;
  bool owns_bits = false;
  cuking_result *d_results = nullptr, *d_gather = nullptr;
  uint32_t *d_counters = nullptr, *d_all = nullptr;
  std::vector<hipEvent_t> events;

  ~RankState() {
    (void)hipSetDevice(device);
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
    if (comm) (void)hipStreamDestroy(comm);
    if (compute) (void)hipStreamDestroy(compute);
    if (copy) (void)hipStreamDestroy(copy);
    if (d_bits && owns_bits) (void)hipFree(d_bits);
    if (d_results) (void)hipFree(d_results);
    if (d_gather) (void)hipFree(d_gather);
    if (d_counters) (void)hipFree(d_counters);
    if (d_all) (void)hipFree(d_all);
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
#define RANK_NCCL(expr)                                                       \
  do {                                                                        \
    const ncclResult_t r_ = (expr);                                           \
    if (r_ != ncclSuccess) {                                                  \
      sh->Fail("INTERNAL", std::string("rank ") + std::to_string(st.rank) +   \
                               ": " + #expr + " failed: " +                   \
                               ncclGetErrorString(r_));                       \
      ok = false;                                                             \
    }                                                                         \
  } while (0)

hipEvent_t NewEvent(RankState &st) {
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess)
    st.events.push_back(e);
  return e;
}

// The whole job of one rank.  Every collective is issued by every rank in the
// same order whatever happens locally: a rank whose own work failed keeps
// taking part (with nothing to contribute) and the failure is agreed on at the
// host barriers, so nobody is left waiting inside RCCL for a rank that gave up.
void RankMain(Shared *sh, int rank) {
  const MultiGpuInput &in = *sh->in;
  const int world = in.num_gpus;
  RankState st;
  st.rank = rank;
  st.device = in.first_device + rank;
  bool ok = true;
  const uint32_t wps = in.words_per_sample;
  const uint32_t stored = cuking_submatrix_num_samples(&in.sm);

  // ---- setup ----------------------------------------------------------------
  RANK_HIP(hipSetDevice(st.device));
  if (ok) RANK_ABI(cuking_ctx_create(st.device, &st.ctx));
  if (ok)
    RANK_ABI(cuking_ctx_set_kernel(st.ctx, in.kernel == "stream" ? CUKING_KERNEL_STREAM
                                                                 : CUKING_KERNEL_TILED));
  if (ok) RANK_ABI(cuking_timing_enable(st.ctx, 1));
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
  if (ok) RANK_HIP(hipMalloc(reinterpret_cast<void **>(&st.d_counters), 2 * sizeof(uint32_t)));
  if (ok) RANK_HIP(hipMalloc(reinterpret_cast<void **>(&st.d_all), 2 * sizeof(uint32_t) * world));
  if (ok) RANK_HIP(hipMemsetAsync(st.d_counters, 0, 2 * sizeof(uint32_t), st.compute));
  sh->barrier->arrive_and_wait();
  if (sh->failed.load()) return;  // nobody has issued a collective yet

  // ---- exchange step 1 + compute ---------------------------------------------
  const double t0 = Now();
  const std::vector<StagedStep> steps =
      StagedSchedule(stored, sh->tile, world, rank, in.chunks);
  for (const StagedStep &s : steps) {
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
    // (with one rank this is RCCL's single-rank broadcast: the same calls run)
    RANK_NCCL(ncclBroadcast(st.d_bits + off, st.d_bits + off, bytes, ncclUint8, 0,
                            sh->comms[rank], st.comm));
    hipEvent_t arrived = NewEvent(st);
    if (ok && arrived) RANK_HIP(hipEventRecord(arrived, st.comm));
    if (ok && arrived) RANK_HIP(hipStreamWaitEvent(st.compute, arrived, 0));
    if (sh->staged && s.has_rect && ok) {
      RANK_ABI(cuking_prepare_samples(st.ctx, &in.sm, wps, st.d_bits,
                                      in.sm.i_begin + s.chunk.begin,
                                      in.sm.i_begin + s.chunk.end, st.compute));
      if (ok)
        RANK_ABI(cuking_compute_king_rect(
            st.ctx, &in.sm, wps, st.d_bits, in.sm.i_begin + s.row_begin,
            in.sm.i_begin + s.row_end, s.row_step, in.sm.i_begin + s.chunk.begin,
            in.sm.i_begin + s.chunk.end, in.kin_threshold, in.max_results, st.d_results,
            st.d_counters, st.d_counters + 1, st.compute));
    }
  }
  if (!sh->staged && ok) {
    if (in.kernel == "stream") {
      // the streaming kernel has no tile enumeration: rank 0 takes the block
      if (rank == 0)
        RANK_ABI(cuking_compute_king(st.ctx, &in.sm, wps, st.d_bits, in.kin_threshold,
                                     in.max_results, st.d_results, st.d_counters,
                                     st.d_counters + 1, st.compute));
    } else {
      const TileRange mine = TilePartition(sh->num_tiles, world)[rank];
      RANK_ABI(cuking_compute_king_tiles(st.ctx, &in.sm, wps, st.d_bits, mine.begin, mine.end,
                                         in.kin_threshold, in.max_results, st.d_results,
                                         st.d_counters, st.d_counters + 1, st.compute));
    }
  }

  // ---- exchange step 2: counts, then records ---------------------------------
  RANK_NCCL(ncclAllGather(st.d_counters, st.d_all, 2, ncclUint32, sh->comms[rank], st.compute));
  std::vector<uint32_t> all(2 * (size_t)world, 0);
  if (ok) RANK_HIP(hipMemcpyAsync(all.data(), st.d_all, all.size() * sizeof(uint32_t),
                                  hipMemcpyDeviceToHost, st.compute));
  if (ok) RANK_HIP(hipStreamSynchronize(st.compute));   // kernel errors surface here
  if (ok) RANK_HIP(hipStreamSynchronize(st.comm));
  const double t1 = Now();
  sh->barrier->arrive_and_wait();
  if (sh->failed.load()) return;  // every rank has finished its collectives so far

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
  if (rank == 0 && world > 1)
    RANK_HIP(hipMalloc(reinterpret_cast<void **>(&st.d_gather),
                       (size_t)(plan.total ? plan.total : 1) * sizeof(cuking_result)));
  // (an allocation failure on rank 0 is agreed on before anybody sends)
  sh->barrier->arrive_and_wait();
  if (sh->failed.load()) return;
  if (world > 1) {
    RANK_NCCL(ncclGroupStart());
    if (rank == 0) {
      for (int r = 1; r < world; ++r)
        if (counts[r])
          RANK_NCCL(ncclRecv(st.d_gather + plan.offset[r], (size_t)counts[r] * 6, ncclUint32, r,
                             sh->comms[rank], st.compute));
    } else if (counts[rank]) {
      RANK_NCCL(ncclSend(st.d_results, (size_t)counts[rank] * 6, ncclUint32, 0,
                         sh->comms[rank], st.compute));
    }
    RANK_NCCL(ncclGroupEnd());
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
    if (rank == 0) {
      sh->out->exchange_and_compute_seconds = t1 - t0;
      sh->out->gather_seconds = t2 - t1;
    }
  }
  sh->barrier->arrive_and_wait();  // nobody tears down while a peer still receives
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
  const int available = cuking_device_count();
  if (available <= 0)
    return fail("INTERNAL", "no HIP device available; this program has no CPU path");
  if (in.num_gpus < 1 || in.first_device < 0 || in.first_device + in.num_gpus > available)
    return fail("INVALID_ARGUMENT",
                "--num_gpus=" + std::to_string(in.num_gpus) + " from device " +
                    std::to_string(in.first_device) + ", but " + std::to_string(available) +
                    " GPU(s) are visible");
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

  std::vector<int> devices(in.num_gpus);
  for (int r = 0; r < in.num_gpus; ++r) devices[r] = in.first_device + r;
  sh.comms.assign(in.num_gpus, nullptr);
  const auto init_t0 = std::chrono::steady_clock::now();
  const ncclResult_t rc = ncclCommInitAll(sh.comms.data(), in.num_gpus, devices.data());
  out->comm_init_seconds =
      std::chrono::duration<double>(std::chrono::steady_clock::now() - init_t0).count();
  if (rc != ncclSuccess)
    return fail("INTERNAL", std::string("ncclCommInitAll failed: ") + ncclGetErrorString(rc));

  std::barrier<> barrier(in.num_gpus);
  sh.barrier = &barrier;
  std::vector<std::thread> threads;
  for (int r = 1; r < in.num_gpus; ++r) threads.emplace_back(RankMain, &sh, r);
  RankMain(&sh, 0);
  for (auto &t : threads) t.join();
  for (ncclComm_t c : sh.comms)
    if (c) (void)ncclCommDestroy(c);
  if (sh.failed.load()) {
    *code = sh.code;
    return sh.error;
  }
  return "";
}

}  // namespace cuking_host
