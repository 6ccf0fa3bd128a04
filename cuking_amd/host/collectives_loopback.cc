// TEST-ONLY implementation of collectives.h (see there): device-to-device
// copies ordered by HIP events, one host rendezvous per call.  Rank threads may
// share a GPU.  Semantics kept from RCCL: every rank makes every call in the
// same order; a call returns once the rank's part is ENQUEUED on its stream;
// a buffer handed to a call must stay valid and unchanged until every rank's
// stream has passed the call (RankMain drains its streams before it frees).
#include "collectives.h"

#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>

namespace cuking_host {

namespace {

class LoopbackCollectives : public Collectives {
 public:
  const char *name() const override { return "loopback"; }

  std::string InitAll(const std::vector<int> &devices) override {
    devices_ = devices;
    seq_.assign(devices.size(), 0);
    return "";
  }

  std::string Broadcast(int rank, void *buf, size_t bytes, int root, void *stream) override {
    std::vector<Post> posts;
    std::string err = Rendezvous(rank, buf, buf, stream, &posts);
    if (!err.empty() || rank == root || bytes == 0) return err;
    return Copy(rank, buf, posts[root], posts[root].send, bytes, stream);
  }

  std::string AllGather(int rank, const void *send, void *recv, size_t bytes_per_rank,
                        void *stream) override {
    std::vector<Post> posts;
    std::string err = Rendezvous(rank, send, recv, stream, &posts);
    for (size_t r = 0; r < posts.size() && err.empty(); ++r)
      err = Copy(rank, static_cast<char *>(recv) + r * bytes_per_rank, posts[r], posts[r].send,
                 bytes_per_rank, stream);
    return err;
  }

  std::string GatherToRoot(int rank, const void *send, void *recv,
                           const std::vector<uint64_t> &bytes,
                           const std::vector<uint64_t> &offset, void *stream) override {
    if (devices_.size() == 1) return "";
    std::vector<Post> posts;
    std::string err = Rendezvous(rank, send, recv, stream, &posts);
    if (rank != 0) return err;
    for (size_t r = 1; r < posts.size() && err.empty(); ++r)
      if (bytes[r])
        err = Copy(rank, static_cast<char *>(recv) + offset[r], posts[r], posts[r].send, bytes[r],
                   stream);
    return err;
  }

  void Destroy() override {
    std::lock_guard<std::mutex> lock(mu_);
    for (auto &e : events_) {
      (void)hipSetDevice(e.first);
      (void)hipEventDestroy(e.second);
    }
    events_.clear();
    ops_.clear();
  }

  ~LoopbackCollectives() override { Destroy(); }

 private:
  struct Post {
    const void *send = nullptr;
    void *recv = nullptr;
    hipEvent_t ready = nullptr;  // everything enqueued before the call, on the poster's stream
    int device = 0;
  };
  struct Op {
    std::vector<Post> posts;
    size_t arrived = 0, left = 0;
  };

  // Posts this rank's buffers for its next call and waits until every rank has
  // posted for the same call.
  std::string Rendezvous(int rank, const void *send, void *recv, void *stream,
                         std::vector<Post> *out) {
    const size_t world = devices_.size();
    Post mine;
    mine.send = send;
    mine.recv = recv;
    mine.device = devices_[rank];
    hipError_t e = hipEventCreateWithFlags(&mine.ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(mine.ready, (hipStream_t)stream);
    // (a rank whose HIP calls fail still posts -- with no event -- so that the
    //  others are not left waiting; its error is reported below)
    std::unique_lock<std::mutex> lock(mu_);
    if (mine.ready) events_.emplace_back(mine.device, mine.ready);
    if (e != hipSuccess) mine.ready = nullptr;
    const uint64_t id = seq_[rank]++;
    Op &op = ops_[id];
    if (op.posts.empty()) op.posts.resize(world);
    op.posts[rank] = mine;
    ++op.arrived;
    cv_.notify_all();
    const bool all = cv_.wait_for(lock, std::chrono::seconds(120),
                                  [&] { return ops_[id].arrived == world; });
    if (!all) return "loopback collectives: a rank did not arrive within 120 s";
    *out = ops_[id].posts;
    if (++ops_[id].left == world) ops_.erase(id);
    lock.unlock();
    if (e != hipSuccess)
      return std::string("loopback collectives: ") + hipGetErrorString(e);
    return "";
  }

  std::string Copy(int rank, void *dst, const Post &from, const void *src, size_t bytes,
                   void *stream) {
    if (bytes == 0 || dst == src) return "";
    if (from.ready == nullptr) return "loopback collectives: the sending rank failed";
    hipError_t e = hipStreamWaitEvent((hipStream_t)stream, from.ready, 0);
    if (e == hipSuccess)
      e = hipMemcpyAsync(dst, src, bytes,
                         from.device == devices_[rank] ? hipMemcpyDeviceToDevice : hipMemcpyDefault,
                         (hipStream_t)stream);
    if (e != hipSuccess) return std::string("loopback collectives: ") + hipGetErrorString(e);
    return "";
  }

  std::vector<int> devices_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<uint64_t> seq_;
  std::map<uint64_t, Op> ops_;
  std::vector<std::pair<int, hipEvent_t>> events_;
};

}  // namespace

std::unique_ptr<Collectives> MakeLoopbackCollectives() {
  return std::unique_ptr<Collectives>(new LoopbackCollectives());
}

}  // namespace cuking_host
