// RCCL implementation of collectives.h: one process, one communicator per GPU.
#include "collectives.h"

#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

namespace cuking_host {

namespace {

class RcclCollectives : public Collectives {
 public:
  const char *name() const override { return "rccl"; }

  std::string InitAll(const std::vector<int> &devices) override {
    comms_.assign(devices.size(), nullptr);
    const ncclResult_t rc = ncclCommInitAll(comms_.data(), (int)devices.size(), devices.data());
    if (rc != ncclSuccess) {
      comms_.clear();
      return std::string("ncclCommInitAll failed: ") + ncclGetErrorString(rc);
    }
    return "";
  }

  std::string Broadcast(int rank, void *buf, size_t bytes, int root, void *stream) override {
    // (with one rank this is RCCL's single-rank broadcast: the same call runs)
    return Check("ncclBroadcast", ncclBroadcast(buf, buf, bytes, ncclUint8, root, comms_[rank],
                                                (hipStream_t)stream));
  }

  std::string AllGather(int rank, const void *send, void *recv, size_t bytes_per_rank,
                        void *stream) override {
    return Check("ncclAllGather", ncclAllGather(send, recv, bytes_per_rank, ncclUint8,
                                                comms_[rank], (hipStream_t)stream));
  }

  std::string GatherToRoot(int rank, const void *send, void *recv,
                           const std::vector<uint64_t> &bytes,
                           const std::vector<uint64_t> &offset, void *stream) override {
    const int world = (int)comms_.size();
    if (world == 1) return "";
    std::string err = Check("ncclGroupStart", ncclGroupStart());
    if (rank == 0) {
      for (int r = 1; r < world && err.empty(); ++r)
        if (bytes[r])
          err = Check("ncclRecv", ncclRecv(static_cast<char *>(recv) + offset[r], bytes[r],
                                           ncclUint8, r, comms_[rank], (hipStream_t)stream));
    } else if (bytes[rank]) {
      err = Check("ncclSend",
                  ncclSend(send, bytes[rank], ncclUint8, 0, comms_[rank], (hipStream_t)stream));
    }
    // (the group is closed whatever happened inside it)
    const std::string end = Check("ncclGroupEnd", ncclGroupEnd());
    return err.empty() ? end : err;
  }

  void Destroy() override {
    for (ncclComm_t c : comms_)
      if (c) (void)ncclCommDestroy(c);
    comms_.clear();
  }

  ~RcclCollectives() override { Destroy(); }

 private:
  static std::string Check(const char *what, ncclResult_t r) {
    if (r == ncclSuccess) return "";
    return std::string(what) + " failed: " + ncclGetErrorString(r);
  }
  std::vector<ncclComm_t> comms_;
};

}  // namespace

std::unique_ptr<Collectives> MakeRcclCollectives() {
  return std::unique_ptr<Collectives>(new RcclCollectives());
}

}  // namespace cuking_host
