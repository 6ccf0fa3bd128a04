// The four collective calls the multi-GPU host makes, behind one interface:
// InitAll, Broadcast, AllGather, GatherToRoot (a grouped send/recv).  Two
// implementations:
//   rccl      RCCL over xGMI, one communicator per GPU from ncclCommInitAll --
//             the product path (`cuking --num_gpus=N`).
//   loopback  TEST ONLY (`--collectives=loopback`): the same calls as
//             device-to-device copies ordered by HIP events, with a host
//             rendezvous per call.  It lets several rank threads share ONE GPU
//             (RCCL refuses two ranks on a device: "Duplicate GPU detected"),
//             so that the whole of RankMain -- offsets of the gather, the cap on
//             the total, the agreement on failures -- runs with N > 1 on a
//             one-GPU box.  No RCCL call is made; nothing about it is fast.
// Reference anchor: the reference fans shards out over VMs and gathers part
// files afterwards (cloud_batch_submit.py:45,73, :111-124); inside a node the
// same two exchange steps are these calls.
#ifndef CUKING_AMD_HOST_COLLECTIVES_H_
#define CUKING_AMD_HOST_COLLECTIVES_H_

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace cuking_host {

// Streams are hipStream_t passed as void* (as in include/cuking_amd.h).
// Every method returns "" or an error message.  InitAll and Destroy are called
// once, from one thread; the others by rank r's own thread, every rank calling
// them in the same order (like RCCL).
class Collectives {
 public:
  virtual ~Collectives() = default;
  virtual const char *name() const = 0;
  // One rank per entry of `devices` (HIP device indices; loopback accepts
  // repeats).
  virtual std::string InitAll(const std::vector<int> &devices) = 0;
  // In place: `buf` of rank `root` to `buf` of every rank.
  virtual std::string Broadcast(int rank, void *buf, size_t bytes, int root, void *stream) = 0;
  // `bytes_per_rank` from every rank's `send` to recv + r * bytes_per_rank.
  virtual std::string AllGather(int rank, const void *send, void *recv, size_t bytes_per_rank,
                                void *stream) = 0;
  // Rank r > 0 sends bytes[r] from `send`; rank 0 receives them at
  // recv + offset[r] (rank 0's own part stays where it is).  bytes / offset
  // are the same on every rank.
  virtual std::string GatherToRoot(int rank, const void *send, void *recv,
                                   const std::vector<uint64_t> &bytes,
                                   const std::vector<uint64_t> &offset, void *stream) = 0;
  virtual void Destroy() = 0;
};

std::unique_ptr<Collectives> MakeRcclCollectives();
std::unique_ptr<Collectives> MakeLoopbackCollectives();

}  // namespace cuking_host

#endif  // CUKING_AMD_HOST_COLLECTIVES_H_
