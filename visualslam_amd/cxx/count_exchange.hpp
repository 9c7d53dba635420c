// The one collective of the multi-GPU path: an all-gather of every rank's
// {harris_count, dog_count} - 2 x uint64 = 16 bytes per rank - over RCCL's C API (librccl, no torch,
// no MPI).  One process per GPU; frames never cross GPUs, so this is the only inter-GPU traffic and it is
// latency-bound (xGMI bandwidth is irrelevant at 16 bytes per rank).
//
// Bootstrap: RCCL needs the ncclUniqueId of rank 0 in every rank before ncclCommInitRank.  Without MPI it
// travels over one TCP connection per rank to rank 0 (single node: 127.0.0.1 by default).  Environment,
// torchrun-compatible: RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR (127.0.0.1), MASTER_PORT (29533); the
// rendezvous listens on VSLAM_RDV_PORT, default MASTER_PORT + 1 (under torchrun MASTER_PORT itself
// belongs to torch's store).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace vslam {

struct RankEnv {
    int rank = 0, world = 1, local_rank = 0;
    std::string addr = "127.0.0.1";
    int rdv_port = 29534;
    static RankEnv from_environment();
};

// rank 0 sends `bytes` to every other rank, the others receive them; 60 s timeout; throws vslam::Error.
// Host-only (no GPU): the rendezvous of the RCCL id, also exercised on its own by `Stream --rdv-selftest`.
void tcp_broadcast_from_rank0(const RankEnv& env, void* bytes, size_t n);

class CountExchange {
public:
    // RCCL: the product path.  Tcp: REHEARSAL ONLY (VSLAM_COUNT_BACKEND=tcp, like bench.py's VSLAM_BENCH_BACKEND=gloo):
    // the same collectives carried by the rendezvous sockets through rank 0, host side, so that the N > 1 logic
    // (rank environment, totals, max over ranks) can run with several ranks on ONE GPU, which RCCL refuses.
    enum class Backend { Rccl, Tcp };
    static Backend backend_from_environment();
    // Collective over all ranks (also with world = 1: the communicator is then a single-rank RCCL one).
    CountExchange(const RankEnv& env, int device, Backend backend = Backend::Rccl);
    ~CountExchange();
    CountExchange(const CountExchange&) = delete;
    CountExchange& operator=(const CountExchange&) = delete;
    Backend backend() const { return backend_; }
    int rank() const { return env_.rank; }
    int world() const { return env_.world; }
    // ncclAllGather of d_local (2 x uint64, device memory) into the object's device buffer, asynchronous on
    // `stream` (a hipStream_t): enqueue it right behind the kernels that produce d_local.
    void all_gather_async(const uint64_t* d_local, void* stream);
    // Waits for `stream` and returns [world][2] = {harris, dog} per rank.
    std::vector<uint64_t> fetch(void* stream);
    // ncclAllReduce(max) of one double (the max-over-ranks step time of the bench contract); synchronous.
    double max_over_ranks(double v, void* stream);
    void barrier(void* stream);

private:
    void tcp_round(const void* mine, size_t n, void* table);  // every rank's n bytes -> rank 0 -> the whole table back to every rank
    RankEnv env_;
    int device_ = 0;
    Backend backend_ = Backend::Rccl;
    std::vector<int> fds_;     // Tcp: rank 0: one socket per peer (index = rank); others: fds_[0] = the socket to rank 0
    void* comm_ = nullptr;     // ncclComm_t
    uint64_t* d_all_ = nullptr;  // [world][2]
    double* d_scratch_ = nullptr;
};

}  // namespace vslam
