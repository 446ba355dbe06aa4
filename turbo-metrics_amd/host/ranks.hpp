// ranks.hpp -- `turbo-metrics --ranks N`: one PROCESS per GPU on one node, frame pairs sharded in contiguous blocks, and ONE
// reduce(sum, f64) of the zero-padded per-frame score vector to rank 0 -- RCCL over xGMI between the GPUs (BASELINE.json
// north_star: "a single RCCL reduce of per-frame scores").  The reference has no counterpart: it is single-GPU, device 0 hard-coded
// (crates/turbo-metrics/src/lib.rs:442); the partitioning follows its Options (lib.rs:39-54): every rank runs the reference's selection
// loop on its block of decode indices.  The Python side of the same arrangement is turbo-metrics_amd/launch.py + shard.py (bench.py).
//
//   launcher   the process the user starts.  It makes NO GPU call: it creates the pipes, starts N copies of itself (fork + exec of
//              /proc/self/exe, same argv; TM_RANK / TM_WORLD / TM_RANK_UP / TM_RANK_DOWN in the environment), forwards nothing -- rank 0
//              inherits stdout, the other ranks write theirs to /dev/null, every rank inherits stderr --, waits for exactly the PIDs it
//              started and returns non-zero as soon as one of them fails (the others are then terminated, by PID).
//   rank r     binds to device r (and that device's NUMA node: init_hip), scores its block, and takes part in the one reduce.
//   transport  "rccl" (default for N > 1): libturbometrics_rccl.so (include/turbo_metrics_comm.h) is loaded at run time -- a
//              single-device run never pays for loading RCCL --, the communicator's unique id travels from rank 0 to the others over the
//              launcher's pipes, an 8-byte reduce warms the communicator up during set-up, then ONE ncclReduce(sum, ncclDouble, root 0) carries the scores.  "pipe" (TM_RANK_TRANSPORT=pipe; the CPU test tier, and
//              ranks that share one device, which RCCL refuses): every rank writes its vector to rank 0, which adds them in rank order.
//              Either way every entry is non-zero on at most one rank and is added to zeros: rank 0 holds the single-device values
//              bit for bit.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "turbo_metrics.hpp"

namespace tm_host {

// contiguous block [lo, hi) of rank `rank`: ceil(n / world) decode indices per rank, the last ranks may be short or empty
// (== shard.py::shard_range and the blocks of `--devices N`)
void shard_range(uint32_t n, uint32_t rank, uint32_t world, uint32_t &lo, uint32_t &hi);

struct RankEnv {
    int rank = -1, world = 0;
    std::vector<int> up;   // rank 0: read ends, [r - 1] = from rank r; rank r > 0: one write end (to rank 0)
    std::vector<int> down; // rank 0: write ends, [r - 1] = to rank r; rank r > 0: one read end (from rank 0)
};
// is this process a rank started by launch_ranks?  (TM_RANK / TM_WORLD / TM_RANK_UP / TM_RANK_DOWN)
bool rank_env(RankEnv &out);

// The launcher: start `world` rank processes of this very program with the same arguments, wait for them, return the exit code (0, or the
// first failing rank's; 124 after timeout_s > 0 seconds).  Must be called before the process has made any GPU call.
// self: the program to start (nullptr = /proc/self/exe)
int launch_ranks(char **argv, int world, double timeout_s = 0.0, const char *self = nullptr);

// A rank that fails only because ANOTHER rank went away (its pipe was closed under the reduce) says so: the process exits with
// RANK_PEER_LOST, and the launcher reports the rank that failed first -- with its own exit code -- when it finds one.
constexpr int RANK_PEER_LOST = 75;
class RankPeerLost : public std::runtime_error {
public:
    using std::runtime_error::runtime_error;
};

// one reduce(sum, f64) to rank 0
class RankTransport {
public:
    virtual ~RankTransport() = default;
    virtual const char *name() const = 0;
    // in: this rank's zero-padded vector; out (rank 0 only): the sum over the ranks.  Same length on every rank.
    virtual void reduce_sum_to_root(std::vector<double> &v) = 0;
};
// kind: "pipe" | "rccl" | "" (= TM_RANK_TRANSPORT from the environment, default rccl).  The RCCL transport needs the calling process
// to be bound to its device already (init_hip); throws std::runtime_error when libturbometrics_rccl.so cannot be loaded or RCCL fails
// (there is no silent fallback to pipes).
std::unique_ptr<RankTransport> make_rank_transport(const RankEnv &env, const std::string &kind = "");

// The score vector: `total` decode indices x (1 + selected metrics) doubles -- [scored flag, psnr?, ssim?, msssim?, ssimulacra2?] -- and one
// last entry, the number of frames this rank decoded.  A block writes its own rows; everything else stays 0.0.
struct ScoreVector {
    Metrics metrics;
    uint32_t total = 0;
    std::vector<double> v;
    ScoreVector(const Metrics &m, uint32_t total_indices);
    size_t stride() const;
    void put(uint32_t decode_index, const FrameScores &s);
    void add_decoded(uint32_t n) { v.back() += (double)n; }
    uint32_t decoded() const { return (uint32_t)v.back(); }
    // the scored frames in decode order
    std::vector<FrameScores> frames() const;
};

} // namespace tm_host
