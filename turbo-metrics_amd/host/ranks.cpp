// ranks.cpp -- see ranks.hpp.  The launcher follows turbo-metrics_amd/launch.py's rules: fresh processes started before any GPU call,
// exact PIDs, non-zero exit as soon as one rank fails.
#include "ranks.hpp"

#include <cerrno>
#include <chrono>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include "../../include/turbo_metrics_comm.h"

namespace tm_host {

void shard_range(uint32_t n, uint32_t rank, uint32_t world, uint32_t &lo, uint32_t &hi)
{
    const uint32_t per = world ? (n + world - 1) / world : n;
    lo = std::min<uint64_t>(n, (uint64_t)rank * per);
    hi = std::min<uint64_t>(n, (uint64_t)lo + per);
}

// ---- environment of a rank ------------------------------------------------------------------------------------------------------
static bool parse_fd_list(const char *s, std::vector<int> &out)
{
    out.clear();
    if (!s) return false;
    while (*s) {
        char *end = nullptr;
        const long v = strtol(s, &end, 10);
        if (end == s || v < 0) return false;
        out.push_back((int)v);
        s = *end == ',' ? end + 1 : end;
        if (*end && *end != ',') return false;
    }
    return true;
}

bool rank_env(RankEnv &out)
{
    const char *r = getenv("TM_RANK"), *w = getenv("TM_WORLD");
    if (!r || !w) return false;
    out.rank = atoi(r); out.world = atoi(w);
    if (out.world < 1 || out.rank < 0 || out.rank >= out.world) throw std::runtime_error("TM_RANK / TM_WORLD: not a rank of this world");
    const size_t want = out.rank == 0 ? (size_t)out.world - 1 : 1;
    if (out.world > 1 && (!parse_fd_list(getenv("TM_RANK_UP"), out.up) || !parse_fd_list(getenv("TM_RANK_DOWN"), out.down) || out.up.size() != want || out.down.size() != want))
        throw std::runtime_error("TM_RANK_UP / TM_RANK_DOWN: the launcher's pipes are missing");
    return true;
}

// ---- launcher ---------------------------------------------------------------------------------------------------------------------
static std::string fd_list(const std::vector<int> &v)
{
    std::string s;
    for (size_t i = 0; i < v.size(); ++i) s += (i ? "," : "") + std::to_string(v[i]);
    return s;
}

// a signal that asks the launcher to stop (SIGTERM, SIGINT, SIGHUP) stops the ranks: they are this process's children and nobody else's
static volatile sig_atomic_t g_launcher_signal = 0;
static void launcher_signal(int sig) { g_launcher_signal = sig; }

int launch_ranks(char **argv, int world, double timeout_s, const char *self)
{
    if (world < 1) return 2;
    for (int sig : {SIGTERM, SIGINT, SIGHUP}) {
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_handler = launcher_signal;
        sigaction(sig, &sa, nullptr);
    }
    // up[r] : rank r -> rank 0 (the vector, pipe transport); down[r] : rank 0 -> rank r (the communicator id, RCCL transport)
    std::vector<int> up_r(world, -1), up_w(world, -1), down_r(world, -1), down_w(world, -1);
    for (int r = 1; r < world; ++r) {
        int a[2], b[2];
        if (pipe(a) != 0 || pipe(b) != 0) { perror("turbo-metrics: pipe"); return 1; }
        up_r[r] = a[0]; up_w[r] = a[1]; down_r[r] = b[0]; down_w[r] = b[1];
    }
    std::vector<pid_t> pids(world, -1);
    for (int r = 0; r < world; ++r) {
        const pid_t pid = fork();
        if (pid < 0) {
            perror("turbo-metrics: fork");
            for (int k = 0; k < r; ++k) kill(pids[k], SIGTERM);
            for (int k = 0; k < r; ++k) waitpid(pids[k], nullptr, 0);
            return 1;
        }
        if (pid == 0) { // the rank: keep its own pipe ends, close the others, exec
            std::vector<int> up, down;
            for (int k = 1; k < world; ++k) {
                if (r == 0) { up.push_back(up_r[k]); down.push_back(down_w[k]); close(up_w[k]); close(down_r[k]); }
                else if (k == r) { up.push_back(up_w[k]); down.push_back(down_r[k]); close(up_r[k]); close(down_w[k]); }
                else { close(up_r[k]); close(up_w[k]); close(down_r[k]); close(down_w[k]); }
            }
            setenv("TM_RANK", std::to_string(r).c_str(), 1);
            setenv("TM_WORLD", std::to_string(world).c_str(), 1);
            setenv("TM_RANK_UP", fd_list(up).c_str(), 1);
            setenv("TM_RANK_DOWN", fd_list(down).c_str(), 1);
            setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0); // dmabuf IPC: what RCCL needs on this driver (kept when already set)
            if (r != 0) { // only rank 0 speaks on stdout
                const int nul = open("/dev/null", O_WRONLY);
                if (nul >= 0) { dup2(nul, STDOUT_FILENO); close(nul); }
            }
            for (int sig : {SIGTERM, SIGINT, SIGHUP}) signal(sig, SIG_DFL); // (the launcher's handlers are not the ranks')
            execv(self ? self : "/proc/self/exe", argv);
            perror("turbo-metrics: exec");
            _exit(127);
        }
        pids[r] = pid;
    }
    for (int r = 1; r < world; ++r) { close(up_r[r]); close(up_w[r]); close(down_r[r]); close(down_w[r]); } // the ranks hold them now
    // wait for exactly these PIDs; the first failure (or the timeout) ends the others
    const auto t0 = std::chrono::steady_clock::now();
    int rc = 0, alive = world;
    std::vector<bool> done(world, false);
    bool told = false;
    int lost_rank = -1; // a rank that exited with RANK_PEER_LOST while no other failure is known yet
    std::chrono::steady_clock::time_point told_at, lost_at;
    while (alive > 0) {
        bool progress = false;
        for (int r = 0; r < world; ++r) {
            if (done[r]) continue;
            int st = 0;
            const pid_t got = waitpid(pids[r], &st, WNOHANG);
            if (got == 0) continue;
            done[r] = true; --alive; progress = true;
            int code = 1;
            if (got == pids[r]) code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
            // the first failure decides -- unless it is a rank that only lost its peer: the peer's own failure is looked for first
            if (code != 0 && !told && (rc == 0 || rc == RANK_PEER_LOST)) {
                if (code == RANK_PEER_LOST && rc == 0) { rc = code; lost_rank = r; lost_at = std::chrono::steady_clock::now(); }
                else if (code != RANK_PEER_LOST) {
                    rc = code; lost_rank = -1;
                    fprintf(stderr, "ERROR turbo_metrics_cli: rank %d of %d failed (exit code %d)\n", r, world, code);
                }
            }
        }
        if (lost_rank >= 0 && (alive == 0 || std::chrono::duration<double>(std::chrono::steady_clock::now() - lost_at).count() > 0.5)) {
            fprintf(stderr, "ERROR turbo_metrics_cli: rank %d of %d failed (it lost another rank)\n", lost_rank, world);
            lost_rank = -1;
        }
        const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rc == 0 && g_launcher_signal) {
            rc = 128 + (int)g_launcher_signal;
            fprintf(stderr, "ERROR turbo_metrics_cli: signal %d: stopping the ranks\n", (int)g_launcher_signal);
        }
        if (rc == 0 && timeout_s > 0.0 && elapsed > timeout_s) {
            rc = 124;
            fprintf(stderr, "ERROR turbo_metrics_cli: the ranks did not finish within %.0f s\n", timeout_s);
        }
        if (rc != 0 && lost_rank < 0 && alive > 0 && !told) {
            for (int r = 0; r < world; ++r) if (!done[r]) kill(pids[r], SIGTERM);
            told = true; told_at = std::chrono::steady_clock::now();
        }
        if (told && alive > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - told_at).count() > 10.0)
            for (int r = 0; r < world; ++r) if (!done[r]) kill(pids[r], SIGKILL);
        if (!progress && alive > 0) std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    return rc;
}

// ---- transports -------------------------------------------------------------------------------------------------------------------
static void write_all(int fd, const void *p, size_t n, const char *what)
{
    const char *c = (const char *)p;
    while (n) {
        const ssize_t k = write(fd, c, n);
        if (k < 0 && errno == EINTR) continue;
        if (k < 0 && errno == EPIPE) throw RankPeerLost(std::string(what) + ": the other rank closed the pipe");
        if (k <= 0) throw std::runtime_error(std::string(what) + ": " + strerror(errno));
        c += k; n -= (size_t)k;
    }
}
static void read_all(int fd, void *p, size_t n, const char *what)
{
    char *c = (char *)p;
    while (n) {
        const ssize_t k = read(fd, c, n);
        if (k < 0 && errno == EINTR) continue;
        if (k == 0) throw RankPeerLost(std::string(what) + ": the other rank closed the pipe");
        if (k < 0) throw std::runtime_error(std::string(what) + ": " + strerror(errno));
        c += k; n -= (size_t)k;
    }
}

namespace {

// RCCL writes a version banner to STDOUT when a communicator is set up (seen with 2.27.7: "RCCL version : ...", five lines), and rank 0's
// stdout is the scores and nothing else: while a call into the library runs, file descriptor 1 points at stderr.
class StdoutToStderr {
public:
    StdoutToStderr()
    {
        fflush(stdout);
        saved_ = dup(STDOUT_FILENO);
        if (saved_ >= 0) dup2(STDERR_FILENO, STDOUT_FILENO);
    }
    ~StdoutToStderr()
    {
        fflush(stdout);
        if (saved_ >= 0) { dup2(saved_, STDOUT_FILENO); close(saved_); }
    }
private:
    int saved_ = -1;
};

class PipeTransport : public RankTransport {
public:
    explicit PipeTransport(const RankEnv &e) : env_(e) { signal(SIGPIPE, SIG_IGN); } // a dead rank 0 is an error message, not a signal
    const char *name() const override { return "pipe"; }
    void reduce_sum_to_root(std::vector<double> &v) override
    {
        if (env_.world == 1) return;
        if (env_.rank != 0) {
            const uint64_t n = v.size();
            write_all(env_.up[0], &n, sizeof n, "reduce (pipe): write");
            write_all(env_.up[0], v.data(), v.size() * sizeof(double), "reduce (pipe): write");
            return;
        }
        std::vector<double> in(v.size());
        for (int r = 1; r < env_.world; ++r) { // rank order: the sum is the same from run to run
            uint64_t n = 0;
            read_all(env_.up[r - 1], &n, sizeof n, "reduce (pipe): read");
            if (n != v.size()) throw std::runtime_error("reduce (pipe): rank " + std::to_string(r) + " sent a vector of another length");
            read_all(env_.up[r - 1], in.data(), in.size() * sizeof(double), "reduce (pipe): read");
            for (size_t i = 0; i < v.size(); ++i) v[i] += in[i];
        }
    }
private:
    RankEnv env_;
};

class RcclTransport : public RankTransport {
public:
    explicit RcclTransport(const RankEnv &e) : env_(e)
    {
        const char *path = getenv("TM_RCCL_LIB");
        lib_ = dlopen(path ? path : "libturbometrics_rccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib_) throw std::runtime_error(std::string("--ranks over RCCL needs libturbometrics_rccl.so: ") + dlerror());
        get_id_ = (int (*)(void *))dlsym(lib_, "tm_comm_get_unique_id");
        init_ = (int (*)(tm_comm **, int, int, const void *))dlsym(lib_, "tm_comm_init");
        reduce_ = (int (*)(tm_comm *, double *, size_t, int))dlsym(lib_, "tm_comm_reduce_sum_f64");
        destroy_ = (void (*)(tm_comm *))dlsym(lib_, "tm_comm_destroy");
        err_ = (const char *(*)(void))dlsym(lib_, "tm_comm_last_error");
        if (!get_id_ || !init_ || !reduce_ || !destroy_ || !err_) throw std::runtime_error("libturbometrics_rccl.so does not export include/turbo_metrics_comm.h");
        signal(SIGPIPE, SIG_IGN);
        // the communicator's id: made by rank 0, carried to the others over the launcher's pipes
        StdoutToStderr quiet;
        char id[TM_COMM_ID_BYTES];
        if (env_.rank == 0) {
            if (get_id_(id)) throw std::runtime_error(std::string("tm_comm_get_unique_id: ") + err_());
            for (int r = 1; r < env_.world; ++r) write_all(env_.down[r - 1], id, sizeof id, "communicator id: write");
        } else read_all(env_.down[0], id, sizeof id, "communicator id: read");
        if (init_(&comm_, env_.world, env_.rank, id)) throw std::runtime_error(std::string("tm_comm_init: ") + err_());
        // the first collective of a communicator pays for RCCL's lazy set-up (channels, kernels: ~0.1 s): an 8-byte reduce HERE, while the ranks are
        // still setting up, so that the one reduce of the scores costs what a reduce costs
        double warm = 0.0;
        if (reduce_(comm_, &warm, 1, 0)) throw std::runtime_error(std::string("tm_comm_reduce_sum_f64 (warm-up): ") + err_());
    }
    ~RcclTransport() override
    {
        StdoutToStderr quiet;
        if (comm_) destroy_(comm_);
        // (the library stays loaded: RCCL's own threads may outlive the communicator)
    }
    const char *name() const override { return "rccl"; }
    void reduce_sum_to_root(std::vector<double> &v) override
    {
        StdoutToStderr quiet;
        if (reduce_(comm_, v.data(), v.size(), 0)) throw std::runtime_error(std::string("tm_comm_reduce_sum_f64: ") + err_());
    }
private:
    RankEnv env_;
    void *lib_ = nullptr;
    tm_comm *comm_ = nullptr;
    int (*get_id_)(void *) = nullptr;
    int (*init_)(tm_comm **, int, int, const void *) = nullptr;
    int (*reduce_)(tm_comm *, double *, size_t, int) = nullptr;
    void (*destroy_)(tm_comm *) = nullptr;
    const char *(*err_)(void) = nullptr;
};

} // namespace

std::unique_ptr<RankTransport> make_rank_transport(const RankEnv &env, const std::string &kind)
{
    std::string k = kind;
    if (k.empty()) { const char *e = getenv("TM_RANK_TRANSPORT"); k = e && *e ? e : "rccl"; }
    if (k == "pipe") return std::make_unique<PipeTransport>(env);
    if (k == "rccl") return std::make_unique<RcclTransport>(env);
    throw std::runtime_error("TM_RANK_TRANSPORT: '" + k + "' (possible values: rccl, pipe)");
}

// ---- score vector -----------------------------------------------------------------------------------------------------------------
ScoreVector::ScoreVector(const Metrics &m, uint32_t total_indices) : metrics(m), total(total_indices)
{
    v.assign((size_t)total * stride() + 1, 0.0);
}

size_t ScoreVector::stride() const
{
    return 1 + (metrics.psnr ? 1 : 0) + (metrics.ssim ? 1 : 0) + (metrics.msssim ? 1 : 0) + (metrics.ssimulacra2 ? 1 : 0);
}

void ScoreVector::put(uint32_t i, const FrameScores &s)
{
    if (i >= total) throw std::out_of_range("ScoreVector::put");
    double *p = v.data() + (size_t)i * stride();
    *p++ = 1.0;
    if (metrics.psnr) *p++ = s.psnr.value_or(0.0);
    if (metrics.ssim) *p++ = s.ssim.value_or(0.0);
    if (metrics.msssim) *p++ = s.msssim.value_or(0.0);
    if (metrics.ssimulacra2) *p++ = s.ssimulacra2.value_or(0.0);
}

std::vector<FrameScores> ScoreVector::frames() const
{
    std::vector<FrameScores> out;
    for (uint32_t i = 0; i < total; ++i) {
        const double *p = v.data() + (size_t)i * stride();
        if (*p++ != 1.0) continue; // every flag is set by exactly one rank
        FrameScores s;
        if (metrics.psnr) s.psnr = *p++;
        if (metrics.ssim) s.ssim = *p++;
        if (metrics.msssim) s.msssim = *p++;
        if (metrics.ssimulacra2) s.ssimulacra2 = *p++;
        out.push_back(s);
    }
    return out;
}

} // namespace tm_host
