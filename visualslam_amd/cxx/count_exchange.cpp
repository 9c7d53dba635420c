#include "count_exchange.hpp"

#include <arpa/inet.h>
#include <hip/hip_runtime_api.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <rccl/rccl.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "vslam_cxx.hpp"

namespace vslam {

namespace {
int env_int(const char* name, int def) {
    const char* v = std::getenv(name);
    return v && *v ? std::atoi(v) : def;
}
[[noreturn]] void fail(const std::string& what) { throw Error(VSLAM_ERR_HIP, what); }
void hipx(hipError_t e, const char* what) {
    if (e != hipSuccess) fail(std::string(what) + ": " + hipGetErrorString(e));
}
void ncclx(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) fail(std::string(what) + ": " + ncclGetErrorString(r));
}
#define HIPX(e) hipx((e), #e)
#define NCCLX(e) ncclx((e), #e)

// 60 s to let slow ranks start; VSLAM_RDV_TIMEOUT_MS overrides (tests)
const int kTimeoutMs = env_int("VSLAM_RDV_TIMEOUT_MS", 60000);

// What a rank says first: the id is handed only to a peer of THIS job (the token comes from VSLAM_JOB_TOKEN, or else from
// what every rank of a torchrun job shares: MASTER_ADDR, MASTER_PORT, WORLD_SIZE and the launcher's run id).
struct Hello {
    uint32_t magic;
    uint32_t token;
    int32_t rank;
};
constexpr uint32_t kHelloMagic = 0x56534c4du;  // "VSLM"
uint32_t job_token() {
    std::string s;
    for (const char* k : {"VSLAM_JOB_TOKEN", "TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE"})
        if (const char* v = std::getenv(k)) s += std::string(k) + "=" + v + ";";
    uint32_t h = 2166136261u;  // FNV-1a
    for (unsigned char c : s) h = (h ^ c) * 16777619u;
    return h;
}

bool io_all(int fd, void* buf, size_t n, bool writing, int timeout_ms = -1) {
    char* p = static_cast<char*>(buf);
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms >= 0 ? timeout_ms : kTimeoutMs);
    while (n) {
        pollfd pf{fd, (short)(writing ? POLLOUT : POLLIN), 0};
        const int left = (int)std::chrono::duration_cast<std::chrono::milliseconds>(t_end - std::chrono::steady_clock::now()).count();
        if (left <= 0 || ::poll(&pf, 1, left) <= 0) return false;
        const ssize_t k = writing ? ::send(fd, p, n, MSG_NOSIGNAL) : ::recv(fd, p, n, 0);
        if (k < 0 && (errno == EINTR || errno == EAGAIN)) continue;
        if (k <= 0) return false;
        p += k;
        n -= (size_t)k;
    }
    return true;
}
}  // namespace

RankEnv RankEnv::from_environment() {
    RankEnv e;
    e.rank = env_int("RANK", 0);
    e.world = env_int("WORLD_SIZE", 1);
    e.local_rank = env_int("LOCAL_RANK", e.rank);
    if (const char* a = std::getenv("MASTER_ADDR"); a && *a) e.addr = a;
    if (e.addr == "localhost") e.addr = "127.0.0.1";
    e.rdv_port = env_int("VSLAM_RDV_PORT", env_int("MASTER_PORT", 29533) + 1);
    if (e.world < 1 || e.rank < 0 || e.rank >= e.world) throw Error(VSLAM_ERR_INVALID, "RANK / WORLD_SIZE out of range");
    return e;
}

// The rendezvous itself; keep != nullptr: the connections stay open (rank 0: keep[rank] = socket of that peer, others:
// keep[0] = the socket to rank 0) instead of being closed after the hand-off.
static void tcp_rendezvous(const RankEnv& env, void* bytes, size_t n, std::vector<int>* keep);
void tcp_broadcast_from_rank0(const RankEnv& env, void* bytes, size_t n) { tcp_rendezvous(env, bytes, n, nullptr); }

static void tcp_rendezvous(const RankEnv& env, void* bytes, size_t n, std::vector<int>* keep) {
    if (keep) keep->assign(env.rank == 0 ? env.world : 1, -1);
    if (env.world <= 1) return;
    sockaddr_in sa{};
    sa.sin_family = AF_INET;
    sa.sin_port = htons((uint16_t)env.rdv_port);
    if (::inet_pton(AF_INET, env.addr.c_str(), &sa.sin_addr) != 1) fail("rendezvous: MASTER_ADDR must be an IPv4 address, got " + env.addr);
    if (env.rank == 0) {
        const int ls = ::socket(AF_INET, SOCK_STREAM, 0);
        if (ls < 0) fail("rendezvous: socket()");
        const int one = 1;
        ::setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        if (::bind(ls, reinterpret_cast<sockaddr*>(&sa), sizeof(sa)) != 0 || ::listen(ls, env.world) != 0) {
            const std::string msg = std::string("rendezvous: cannot listen on ") + env.addr + ":" + std::to_string(env.rdv_port) + ": " + std::strerror(errno);
            ::close(ls);
            fail(msg);
        }
        // every peer introduces itself with {magic, job token, rank}; each rank is served exactly once.  A connection
        // that does not (a port scanner, a late rank of another job, a duplicate) is closed and ignored: the loop keeps
        // accepting until every rank has been served or the deadline passes.
        std::vector<char> seen(env.world, 0);
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(kTimeoutMs);
        int served = 1, ignored = 0;
        while (served < env.world) {
            pollfd pf{ls, POLLIN, 0};
            const int left = (int)std::chrono::duration_cast<std::chrono::milliseconds>(t_end - std::chrono::steady_clock::now()).count();
            if (left <= 0 || ::poll(&pf, 1, left) <= 0) {
                ::close(ls);
                fail("rendezvous: rank 0 timed out waiting for " + std::to_string(env.world - served) + " rank(s)" +
                     (ignored ? " (" + std::to_string(ignored) + " foreign connection(s) ignored)" : ""));
            }
            const int fd = ::accept(ls, nullptr, nullptr);
            if (fd < 0) {
                // transient: the peer gave up between poll() and accept(), or a signal.  Anything else (EMFILE, ENFILE,
                // ENOMEM ...) leaves the listening socket readable - retrying would spin to the deadline and then blame the ranks
                if (errno == EINTR || errno == ECONNABORTED || errno == EAGAIN || errno == EWOULDBLOCK) continue;
                const std::string msg = std::string("rendezvous: accept(): ") + std::strerror(errno);
                ::close(ls);
                fail(msg);
            }
            Hello h{};
            const bool hello_ok = io_all(fd, &h, sizeof(h), false, 2000) && h.magic == kHelloMagic && h.token == job_token() && h.rank > 0 &&
                                  h.rank < env.world && !seen[h.rank];
            if (!hello_ok) {  // not one of this job's ranks
                ::close(fd);
                ++ignored;
                continue;
            }
            if (!io_all(fd, bytes, n, true)) {  // a rank of this job whose payload could not be sent: it will not come back
                const std::string msg = "rendezvous: sending to rank " + std::to_string(h.rank) + " failed: " + std::strerror(errno);
                ::close(fd);
                ::close(ls);
                fail(msg);
            }
            if (keep) {
                const int one_ = 1;
                ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one_, sizeof(one_));
                (*keep)[h.rank] = fd;
            } else {
                ::close(fd);
            }
            seen[h.rank] = 1;
            ++served;
        }
        ::close(ls);
    } else {
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(kTimeoutMs);
        for (;;) {  // rank 0 may not be listening yet
            const int fd = ::socket(AF_INET, SOCK_STREAM, 0);
            if (fd < 0) fail("rendezvous: socket()");
            if (::connect(fd, reinterpret_cast<sockaddr*>(&sa), sizeof(sa)) == 0) {
                Hello me{kHelloMagic, job_token(), env.rank};
                const bool ok = io_all(fd, &me, sizeof(me), true) && io_all(fd, bytes, n, false);
                if (!ok) {
                    ::close(fd);
                    fail("rendezvous: rank " + std::to_string(env.rank) + " lost the connection to rank 0");
                }
                if (keep) {
                    const int one_ = 1;
                    ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one_, sizeof(one_));
                    (*keep)[0] = fd;
                } else {
                    ::close(fd);
                }
                return;
            }
            ::close(fd);
            if (std::chrono::steady_clock::now() > t_end)
                fail("rendezvous: rank " + std::to_string(env.rank) + " could not reach rank 0 at " + env.addr + ":" + std::to_string(env.rdv_port));
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
        }
    }
}

CountExchange::Backend CountExchange::backend_from_environment() {
    const char* e = std::getenv("VSLAM_COUNT_BACKEND");
    return e && std::string(e) == "tcp" ? Backend::Tcp : Backend::Rccl;
}

CountExchange::CountExchange(const RankEnv& env, int device, Backend backend) : env_(env), device_(device), backend_(backend) {
    HIPX(hipSetDevice(device_));
    if (backend_ == Backend::Rccl) {
        ncclUniqueId id;
        std::memset(&id, 0, sizeof(id));
        if (env_.rank == 0) NCCLX(ncclGetUniqueId(&id));
        tcp_broadcast_from_rank0(env_, &id, sizeof(id));
        ncclComm_t comm;
        NCCLX(ncclCommInitRank(&comm, env_.world, id, env_.rank));
        comm_ = comm;
    } else {
        char hello[8] = "vslamrh";  // same hand-off as the RCCL id, the sockets stay open for the rounds
        tcp_rendezvous(env_, hello, sizeof(hello), &fds_);
    }
    HIPX(hipMalloc((void**)&d_all_, sizeof(uint64_t) * 2 * (size_t)env_.world));
    HIPX(hipMemset(d_all_, 0, sizeof(uint64_t) * 2 * (size_t)env_.world));
    HIPX(hipMalloc((void**)&d_scratch_, 2 * sizeof(double)));
}

CountExchange::~CountExchange() {
    (void)hipSetDevice(device_);
    (void)hipDeviceSynchronize();
    if (comm_) (void)ncclCommDestroy((ncclComm_t)comm_);
    for (int fd : fds_)
        if (fd >= 0) ::close(fd);
    if (d_all_) (void)hipFree(d_all_);
    if (d_scratch_) (void)hipFree(d_scratch_);
}

void CountExchange::tcp_round(const void* mine, size_t n, void* table) {
    char* T = static_cast<char*>(table);
    if (env_.rank == 0) {
        std::memcpy(T, mine, n);
        for (int r = 1; r < env_.world; ++r)
            if (!io_all(fds_[r], T + (size_t)r * n, n, false)) fail("tcp exchange: lost rank " + std::to_string(r));
        for (int r = 1; r < env_.world; ++r)
            if (!io_all(fds_[r], T, n * (size_t)env_.world, true)) fail("tcp exchange: lost rank " + std::to_string(r));
    } else if (env_.world > 1) {
        if (!io_all(fds_[0], const_cast<void*>(mine), n, true) || !io_all(fds_[0], T, n * (size_t)env_.world, false)) fail("tcp exchange: lost rank 0");
    } else {
        std::memcpy(T, mine, n);
    }
}

void CountExchange::all_gather_async(const uint64_t* d_local, void* stream) {
    if (backend_ == Backend::Rccl) {
        NCCLX(ncclAllGather(d_local, d_all_, 2, ncclUint64, (ncclComm_t)comm_, (hipStream_t)stream));
        return;
    }
    // rehearsal backend: through the host, synchronous
    uint64_t mine[2];
    HIPX(hipMemcpyAsync(mine, d_local, 16, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPX(hipStreamSynchronize((hipStream_t)stream));
    std::vector<uint64_t> table(2 * (size_t)env_.world);
    tcp_round(mine, 16, table.data());
    HIPX(hipMemcpyAsync(d_all_, table.data(), table.size() * 8, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIPX(hipStreamSynchronize((hipStream_t)stream));
}

std::vector<uint64_t> CountExchange::fetch(void* stream) {
    std::vector<uint64_t> all(2 * (size_t)env_.world);
    HIPX(hipMemcpyAsync(all.data(), d_all_, all.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPX(hipStreamSynchronize((hipStream_t)stream));
    return all;
}

double CountExchange::max_over_ranks(double v, void* stream) {
    const hipStream_t s = (hipStream_t)stream;
    if (backend_ == Backend::Tcp) {
        HIPX(hipStreamSynchronize(s));
        std::vector<double> table((size_t)env_.world);
        tcp_round(&v, sizeof(double), table.data());
        double m = table[0];
        for (double x : table) m = x > m ? x : m;
        return m;
    }
    HIPX(hipMemcpyAsync(d_scratch_, &v, sizeof(double), hipMemcpyHostToDevice, s));
    NCCLX(ncclAllReduce(d_scratch_, d_scratch_ + 1, 1, ncclDouble, ncclMax, (ncclComm_t)comm_, s));
    double out = v;
    HIPX(hipMemcpyAsync(&out, d_scratch_ + 1, sizeof(double), hipMemcpyDeviceToHost, s));
    HIPX(hipStreamSynchronize(s));
    return out;
}

void CountExchange::barrier(void* stream) { (void)max_over_ranks(0.0, stream); }

}  // namespace vslam
