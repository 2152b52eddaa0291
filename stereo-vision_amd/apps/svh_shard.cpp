// svh_shard -- the multi-GPU host driver in C++ (BASELINE north_star: "stereo pairs shard naturally
// one-pair-per-GPU across the 8 GPUs of one node with RCCL over xGMI only for the tiny pose/result gather").
//
// One process per GPU, as everywhere in this repo.  A launcher process forks + execs N rank processes of itself
// (it never touches HIP); rank r binds device r mod <devices>, takes its slice of the pairs (Elas::process keeps no
// state between calls -- libelas/src/elas.cpp:32-170 -- so the slices are independent and the data path has NO
// collective), runs them through the C-ABI (svh_elas_process_batch_device: images and maps resident in HBM, the
// contract of bench.py's `value`), and the ranks exchange one small record each:
//   gather = rccl   ncclAllGather of the records over an RCCL communicator (ncclCommInitRank; the unique id travels
//                   from rank 0 through the launcher's sockets -- the only thing those sockets carry in this mode)
//   gather = pipes  the same all-gather through the launcher's sockets: RCCL refuses two ranks on one device, so this
//                   is what runs when ranks share a GPU (1-GPU boxes, tests)
//   gather = auto   rccl when every rank has a device of its own, else pipes
// Timing follows bench.py's contract: warm-up steps, a gather as barrier, K timed steps, max over ranks.
// Rank 0 prints ONE JSON line: whole-job pairs/s, per-rank records, and FNV-1a sums of D1 / D2 of the first pairs of
// every rank (replicated inputs: they must be equal across ranks; tests compare them with tests/golden/*.npz).
//
//   svh_shard --ranks N [--pairs-per-rank B | --total P] [--steps K] [--warmup W] [--gather auto|rccl|pipes]
//             [--images DIR] [--lanes L] [--cores-per-rank C] [--no-bind] [--spinup-ms T] [--timeout-s S] [--selftest-gather]
// Round 6 (8-GPU readiness): a rank binds its threads to at most C (16) CPUs of its GPU's NUMA node, the record carries
// the GPU's PCI address, NUMA node, CPUs bound and CPU time; rank 0 checks "one rank per device, distinct PCI addresses"
// whenever there are at least as many devices as ranks, and prints per-rank host cores beside the budget.
#include <errno.h>
#include <poll.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "svh.h"

namespace {

constexpr int kRecWords = 24;   // one record = 24 x uint64 (192 bytes)
// R_BUS: the GPU's PCI address (domain << 16 | bus << 8 | device << 3 | function); R_NUMA: NUMA node + 1 (0: unknown) |
// CPUs the rank bound itself to << 16 | CPUs of the node << 32; R_CPU_NS: process CPU time over the timed steps
enum { R_RANK, R_DEVICE, R_PAIRS, R_NS, R_STATUS, R_D1, R_D2 = R_D1 + 4, R_LO = R_D2 + 4, R_HI, R_BUS, R_NUMA, R_CPU_NS, R_SPARE };
static_assert(R_SPARE < kRecWords, "record layout");

uint64_t pack_bus_id(const char* s) {   // "0000:c5:00.0"
    unsigned dom = 0, bus = 0, dev = 0, fn = 0;
    if (sscanf(s, "%x:%x:%x.%x", &dom, &bus, &dev, &fn) != 4) return 0;
    return (uint64_t)dom << 16 | (uint64_t)(bus & 0xFF) << 8 | (uint64_t)(dev & 0x1F) << 3 | (uint64_t)(fn & 7);
}
double cpu_s() {
    timespec t;
    clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

struct Options {
    int ranks = 1, per_rank = 256, lanes = 6, total = 0, steps = 5, warmup = 2, rank = -1, fd = -1, timeout_s = 900, spinup_ms = 1000;
    int cores_per_rank = 16, bind = 1;   // host-core budget of a rank (its threads are bound to that many CPUs of its GPU's NUMA node)
    std::string gather = "auto", images = "tests/golden";
    bool selftest = false;
};

double now_s() {
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

bool write_all(int fd, const void* p, size_t n) {
    const char* c = static_cast<const char*>(p);
    while (n) {
        const ssize_t k = write(fd, c, n);
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) return false;
        c += k;
        n -= (size_t)k;
    }
    return true;
}
bool read_all(int fd, void* p, size_t n) {
    char* c = static_cast<char*>(p);
    while (n) {
        const ssize_t k = read(fd, c, n);
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) return false;
        c += k;
        n -= (size_t)k;
    }
    return true;
}

// contiguous slice [lo, hi) of n items for `rank`; sizes differ by at most one (svhip/shard.py: shard_range)
void shard_range(int n, int rank, int world, int* lo, int* hi) {
    const int base = n / world, rem = n % world;
    *lo = rank * base + (rank < rem ? rank : rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
}

uint64_t fnv1a(const void* p, size_t n) {
    const unsigned char* c = static_cast<const unsigned char*>(p);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) h = (h ^ c[i]) * 1099511628211ull;
    return h;
}

bool read_pgm(const std::string& path, std::vector<uint8_t>& img, int32_t& w, int32_t& h) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[3] = {0, 0, 0};
    int maxv = 0;
    bool ok = fscanf(f, "%2s %d %d %d", magic, &w, &h, &maxv) == 4 && !strcmp(magic, "P5") && w > 0 && h > 0;
    if (ok) {
        fgetc(f);
        img.resize((size_t)w * h);
        ok = fread(img.data(), 1, img.size(), f) == img.size();
    }
    fclose(f);
    return ok;
}

// ---------------------------------------------------------------------------------------------------------------
// launcher: relays messages between the ranks' sockets.  A message = uint32 words (count of uint64 payload words)
// followed by the payload; every round, each rank sends one message and receives the concatenation of all of them in
// rank order.  The unique id of RCCL travels as one such round (rank 0's payload is the id, the others' is empty).
// ---------------------------------------------------------------------------------------------------------------
int launcher(const Options& o, char** argv) {
    const int N = o.ranks;
    std::vector<pid_t> pid(N, -1);
    std::vector<int> fd(N, -1);
    std::vector<char> reaped(N, 0);
    signal(SIGPIPE, SIG_IGN);
    for (int r = 0; r < N; r++) {
        int sv[2];
        if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv) != 0) {
            perror("socketpair");
            return 1;
        }
        const pid_t p = fork();
        if (p < 0) {
            perror("fork");
            return 1;
        }
        if (p == 0) {
            close(sv[0]);
            for (int q = 0; q < r; q++) close(fd[q]);
            std::vector<std::string> a;
            for (char** s = argv; *s; s++) a.push_back(*s);
            a.push_back("--rank");
            a.push_back(std::to_string(r));
            a.push_back("--fd");
            a.push_back(std::to_string(sv[1]));
            std::vector<char*> av;
            for (auto& s : a) av.push_back(const_cast<char*>(s.c_str()));
            av.push_back(nullptr);
            execv("/proc/self/exe", av.data());
            perror("execv");
            _exit(127);
        }
        close(sv[1]);
        pid[r] = p;
        fd[r] = sv[0];
    }
    auto stop_all = [&]() {
        for (int r = 0; r < N; r++)
            if (pid[r] > 0 && !reaped[r]) kill(pid[r], SIGTERM);   // (exact PIDs of our own children)
    };
    // a rank that dies leaves the others waiting in a collective: the launcher watches the children while it waits
    auto reap = [&]() -> bool {   // false: a rank ended badly
        bool ok = true;
        for (int r = 0; r < N; r++) {
            if (reaped[r]) continue;
            int st = 0;
            if (waitpid(pid[r], &st, WNOHANG) == pid[r]) {
                reaped[r] = 1;
                if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
                    fprintf(stderr, "svh_shard: rank %d failed (status 0x%x)\n", r, st);
                    ok = false;
                }
            }
        }
        return ok;
    };
    const double deadline = now_s() + (double)o.timeout_s;
    auto readable = [&](int f) -> bool {
        for (;;) {
            pollfd pf = {f, POLLIN, 0};
            const int k = poll(&pf, 1, 200);
            if (k > 0) return true;
            if (k < 0 && errno != EINTR) return false;
            if (!reap()) return false;
            if (now_s() > deadline) {
                fprintf(stderr, "svh_shard: timeout after %d s\n", o.timeout_s);
                return false;
            }
        }
    };
    bool failed = false;
    while (!failed) {
        // one round: a message from every rank (or the end of every rank)
        std::vector<std::vector<uint64_t>> msg(N);
        int closed = 0;
        for (int r = 0; r < N && !failed; r++) {
            uint32_t words = 0;
            if (!readable(fd[r])) { failed = true; break; }
            if (!read_all(fd[r], &words, sizeof words)) {   // EOF: the rank is done (all of them finish in the same round)
                closed++;
                continue;
            }
            if (words > 4096) { failed = true; break; }
            msg[r].resize(words);
            if (words && !read_all(fd[r], msg[r].data(), words * sizeof(uint64_t))) failed = true;
        }
        if (failed || closed == N) break;
        if (closed) { failed = true; break; }   // some ranks ended while others still gather
        std::vector<uint64_t> all;
        for (int r = 0; r < N; r++) all.insert(all.end(), msg[r].begin(), msg[r].end());
        const uint32_t words = (uint32_t)all.size();
        for (int r = 0; r < N && !failed; r++)
            if (!write_all(fd[r], &words, sizeof words) || (words && !write_all(fd[r], all.data(), words * sizeof(uint64_t))))
                failed = true;
    }
    if (failed) stop_all();
    int rc = failed ? 1 : 0;
    for (int r = 0; r < N; r++) {
        int st = 0;
        if (!reaped[r] && (waitpid(pid[r], &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0)) {
            if (!rc) fprintf(stderr, "svh_shard: rank %d failed (status 0x%x)\n", r, st);
            rc = 1;
        }
        close(fd[r]);
    }
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------
// rank side
// ---------------------------------------------------------------------------------------------------------------
struct Gather {
    int world = 1, rank = 0, fd = -1;
    bool rccl = false;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    uint64_t* d_send = nullptr;
    uint64_t* d_recv = nullptr;
    long rounds_rccl = 0, rounds_pipes = 0;

    // all-gather through the launcher: `words` uint64 from every rank, in rank order
    bool pipes(const uint64_t* mine, uint32_t words, std::vector<uint64_t>& all) {
        if (!write_all(fd, &words, sizeof words) || (words && !write_all(fd, mine, words * sizeof(uint64_t)))) return false;
        uint32_t got = 0;
        if (!read_all(fd, &got, sizeof got)) return false;
        all.resize(got);
        rounds_pipes++;
        return !got || read_all(fd, all.data(), got * sizeof(uint64_t));
    }
    bool init_rccl() {
        ncclUniqueId id;
        memset(&id, 0, sizeof id);
        static_assert(sizeof(ncclUniqueId) % sizeof(uint64_t) == 0, "id travels as uint64 words");
        if (rank == 0 && ncclGetUniqueId(&id) != ncclSuccess) return false;
        std::vector<uint64_t> all;
        if (!pipes(reinterpret_cast<const uint64_t*>(&id), rank == 0 ? sizeof id / sizeof(uint64_t) : 0, all)) return false;
        if (all.size() != sizeof id / sizeof(uint64_t)) return false;
        memcpy(&id, all.data(), sizeof id);
        if (ncclCommInitRank(&comm, world, id, rank) != ncclSuccess) return false;
        if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return false;
        if (hipMalloc((void**)&d_send, kRecWords * sizeof(uint64_t)) != hipSuccess) return false;
        if (hipMalloc((void**)&d_recv, (size_t)world * kRecWords * sizeof(uint64_t)) != hipSuccess) return false;
        rccl = true;
        return true;
    }
    // the records of all ranks, [world][kRecWords]
    bool run(const uint64_t* mine, std::vector<uint64_t>& all) {
        if (!rccl) return pipes(mine, kRecWords, all) && all.size() == (size_t)world * kRecWords;
        all.resize((size_t)world * kRecWords);
        if (hipMemcpyAsync(d_send, mine, kRecWords * sizeof(uint64_t), hipMemcpyHostToDevice, stream) != hipSuccess) return false;
        if (ncclAllGather(d_send, d_recv, kRecWords, ncclUint64, comm, stream) != ncclSuccess) return false;
        if (hipMemcpyAsync(all.data(), d_recv, all.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, stream) != hipSuccess) return false;
        rounds_rccl++;
        return hipStreamSynchronize(stream) == hipSuccess;
    }
    void close_all() {
        if (comm) ncclCommDestroy(comm);
        if (d_send) (void)hipFree(d_send);
        if (d_recv) (void)hipFree(d_recv);
        if (stream) (void)hipStreamDestroy(stream);
        if (fd >= 0) close(fd);
        fd = -1;
    }
};

int fail(Gather& g, const char* what) {
    fprintf(stderr, "svh_shard rank %d: %s (%s)\n", g.rank, what, svh_last_error());
    g.close_all();
    return 1;
}

// orchestration only (no device): three rounds of records through the launcher, checked on every rank
int selftest(Gather& g) {
    for (int round = 0; round < 3; round++) {
        uint64_t mine[kRecWords];
        for (int k = 0; k < kRecWords; k++) mine[k] = (uint64_t)g.rank * 1000 + (uint64_t)round * 100 + (uint64_t)k;
        std::vector<uint64_t> all;
        if (!g.run(mine, all)) return fail(g, "gather failed");
        for (int r = 0; r < g.world; r++)
            for (int k = 0; k < kRecWords; k++)
                if (all[(size_t)r * kRecWords + k] != (uint64_t)r * 1000 + (uint64_t)round * 100 + (uint64_t)k)
                    return fail(g, "gathered records differ from what the ranks sent");
    }
    if (g.rank == 0) printf("{\"selftest_gather\": \"ok\", \"ranks\": %d, \"rounds\": %ld}\n", g.world, g.rounds_pipes);
    g.close_all();
    return 0;
}

int rank_main(const Options& o) {
    Gather g;
    g.world = o.ranks;
    g.rank = o.rank;
    g.fd = o.fd;
    if (o.selftest) return selftest(g);

    const int ndev = svh_device_count();
    if (ndev < 1) return fail(g, "no HIP device: libsvhip has no CPU path");
    const int device = o.rank % ndev;
    if (svh_set_device(device) != SVH_OK || hipSetDevice(device) != hipSuccess) return fail(g, "cannot bind the device");
    const bool own_device = o.ranks <= ndev;
    // where the GPU sits, and this rank's threads next to it: the CPUs of the GPU's NUMA node, at most the rank's core
    // budget of them, inside whatever mask the process already has (include/svh.h: svh_bind_host_to_device).  Done before
    // the engine starts its workers, which inherit the mask.
    svh_device_topology topo;
    memset(&topo, 0, sizeof topo);
    topo.numa_node = -1;
    (void)svh_get_device_topology(device, &topo);
    const int bound = o.bind ? svh_bind_host_to_topology(&topo, o.cores_per_rank) : 0;
    if (o.gather == "rccl" && !own_device) return fail(g, "--gather rccl needs one device per rank");
    if ((o.gather == "rccl" || (o.gather == "auto" && own_device)) && !g.init_rccl()) return fail(g, "RCCL communicator");

    // the four 1242x375 crops of the golden set, cycled over this rank's slice
    static const char* names[4] = {"urban1", "urban2", "urban3", "urban4"};
    std::vector<uint8_t> L[4], R[4];
    int32_t W = 0, H = 0;
    for (int k = 0; k < 4; k++) {
        int32_t w = 0, h = 0, w2 = 0, h2 = 0;
        if (!read_pgm(o.images + "/" + names[k] + "_1242x375_left.pgm", L[k], w, h) ||
            !read_pgm(o.images + "/" + names[k] + "_1242x375_right.pgm", R[k], w2, h2) || w != w2 || h != h2 ||
            (k && (w != W || h != H)))
            return fail(g, "cannot read the golden crops (--images)");
        W = w;
        H = h;
    }
    int lo = 0, hi = o.per_rank;
    if (o.total > 0) shard_range(o.total, o.rank, o.ranks, &lo, &hi);
    else { lo = o.rank * o.per_rank; hi = lo + o.per_rank; }
    const int n = hi - lo;
    const size_t N = (size_t)W * H;
    uint8_t *dI1 = nullptr, *dI2 = nullptr;
    float *dD1 = nullptr, *dD2 = nullptr;
    const size_t nn = n > 0 ? (size_t)n : 1;
    if (hipMalloc((void**)&dI1, nn * N) != hipSuccess || hipMalloc((void**)&dI2, nn * N) != hipSuccess ||
        hipMalloc((void**)&dD1, nn * N * sizeof(float)) != hipSuccess || hipMalloc((void**)&dD2, nn * N * sizeof(float)) != hipSuccess)
        return fail(g, "hipMalloc of the resident images / maps");
    for (int i = 0; i < n; i++) {   // pair (lo + i) of the job is crop (lo + i) mod 4, whatever rank it lands on
        const int k = (lo + i) & 3;
        if (hipMemcpy(dI1 + (size_t)i * N, L[k].data(), N, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(dI2 + (size_t)i * N, R[k].data(), N, hipMemcpyHostToDevice) != hipSuccess)
            return fail(g, "upload of the crops");
    }
    svh_elas_params prm;
    svh_elas_params_default(&prm, SVH_ELAS_ROBOTICS);   // libelas/src/elas.h:86-116: the preset as it is, like bench.py
    // six workers per GPU, each double-buffered: the depth bench.py measures the device stage with (more only
    // stretches every kernel's in-run duration)
    svh_elas_set_lanes(o.lanes);
    svh_elas* e = svh_elas_create(&prm);
    if (!e) return fail(g, "svh_elas_create");
    const int32_t dims[3] = {W, H, W};
    std::vector<int32_t> status(nn, 0);
    int64_t bad = 0;
    auto step = [&]() -> bool {
        if (n <= 0) return true;
        const int32_t rc = svh_elas_process_batch_device(e, n, dI1, dI2, N, dD1, dD2, N * sizeof(float), dims, status.data());
        for (int i = 0; i < n; i++) bad += status[i] != SVH_OK;
        return rc == SVH_OK;
    };
    // untimed: lanes allocated, clocks up (bench.py: --spinup), then the warm-up steps
    const double t_spin = now_s();
    do {
        if (!step()) return fail(g, "spin-up step");
    } while (n > 0 && (now_s() - t_spin) * 1e3 < (double)o.spinup_ms);
    for (int i = 0; i < o.warmup; i++)
        if (!step()) return fail(g, "warm-up step");
    uint64_t rec[kRecWords];
    memset(rec, 0, sizeof rec);
    std::vector<uint64_t> all;
    if (!g.run(rec, all)) return fail(g, "barrier gather");   // every rank is warm: start together
    const double t0 = now_s(), c0 = cpu_s();
    for (int i = 0; i < o.steps; i++)
        if (!step()) return fail(g, "timed step");
    const double t1 = now_s(), c1 = cpu_s();
    // sums of the maps of this rank's first pair of every crop
    std::vector<float> h1(N), h2(N);
    rec[R_RANK] = (uint64_t)o.rank;
    rec[R_DEVICE] = (uint64_t)device;
    rec[R_PAIRS] = (uint64_t)n * (uint64_t)o.steps;
    rec[R_NS] = (uint64_t)((t1 - t0) * 1e9);
    rec[R_STATUS] = (uint64_t)bad;
    rec[R_LO] = (uint64_t)lo;
    rec[R_HI] = (uint64_t)hi;
    rec[R_BUS] = pack_bus_id(topo.pci_bus_id);
    rec[R_NUMA] = (uint64_t)(topo.numa_node + 1) | (uint64_t)(bound > 0 ? bound : 0) << 16 | (uint64_t)(topo.n_cpus > 0 ? topo.n_cpus : 0) << 32;
    rec[R_CPU_NS] = (uint64_t)((c1 - c0) * 1e9);
    for (int i = 0; i < n && i < 4; i++) {
        const int k = (lo + i) & 3;
        if (hipMemcpy(h1.data(), dD1 + (size_t)i * N, N * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(h2.data(), dD2 + (size_t)i * N, N * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(g, "download of the maps");
        rec[R_D1 + k] = fnv1a(h1.data(), N * sizeof(float));
        rec[R_D2 + k] = fnv1a(h2.data(), N * sizeof(float));
    }
    if (!g.run(rec, all)) return fail(g, "result gather");
    int rc = 0;
    if (o.rank == 0) {
        uint64_t pairs = 0, ns_max = 0, badsum = 0, ref1[4] = {0, 0, 0, 0}, ref2[4] = {0, 0, 0, 0};
        bool equal = true;
        for (int r = 0; r < o.ranks; r++) {
            const uint64_t* q = all.data() + (size_t)r * kRecWords;
            pairs += q[R_PAIRS];
            ns_max = q[R_NS] > ns_max ? q[R_NS] : ns_max;
            badsum += q[R_STATUS];
            for (int k = 0; k < 4; k++) {
                if (q[R_D1 + k] && !ref1[k]) { ref1[k] = q[R_D1 + k]; ref2[k] = q[R_D2 + k]; }
                if (q[R_D1 + k] && (q[R_D1 + k] != ref1[k] || q[R_D2 + k] != ref2[k])) equal = false;
            }
        }
        // one rank per device?  (RCCL needs it; a scaling run is only what it says if it holds)
        bool distinct = true;
        for (int r = 0; r < o.ranks; r++)
            for (int q2 = 0; q2 < r; q2++) {
                const uint64_t* a = all.data() + (size_t)r * kRecWords;
                const uint64_t* b = all.data() + (size_t)q2 * kRecWords;
                if (a[R_DEVICE] == b[R_DEVICE] || (a[R_BUS] && a[R_BUS] == b[R_BUS])) distinct = false;
            }
        if (own_device && !distinct) {
            fprintf(stderr, "svh_shard: %d ranks on %d devices, yet two ranks report the same device / PCI address\n", o.ranks, ndev);
            rc = 1;
        }
        const double secs = 1e-9 * (double)ns_max;
        printf("{\"driver\": \"svh_shard (C++ over the C-ABI)\", \"library\": \"%s\", \"metric\": \"stereo pairs/sec (ELAS %dx%d, ROBOTICS, D1+D2+LR)\", "
               "\"value\": %.1f, \"unit\": \"pairs/s\", \"ranks\": %d, \"devices\": %d, \"gather\": \"%s\", \"gather_rounds\": %ld, "
               "\"lanes\": %d, \"hw_queues\": \"%s\", \"steps\": %d, \"warmup\": %d, \"pairs\": %llu, \"seconds_max_over_ranks\": %.6f, \"scaling\": \"%s\", "
               "\"pairs_failed\": %llu, \"maps_equal_across_ranks\": %s, \"ranks_seen\": %d, \"one_rank_per_device\": %s, "
               "\"host_core_budget_per_rank\": %d, \"record_bytes\": %d, ",
               svh_version(), W, H, secs > 0 ? (double)pairs / secs : 0.0, o.ranks, ndev, g.rccl ? "rccl" : "pipes",
               g.rccl ? g.rounds_rccl : g.rounds_pipes, o.lanes, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "runtime default",
               o.steps, o.warmup, (unsigned long long)pairs, secs,
               o.total > 0 ? "strong" : "weak", (unsigned long long)badsum, equal ? "true" : "false", (int)(all.size() / kRecWords),
               distinct ? "true" : "false", o.cores_per_rank, (int)(kRecWords * sizeof(uint64_t)));
        printf("\"d1_fnv1a\": [");
        for (int k = 0; k < 4; k++) printf("%s\"%016llx\"", k ? ", " : "", (unsigned long long)ref1[k]);
        printf("], \"d2_fnv1a\": [");
        for (int k = 0; k < 4; k++) printf("%s\"%016llx\"", k ? ", " : "", (unsigned long long)ref2[k]);
        printf("], \"per_rank\": [");
        for (int r = 0; r < o.ranks; r++) {
            const uint64_t* q = all.data() + (size_t)r * kRecWords;
            const double s = 1e-9 * (double)q[R_NS];
            printf("%s{\"rank\": %llu, \"device\": %llu, \"pci_bus_id\": \"%04llx:%02llx:%02llx.%llx\", \"numa_node\": %d, \"cpus_of_node\": %d, "
                   "\"cpus_bound\": %d, \"host_cores_used\": %.2f, \"slice\": [%llu, %llu], \"pairs\": %llu, \"seconds\": %.6f, \"pairs_per_s\": %.1f}",
                   r ? ", " : "", (unsigned long long)q[R_RANK], (unsigned long long)q[R_DEVICE],
                   (unsigned long long)(q[R_BUS] >> 16), (unsigned long long)(q[R_BUS] >> 8 & 0xFF), (unsigned long long)(q[R_BUS] >> 3 & 0x1F),
                   (unsigned long long)(q[R_BUS] & 7), (int)(q[R_NUMA] & 0xFFFF) - 1, (int)(q[R_NUMA] >> 32), (int)(q[R_NUMA] >> 16 & 0xFFFF),
                   s > 0 ? 1e-9 * (double)q[R_CPU_NS] / s : 0.0, (unsigned long long)q[R_LO],
                   (unsigned long long)q[R_HI], (unsigned long long)q[R_PAIRS], s, s > 0 ? (double)q[R_PAIRS] / s : 0.0);
        }
        printf("]}\n");
        fflush(stdout);
        rc = (equal && !badsum && !(own_device && !distinct)) ? 0 : 1;
    }
    svh_elas_destroy(e);
    (void)hipFree(dI1);
    (void)hipFree(dI2);
    (void)hipFree(dD1);
    (void)hipFree(dD2);
    g.close_all();
    return rc;
}

bool parse(int argc, char** argv, Options& o) {
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&](int* dst) { if (i + 1 >= argc) return false; *dst = atoi(argv[++i]); return true; };
        if (a == "--ranks") { if (!val(&o.ranks)) return false; }
        else if (a == "--pairs-per-rank") { if (!val(&o.per_rank)) return false; }
        else if (a == "--total") { if (!val(&o.total)) return false; }
        else if (a == "--steps") { if (!val(&o.steps)) return false; }
        else if (a == "--warmup") { if (!val(&o.warmup)) return false; }
        else if (a == "--rank") { if (!val(&o.rank)) return false; }
        else if (a == "--fd") { if (!val(&o.fd)) return false; }
        else if (a == "--timeout-s") { if (!val(&o.timeout_s)) return false; }
        else if (a == "--lanes") { if (!val(&o.lanes)) return false; }
        else if (a == "--cores-per-rank") { if (!val(&o.cores_per_rank)) return false; }
        else if (a == "--no-bind") o.bind = 0;
        else if (a == "--spinup-ms") { if (!val(&o.spinup_ms)) return false; }
        else if (a == "--gather" && i + 1 < argc) o.gather = argv[++i];
        else if (a == "--images" && i + 1 < argc) o.images = argv[++i];
        else if (a == "--selftest-gather") o.selftest = true;
        else return false;
    }
    return o.ranks >= 1 && o.ranks <= 64 && o.per_rank >= 0 && o.per_rank <= 16384 && o.lanes >= 1 && o.lanes <= 64 && o.total >= 0 && o.steps >= 1 &&
           o.warmup >= 0 && (o.gather == "auto" || o.gather == "rccl" || o.gather == "pipes");
}

}  // namespace

int main(int argc, char** argv) {
    Options o;
    if (!parse(argc, argv, o)) {
        fprintf(stderr,
                "usage: %s --ranks N [--pairs-per-rank B | --total P] [--steps K] [--warmup W] [--gather auto|rccl|pipes]\n"
                "          [--images DIR] [--lanes L] [--cores-per-rank C] [--no-bind] [--spinup-ms T] [--timeout-s S] [--selftest-gather]\n", argv[0]);
        return 2;
    }
    if (o.selftest) o.gather = "pipes";
    if (o.rank >= 0) {
        // a rank: fix the library's process settings first (before RCCL or anything else starts the HIP runtime, so that
        // the hardware-queue count the engine wants can still be asked for) -- svh_init, include/svh.h
        svh_config cfg;
        svh_config_default(&cfg);
        if (o.lanes > 0) cfg.elas_workers = o.lanes;
        svh_init(&cfg);
    }
    return o.rank < 0 ? launcher(o, argv) : rank_main(o);
}
