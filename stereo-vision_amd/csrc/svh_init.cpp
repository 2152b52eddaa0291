// svh_init / svh_config / svh_get_runtime_info (include/svh.h): the library's process-wide settings, fixed explicitly by
// the host program or implicitly at the first use of the library -- never when the library is loaded.
//
// Round 5 set GPU_MAX_HW_QUEUES from a load-time constructor (a library editing its host's environment, racing with
// getenv in other threads, for a runtime that may already have started).  The measurement behind it stands
// (profiles/r05_hw_queues_foreign_streams.txt: the HIP runtime multiplexes streams onto that many hardware queues,
// default 4; 12 worker streams + a few spare want 20, more costs the device); what changed is who decides and when:
//   * svh_init(&cfg) at program start is the supported way: cfg.hw_queues asks for a count (0: the measured default,
//     < 0: hands off), and the variable is only written if the HIP runtime has not started in this process and the
//     process has not set it itself;
//   * a process that never calls svh_init gets svh_init(NULL) at its first svh_* call that needs the device;
//   * svh_get_runtime_info() says what happened (asked / applied / too late / left to the caller).
#include <dirent.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "svh_config.h"

namespace svh {
namespace {

std::atomic<int> g_fixed{0};          // the configuration has been fixed (explicitly or implicitly)
std::atomic<int> g_ready{0};          // ... and every subscriber has applied it (what ensure_init's fast path waits for)
thread_local bool t_in_fix = false;   // a subscriber that reaches ensure_init while it is being called
std::atomic<int> g_read_env{1};
std::mutex g_mu;
svh_config g_cfg;                     // effective configuration
svh_runtime_info g_info;
int g_explicit = 0;

std::vector<void (*)(const svh_config&, int)>& hooks() {
    static std::vector<void (*)(const svh_config&, int)>* v = new std::vector<void (*)(const svh_config&, int)>();
    return *v;
}

void defaults(svh_config* c) {
    memset(c, 0, sizeof *c);
    c->size = (uint32_t)sizeof *c;
    c->hw_queues = 0;
    c->elas_workers = 0;
    c->elas_pairs_per_launch = 0;
    c->elas_stage = -1;
    c->wait_us = -1;
    c->read_env = 1;
}

// Has the ROCm runtime started in this process?  It opens /dev/kfd when it does (and never closes it).
bool hip_runtime_started() {
    DIR* d = opendir("/proc/self/fd");
    if (!d) return false;
    bool found = false;
    char path[64], target[256];
    while (struct dirent* e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        snprintf(path, sizeof path, "/proc/self/fd/%s", e->d_name);
        const ssize_t n = readlink(path, target, sizeof target - 1);
        if (n <= 0) continue;
        target[n] = 0;
        if (!strcmp(target, "/dev/kfd")) { found = true; break; }
    }
    closedir(d);
    return found;
}

int fix(const svh_config* in, bool explicit_call) {
    std::lock_guard<std::mutex> lk(g_mu);
    const bool first = !g_fixed.load();
    if (!first && !explicit_call) return SVH_OK;   // (two first users at once: the second finds it done)
    svh_config c;
    defaults(&c);
    if (in) {
        if (in->size < 8 || in->size > 4096) return SVH_ERR_BAD_ARG;
        memcpy(&c, in, in->size < sizeof c ? in->size : sizeof c);
        c.size = (uint32_t)sizeof c;
    }
    g_read_env.store(c.read_env != 0);
    if (first) {
        // ---- hardware queues: only now, only if it can still take effect, only if nobody else decided
        int want = c.hw_queues;
        if (want == 0) {
            const char* own = env("SVH_HW_QUEUES");          // A/B override of the default
            want = own ? atoi(own) : 20;
            if (own && want <= 0) want = -1;
        }
        g_info.hw_queues_asked = want;
        const char* cur = getenv("GPU_MAX_HW_QUEUES");
        g_info.hip_started_before = hip_runtime_started() ? 1 : 0;
        if (want < 0) {
            g_info.hw_queues_state = SVH_HWQ_HANDS_OFF;
        } else if (cur && *cur) {
            g_info.hw_queues_state = SVH_HWQ_CALLER_SET;
            g_info.hw_queues_env = atoi(cur);
        } else if (g_info.hip_started_before) {
            g_info.hw_queues_state = SVH_HWQ_TOO_LATE;
        } else {
            char buf[16];
            snprintf(buf, sizeof buf, "%d", want);
            setenv("GPU_MAX_HW_QUEUES", buf, /*overwrite=*/0);
            g_info.hw_queues_state = SVH_HWQ_APPLIED;
            g_info.hw_queues_env = want;
            g_info.env_modified = 1;
        }
        g_info.implicit = explicit_call ? 0 : 1;
    }
    g_cfg = c;
    g_fixed.store(1, std::memory_order_release);
    g_explicit = explicit_call ? 1 : 0;
    t_in_fix = true;
    for (auto fn : hooks()) fn(g_cfg, g_explicit);
    t_in_fix = false;
    g_ready.store(1, std::memory_order_release);
    return SVH_OK;
}

}   // namespace

const char* env(const char* name) { return g_read_env.load(std::memory_order_relaxed) ? getenv(name) : nullptr; }

void ensure_init() {
    // (a second first user waits on the mutex inside fix() until the first one's subscribers are through)
    if (g_ready.load(std::memory_order_acquire) || t_in_fix) return;
    (void)fix(nullptr, false);
}

void on_config(void (*fn)(const svh_config&, int)) {
    std::lock_guard<std::mutex> lk(g_mu);
    hooks().push_back(fn);
    if (g_fixed.load()) fn(g_cfg, g_explicit);
}

}   // namespace svh

extern "C" {

void svh_config_default(svh_config* c) {
    if (c) svh::defaults(c);
}

int32_t svh_init(const svh_config* c) { return svh::fix(c, true); }

int32_t svh_get_runtime_info(svh_runtime_info* out) {
    if (!out) return SVH_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(svh::g_mu);
    *out = svh::g_info;
    out->initialised = svh::g_fixed.load() ? 1 : 0;
    out->read_env = svh::g_read_env.load();
    return SVH_OK;
}

}   // extern "C"
