// svh_get_device_topology / svh_bind_host_to_device (include/svh.h): where a GPU sits -- PCI bus id, NUMA node, the CPUs of
// that node -- and the host side of "one process per GPU": a rank's threads (the engine's workers, helper pools and
// streams' consumers are all created by the rank's own calls and inherit its mask) stay on the cores next to its GPU.
// SURVEY 8(e); BASELINE north_star "scaling reported at 1/2/4/8 GPUs".  Host code only; the sysfs parsing is exercised
// on the CPU through a directory of the test's making (svh_topology_from_sysfs).
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <ctype.h>
#include <hip/hip_runtime_api.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "svh_config.h"

namespace svh {
std::string& topo_error() {
    static thread_local std::string e;
    return e;
}
namespace {

bool read_line(const std::string& path, char* buf, size_t cap) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    const bool ok = fgets(buf, (int)cap, f) != nullptr;
    fclose(f);
    if (ok) buf[strcspn(buf, "\r\n")] = 0;
    return ok;
}

// "0-15,32-47" -> bits; returns the number of CPUs, -1 on a malformed list
int parse_cpulist(const char* s, uint64_t* mask, int words) {
    memset(mask, 0, sizeof(uint64_t) * (size_t)words);
    int n = 0;
    while (*s) {
        while (*s == ',' || isspace((unsigned char)*s)) s++;
        if (!*s) break;
        char* e = nullptr;
        const long a = strtol(s, &e, 10);
        if (e == s || a < 0) return -1;
        long b = a;
        s = e;
        if (*s == '-') {
            b = strtol(s + 1, &e, 10);
            if (e == s + 1 || b < a) return -1;
            s = e;
        }
        for (long c = a; c <= b; c++) {
            if (c >= 64L * words) return -1;
            if (!(mask[c >> 6] >> (c & 63) & 1)) n++;
            mask[c >> 6] |= 1ull << (c & 63);
        }
    }
    return n;
}

}   // namespace
}   // namespace svh

extern "C" {

int32_t svh_topology_from_sysfs(const char* sysfs_root, const char* pci_bus_id, svh_device_topology* out) {
    if (!sysfs_root || !pci_bus_id || !out) return SVH_ERR_BAD_ARG;
    memset(out, 0, sizeof *out);
    out->numa_node = -1;
    snprintf(out->pci_bus_id, sizeof out->pci_bus_id, "%s", pci_bus_id);
    for (char* c = out->pci_bus_id; *c; c++) *c = (char)tolower((unsigned char)*c);   // sysfs names are lower case
    char buf[4096];
    const std::string dev = std::string(sysfs_root) + "/bus/pci/devices/" + out->pci_bus_id;
    if (svh::read_line(dev + "/numa_node", buf, sizeof buf)) out->numa_node = atoi(buf);
    // the node's CPUs; a machine without NUMA information (node -1) reports every online CPU
    bool have = false;
    if (out->numa_node >= 0)
        have = svh::read_line(std::string(sysfs_root) + "/devices/system/node/node" + std::to_string(out->numa_node) + "/cpulist",
                              buf, sizeof buf);
    if (!have) have = svh::read_line(dev + "/local_cpulist", buf, sizeof buf);
    if (!have) have = svh::read_line(std::string(sysfs_root) + "/devices/system/cpu/online", buf, sizeof buf);
    if (!have) return SVH_OK;                         // nothing known: n_cpus = 0, the caller leaves its mask alone
    snprintf(out->cpulist, sizeof out->cpulist, "%s", buf);
    const int n = svh::parse_cpulist(buf, out->cpu_mask, SVH_TOPO_MASK_WORDS);
    if (n < 0) {
        svh::topo_error() = std::string("malformed CPU list: ") + buf;
        return SVH_ERR_BAD_ARG;
    }
    out->n_cpus = n;
    return SVH_OK;
}

int32_t svh_get_device_topology(int32_t device, svh_device_topology* out) {
    if (!out) return SVH_ERR_BAD_ARG;
    svh::ensure_init();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return SVH_ERR_NO_DEVICE;
    char id[64] = "";
    if (hipDeviceGetPCIBusId(id, (int)sizeof id, device) != hipSuccess) return SVH_ERR_HIP;
    const char* root = svh::env("SVH_SYSFS_ROOT");
    const int32_t rc = svh_topology_from_sysfs(root ? root : "/sys", id, out);
    out->device = device;
    return rc;
}

int32_t svh_bind_host_to_topology(const svh_device_topology* t, int32_t max_cpus) {
    if (!t) return SVH_ERR_BAD_ARG;
    if (t->n_cpus <= 0) return 0;
    cpu_set_t allowed, want;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return 0;
    CPU_ZERO(&want);
    int n = 0;
    for (int c = 0; c < 64 * SVH_TOPO_MASK_WORDS && c < CPU_SETSIZE; c++)
        if ((t->cpu_mask[c >> 6] >> (c & 63) & 1) && CPU_ISSET(c, &allowed) && (max_cpus <= 0 || n < max_cpus)) {
            CPU_SET(c, &want);
            n++;
        }
    if (n == 0) return 0;                       // the node's CPUs are outside this process' quota: leave the mask alone
    if (sched_setaffinity(0, sizeof want, &want) != 0) return 0;
    return n;
}

int32_t svh_bind_host_to_device(int32_t device, int32_t max_cpus) {
    svh_device_topology t;
    const int32_t rc = svh_get_device_topology(device, &t);
    if (rc != SVH_OK) return rc;
    return svh_bind_host_to_topology(&t, max_cpus);
}

}   // extern "C"
