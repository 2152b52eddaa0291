// Device-wide hardware counters over a region of an UNSERIALISED run (measurement helper; does not ship).
//
// rocprofv3 --pmc collects per dispatch and runs the kernels one after the other, so its numbers say nothing
// about the steady state in which ~10 kernels of different workers share the device.  This tool library uses
// rocprofiler-sdk's device counting service instead: the counters run free on the whole device while the
// application runs as it always does, and the application brackets the region it wants:
//
//     ROCP_TOOL_LIBRARIES=tools/libdevcount.so   (set before the HIP runtime starts; bench.py --devcount does it)
//     svh_devcount_start("GRBM_GUI_ACTIVE,SQ_BUSY_CYCLES,SQ_INSTS_VALU,...")   -> 0 or a negative status
//     svh_devcount_sample(values, n)     values[i] = counter i summed over its instances (XCDs, SEs ...)
//     svh_devcount_stop()
//
// One counter set per start (the job fails if the set does not fit one pass, like rocprofv3 --pmc).
// Build: make -C tools libdevcount.so
#include <rocprofiler-sdk/registration.h>
#include <rocprofiler-sdk/rocprofiler.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {
rocprofiler_context_id_t g_ctx = {};
rocprofiler_agent_id_t g_agent = {};
bool g_have_agent = false, g_configured = false, g_running = false;
rocprofiler_counter_config_id_t g_cfg = {.handle = 0};
std::vector<uint64_t> g_ids;            // counter handle per requested name (0 = not found)
std::vector<std::string> g_names;
size_t g_records = 0;

#define DC_TRY(call)                                                                                   \
    do {                                                                                               \
        rocprofiler_status_t s_ = (call);                                                              \
        if (s_ != ROCPROFILER_STATUS_SUCCESS) {                                                        \
            fprintf(stderr, "[devcount] %s: %s\n", #call, rocprofiler_get_status_string(s_));          \
            return -(int)s_ - 1000;                                                                    \
        }                                                                                              \
    } while (0)

void on_start(rocprofiler_context_id_t ctx, rocprofiler_agent_id_t, rocprofiler_device_counting_agent_cb_t set, void*) {
    if (g_cfg.handle) set(ctx, g_cfg);
}

int tool_init(rocprofiler_client_finalize_t, void*) {
    auto cb = [](rocprofiler_agent_version_t, const void** arr, size_t n, void*) {
        for (size_t i = 0; i < n && !g_have_agent; i++) {
            const auto* a = static_cast<const rocprofiler_agent_v0_t*>(arr[i]);
            if (a->type == ROCPROFILER_AGENT_TYPE_GPU) {
                g_agent = a->id;
                g_have_agent = true;
            }
        }
        return ROCPROFILER_STATUS_SUCCESS;
    };
    if (rocprofiler_query_available_agents(ROCPROFILER_AGENT_INFO_VERSION_0, cb, sizeof(rocprofiler_agent_t), nullptr) !=
            ROCPROFILER_STATUS_SUCCESS || !g_have_agent) {
        fprintf(stderr, "[devcount] no GPU agent\n");
        return 0;
    }
    if (rocprofiler_create_context(&g_ctx) != ROCPROFILER_STATUS_SUCCESS) return 0;
    if (rocprofiler_configure_device_counting_service(g_ctx, rocprofiler_buffer_id_t{.handle = 0}, g_agent, on_start,
                                                      nullptr) == ROCPROFILER_STATUS_SUCCESS)
        g_configured = true;
    else
        fprintf(stderr, "[devcount] device counting service refused\n");
    return 0;
}
void tool_fini(void*) {}
}   // namespace

extern "C" {

rocprofiler_tool_configure_result_t* rocprofiler_configure(uint32_t, const char*, uint32_t, rocprofiler_client_id_t* id) {
    id->name = "svh-devcount";
    static rocprofiler_tool_configure_result_t cfg{sizeof(rocprofiler_tool_configure_result_t), &tool_init, &tool_fini, nullptr};
    return &cfg;
}

// 0 = ok; -1 = tool not registered (library not named in ROCP_TOOL_LIBRARIES before HIP started)
int svh_devcount_start(const char* csv) {
    if (!g_configured) return -1;
    if (g_running) return -2;
    std::map<std::string, rocprofiler_counter_id_t> avail;
    auto it_cb = [](rocprofiler_agent_id_t, rocprofiler_counter_id_t* c, size_t n, void* ud) {
        auto* m = static_cast<std::map<std::string, rocprofiler_counter_id_t>*>(ud);
        for (size_t i = 0; i < n; i++) {
            rocprofiler_counter_info_v0_t info;
            if (rocprofiler_query_counter_info(c[i], ROCPROFILER_COUNTER_INFO_VERSION_0, &info) == ROCPROFILER_STATUS_SUCCESS)
                (*m)[info.name] = c[i];
        }
        return ROCPROFILER_STATUS_SUCCESS;
    };
    DC_TRY(rocprofiler_iterate_agent_supported_counters(g_agent, it_cb, &avail));
    g_names.clear();
    g_ids.clear();
    std::vector<rocprofiler_counter_id_t> want;
    g_records = 0;
    for (const char* p = csv; *p;) {
        const char* q = strchr(p, ',');
        std::string name(p, q ? (size_t)(q - p) : strlen(p));
        p = q ? q + 1 : p + name.size();
        auto f = avail.find(name);
        g_names.push_back(name);
        if (f == avail.end()) {
            fprintf(stderr, "[devcount] counter %s not available\n", name.c_str());
            g_ids.push_back(0);
            continue;
        }
        g_ids.push_back(f->second.handle);
        want.push_back(f->second);
        rocprofiler_counter_info_v1_t v1;
        DC_TRY(rocprofiler_query_counter_info(f->second, ROCPROFILER_COUNTER_INFO_VERSION_1, &v1));
        g_records += v1.dimensions_instances_count;
    }
    if (want.empty()) return -3;
    DC_TRY(rocprofiler_create_counter_config(g_agent, want.data(), want.size(), &g_cfg));
    DC_TRY(rocprofiler_start_context(g_ctx));
    g_running = true;
    return 0;
}

// values[i] for the i-th requested counter (sum over its instances); returns the number of names or < 0
int svh_devcount_sample(double* values, int n) {
    if (!g_running) return -1;
    std::vector<rocprofiler_counter_record_t> rec(g_records + 64);
    size_t cnt = rec.size();
    DC_TRY(rocprofiler_sample_device_counting_service(g_ctx, {}, ROCPROFILER_COUNTER_FLAG_NONE, rec.data(), &cnt));
    for (int i = 0; i < n && i < (int)g_ids.size(); i++) values[i] = 0.0;
    for (size_t r = 0; r < cnt; r++) {
        rocprofiler_counter_id_t cid = {.handle = 0};
        rocprofiler_query_record_counter_id(rec[r].id, &cid);
        for (int i = 0; i < n && i < (int)g_ids.size(); i++)
            if (g_ids[i] == cid.handle) values[i] += rec[r].counter_value;
    }
    return (int)g_ids.size();
}

int svh_devcount_stop(void) {
    if (!g_running) return -1;
    g_running = false;
    DC_TRY(rocprofiler_stop_context(g_ctx));
    return 0;
}
}
