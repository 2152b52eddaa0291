/*
 * svh.h -- C-ABI of libsvhip.so, the MI355X (gfx950) implementation of the
 * dense/sparse stereo hot path of willSapgreen/stereo-vision.
 *
 * Every entry point is plain C: opaque handles, raw pointers, sizes.  No C++
 * or torch types cross this boundary.  Each function names the reference
 * interface it replaces (paths relative to the reference checkout).
 *
 * The C++ drop-in classes in include/elas.h and include/matcher.h are thin
 * wrappers over these calls; INTEGRATION.md shows the reference-side binding.
 */
#ifndef SVH_H
#define SVH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------ */
/* return codes                                                              */
/* ------------------------------------------------------------------------ */
#define SVH_OK                 0
#define SVH_ERR_FEW_SUPPORT    1  /* <3 support points: outputs untouched, message on
                                     stdout (libelas/src/elas.cpp:69-75)              */
#define SVH_ERR_BAD_ARG       -1
#define SVH_ERR_HIP           -2  /* HIP runtime failure; svh_last_error() has text   */
#define SVH_ERR_UNSUPPORTED   -3  /* parameter combination not implemented on device  */
#define SVH_ERR_NO_DEVICE     -4
#define SVH_ERR_BAD_DIMS      -7  /* Matcher::pushBack with bad dimensions: "ERROR: Image dimension mismatch!" on
                                     stderr and the call is ignored, as in the reference (matcher.cpp:110-114) */

/* ------------------------------------------------------------------------ */
/* ELAS parameters: field-for-field Elas::parameters (libelas/src/elas.h:59-148),
 * bools widened to int32 for a stable ABI.                                   */
/* ------------------------------------------------------------------------ */
typedef struct svh_elas_params {
    int32_t disp_min;
    int32_t disp_max;
    float   support_threshold;
    int32_t support_texture;
    int32_t candidate_stepsize;
    int32_t incon_window_size;
    int32_t incon_threshold;
    int32_t incon_min_support;
    int32_t add_corners;
    int32_t grid_size;
    float   beta;
    float   gamma;
    float   sigma;
    float   sradius;
    int32_t match_texture;
    int32_t lr_threshold;
    float   speckle_sim_threshold;
    int32_t speckle_size;
    int32_t ipol_gap_width;
    int32_t filter_median;
    int32_t filter_adaptive_mean;
    int32_t postprocess_only_left;
    int32_t subsampling;
} svh_elas_params;

#define SVH_ELAS_ROBOTICS   0
#define SVH_ELAS_MIDDLEBURY 1

/* Elas::parameters::parameters(setting) -- libelas/src/elas.h:86-147 */
void svh_elas_params_default(svh_elas_params* p, int32_t setting);

/* ------------------------------------------------------------------------ */
/* library / device                                                          */
/* ------------------------------------------------------------------------ */
const char* svh_version(void);
/* thread-local text of the last failure on the calling thread */
const char* svh_last_error(void);
/* ---- process configuration -------------------------------------------------------------------------
 * Loading libsvhip.so does nothing: no environment variable is read or written, no thread is started, the HIP
 * runtime is not touched.  The settings below are fixed ONCE, by svh_init() -- call it at program start, before the
 * process creates a HIP context -- or, for a program that never calls it, by an implicit svh_init(NULL) inside the
 * first svh_* entry that needs the device (svh_device_count, svh_set_device, the *_create entries).
 *   hw_queues   hardware queues the HIP runtime should multiplex its streams onto (its GPU_MAX_HW_QUEUES, default
 *               4): 0 = the measured default, 20 (12 ELAS worker streams + spare; profiles/r05_hw_queues_*.txt),
 *               n > 0 = n, < 0 = hands off.  The runtime reads the variable when it starts, so it is written only
 *               if the runtime has NOT started in this process and the process has not set it itself;
 *               svh_get_runtime_info() tells which of the four cases applied.
 *   elas_workers, elas_pairs_per_launch, elas_stage, wait_us   = svh_elas_set_lanes / _set_group / _set_stage and
 *               the workers' poll interval (0 / 0 / -1 / -1 = defaults: 6, automatic, automatic, 40 us).
 *   read_env    1 (default): the SVH_* environment switches listed below are honoured (A/B measurements);
 *               0: the library never calls getenv (GPU_MAX_HW_QUEUES, read once by svh_init, excepted).
 * A later svh_init() may change everything except hw_queues.  Returns SVH_OK or SVH_ERR_BAD_ARG.          */
typedef struct svh_config {
    uint32_t size;                    /* sizeof(svh_config) of the caller's header (svh_config_default sets it) */
    int32_t  hw_queues;
    int32_t  elas_workers;
    int32_t  elas_pairs_per_launch;
    int32_t  elas_stage;
    int32_t  wait_us;
    int32_t  read_env;
    int32_t  reserved_[9];
} svh_config;
void    svh_config_default(svh_config* c);
int32_t svh_init(const svh_config* c);          /* NULL: defaults */
#define SVH_HWQ_NONE        0   /* not initialised yet                                          */
#define SVH_HWQ_APPLIED     1   /* GPU_MAX_HW_QUEUES was set by the library (env_modified = 1)  */
#define SVH_HWQ_CALLER_SET  2   /* the process had set it: left alone (hw_queues_env = its value) */
#define SVH_HWQ_TOO_LATE    3   /* the HIP runtime had already started: left alone              */
#define SVH_HWQ_HANDS_OFF   4   /* hw_queues < 0 (or SVH_HW_QUEUES=0)                           */
typedef struct svh_runtime_info {
    int32_t initialised;          /* the configuration is fixed                                   */
    int32_t implicit;             /* ... by the implicit svh_init(NULL) of a first use            */
    int32_t hw_queues_asked;      /* count asked for (-1: hands off)                              */
    int32_t hw_queues_state;      /* SVH_HWQ_*                                                    */
    int32_t hw_queues_env;        /* value of GPU_MAX_HW_QUEUES after svh_init (0: unset)         */
    int32_t hip_started_before;   /* the HIP runtime was already up when the configuration was fixed */
    int32_t env_modified;         /* the library wrote GPU_MAX_HW_QUEUES                          */
    int32_t read_env;
    int32_t reserved_[8];
} svh_runtime_info;
int32_t svh_get_runtime_info(svh_runtime_info* out);
int32_t     svh_device_count(void);
/* ---- where a GPU sits (one process per GPU) ------------------------------------------------------
 * svh_get_device_topology: PCI bus id (hipDeviceGetPCIBusId), NUMA node (sysfs numa_node of that PCI device, -1 when the
 * machine reports none) and the CPUs of that node (cpu_mask bit c = CPU c; cpulist as sysfs prints it; falls back to
 * the device's local_cpulist, then to every online CPU).
 * svh_bind_host_to_device: restricts the CALLING process' CPU mask to those CPUs (at most max_cpus of them, <= 0: all;
 * intersected with the mask the process already has -- a container's quota is respected); threads created afterwards,
 * the engine's included, inherit it.  Call it once per rank, before the first batch.  Returns the number of CPUs
 * bound (0: nothing known or nothing allowed -- the mask is left alone) or a negative error.
 * svh_topology_from_sysfs / svh_bind_host_to_topology: the two halves, for tests and for callers that know the bus id. */
#define SVH_TOPO_MASK_WORDS 16
typedef struct svh_device_topology {
    int32_t  device;
    int32_t  numa_node;
    int32_t  n_cpus;
    int32_t  reserved_;
    char     pci_bus_id[32];
    char     cpulist[256];
    uint64_t cpu_mask[SVH_TOPO_MASK_WORDS];
} svh_device_topology;
int32_t     svh_get_device_topology(int32_t device, svh_device_topology* out);
int32_t     svh_topology_from_sysfs(const char* sysfs_root, const char* pci_bus_id, svh_device_topology* out);
int32_t     svh_bind_host_to_topology(const svh_device_topology* t, int32_t max_cpus);
int32_t     svh_bind_host_to_device(int32_t device, int32_t max_cpus);
/* bind the calling thread's subsequent svh_* objects to a HIP device */
int32_t     svh_set_device(int32_t device);

/* ------------------------------------------------------------------------ */
/* Elas                                                                      */
/* ------------------------------------------------------------------------ */
typedef struct svh_elas svh_elas;

/* Environment switches read by the library (all optional; none is read with svh_config::read_env = 0):
 *   SVH_MATCHER_WAIT=0|1   Matcher / visual odometry waits: 0 spin in the driver, 1 sleep between polls
 *                          (default: spin while at most two threads are inside the library, poll otherwise)
 *   SVH_WAIT_US=n          sleep of the ELAS batch / stream workers between completion polls (40; 0 = spin)
 *   SVH_H2D_STRIDED=0, SVH_D2H_STRIDED=0   one copy per image / map instead of one strided copy per group
 *   SVH_MATCH_LIST=0       dense matching with round 3's k_match_keyed instead of k_match_list (A/B runs)
 *   SVH_DESC_FLY=0         E1 writes the 16-byte descriptor maps; default: only the two Sobel planes, the support and
 *                          dense matchers assemble the descriptor rows they stage (same results, 27 % less HBM traffic)
 *   SVH_MATCHER_COPY_KERNEL=0   small pinned transfers of the Matcher / visual odometry by hipMemcpyAsync instead of a
 *                          copy kernel in the stream's own queue
 *   SVH_UPLOAD_BATCH=0     lockstep pushBack / prefetch: one k_upload launch per image from the packing threads instead
 *                          of one recorded launch per camera over the call's objects
 *   SVH_POOL_SERIAL=1      helper pool of the lockstep entries: parallel_for calls take turns (the form before the job
 *                          list: A/B runs, profiles/r05_pool_jobs_ab.txt)
 *   SVH_DT_THREADS=256|512|1024, SVH_DT_SPREAD=n, SVH_DT_LDS_KB=n, SVH_DT_SPLIT=0   shape of the device
 *                          triangulation (defaults 512, 64, 96, split on for large point sets)
 *   SVH_DT_UNIFORM=0|n     scalar (wave-uniform) seam walk of the device triangulation at depths with at most n nodes per
 *                          wave; default: n = 1 for a group that is the only one of its call (single call, batch of
 *                          one group: +5 %), off otherwise (costs the throughput path 1.2 %:
 *                          profiles/r05_delaunay_scalar_walk.txt)
 *   SVH_DESC_FLY_KEYED=0   subsampling / disp_max > 255 keep the stored descriptor maps
 *   SVH_HW_QUEUES=n        the count svh_init asks for when svh_config::hw_queues is 0 (default 20); 0: hands off
 *   SVH_MATCH_WIDE768=0    rows of 1281-1920 px: 512-thread blocks (8 pixels per thread) in k_match_list instead of 768
 *   SVH_GAP_SEQ=1          wide interpolation gaps / add_corners: one thread per line (k_gap_lines) instead of the
 *                          per-row scan and the segmented column pass
 * (the full list with defaults: INTEGRATION.md, "Environment switches")                                      */
/* Elas::Elas(parameters) -- libelas/src/elas.h:151.  Cheap: callers build one
 * per frame (stereomapper/stereothread.cpp:113); device buffers live in a
 * process-wide pool keyed by (device, width, height).                        */
svh_elas* svh_elas_create(const svh_elas_params* p);
/* Elas::~Elas() -- libelas/src/elas.h:154 */
void      svh_elas_destroy(svh_elas* e);

/* Elas::process(I1,I2,D1,D2,dims) -- libelas/src/elas.h:165, elas.cpp:32-170.
 * Host pointers; dims = {width, height, bytes_per_line}; D1/D2 tightly packed
 * width x height float (width/2 x height/2 when subsampling).  Synchronous:
 * inputs are consumed and outputs complete on return.                        */
int32_t svh_elas_process(svh_elas* e, const uint8_t* I1, const uint8_t* I2,
                         float* D1, float* D2, const int32_t* dims);

/* n independent pairs of identical dims, pipelined over the engine's lanes
 * (one HIP stream + one host worker each).  status[i] receives the per-pair
 * return code (may be NULL).  Returns the first non-OK status or SVH_OK.     */
int32_t svh_elas_process_batch(svh_elas* e, int32_t n,
                               const uint8_t* const* I1, const uint8_t* const* I2,
                               float* const* D1, float* const* D2,
                               const int32_t* dims, int32_t* status);

/* Same, with images and disparity maps resident in device memory (HBM):
 * dI1/dI2 point at n images of dims[2]*dims[1] bytes spaced in_stride bytes
 * apart, dD1/dD2 at n maps spaced out_stride bytes apart.                   */
int32_t svh_elas_process_batch_device(svh_elas* e, int32_t n,
                                      const uint8_t* dI1, const uint8_t* dI2, size_t in_stride,
                                      float* dD1, float* dD2, size_t out_stride,
                                      const int32_t* dims, int32_t* status);

/* ---- streaming submission (round 4) -----------------------------------------
 * The reference's producer hands over ONE pair at a time (stereomapper/
 * readfromfilesthread.cpp:63,104 -> stereothread.cpp:68-176, maindialog.cpp:456-465) and
 * BASELINE configs[2] is a sequence "streamed" frame by frame.  A stream keeps the
 * engine's lanes full ACROSS calls: pairs enter a bounded queue, are cut into groups of
 * consecutive pairs (svh_elas_set_group) that workers pipeline over their lanes exactly
 * as the batch entries do, and come back in submission order with a status each.
 * Elas::process semantics per pair are unchanged (same maps, same "<3 support points"
 * message and status, outputs untouched then).
 *   open   dims = {width, height, bytes_per_line}; depth = pairs that may be in flight
 *          (0: lanes x 2 groups).  NULL on error (svh_last_error()).
 *   push   host buffers (any kind, pinned is faster): they must stay valid, and D1/D2
 *          unread, until the pair is popped.  Blocks while `depth` pairs are in flight.
 *          *ticket (may be NULL) receives the pair's sequence number, 0, 1, 2 ...
 *   push_device  the same with images / maps resident in device memory; pairs whose four
 *          pointers continue an arithmetic progression share a kernel launch.
 *   flush  the pairs pushed so far start now (a group otherwise starts when it is full
 *          or when pop has nothing older to wait for).
 *   pop    the next pair in submission order: blocks until it is done.  Returns SVH_OK
 *          and fills *ticket / *status, SVH_ERR_EMPTY when nothing is in flight,
 *          SVH_ERR_TIMEOUT after timeout_ms (< 0: no limit).
 *   close  drains (every pushed pair completes), then frees the stream.
 * push / flush / pop may be called from different threads (a producer and a consumer, or
 * several producers: tickets define the order).                                          */
#define SVH_ERR_EMPTY         -5
#define SVH_ERR_TIMEOUT       -6
typedef struct svh_elas_stream svh_elas_stream;
svh_elas_stream* svh_elas_stream_open(svh_elas* e, const int32_t* dims, int32_t depth);
int32_t svh_elas_stream_push(svh_elas_stream* s, const uint8_t* I1, const uint8_t* I2,
                             float* D1, float* D2, uint64_t* ticket);
int32_t svh_elas_stream_push_device(svh_elas_stream* s, const uint8_t* dI1, const uint8_t* dI2,
                                    float* dD1, float* dD2, uint64_t* ticket);
int32_t svh_elas_stream_flush(svh_elas_stream* s);
int32_t svh_elas_stream_pop(svh_elas_stream* s, uint64_t* ticket, int32_t* status, int32_t timeout_ms);
int32_t svh_elas_stream_close(svh_elas_stream* s);
/* n pairs in one call (n images spaced in_stride bytes apart, n maps out_stride bytes apart, as in
 * svh_elas_process_batch_device): push_device_n blocks while the stream is full, pop_n until n pairs
 * are done -- it WAITS for pairs that have not been pushed yet, so the consumer may start before the
 * producer (status may be NULL; returns the first non-OK status; *popped says how many came back).
 * A producer thread in push_device_n and a consumer in pop_n keep the lanes full without a call per
 * frame.                                                                                             */
int32_t svh_elas_stream_push_device_n(svh_elas_stream* s, int32_t n, const uint8_t* dI1, const uint8_t* dI2,
                                      size_t in_stride, float* dD1, float* dD2, size_t out_stride,
                                      uint64_t* first_ticket);
int32_t svh_elas_stream_pop_n(svh_elas_stream* s, int32_t n, int32_t* status, int32_t* popped);
/* the host-buffer form of push_device_n: n pairs through pointer arrays, as svh_elas_process_batch takes them
 * (readfromfilesthread.cpp:63,104 hands host frames over one by one; a producer with a ring of n decoded frames hands
 * them over in one call).  Frames whose buffers follow one another in memory travel in one strided copy per camera and
 * group, finished groups come back the same way, on the lanes' own streams: both PCIe directions are busy at once. */
int32_t svh_elas_stream_push_n(svh_elas_stream* s, int32_t n, const uint8_t* const* I1, const uint8_t* const* I2,
                               float* const* D1, float* const* D2, uint64_t* first_ticket);

/* Device buffers, pinned staging, streams and events live in a per-device pool of "lanes" that
 * outlives the svh_elas handles (callers build an Elas per frame).  svh_elas_trim() releases
 * every lane that is not in use at the moment -- e.g. after one large batch in a long-lived
 * process -- and returns how many it released; the pool regrows on demand. */
int64_t svh_elas_trim(void);

/* number of batch workers the engine runs per device (default 6; each is double-buffered: two HIP
 * streams and buffer sets, the host stage of one group overlaps the device stages of the next) */
int32_t svh_elas_set_lanes(int32_t lanes);
/* pairs a lane pushes through each kernel launch (1..32; default 32 for images up to ~1
 * Mpixel, proportionally fewer for larger ones -- 16 at 1920x1080 -- until this is called):
 * batches are cut into groups of this many consecutive pairs */
int32_t svh_elas_set_group(int32_t pairs);

/* where the stages between the two matching phases run (lattice filters elas.cpp:174-279, support
 * list :495-523, the two Delaunay triangulations :534-600): 1 = on the device (k_lattice,
 * k_delaunay: no host round trip inside a pair), 0 = on the host (elas_host.cpp, delaunay.cpp),
 * -1 = automatic (the default: batches of images whose candidate lattice fits the device
 * kernel's LDS -- up to ~20 k lattice cells, e.g. 1242x375 -- and batches of 32 or more pairs of
 * larger images on the device; a single svh_elas_process and small batches of large images on
 * the host, where two host threads are the shorter path for one pair).  Use 1 where host cores
 * are scarce (several ranks per node).  Results are identical.  Returns the mode in effect. */
int32_t svh_elas_set_stage(int32_t where);
/* diagnostics: groups of pairs that took the device stage since the library was loaded, and how
 * many of them it handed back to the host path (coincident support points in a triangulation,
 * whose survivor depends on Triangle's pivot stream -- triangle.cpp:5446-5501, 6179-6196) */
void svh_elas_stage_stats(int64_t* device_groups, int64_t* handed_back);
/* the engine settings in effect: out[0] workers (svh_elas_set_lanes), out[1] pairs per launch (0: automatic by image
 * size), out[2] stage (-1 / 0 / 1, svh_elas_set_stage), out[3] the workers' poll interval in microseconds */
void svh_elas_get_settings(int32_t out[4]);

/* Stage taps for parity tests: after a successful svh_elas_process() the
 * intermediate of the given stage (of the last pair processed through handle
 * e) is copied to buf.  *size receives the byte size; returns SVH_ERR_BAD_ARG
 * when cap is too small.  Taps are recorded only after svh_elas_set_taps(e,1). */
enum svh_elas_stage {
    SVH_ELAS_DESC1 = 0,      /* u8  [H][W][16]        descriptor.cpp:88-119          */
    SVH_ELAS_DESC2,
    SVH_ELAS_DCAN_RAW,       /* i16 [Hc][Wc]          elas.cpp:471-493               */
    SVH_ELAS_SUPPORT,        /* i32 [n][3] (u,v,d)    elas.cpp:495-523               */
    SVH_ELAS_TRI1,           /* i32 [n][3]            elas.cpp:534-600 (left)        */
    SVH_ELAS_TRI2,           /* i32 [n][3]            (right)                        */
    SVH_ELAS_PLANES1,        /* f32 [n][6] t1a..t2c   elas.cpp:605-680               */
    SVH_ELAS_PLANES2,
    SVH_ELAS_GRID1,          /* i32 [gh][gw][disp_max+2] elas.cpp:684-780            */
    SVH_ELAS_GRID2,
    SVH_ELAS_D1_RAW,         /* f32 [H][W]            elas.cpp:960-1118              */
    SVH_ELAS_D2_RAW,
    SVH_ELAS_D1_LR,          /* after leftRightConsistencyCheck elas.cpp:1122        */
    SVH_ELAS_D2_LR,
    SVH_ELAS_D1_SEG,         /* after removeSmallSegments elas.cpp:1208              */
    SVH_ELAS_D2_SEG,
    SVH_ELAS_D1_GAP,         /* after gapInterpolation elas.cpp:1330                 */
    SVH_ELAS_D2_GAP,
    SVH_ELAS_STAGE_COUNT
};
int32_t svh_elas_set_taps(svh_elas* e, int32_t enable);
int32_t svh_elas_get_stage(svh_elas* e, int32_t stage, void* buf, size_t cap, size_t* size);

/* per-stage GPU/host milliseconds of the last single svh_elas_process() call;
 * names mirror the -DPROFILE labels of elas.cpp:58-163.  Returns the number
 * of entries written (<= cap).                                               */
int32_t svh_elas_last_timing(svh_elas* e, const char** names, float* ms, int32_t cap);

/* ------------------------------------------------------------------------ */
/* per-kernel timing: when enabled every kernel launch is bracketed by HIP
 * events on the stream it is launched on; totals accumulate per kernel name
 * over all lanes until reset.  svh_profile_get(-1,..) returns the entry count. */
/* ------------------------------------------------------------------------ */
int32_t svh_profile_enable(int32_t on);
void    svh_profile_only(const char* kernel);   /* time just this kernel (NULL: all) */
void    svh_profile_reset(void);
int32_t svh_profile_get(int32_t index, const char** name, double* total_ms, int64_t* launches);

/* ------------------------------------------------------------------------ */
/* host-side geometry helpers that stay on the CPU; exported
 * so they can be parity-tested directly.                                     */
/* ------------------------------------------------------------------------ */
/* Delaunay triangulation reproducing Triangle 1.6 "zQB" output order
 * (libelas/src/triangle.cpp:8499; elas.cpp:534-600).  pts = n (x,y) floats.
 * tri receives up to cap triangles (3 input-order vertex indices each).
 * Returns the triangle count or a negative error.                            */
int32_t svh_delaunay(const float* pts, int32_t n, int32_t* tri, int32_t cap);
/* same, with the top `par_depth` levels of the divide-and-conquer on two threads each
 * (2^par_depth threads); the triangle list is identical to svh_delaunay's */
int32_t svh_delaunay_mt(const float* pts, int32_t n, int32_t* tri, int32_t cap, int32_t par_depth);
/* The halves of a parallel triangulation (and the four parts of the Matcher's outlier vote, the two triangulations
 * of one svh_elas_process call) run on up to seven helper threads of the library.  They exist only on the latency
 * paths -- one sequence per process side; batch and lockstep entries never use them --, sleep between uses, poll for
 * work during a parallel section (at most ~0.3 ms per frame / call) and pin themselves to the cores that share the
 * calling thread's L3 cache; there are no more of them than the process has CPUs to spare (none below three CPUs).  Counters since the library was loaded: out[0] tasks run, out[1] of them after the helper
 * moved to another L3 domain, out[2] handed to a polling helper, out[3] to a sleeping one (a futex wake, 30-50 us late) */
void svh_host_helper_stats(int64_t out[4]);

/* Lattice filters + support list (libelas/src/elas.cpp:174-318, 495-523) as the
 * engine runs them between the two device phases.  dcan = candidate lattice
 * [Hc][Wc] as produced by the support kernel, modified in place.  support
 * receives up to cap (u,v,d) triples; returns the point count.               */
int32_t svh_elas_support_from_candidates(const svh_elas_params* p, int32_t width, int32_t height,
                                         int16_t* dcan, int32_t* support, int32_t cap);

/* ======================================================================== */
/* Matcher (libviso2)                                                        */
/* ======================================================================== */
/* Matcher::parameters -- libviso2/src/matcher.h:41-69 (defaults: constructor) */
typedef struct svh_matcher_params {
    int32_t nms_n;
    int32_t nms_tau;
    int32_t match_binsize;
    int32_t match_radius;
    int32_t match_disp_tolerance;
    int32_t outlier_disp_tolerance;
    int32_t outlier_flow_tolerance;
    int32_t multi_stage;
    int32_t half_resolution;
    int32_t refinement;
    double  f, cu, cv, base;       /* calibration, only used for match prediction */
} svh_matcher_params;
void svh_matcher_params_default(svh_matcher_params* p);

/* Matcher::p_match -- libviso2/src/matcher.h:87-102, same field order */
typedef struct svh_p_match {
    float u1p, v1p; int32_t i1p;
    float u2p, v2p; int32_t i2p;
    float u1c, v1c; int32_t i1c;
    float u2c, v2c; int32_t i2c;
} svh_p_match;

typedef struct svh_matcher svh_matcher;

/* Matcher::Matcher(parameters) / ~Matcher() -- matcher.cpp:33-98 */
svh_matcher* svh_matcher_create(const svh_matcher_params* p);
void         svh_matcher_destroy(svh_matcher* m);
/* Per-call timeline of a Matcher (round 6): svh_matcher_set_timing(1) switches the collection on for the process (the
 * same clocks SVH_MATCHER_TIMING=1 prints at destroy); svh_matcher_get_timing returns averages per call, in ms: seven
 * host wall-clock steps (pushBack: pack + enqueue, wait; matchFeatures: sparse matching, sparse vote, prior statistics,
 * dense matching + refinement, dense vote -- the matching steps include their wait for the device) and the device time of
 * the three device phases from HIP events on the object's streams (first launch to last copy).  Returns the number of
 * entries written (<= cap); reset != 0 clears the sums. */
void    svh_matcher_set_timing(int32_t on);
int32_t svh_matcher_get_timing(svh_matcher* m, const char** names, double* ms, int32_t cap, int32_t reset);
/* Matcher::setIntrinsics -- matcher.h:78-84 */
void svh_matcher_set_intrinsics(svh_matcher* m, double f, double cu, double cv, double base);
/* Matcher::pushBack(I1,I2,dims,replace) -- matcher.cpp:102-205.  I2 may be NULL
 * (single-image variant, matcher.h:118).  dims = {width,height,bytes_per_line}. */
int32_t svh_matcher_push_back(svh_matcher* m, const uint8_t* I1, const uint8_t* I2,
                              const int32_t* dims, int32_t replace);
/* Matcher::matchFeatures(method, Tr_delta) -- matcher.cpp:209-293.  method 0 flow,
 * 1 stereo, 2 quad.  Tr_delta: 16 doubles row-major (4x4) or NULL.               */
int32_t svh_matcher_match_features(svh_matcher* m, int32_t method, const double* Tr_delta);
/* One frame of K sequences: the two calls above for K Matchers in lockstep (same parameters and image size).
 * The device work of all K objects is ONE launch per kernel (blockIdx.z = object), the per-object host steps
 * (row packing, outlier votes, prior statistics) run on helper threads.  Results per object are exactly those of
 * K separate calls; objects that cannot run in lockstep (different parameters, taps on) are run one by one.
 * I1 / I2 / Tr_delta: K pointers each (I2, Tr_delta and their entries may be NULL as in the single calls).
 * The reference has no batched form: this is the multi-sequence counterpart of demo.cpp's frame loop
 * (viso_stereo.cpp:41-68 per object). */
int32_t svh_matcher_push_back_batch(svh_matcher* const* ms, int32_t K, const uint8_t* const* I1,
                                    const uint8_t* const* I2, const int32_t* dims, int32_t replace);
int32_t svh_matcher_match_features_batch(svh_matcher* const* ms, int32_t K, int32_t method,
                                         const double* const* Tr_delta);
/* The NEXT frame of the K Matchers handed over early: its rows are packed, uploaded and its features computed into
 * a third set of per-frame buffers on the objects' second streams, and the call returns WITHOUT waiting -- that
 * work then overlaps the matchFeatures (and, under the visual odometry, the motion estimate) of the frame before
 * it.  The following svh_matcher_push_back / svh_matcher_push_back_batch of these objects is called with
 * I1 = I2 = NULL and takes the prefetched frame (ring-buffer rotation and `replace` as usual); results are those
 * of the plain calls.  One prefetched frame per object at a time.  (No counterpart in the reference, whose
 * pushBack computes the features synchronously, matcher.cpp:102-205.) */
int32_t svh_matcher_prefetch_batch(svh_matcher* const* ms, int32_t K, const uint8_t* const* I1,
                                   const uint8_t* const* I2, const int32_t* dims);
/* Matcher::bucketFeatures -- matcher.cpp:297-343 (std::random_shuffle on the host) */
int32_t svh_matcher_bucket_features(svh_matcher* m, int32_t max_features, float bucket_width,
                                    float bucket_height);
/* Matcher::getMatches -- matcher.h:146: copies up to cap matches, returns the count */
int32_t svh_matcher_get_matches(svh_matcher* m, svh_p_match* out, int32_t cap);
/* Matcher::getGain -- matcher.cpp:347-389 */
float   svh_matcher_get_gain(svh_matcher* m, const int32_t* inliers, int32_t n);

/* parity taps */
enum svh_matcher_table {            /* feature tables, 12 x int32 per feature      */
    SVH_M_1P1 = 0, SVH_M_1P2, SVH_M_2P1, SVH_M_2P2,   /* previous: left sparse/dense, right .. */
    SVH_M_1C1, SVH_M_1C2, SVH_M_2C1, SVH_M_2C2        /* current                               */
};
int32_t svh_matcher_get_features(svh_matcher* m, int32_t table, int32_t* out, int32_t cap);
enum svh_matcher_stage {
    SVH_M_SPARSE_RAW = 0,   /* svh_p_match[]  after matching(), 1st pass (matcher.cpp:269)     */
    SVH_M_SPARSE,           /* after removeOutliers (matcher.cpp:270)                          */
    SVH_M_RANGES,           /* float[bins][16]: u_min[4],u_max[4],v_min[4],v_max[4] (:273)     */
    SVH_M_DENSE_RAW,        /* after matching(), 2nd pass (:276)                               */
    SVH_M_DENSE_REFINED,    /* after refinement (:279)                                         */
    SVH_M_DENSE,            /* after removeOutliers (:281) == getMatches before bucketing      */
    SVH_M_STAGE_COUNT
};
/* stages are recorded only after svh_matcher_set_taps(m, 1) (extra device->host copies) */
int32_t svh_matcher_set_taps(svh_matcher* m, int32_t enable);
int32_t svh_matcher_get_stage(svh_matcher* m, int32_t stage, void* buf, size_t cap, size_t* size);
/* filter images of the current left frame for parity checks:
 * 0 du, 1 dv (matching resolution), 2 du_full, 3 dv_full (u8); 4 f1 blob, 5 f2 checkerboard (i16) */
int32_t svh_matcher_get_filter(svh_matcher* m, int32_t which, void* buf, size_t cap, size_t* size,
                               int32_t* dims3);

/* ===========================================================================
 * libviso2 VisualOdometryStereo (the caller of the Matcher)
 *   VisualOdometryStereo::parameters        libviso2/src/viso_stereo.h:30-44
 *   VisualOdometry::parameters/bucketing/calibration  libviso2/src/viso.h:30-66
 * =========================================================================== */
typedef struct svh_vo_params {
    svh_matcher_params match;      /* parameters::match                                       */
    int32_t bucket_max_features;   /* bucketing::max_features   (2)                           */
    double  bucket_width;          /* bucketing::bucket_width   (50)                          */
    double  bucket_height;         /* bucketing::bucket_height  (50)                          */
    double  f, cu, cv;             /* calibration               (1, 0, 0)                     */
    double  base;                  /* baseline in metres        (1.0)                         */
    int32_t ransac_iters;          /* (200)                                                   */
    double  inlier_threshold;      /* (2.0)                                                   */
    int32_t reweighting;           /* (1)                                                     */
} svh_vo_params;
typedef struct svh_vo svh_vo;

void    svh_vo_params_default(svh_vo_params* p);
/* VisualOdometryStereo(param): creates the Matcher, sets its intrinsics, Tr_delta = I and
 * calls srand(0) like the reference (viso.cpp:28-37, viso_stereo.cpp:26-31) */
svh_vo* svh_vo_create(const svh_vo_params* p);
void    svh_vo_destroy(svh_vo* v);
/* bool VisualOdometryStereo::process(I1,I2,dims,replace) -- viso_stereo.cpp:41-68.
 * returns 1 (true), 0 (false: motion estimate failed) or a negative SVH_ERR_* */
int32_t svh_vo_process(svh_vo* v, const uint8_t* I1, const uint8_t* I2, const int32_t* dims,
                       int32_t replace);
/* The random numbers of bucketFeatures (matcher.cpp:297-343) and getRandomSample (viso.cpp:130-153).  Default
 * (enable = 0): libc rand(), the process-wide stream the reference draws from after its srand(0) (viso.cpp:36).
 * enable = 1: the object draws from a PRIVATE generator that reproduces glibc's srand(seed) / rand() sequence,
 * i.e. it sees exactly the numbers a reference object sees when it has the process to itself (seed 0 = the
 * reference's constructor), independent of other objects and threads, and without glibc's rand() lock, for which
 * K threads otherwise contend.  With private streams svh_vo_process_batch also runs the bucketing of its objects
 * in parallel. */
void svh_vo_set_private_rand(svh_vo* v, int32_t enable, uint32_t seed);
/* the first n numbers of that generator for `seed` (== srand(seed); rand() x n with glibc): lets tests pin it */
void svh_rand_sequence(uint32_t seed, int32_t* out, int32_t n);
/* svh_vo_process for K objects in lockstep (one frame of K sequences): batched Matcher steps, the K motion
 * estimates in two launches.  libc rand() is drawn in the order of K svh_vo_process calls, so with the same
 * srand() the results are bit-identical to that loop.  ok[i] (optional) receives the per-object return value;
 * the call returns the number of objects whose motion was updated, or a negative SVH_ERR_*.  Objects that are
 * still bootstrapping (viso_stereo.cpp:47-53) or differ in parameters are processed one after the other. */
int32_t svh_vo_process_batch(svh_vo* const* vs, int32_t K, const uint8_t* const* I1, const uint8_t* const* I2,
                             const int32_t* dims, int32_t replace, int32_t* ok);
/* The pipelined frame loop.  svh_vo_prefetch_batch hands over the FIRST frame (svh_matcher_prefetch_batch for the
 * objects' Matchers; returns without waiting).  Then, per frame, svh_vo_process_next_batch processes the frame
 * handed over before and hands over the next one (next_I1 / next_I2; NULL after the last frame) as soon as the ring
 * buffers have rotated: the next frame's row packing, upload and feature extraction overlap this frame's matching
 * and motion estimate.  Results and return values are those of svh_vo_process_batch with the images passed
 * directly.  svh_vo_process / svh_vo_process_batch with I1 = I2 = NULL also take a frame handed over before. */
int32_t svh_vo_prefetch_batch(svh_vo* const* vs, int32_t K, const uint8_t* const* I1, const uint8_t* const* I2,
                              const int32_t* dims);
int32_t svh_vo_process_next_batch(svh_vo* const* vs, int32_t K, const uint8_t* const* next_I1,
                                  const uint8_t* const* next_I2, const int32_t* dims, int32_t replace, int32_t* ok);
/* bool VisualOdometry::process(p_matched) -- viso.h:87-91: motion from given matches */
int32_t svh_vo_process_matches(svh_vo* v, const svh_p_match* matches, int32_t n);
/* vector<double> estimateMotion(p_matched) -- viso_stereo.cpp:72-228 (RANSAC + Gauss-Newton on
 * the device).  returns 1 and tr_delta[6] = rx,ry,rz,tx,ty,tz, or 0 (empty vector).  Consumes
 * 3 libc rand() values per RANSAC iteration exactly like getRandomSample (viso.cpp:130-153). */
int32_t svh_vo_estimate_motion(svh_vo* v, const svh_p_match* matches, int32_t n, double* tr_delta6);
void    svh_vo_get_motion(svh_vo* v, double* Tr16);              /* getDeltaMotion, row major  */
int32_t svh_vo_get_matches(svh_vo* v, svh_p_match* out, int32_t cap);   /* _matcher->getMatches() */
int32_t svh_vo_num_matches(svh_vo* v);                           /* getNumberOfMatches         */
int32_t svh_vo_get_inliers(svh_vo* v, int32_t* out, int32_t cap);/* getInlierIndices           */
float   svh_vo_get_gain(svh_vo* v, const int32_t* inliers, int32_t n);
svh_matcher* svh_vo_matcher(svh_vo* v);                          /* the owned Matcher (taps)   */

#ifdef __cplusplus
}
#endif
#endif /* SVH_H */
