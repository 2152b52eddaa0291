// Internal declarations shared by the host stages, the HIP kernels and the
// engine of libsvhip.so.  Nothing here is part of the public C-ABI.
#ifndef SVH_INTERNAL_H
#define SVH_INTERNAL_H

#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <vector>

#include "../../include/svh.h"
#include "svh_config.h"

namespace svh {

// ---------------------------------------------------------------- geometry
struct Dims {
    int32_t W, H;        // image
    int32_t DW, DH;      // disparity map (W/2,H/2 when subsampling)
    int32_t step;        // candidate lattice step (elas.cpp:453-457)
    int32_t Wc, Hc;      // candidate lattice (elas.cpp:460-463)
    int32_t gw, gh;      // disparity grid (elas.cpp:98-99)
    int32_t gwords;      // 32-bit words of one grid cell's disparity bit set
};
Dims make_dims(const svh_elas_params& p, int32_t W, int32_t H);

// A lane processes a GROUP of up to kMaxGroup independent pairs per kernel
// launch (blockIdx.z / .y = pair): per-launch work is large enough to fill 256
// CUs and launch gaps are paid once per group, not once per pair.
// (64 measured in round 5: 33.7 k pairs/s at 32, 48 and 64 pairs per launch -- the device is full at 32)
constexpr int kMaxGroup = 32;

// ---------------------------------------------------------------- fault injection (tests)
// TEST HOOK, not part of the public C-ABI (include/svh.h does not declare it; tests bind it by name, the sanitizer
// drivers pass their SVH_TEST_FAIL_AT on): svh_test_fail_at("<kind>:<n>[:<count>]") arms it, "" / NULL disarms; the
// library itself never reads a specification from the environment.  The n-th (1-based) HIP call of that kind in this process, counted from the moment the specification is set, is
// NOT issued and reports an error instead, and so do the count-1 calls of the kind after it (count 0: every one from the
// n-th on).  Kinds, as the engines' error macros see their calls:
//   malloc  hipMalloc / hipHostMalloc              copy  hipMemcpy*Async / hipMemset*Async
//   launch  the hipGetLastError() after a phase's launches      wait  stream / event waits (and queries in sleep-polls)
// fi_filter() is what HIP_TRY / VO_TRY / MAP_TRY put in front of the call; when nothing is armed it costs one relaxed
// load.  Boundary behaviour under a failure (tests/test_faults_gpu.py): the entry returns SVH_ERR_HIP, svh_last_error()
// names the call, one line goes to stderr, the lane / object is usable for the next call, nothing leaks.
bool fi_armed();
bool fi_hit(const char* expr_text);
void report_hip_failure(const char* entry);   // "svhip: <entry>: <last error>" on stderr, once per failing call
}   // namespace svh
extern "C" int32_t svh_test_fail_at(const char* spec);
namespace svh {

// Per-pair header, uploaded with the support points and triangle lists after
// the host stage.  Triangles of all pairs and both sides are packed in one
// array; tri_end[] are running ends in (pair, side) order.
struct GroupHdr {
    int32_t npairs;
    int32_t active[kMaxGroup];          // 0: fewer than 3 support points -> outputs untouched
    int32_t sup_off[kMaxGroup + 1];     // support point offsets (points, not ints)
    int32_t tri_end[2 * kMaxGroup];     // cumulative triangle count after (pair, side)
    int32_t total_sup, total_tri;       // sup_off[npairs], tri_end[2*npairs-1] (device-built headers)
};

// One rasterisation record per triangle and image side, computed on the device
// by k_prior with the reference's arithmetic (elas.cpp:605-680, 1026-1072).
struct alignas(16) TriRaster {
    float pa, pb, pc;          // disparity plane of this side   } one 16-byte load in the
    int32_t valid;             // |plane_a|<0.7 && |plane_d|<0.7 } matcher
    int32_t uA, uB, uC;        // (int32)A_u, (int32)B_u, (int32)C_u
    float ACa, ACb;            // long edge
    float ABa, ABb;            // first part
    float BCa, BCb;            // second part
    int32_t slot, first;   // (pair, side) slot of the triangle and the index of the slot's first triangle
    int32_t pad_[1];
};
static_assert(sizeof(TriRaster) == 64, "TriRaster is one 64-byte record");

// Host-side result of stages E4(second half)..E7 for one pair.
struct HostPrior {
    std::vector<int32_t> support;            // n x (u,v,d)
    std::vector<int32_t> tri[2];             // n x 3
};

// E5/E6 + list + corners (elas.cpp:174-318, 495-523); with write_back dcan receives the filtered
// lattice (the reference filters D_can in place), otherwise it is only read
void support_from_candidates(const svh_elas_params& p, const Dims& d, int16_t* dcan,
                             std::vector<int32_t>& support, bool write_back = true);
// E7, both sides.  Returns false when a triangulation fails.
bool triangulate_support(HostPrior& hp, bool parallel = false);
// runs a() on the calling thread and b() on a parked helper thread (or both here if none is free)
void run_pair(const std::function<void()>& a, const std::function<void()>& b);
// fn(0) on the calling thread, fn(1) .. fn(k-1) on helper threads (or here, one after the other, if none is free); k <= 8
void run_many(int k, const std::function<void(int)>& fn);
// `want` helper threads poll for tasks for the next `us` microseconds instead of sleeping (0, 0: back to sleep once
// idle).  Called a little ahead of a parallel section on a latency path; a polling helper is a busy core.
void helpers_warm(int want, int us);
// prior table + plane radius (elas.cpp:984-993)
void prior_table(const svh_elas_params& p, std::vector<int32_t>& P, int32_t* plane_radius);
// reference-layout grid (int32 [gh][gw][disp_max+2]) from the device bit sets, for the tap
void expand_grid(const svh_elas_params& p, const Dims& d, const uint32_t* mask,
                 std::vector<int32_t>& grid);

// par_depth > 0: the top `par_depth` levels of the divide-and-conquer run their halves on two
// threads (2^par_depth threads in all); the output is identical to the sequential run
int32_t delaunay(const float* pts, int32_t n, int32_t* tri, int32_t cap, int par_depth = 0, bool expect_dups = false);

// ---------------------------------------------------------------- device-side E5-E7
// The stages between the two matching phases on the GPU (elas_stage_kernels.hip): lattice filters
// + support list, both Delaunay triangulations, packed lists + header.  Per-lane scratch:
struct StageCounts {
    int32_t nsup[kMaxGroup];            // support points per pair (incl. corner points)
    int32_t ntri[2 * kMaxGroup];        // triangles per (pair, side)
    int32_t flags[kMaxGroup];           // STG_*
    int64_t dbg[32];                    // phase time stamps of slot 0 (wall_clock64, 10 ns ticks)
    // k_delaunay in two launches (large point sets): what the ordering launch hands to the build launch, per slot
    int32_t dt_m[2 * kMaxGroup];        // points of the triangulation after coincident ones were dropped; 0 = nothing to build
    int32_t dt_depth[2 * kMaxGroup];    // depth at which every node is a leaf | 0x10000 when points were dropped (dmap)
};
enum { STG_DUP = 1,        // coincident points in a triangulation: the host path must decide
       STG_OVERFLOW = 2,   // more support points than the scratch holds / coordinates out of range
       STG_FEW = 4 };      // fewer than 3 support points (elas.cpp:69-75)
struct StageDev {
    int16_t* dcan;          // [g][Wc*Hc] candidate lattice (filtered in place when it does not fit LDS)
    int32_t* sup_raw;       // [g][3*sup_cap] support list at a fixed stride
    StageCounts* counts;
    int32_t* ids;           // [2g][4*rec_cap] triangle records: corner indices,
    int32_t* xys;           //                  their packed coordinates,
    uint32_t* nbr;          //                  neighbour handles
    int32_t *pxy, *buck, *buck2, *byx, *order, *oxy;   // [2g][sup_cap]
    int32_t* dmap;          // [2g][sup_cap] surviving point -> support index, after coincident points were dropped
    uint32_t *lx, *ly, *tmp, *P;                        // [2g][sup_cap]
    uint32_t *fl, *fr;                                  // [2g][2][sup_cap] hull handles per node
    int32_t* wl;            // [g][3*Wc*Hc] k_lattice: valid cells, two work lists of fresh drops
    uint32_t* cntw;         // [g][Wc*Hc/4+1] k_lattice count bytes when the lattice does not fit LDS
    int32_t sup_cap, rec_cap;
};
struct LaunchCtx;
// the image geometry fits the device stage (LDS histograms, 14-bit coordinates)
bool stage_device_ok(const svh_elas_params& p, const Dims& d);
// ... and it is expected to beat the host stage there (what automatic mode asks)
bool stage_device_preferred(const svh_elas_params& p, const Dims& d, bool deep_batch = false);

// ---------------------------------------------------------------- device
// Kernel launchers (elas_kernels.hip).  LaunchCtx carries the hipStream_t (as
// void*) and an optional per-kernel timer (HIP events on that same stream).
struct Profiler {
    virtual void begin(const char* kernel) = 0;
    virtual void end() = 0;
    virtual ~Profiler() {}
};
struct LaunchCtx {
    void* stream;
    Profiler* prof;
    bool latency = false;   // the group is the only one of its call (single call, batch of one group): nothing else on the device
};

// images of a group: image k of pair j starts at I[k] + j*stride, rows pitch apart
struct DevImages {
    const uint8_t* I[2];
    size_t stride;
    int32_t pitch;
};
// disparity maps of a group: map k of pair j starts at D[k] + j*stride[k] (floats)
struct DevMaps {
    float* D[2];
    size_t stride[2];
};

// everything phase B needs, all device pointers
struct GroupDev {
    const GroupHdr* hdr;       // device copy
    const int32_t* support;    // packed (u,v,d)
    const int32_t* tri;        // packed corner triples, (pair, side) order
    const int32_t* P;          // prior table
    TriRaster* raster;         // packed like tri
    float* planes;             // 6 floats per triangle (t1a..t2c), packed like tri
    uint32_t* seed;            // [g][2][cells][gwords]
    uint32_t* mask;            // [g][2][cells][gwords] dilated
    uint16_t* lists;           // [g][2][cells][32] candidate records of k_grid_list (disp_max <= 255), else null
    const uint8_t* desc;       // [g][2][N*16]
    int32_t* owner;            // [g][2][N]
    float* Draw;               // [g][2][DN]
    int32_t plane_radius;
    int32_t owner_base;        // owner[] holds owner_base + 1 + triangle; values <= owner_base are stale
    int32_t prior_absmax;      // max |P[dd]|, dd <= plane_radius (selects the keyed match kernel)
    int32_t desc_fly;          // `desc` holds the Sobel planes only (k_sobel_planes): the matchers assemble descriptors
};

// k_lattice + k_delaunay + k_stage_pack: from S.dcan to the packed support / triangle lists and
// the group header in device memory
void launch_stage_device(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
                         const StageDev& S, GroupHdr* hdr, int32_t* support, int32_t* tri);

// fly: E1 writes the two Sobel planes into `desc` instead of the descriptors (see descriptors_on_the_fly)
void launch_descriptor(const LaunchCtx& cx, const DevImages& img, int32_t g, int32_t W, int32_t H,
                       int32_t half, uint8_t* desc, bool fly);
// true when both matchers will stage their descriptor rows from the Sobel planes for these parameters
bool descriptors_on_the_fly(const svh_elas_params& p, const Dims& d, int32_t prior_absmax, int32_t plane_radius,
                            bool have_lists);
void launch_support(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
                    const uint8_t* desc, int16_t* dcan, bool fly);
// planes + raster records + grid bit sets for the whole group.  total_sup / total_tri < 0: the
// counts are in the device header (device-built), the launch is sized for `tri_bound` triangles
// and the kernels stride over whatever is there
void launch_prior(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
                  int32_t total_sup, int32_t total_tri, const GroupDev& G);
void launch_owner(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
                  int32_t total_tri, const GroupDev& G);
// dense matching; when `lr_out` is given and the row kernel applies, the L/R check is fused in
// (returns true: the checked maps are in *lr_out, launch_lr must be skipped); `write_raw` keeps
// the raw maps in G.Draw as well (parity taps)
bool launch_match(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
                  const GroupDev& G, const DevMaps* lr_out, bool write_raw, const char** error);
void launch_lr(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
               const GroupDev& G, const DevMaps& out);
// post-processing of nside maps per pair, in place on `out`; scratch arrays are
// [g][nside][DN]
struct PostScratch {
    float* tmp;
    int32_t* labels;
    int32_t* counts;   // tile-local sizes at tile roots (0 elsewhere), then component sizes at roots
    int32_t* nroots;   // [g * nside][tiles] tile-local roots per 64 x 16 tile; their pixel indices are listed in `tmp`
};
// default configuration only (see post_tiles_ok): gap interpolation + adaptive mean as two tile
// kernels (D -> tmp -> D); the caller must skip launch_gap / launch_adaptive_mean then
bool post_tiles_ok(const svh_elas_params& p);
void launch_gap_mean_tiles(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
                           int32_t nside, const GroupDev& G, const DevMaps& out, const PostScratch& S);
void launch_segments(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
                     int32_t nside, const GroupDev& G, const DevMaps& out, const PostScratch& s, bool mask = true);
void launch_gap(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
                int32_t nside, const GroupDev& G, const DevMaps& out, const PostScratch& s);
void launch_adaptive_mean(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
                          int32_t nside, const GroupDev& G, const DevMaps& out, const PostScratch& s);
void launch_median(const LaunchCtx& cx, const Dims& d, int32_t g, int32_t nside, const GroupDev& G,
                   const DevMaps& out, const PostScratch& s);
// speckle labelling (tile union-find, seam merge, component sizes) without the mask pass
void launch_segments_label(const LaunchCtx& cx, const svh_elas_params& p, const Dims& d, int32_t g,
                           int32_t nside, const GroupDev& G, const DevMaps& in, const PostScratch& s);

}  // namespace svh
#endif
